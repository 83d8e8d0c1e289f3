"""Build libunipose_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv_igemm.hip", "norm_act.hip", "spatial.hip", "plan.hip"]
HEADERS = ["up_common.h", "bf16s_glds.h", "bf16s_big.h", "f32_glds.h", "stem_f32.h", "bn_fold.h"]
OUT = os.path.join(HERE, "libunipose_hip.so")
STAMP = OUT + ".stamp"          # sha256 of the sources the library next to it was built from
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "unipose_hip.h")]


def source_hash() -> str:
    """sha256 over the kernel sources, headers and build flags: what a library must have been built from to be the tree's."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in _deps():
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def binary_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    with open(OUT, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def library_is_current() -> bool:
    """True when libunipose_hip.so exists, its stamp names exactly the sources of this tree AND the sha256 of the binary the
    build wrote (a stamp that only names sources is a statement about a text file — a library swapped in afterwards,
    e.g. by tools/gpu/ab.sh, kept passing).  Not a time-stamp comparison: a shipped binary must be provably the tree's."""
    try:
        with open(STAMP) as f:
            words = f.read().split()
        return os.path.exists(OUT) and len(words) == 2 and words[0] == source_hash() and words[1] == binary_hash()
    except OSError:
        return False


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and library_is_current():
        return OUT
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [_hipcc(), *FLAGS, *srcs, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(STAMP):
        os.remove(STAMP)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + " " + binary_hash() + "\n")
    return OUT


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
