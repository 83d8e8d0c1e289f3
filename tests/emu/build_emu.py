"""TEST INFRASTRUCTURE: compile unipose_amd/csrc/*.hip for the host with the fiber emulator
(tests/emu/hip_emu.h) into tests/emu/libunipose_emu.so.  Used only by tests/ (CPU, no GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "unipose_amd", "csrc")
OUT = os.path.join(HERE, "libunipose_emu.so")
SOURCES = ["conv_igemm.hip", "norm_act.hip", "spatial.hip", "plan.hip"]


def build(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "up_common.h"), os.path.join(CSRC, "bf16s_glds.h"), os.path.join(CSRC, "bf16s_big.h"), os.path.join(CSRC, "f32_glds.h"), os.path.join(CSRC, "stem_f32.h"), os.path.join(CSRC, "bn_fold.h"),
                   os.path.join(HERE, "hip_emu.h"),
                   os.path.join(HERE, "emu_switch.cpp"), os.path.join(ROOT, "include", "unipose_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DUP_EMU", "-Wno-unknown-pragmas", "-Wno-psabi",
           "-include", os.path.join(HERE, "hip_emu.h")]
    for s in srcs:
        cmd += ["-x", "c++", s]
    cmd += ["-x", "c++", os.path.join(HERE, "emu_switch.cpp"), "-o", OUT, "-lpthread"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
