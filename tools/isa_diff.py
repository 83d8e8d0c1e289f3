"""Which gfx950 kernels differ between two builds of unipose_amd/csrc?  (claims like "the default
kernels are ISA-identical to the profiled build" are checked with this, not asserted.)

    python tools/isa_diff.py --ref <git-rev>            # <git-rev> vs the working tree
    python tools/isa_diff.py --ref a0c0134 --new HEAD   # two commits
    python tools/isa_diff.py --ref HEAD --only 'igemm_kernel<64, 64'   # restrict the report

Each side's three .hip files are compiled device-only to AMDGCN assembly with the flags of unipose_amd/build.py
(hipcc --offload-arch=gfx950 -O3 -std=c++17); the text is cut into one body per kernel symbol, labels and comments are
normalised away, and bodies are compared by hash.  Output: one line per kernel that is new / gone / changed (with the
instruction-count delta and the VGPR/SGPR/LDS/scratch figures of both sides) and the number of identical kernels; a kernel whose
symbol changed (template parameter list) but whose body did not is reported as RENAMED, not as a change.
Exit code 0 if nothing in the selection changed, 1 otherwise.  Runs without a GPU (hipcc cross-compiles)."""
import argparse
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "unipose_amd/csrc"
HDRS = ("include/unipose_hip.h",)


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise SystemExit("hipcc not found")


def demangle(names):
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt")
    if not filt:
        return {n: n for n in names}
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return dict(zip(names, out))


def materialise(rev, dst):
    """The sources of `rev` (None = working tree) under dst/, with the directory layout the #includes expect."""
    os.makedirs(os.path.join(dst, SRC), exist_ok=True)
    os.makedirs(os.path.join(dst, "include"), exist_ok=True)
    if rev is None:
        for f in os.listdir(os.path.join(ROOT, SRC)):
            shutil.copy(os.path.join(ROOT, SRC, f), os.path.join(dst, SRC, f))
        for h in HDRS:
            shutil.copy(os.path.join(ROOT, h), os.path.join(dst, h))
        return
    files = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", rev, SRC + "/"], capture_output=True, text=True,
                           check=True).stdout.split()
    for f in files + list(HDRS):
        blob = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{f}"], capture_output=True, check=True).stdout
        with open(os.path.join(dst, f), "wb") as fh:
            fh.write(blob)


LABEL = re.compile(r"\.L[A-Za-z_]*\d+(_\d+)?")


def kernels_of(asm):
    """{symbol: (normalised body, n_instructions, {resource: value})} for every .amdhsa_kernel in the assembly text."""
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        kv = dict(re.findall(r"\.amdhsa_(\w+) (\S+)", m.group(2)))
        meta[m.group(1)] = {"vgpr": kv.get("next_free_vgpr"), "sgpr": kv.get("next_free_sgpr"), "agpr_off": kv.get("accum_offset"),
                            "lds": kv.get("group_segment_fixed_size"), "scratch": kv.get("private_segment_fixed_size")}
    out = {}
    for sym in meta:
        m = re.search(r"^%s:[^\n]*\n(.*?)^\s*\.amdhsa_kernel %s$" % (re.escape(sym), re.escape(sym)), asm, re.S | re.M)
        if not m:
            continue
        lines, labels = [], {}
        for ln in m.group(1).split("\n"):
            ln = ln.split(";")[0].strip()
            if not ln or ln.startswith(".") and not ln.endswith(":"):
                continue
            lines.append(ln)
        body = "\n".join(lines)
        for mm in LABEL.finditer(body):
            labels.setdefault(mm.group(0), "L%d" % len(labels))
        body = LABEL.sub(lambda mm: labels[mm.group(0)], body)
        n_inst = sum(1 for ln in lines if not ln.endswith(":"))
        out[sym] = (body, n_inst, meta[sym])
    return out


def build_side(rev, tmp, tag):
    d = os.path.join(tmp, tag)
    materialise(rev, d)
    res = {}
    for f in sorted(os.listdir(os.path.join(d, SRC))):
        if not f.endswith(".hip"):
            continue
        s = os.path.join(d, SRC, f[:-4] + ".s")
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", os.path.join(d, SRC, f), "-o", s]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(f"{tag}: {f} does not compile:\n{r.stderr[-2000:]}")
        res.update(kernels_of(open(s).read()))
    return res


def fmt_meta(m):
    return "v%s s%s lds%s scr%s" % (m["vgpr"], m["sgpr"], m["lds"], m["scratch"])


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ref", required=True, help="git revision of the reference side")
    ap.add_argument("--new", default=None, help="git revision of the other side (default: the working tree)")
    ap.add_argument("--only", default=None, help="regex on the demangled kernel name")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        a = build_side(args.ref, tmp, "ref")
        b = build_side(args.new, tmp, "new")
    names = demangle(sorted(set(a) | set(b)))
    sel = re.compile(args.only) if args.only else None
    same = changed = renamed = 0
    # a kernel whose template parameter list changed has a new symbol: pair GONE / NEW symbols with identical bodies
    gone = {}
    for sym in set(a) - set(b):
        gone.setdefault(hashlib.sha1(a[sym][0].encode()).digest(), []).append(sym)
    moved = {}
    for sym in sorted(set(b) - set(a)):
        cands = gone.get(hashlib.sha1(b[sym][0].encode()).digest())
        if cands:
            moved[sym] = cands.pop()
    moved_old = set(moved.values())
    for sym in sorted(names, key=lambda s: names[s]):
        nm = names[sym]
        if sel and not sel.search(nm):
            continue
        if sym in moved_old:
            continue
        if sym in moved:
            print(f"RENAMED  {names[moved[sym]]}  ->  {nm}  (identical body)")
            renamed += 1
            continue
        if sym not in a:
            print(f"NEW      {nm}  [{fmt_meta(b[sym][2])}, {b[sym][1]} instr]")
            changed += 1
        elif sym not in b:
            print(f"GONE     {nm}")
            changed += 1
        elif hashlib.sha1(a[sym][0].encode()).digest() != hashlib.sha1(b[sym][0].encode()).digest():
            print(f"CHANGED  {nm}  [{fmt_meta(a[sym][2])}, {a[sym][1]} instr] -> [{fmt_meta(b[sym][2])}, {b[sym][1]} instr]")
            changed += 1
        else:
            same += 1
    print(f"# {args.ref} vs {args.new or 'working tree'}: {same} kernels identical, {renamed} renamed with identical bodies, "
          f"{changed} new/gone/changed"
          + (f" (selection /{args.only}/)" if args.only else ""))
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
