"""Reduce rocprofv3 --pmc CSV passes (one directory per pass) to a per-kernel table.
FETCH_SIZE / WRITE_SIZE are KB per dispatch; gfx950 note (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-B
requests at 64 B, so wide coalesced reads are reported at 1/2 — both raw and x2 are printed.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs)."""
import collections
import csv
import os
import re
import sys


def load(d):
    f = [x for x in os.listdir(d) if x.endswith("counter_collection.csv")][0]
    return list(csv.DictReader(open(os.path.join(d, f))))


def main(base):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in sorted(os.listdir(base)):
        p = os.path.join(base, tag)
        if not os.path.isdir(p):
            continue
        for r in load(p):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:58]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{'kernel':58s} {'n':>5} {'FETCH MB':>9} {'(x2)':>8} {'WRITE MB':>9} {'MfmaUtil%':>9}")
    rows = []
    for k, c in per.items():
        n = max(len(v) for v in c.values())
        avg = {m: sum(v) / len(v) for m, v in c.items()}
        util = None
        if avg.get("GRBM_GUI_ACTIVE"):
            util = 100.0 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024)
        rows.append((sum(c.get("GRBM_GUI_ACTIVE", [0])), k, n, avg.get("FETCH_SIZE", 0) / 1e3, avg.get("WRITE_SIZE", 0) / 1e3, util))
    for _, k, n, f, w, u in sorted(rows, reverse=True)[:24]:
        print(f"{k:58s} {n:5d} {f:9.1f} {2 * f:8.1f} {w:9.1f} {'' if u is None else format(u, '9.1f')}")


if __name__ == "__main__":
    main(sys.argv[1])
