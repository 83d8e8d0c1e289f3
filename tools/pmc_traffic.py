"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes reduced by tools/pmc_summary.py: per bench-level kernel
variant (the names bench.py's roofline.by_kernel uses) the launch-weighted average FETCH_SIZE / WRITE_SIZE and the
MFMA-busy fraction.   usage: python tools/pmc_traffic.py <pmc dir> <tag> > profiles/pmc_traffic.json"""
import collections
import csv
import json
import os
import re
import sys


def bench_name(k):
    m = re.match(r"(?:void )?up::(?:glds::)?(wgrad_kernel|igemm_kernel|igemm_bf16_kernel|igemm_glds32_kernel|wgrad_glds32_kernel|"
                 r"igemm_glds_kernel|wgrad_glds_kernel)<(\d+), (\d+)(?:, (\d+))?", k)
    if not m:
        return None
    kind, bm, bn, mode = m.groups()
    if kind in ("igemm_glds32_kernel", "wgrad_glds32_kernel"):        # round 4: the names of bench.py's profile variants
        return f"{kind}<{bm},{bn}>"
    if kind in ("igemm_glds_kernel", "wgrad_glds_kernel"):
        return f"{kind}<{bm},{bn}> (bf16)"
    if kind == "wgrad_kernel":
        return f"wgrad_kernel<{bm},{bn}>"
    if kind == "igemm_kernel":
        return f"igemm_kernel<{bm},{bn},{'generic' if mode == '0' else 'aligned'}>"
    return f"igemm_bf16_kernel<{bm},{bn}>"


def main(base, tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(os.listdir(base)):
        p = os.path.join(base, d)
        if not os.path.isdir(p):
            continue
        f = [x for x in os.listdir(p) if x.endswith("counter_collection.csv")][0]
        for r in csv.DictReader(open(os.path.join(p, f))):
            n = bench_name(r["Kernel_Name"])
            if n:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for n, c in sorted(acc.items()):
        avg = {m: sum(v) / len(v) for m, v in c.items()}
        fetch, write = avg.get("FETCH_SIZE", 0.0) / 1e3, avg.get("WRITE_SIZE", 0.0) / 1e3      # KB -> MB
        rec = {"launches": len(c.get("FETCH_SIZE", [])), "fetch_mb_raw": round(fetch, 1), "write_mb": round(write, 1),
               "traffic_mb": round(2 * fetch + write, 1)}
        if avg.get("GRBM_GUI_ACTIVE"):
            rec["mfma_busy_pct"] = round(100.0 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024), 1)
        out[n] = rec
    print(json.dumps({
        "_comment": "HBM-side traffic per launch from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy; one "
                    "pass each, with --kernel-trace only) of `UNIPOSE_SYNC_WGRAD=1 python bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline --no-profile --no-alt-math`, launch-weighted per bench-level kernel variant by "
                    "tools/pmc_traffic.py.  fetch_mb_raw is the counter; per MI355X_MICROARCH.md (HBM / rocprofv3 section) "
                    "FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so traffic_mb = 2*fetch + write.  "
                    "bench.py copies the entry of its dominant kernel into roofline.traffic.",
        "tag": tag, "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
