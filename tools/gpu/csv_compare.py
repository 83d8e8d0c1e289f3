"""Compare two per-launch CSVs written through UP_PROFILE_CSV (tools/gpu/csv.sh): launches grouped by
kernel family and GEMM shape.   python tools/gpu/csv_compare.py a.csv b.csv [min_ms]"""
import collections
import csv
import sys


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        fam = r["kernel"].split("<")[0].replace("igemm_glds32_kernel", "igemm_kernel").replace("wgrad_glds32_kernel", "wgrad_kernel")   # the two fp32 generations line up
        key = (fam, int(r["M"]), int(r["N"]), int(r["K"]))
        e = d.setdefault(key, [0, 0.0, set(), 0.0])
        e[0] += 1
        e[1] += float(r["ms"])
        e[2].add(r["kernel"].split("<")[1].rstrip(">") + "/" + r["workgroups"])
        e[3] += float(r["tflops"]) * float(r["ms"])
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
ta = tb = 0.0
rows = []
for k in a:
    if k not in b:
        continue
    ta += a[k][1]
    tb += b[k][1]
    rows.append((b[k][1] - a[k][1], k, a[k], b[k]))
rows.sort(key=lambda r: -abs(r[0]))
print(f"total matched: {ta:.2f} ms -> {tb:.2f} ms")
for fam in sorted({k[0] for k in a}):
    sa = sum(v[1] for k, v in a.items() if k[0] == fam and k in b)
    sb = sum(v[1] for k, v in b.items() if k[0] == fam and k in a)
    print(f"  {fam}: {sa:.2f} -> {sb:.2f}")
for d, k, x, y in rows:
    if abs(d) < floor:
        break
    print(f"{d:+7.3f} ms  {k[0]:13s} M={k[1]:7d} N={k[2]:5d} K={k[3]:6d} x{x[0]:3d}  {x[1]:7.3f} -> {y[1]:7.3f}   "
          f"{x[3] / x[1]:6.1f} -> {y[3] / y[1]:6.1f} TF   {sorted(x[2])[:2]} -> {sorted(y[2])[:2]}")
