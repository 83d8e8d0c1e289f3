"""Rank the GEMM shapes of a per-launch CSV (UP_PROFILE_CSV, see csv.sh) by the time they lose against
the fp32 MFMA peak: which layers to look at first.   python tools/gpu/csv_loss.py launches.csv [peak_tflops] [rows]"""
import collections
import csv
import sys

peak = float(sys.argv[2]) if len(sys.argv) > 2 else 157.3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["kernel"].split("<")[0], int(r["M"]), int(r["N"]), int(r["K"]))
    e = d.setdefault(key, [0, 0.0, 0.0, r["kernel"], r["workgroups"]])
    ms = float(r["ms"])
    e[0] += 1
    e[1] += ms
    e[2] += float(r["tflops"]) * ms          # TFLOP/s * ms = GFLOP
total = sum(e[1] for e in d.values())
ideal = sum(e[2] / peak for e in d.values())
print(f"total {total:.2f} ms, at the {peak} TFLOP/s peak {ideal:.2f} ms, lost {total - ideal:.2f} ms")
acc = 0.0
for k, e in sorted(d.items(), key=lambda kv: -(kv[1][1] - kv[1][2] / peak))[:top]:
    lost = e[1] - e[2] / peak
    acc += lost
    print(f"lost {lost:6.3f} (cum {acc:6.2f})  {k[0]:12s} M={k[1]:7d} N={k[2]:5d} K={k[3]:7d} x{e[0]:3d} {e[1]:7.3f} ms "
          f"{e[2] / e[1]:6.1f} TF  {e[3].split('<')[1].rstrip('>')} wg={e[4]}")
