"""Summarise a rocprofv3 --kernel-trace rocpd database (bench_results.db) into a per-kernel table."""
import re
import sqlite3
import sys


def main(path, steps):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                            "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# {path}: total kernel time {tot:.2f} ms over {steps} steps = {tot / steps:.2f} ms/step")
    print(f"{'ms/step':>9} {'%':>6} {'calls/step':>10} {'avg us':>9} {'min us':>9} {'max us':>9}  kernel")
    for name, n, ms, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)[:100]
        print(f"{ms / steps:9.3f} {100 * ms / tot:6.2f} {n / steps:10.1f} {avg:9.1f} {mn:9.1f} {mx:9.1f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
