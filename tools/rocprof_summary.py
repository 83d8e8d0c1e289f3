"""Summarise a rocprofv3 --kernel-trace rocpd database (bench_results.db) into a per-kernel table.

    python tools/rocprof_summary.py bench_results.db STEPS            every dispatch of the run, divided by STEPS
    python tools/rocprof_summary.py bench_results.db --steady [SKIP]  only the steady-state steps: the window between the
        (SKIP+1)-th and the last launch of the once-per-step loss kernel (up::mse_partial_kernel; default SKIP = 1).  A window
        [loss of step k, loss of step k+1) holds exactly one backward + optimiser + forward, i.e. one step's launches; the first
        step's one-off work (weight packing, tap-order / rectangle tables, allocator growth: ~260 fills and ~150 copies) stays out.
"""
import re
import sqlite3
import sys

MARKER = "mse_partial_kernel"


def main(path, steps, steady_skip=None):
    cur = sqlite3.connect(path).cursor()
    where, note = "", ""
    if steady_skip is not None:
        marks = [r[0] for r in cur.execute("select start from kernels where name like ? order by start", (f"%{MARKER}%",))]
        if len(marks) - 1 - steady_skip < 1:
            raise SystemExit(f"only {len(marks)} launches of {MARKER}: nothing left after skipping {steady_skip}")
        lo, hi = marks[steady_skip], marks[-1]
        steps = len(marks) - 1 - steady_skip
        where = f" where start >= {lo} and start < {hi}"
        note = f" (steady state: {steps} step windows between launches {steady_skip + 1} and {len(marks)} of {MARKER})"
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                            f"max(end-start)/1e3 from kernels{where} group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    launches = sum(r[1] for r in rows)
    own = sum(r[1] for r in rows if "up::" in r[0])
    own_ms = sum(r[2] for r in rows if "up::" in r[0])
    print(f"# {path}: total kernel time {tot:.2f} ms over {steps} steps = {tot / steps:.2f} ms/step{note}")
    print(f"# launches per step: {launches / steps:.1f}, of which up:: {own / steps:.1f}, others {(launches - own) / steps:.1f} "
          f"({(tot - own_ms) / steps:.3f} ms/step)")
    print(f"{'ms/step':>9} {'%':>6} {'calls/step':>10} {'avg us':>9} {'min us':>9} {'max us':>9}  kernel")
    for name, n, ms, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)[:100]
        print(f"{ms / steps:9.3f} {100 * ms / tot:6.2f} {n / steps:10.1f} {avg:9.1f} {mn:9.1f} {mx:9.1f}  {short}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--steady":
        main(sys.argv[1], 0, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
