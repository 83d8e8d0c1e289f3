"""Where does the wall time of a training step go?  Per-stream accounting of a rocprofv3 --kernel-trace database.

    python tools/wall_accounting.py bench_results.db [--skip N]

For every steady-state step window (between two launches of the once-per-step loss kernel, like rocprof_summary.py --steady)
the script reports, per step:
  * per stream (queue): number of kernels, sum of kernel time, sum and count of idle gaps between consecutive kernels of that
    stream (gap = next.start - prev.end when positive), the largest gaps;
  * device level (sweep line over all kernels): time with >= 1 kernel running, with kernels of two streams running, idle;
  * the overlap matrix: time during which a kernel of class {MFMA, memory-bound} of the MAIN stream and one of class
    {MFMA, memory-bound} of the SIDE stream were running together, and the time each class ran with the other stream empty;
  * per kernel class: in-step duration vs the same kernels' count, for the stretch of a kernel under sharing.
The main stream is the one that launched the loss kernel; MFMA kernels are the igemm / wgrad variants.
"""
import re
import sqlite3
import sys
from collections import defaultdict

MARKER = "mse_partial_kernel"
MFMA = re.compile(r"igemm|wgrad_(glds|kernel|bf16)|wgrad_glds")


def klass(name):
    return "mfma" if MFMA.search(name) and "reduce" not in name else "mem"


def stream_column(cur):
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    for c in ("stream_id", "stream", "queue_id", "queue"):
        if c in cols:
            return c, cols
    return None, cols


def main(path, skip=1):
    cur = sqlite3.connect(path).cursor()
    scol, cols = stream_column(cur)
    print(f"# {path}: columns of `kernels`: {', '.join(cols)}")
    print(f"# stream column: {scol}")
    sel = f"select name, start, end, {scol or '0'} from kernels order by start"
    rows = [(n, s, e, q) for n, s, e, q in cur.execute(sel)]
    marks = [(s, q) for n, s, e, q in rows if MARKER in n]
    if len(marks) - 1 - skip < 1:
        raise SystemExit(f"only {len(marks)} launches of {MARKER}")
    main_q = marks[0][1]
    lo, hi = marks[skip][0], marks[-1][0]
    steps = len(marks) - 1 - skip
    ks = [(n, s, e, q) for n, s, e, q in rows if lo <= s < hi]
    print(f"# steady state: {steps} step windows, {len(ks) / steps:.1f} kernels per step, window {(hi - lo) / steps / 1e6:.3f} ms per step")
    streams = sorted({q for _, _, _, q in ks}, key=lambda q: (q != main_q, q))
    print(f"# main stream = {main_q}; streams seen: {streams}\n")

    print("## per stream, per step")
    print(f"{'stream':>8} {'kernels':>8} {'busy ms':>9} {'gaps':>6} {'gap ms':>8} {'mean gap us':>12} {'gaps<5us':>9} {'ms in <5us':>11} {'gaps 5-50':>9} {'ms':>7} {'gaps>50us':>9} {'ms':>7}")
    for q in streams:
        mine = sorted([(s, e) for _, s, e, qq in ks if qq == q])
        busy = sum(e - s for s, e in mine)
        gaps = [b[0] - a[1] for a, b in zip(mine, mine[1:]) if b[0] > a[1]]
        small = [g for g in gaps if g < 5e3]
        mid = [g for g in gaps if 5e3 <= g < 5e4]
        big = [g for g in gaps if g >= 5e4]
        tag = "main" if q == main_q else "side"
        print(f"{str(q) + ' ' + tag:>8} {len(mine) / steps:8.1f} {busy / steps / 1e6:9.3f} {len(gaps) / steps:6.0f} {sum(gaps) / steps / 1e6:8.3f} "
              f"{(sum(gaps) / max(len(gaps), 1)) / 1e3:12.2f} {len(small) / steps:9.0f} {sum(small) / steps / 1e6:11.3f} "
              f"{len(mid) / steps:9.0f} {sum(mid) / steps / 1e6:7.3f} {len(big) / steps:9.0f} {sum(big) / steps / 1e6:7.3f}")

    # sweep line: at every instant the set of running kernels per (stream role, class)
    ev = []
    for n, s, e, q in ks:
        role = "main" if q == main_q else "side"
        ev.append((s, +1, role, klass(n)))
        ev.append((e, -1, role, klass(n)))
    ev.sort(key=lambda t: (t[0], t[1]))
    live = defaultdict(int)
    acc = defaultdict(float)
    prev = ev[0][0]
    for t, d, role, kc in ev:
        if t > prev:
            dt = t - prev
            m = "mfma" if live[("main", "mfma")] else ("mem" if live[("main", "mem")] else None)
            s_ = "mfma" if live[("side", "mfma")] else ("mem" if live[("side", "mem")] else None)
            acc[(m, s_)] += dt
            prev = t
        live[(role, kc)] += d
    tot = (hi - lo) / steps / 1e6
    print("\n## device level, ms per step (main-stream class x side-stream class running together)")
    print(f"{'main / side':>14} {'(none)':>9} {'mfma':>9} {'mem':>9}")
    for m in (None, "mfma", "mem"):
        print(f"{str(m):>14} " + " ".join(f"{acc[(m, s_)] / steps / 1e6:9.3f}" for s_ in (None, "mfma", "mem")))
    idle = acc[(None, None)] / steps / 1e6
    both = sum(v for (m, s_), v in acc.items() if m and s_) / steps / 1e6
    print(f"\nwindow {tot:.3f} ms = busy {tot - idle:.3f} (two streams together {both:.3f}) + idle {idle:.3f}")

    # phases of a window [loss_k, loss_k+1): backward = loss .. first optimizer kernel, optimizer, forward = rest
    marks_t = [s for s, _ in marks[skip:]]
    ph = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])       # phase -> wall, main busy, side busy, main mfma
    for w0, w1 in zip(marks_t, marks_t[1:]):
        win = [(n, s, e, q) for n, s, e, q in ks if w0 <= s < w1]
        opt = [(s, e) for n, s, e, q in win if "Adam" in n or "multi_tensor_apply" in n]
        if not opt:
            continue
        o0, o1 = min(s for s, _ in opt), max(e for _, e in opt)
        for name, a, b in (("backward", w0, o0), ("optimizer", o0, o1), ("forward", o1, w1)):
            p = ph[name]
            p[0] += b - a
            for n, s, e, q in win:
                if a <= s < b:
                    p[1 if q == main_q else 2] += e - s
                    if q == main_q and klass(n) == "mfma":
                        p[3] += e - s
    if ph:
        print("\n## phases of a step window (ms per step): wall, kernel time on the main stream (of which MFMA), on the side stream(s)")
        for name in ("backward", "optimizer", "forward"):
            w, mb, sb, mm = (v / steps / 1e6 for v in ph[name])
            print(f"{name:>10}: wall {w:7.3f}  main {mb:7.3f} (mfma {mm:7.3f}, other {mb - mm:6.3f}, not running {w - mb:6.3f})  side {sb:7.3f}")

    print("\n## per kernel class and stream: in-step time of its kernels (sharing the GPU stretches them)")
    by = defaultdict(lambda: [0, 0.0])
    for n, s, e, q in ks:
        short = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:70]
        k = ("main" if q == main_q else "side", short)
        by[k][0] += 1
        by[k][1] += e - s
    for (role, short), (cnt, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{role:>5} {cnt / steps:7.1f} x {t / cnt / 1e3:8.1f} us = {t / steps / 1e6:7.3f} ms  {short}")


if __name__ == "__main__":
    sk = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 1
    main(sys.argv[1], sk)
