"""tools/rocprof_summary.py --steady: only the windows between two launches of the once-per-step loss kernel are counted."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rocprof_summary_steady_window(tmp_path):
    db = tmp_path / "bench_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    rows = []
    for step in range(4):
        n_fill = 50 if step == 0 else 2                       # the first step's one-off work
        for _ in range(n_fill):
            rows.append(("void at::native::FillFunctor<float>(...)", t, t + 1000)); t += 2000
        for _ in range(10):
            rows.append(("void up::glds::igemm_glds32_kernel<64, 64>(up::IgemmArgs)", t, t + 100000)); t += 110000
        rows.append(("up::mse_partial_kernel(float const*)", t, t + 5000)); t += 6000
        for _ in range(10):
            rows.append(("void up::glds::wgrad_glds32_kernel<128, 128>(up::WgradArgs)", t, t + 200000)); t += 210000
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), str(db), "--steady", "1"],
                         capture_output=True, text=True, check=True).stdout
    head = out.splitlines()
    assert "2 step windows" in head[0]
    # per window: 10 wgrad (behind the marker) + 2 fills + 10 igemm (of the next step) + the marker itself
    assert "launches per step: 23.0, of which up:: 21.0, others 2.0" in head[1], head[1]
    full = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), str(db), "4"],
                          capture_output=True, text=True, check=True).stdout
    assert "FillFunctor" in full and "14.0" in [ln for ln in full.splitlines() if "FillFunctor" in ln][0]     # (50 + 6) / 4
