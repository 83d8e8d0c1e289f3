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


def test_wall_accounting_two_streams(tmp_path):
    """tools/wall_accounting.py: per-stream busy / gap accounting, the overlap matrix and the phase split of a step window."""
    db = tmp_path / "bench_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, stream_id integer)")
    rows, t = [], 0
    for step in range(4):
        rows.append(("up::mse_partial_kernel(float const*)", t, t + 5000, 0)); t += 7000
        for _ in range(10):                                            # backward: data gradient on main, weight gradient beside it
            rows.append(("void up::glds::igemm_glds32_kernel<64, 64>(up::IgemmArgs)", t, t + 100000, 0))
            rows.append(("void up::glds::wgrad_glds32_kernel<128, 128>(up::WgradArgs)", t + 20000, t + 150000, 1))
            t += 102000
            rows.append(("void up::bn_bwd_apply_rows_kernel<float>(...)", t, t + 30000, 0)); t += 50000
        rows.append(("void at::native::multi_tensor_apply_kernel<FusedAdamMathFunctor>(...)", t, t + 40000, 0)); t += 100000
        for _ in range(10):                                            # forward: main stream alone
            rows.append(("void up::glds::igemm_glds32_kernel<64, 64>(up::IgemmArgs)", t, t + 90000, 0)); t += 91000
    con.executemany("insert into kernels values (?, ?, ?, ?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wall_accounting.py"), str(db)], capture_output=True, text=True,
                         check=True).stdout
    assert "2 step windows" in out and "main stream = 0; streams seen: [0, 1]" in out
    bwd = [ln for ln in out.splitlines() if ln.strip().startswith("backward:")][0]
    fwd = [ln for ln in out.splitlines() if ln.strip().startswith("forward:")][0]
    assert "side   1.300" in bwd and "mfma   1.000" in bwd, bwd          # 10 x 130 us beside 10 x 100 us of MFMA on main
    assert "mfma   0.900" in fwd and "side   0.000" in fwd, fwd


def test_build_stamp_names_sources_and_binary(tmp_path, monkeypatch):
    """build.library_is_current(): the stamp must name the tree's sources AND the sha256 of the binary next to it."""
    sys.path.insert(0, ROOT)
    from unipose_amd import build
    so, stamp = tmp_path / "lib.so", tmp_path / "lib.so.stamp"
    monkeypatch.setattr(build, "OUT", str(so))
    monkeypatch.setattr(build, "STAMP", str(stamp))
    so.write_bytes(b"\x7fELF binary one")
    stamp.write_text(build.source_hash() + " " + build.binary_hash() + "\n")
    assert build.library_is_current()
    so.write_bytes(b"\x7fELF another binary")                            # swapped in after the build: same stamp, other bits
    assert not build.library_is_current()
    stamp.write_text(build.source_hash() + "\n")                         # a round-5 stamp (sources only) no longer passes
    assert not build.library_is_current()
    stamp.write_text("0" * 64 + " " + build.binary_hash() + "\n")         # right binary, other sources
    assert not build.library_is_current()
