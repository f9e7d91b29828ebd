"""a sanitizer build of the host-side code.  The CPU oracle (oracle/wol_oracle.c, test infrastructure)
is compiled with AddressSanitizer + UndefinedBehaviorSanitizer together with a small driver that runs its entry points on
a golden frame of the reference; the run must be clean and its outputs must equal the regular build's bit for bit (and the
reference's golden q).  GPU sanitizers are not available on this pool; device code is covered by the parity tests."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden, ROOT


@pytest.fixture(scope="module")
def asan_binary(tmp_path_factory):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    out = str(tmp_path_factory.mktemp("asan") / "asan_driver")
    cmd = [gcc, "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", os.path.join(ROOT, "oracle", "wol_oracle.c"), os.path.join(ROOT, "oracle", "asan_driver.c"),
           "-o", out, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this gcc cannot link the sanitizer runtimes: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("case", ["ice512", "sparse_clusters", "tiny_box"])
def test_oracle_is_clean_under_asan_and_ubsan(asan_binary, oracle, tmp_path, case):
    g = golden(case)
    pos = np.ascontiguousarray(g["pos"], dtype=np.float64)
    box = np.ascontiguousarray(g["box"], dtype=np.float64).reshape(-1)[:3]
    sub = np.ascontiguousarray(pos[:min(len(pos), 200)])
    n, ns = len(pos), len(sub)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        fh.write(np.array([n, ns], dtype=np.int32).tobytes()); fh.write(box.tobytes()); fh.write(pos.tobytes()); fh.write(sub.tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([asan_binary, fin, fout], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    out = np.fromfile(fout, dtype=np.float64)
    # the regular oracle build on the same inputs, bit for bit
    q_lit = oracle.getOrderParamq(sub, pos, box)
    q_c, idx, dist, ang = oracle.getOrderParamq(sub, pos, box, full=True, cells=True)
    o = 0
    assert np.array_equal(out[o:o + ns], q_lit, equal_nan=True); o += ns
    assert np.array_equal(out[o:o + ns], q_c, equal_nan=True); o += ns
    assert np.array_equal(out[o:o + 4 * ns], idx.astype(np.float64).ravel()); o += 4 * ns
    assert np.array_equal(out[o:o + 4 * ns], dist.ravel(), equal_nan=True); o += 4 * ns
    assert np.array_equal(out[o:o + 6 * ns], ang.ravel(), equal_nan=True); o += 6 * ns
    angs, _ = oracle.getCosAngs(sub, pos, box)
    assert int(out[o]) == len(angs); o += 1
    assert np.array_equal(out[o:o + len(angs)], angs); o += len(angs)
    lsi_vals, _ = oracle.getLSI(sub, pos, box)
    lsi_full = out[o:o + ns]; o += ns
    assert np.array_equal(lsi_full[~np.isnan(lsi_full)], lsi_vals)
    assert out[o] == oracle.nearneighbors(sub, pos, box, 0.0, 3.5).sum(); o += 1
    assert out[o] == oracle.allnearneighbors(pos, box, 0.0, 3.5).sum(); o += 1
    assert np.array_equal(out[o:o + 60], oracle.radialdistsame(pos, 0.1, 60, 1.0, box)); o += 60
    donh = pos + np.array([0.96, 0.0, 0.0])
    assert out[o] == oracle.generalhbonds(sub, pos, donh, box, 3.5, 120.0).sum(); o += 1
    m, v, qh, dh = oracle.tet_order_traj(np.stack([pos, pos]), np.stack([box, box]))
    assert np.array_equal(out[o:o + 2], m) and np.array_equal(out[o + 2:o + 4], v) and np.array_equal(out[o + 4:o + 504], qh.astype(np.float64)); o += 504
    # and the reference's own golden q for the frame (the oracle is pinned to it elsewhere; here: the sanitized build too)
    if "sub" not in g.files and float(g["low"]) == 0.0 and float(g["high"]) == 10.0:
        assert np.array_equal(out[:ns], g["q"][:ns], equal_nan=True)
