"""Parity on inputs full of exact and near ties: perfect lattices (simple cubic: six equidistant neighbours, fcc: twelve,
diamond: four + twelve), with and without a jitter at the level of the last bits, whole-box shifts, and twins of random
waters.  The 4th / 5th neighbour tie sends (nearly) every water to the exact kernel, whose choice among equals must be
the reference's (the oracle's).      python tools/soak_lattice.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from waterorderlib_amd import _lib, water_properties as wp
import wol_oracle
import soak_gen
wol_oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for seed in range(first, first + count):
    pos, box, kind, jitter = soak_gen.lattice_case(seed)
    a = 0.0
    n = len(pos)
    q_o, idx_o, dist_o, ang_o = wol_oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    for rows in (1, 0):
        c = _lib.Context(0); c.set_option("path", 3); c.set_option("search_rows", rows)
        q, idx, dist, ang = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
        slow = c.counter("slow_waters")
        ok = (np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o, equal_nan=True) and
              np.allclose(ang, ang_o, rtol=0, atol=1e-9, equal_nan=True) and np.allclose(q, q_o, rtol=0, atol=1e-12))
        if not ok:
            bad += 1
            print("MISMATCH seed %d rows %d: %d waters differ in idx, %d in dist, max |dq| %.2e" % (
                seed, rows, int((idx != idx_o).any(axis=1).sum()), int((dist != dist_o).any(axis=1).sum()), float(np.abs(q - q_o).max())))
        c.close()
    print("seed %3d %-7s n %5d a %.3f jitter %.0e exact-path %.1f %% %s" % (seed, kind, n, a, jitter, 100.0 * slow / n, "ok" if bad == 0 else ""), flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
