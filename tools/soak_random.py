"""Wider randomized parity sweep than the test-suite's ten seeds: box shapes, densities, clustering, both proposal
kernels and whichever instantiation the host picks; every water's indices, distances, angles and q against the oracle.
    python tools/soak_random.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from waterorderlib_amd import _lib, water_properties as wp
import wol_oracle
import soak_gen
wol_oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    pos, box = soak_gen.random_case(seed)
    n = len(pos); rho = n / np.prod(box)
    q_o, idx_o, dist_o, ang_o = wol_oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    for rows in (1, 0):
        c = _lib.Context(0); c.set_option("path", 3); c.set_option("search_rows", rows)
        q, idx, dist, ang = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
        if rows == 1: blk = c.counter("rows_block"); slow = c.counter("slow_waters")
        ok = (np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o, equal_nan=True) and
              np.allclose(ang, ang_o, rtol=0, atol=1e-9, equal_nan=True) and np.allclose(q, q_o, rtol=0, atol=1e-12))
        if not ok:
            bad += 1
            print("MISMATCH seed %d rows %d: %d waters differ" % (seed, rows, int((idx != idx_o).any(axis=1).sum())))
        c.close()
    print("seed %3d n %5d box %s rho %.3f block %d exact-path %.2f %% %s" % (seed, n, np.round(box, 1), rho, blk, 100.0 * slow / n, "ok" if bad == 0 else ""))
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
