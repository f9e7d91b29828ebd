"""Randomized parity of the shell functions next to q: getCosAngs (three-body angles inside a shell) and getLSI,
against the oracle -- box shapes, densities, clustering, shell cuts, subsets, twins.
    python tools/soak_shell.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import water_properties as wp
import wol_oracle
wol_oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    box = rng.uniform(13.6, 60.0, size=3)
    rho = rng.choice([0.0334, 0.0334, 0.02, 0.045])
    n = int(min(max(rho * np.prod(box), 200), 8000))
    pos = rng.random((n, 3)) * box
    if seed % 2 == 1:
        k = n // 5
        pos[:k] = rng.random(3) * box + rng.normal(0.0, 4.0, size=(k, 3))
    if seed % 3 == 0:
        w = rng.permutation(n)[:20].reshape(10, 2)
        pos[w[:, 1]] = pos[w[:, 0]] + rng.normal(0.0, 1e-3, size=(10, 3))
    pos += rng.integers(-1, 2, size=pos.shape) * box
    sub = pos if seed % 4 else pos[::5] + 0.013
    lo, hi = [(0.0, 3.413), (2.6, 3.8), (0.0, 4.5), (1.0, 3.0)][seed % 4]
    ca, cn = wp.getCosAngs(sub, pos, box, lo, hi)
    ra, rn = wol_oracle.getCosAngs(sub, pos, box, lo, hi)
    ok = np.array_equal(cn, rn) and ca.shape == ra.shape and np.allclose(ca, ra, rtol=0, atol=1e-9)
    li, ln = wp.getLSI(sub, pos, box, 0.0, 3.7)
    ri, rn2 = wol_oracle.getLSI(sub, pos, box, 0.0, 3.7)
    ok2 = np.array_equal(ln, rn2) and li.shape == ri.shape and np.allclose(li, ri, rtol=0, atol=1e-12, equal_nan=True)
    if not (ok and ok2):
        bad += 1
        print("MISMATCH seed %d: cosangs %s (counts differ for %d) lsi %s (counts differ for %d)" % (
            seed, ok, int((cn != rn).sum()), ok2, int((ln != rn2).sum())))
    print("seed %3d n %5d angles %7d cuts %.1f-%.1f %s" % (seed, n, len(ca), lo, hi, "ok" if bad == 0 else ""), flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
