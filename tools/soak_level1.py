"""Randomized parity of the waterlib-level functions built next to the q path: nearneighbors masks, hydrogen
bonds, pair-distance histograms / RDFs, water orientation, Willard-Chandler density -- integer outputs bit for bit.
    python tools/soak_level1.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import waterlib as wl
import wol_oracle as oracle
oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(13000 + seed)
    box = rng.uniform(13.6, 50.0, size=3)
    n = int(min(max(0.0334 * np.prod(box), 150), 3000))
    O = rng.random((n, 3)) * box
    if seed % 2: O[: n // 4] = rng.random(3) * box + rng.normal(0.0, 3.0, size=(n // 4, 3))
    u = rng.normal(size=(n, 2, 3)); u /= np.linalg.norm(u, axis=2, keepdims=True)
    H = (O[:, None, :] + 0.9572 * u).reshape(-1, 3)
    don = np.repeat(O, 2, axis=0)
    shift = rng.integers(-1, 2, size=(n, 1, 3)) * box
    O2 = O + shift[:, 0]; H2 = (H.reshape(n, 2, 3) + shift).reshape(-1, 3)
    fails = []
    lo, hi = [(0.0, 3.5), (2.5, 4.2), (0.0, 6.0)][seed % 3]
    sub = O2[:: 1 + seed % 4]
    if not np.array_equal(wl.nearneighbors(sub, O2, box, lo, hi), oracle.nearneighbors(sub, O2, box, lo, hi)): fails.append("nearneighbors")
    if not np.array_equal(wl.allnearneighbors(O2, box, lo, hi), oracle.allnearneighbors(O2, box, lo, hi)): fails.append("allnearneighbors")
    dc, ac = [(3.5, 120.0), (3.0, 150.0), (3.5, 30.0), (4.0, 179.0)][seed % 4]
    acc = O2[: min(n, 1200)]
    if not np.array_equal(wl.generalhbonds(acc, np.repeat(O2, 2, axis=0), H2, box, dc, ac), oracle.generalhbonds(acc, np.repeat(O2, 2, axis=0), H2, box, dc, ac)): fails.append("generalhbonds")
    bw, nb = [(0.05, 200), (0.01, 1000), (0.1, 90)][seed % 3]
    if not np.array_equal(wl.pairdistancehistogram(O2, H2, bw, nb, box), oracle.pairdistancehistogram(O2, H2, bw, nb, box)): fails.append("pairdistancehistogram")
    if not np.allclose(wl.radialdistsame(O2, bw, nb, 0.0334, box), oracle.radialdistsame(O2, bw, nb, 0.0334, box), rtol=1e-13, atol=0): fails.append("radialdistsame")
    if not np.allclose(wl.radialdist(O2, H2, bw, nb, 0.0668, box), oracle.radialdist(O2, H2, bw, nb, 0.0668, box), rtol=1e-13, atol=0): fails.append("radialdist")
    rv = rng.normal(size=3)
    a1, p1 = wl.watorient(O2, H2, rv, box); a2, p2 = oracle.watorient(O2, H2, rv, box)
    if not (np.allclose(a1, a2, rtol=0, atol=1e-9) and np.allclose(p1, p2, rtol=0, atol=1e-9)): fails.append("watorient")
    if not np.array_equal(wl.watohvec(O2, H2, box), oracle.watohvec(O2, H2, box)): fails.append("watohvec")
    pts = rng.random((500, 3)) * box
    v1, g1 = wl.willarddensitypoints(O2, pts, box, 2.4); v2, g2 = oracle.willarddensitypoints(O2, pts, box, 2.4)
    if not (np.allclose(v1, v2, rtol=1e-12, atol=1e-300) and np.allclose(g1, g2, rtol=1e-10, atol=1e-16)): fails.append("willarddensitypoints")
    if fails:
        bad += 1
        print("MISMATCH seed %d: %s" % (seed, ", ".join(fails)))
    print("seed %3d n %5d %s" % (seed, n, "ok" if bad == 0 else ""), flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
