"""Randomized parity sweep of the TRAJECTORY entry (wol_q_traj): frame counts, water counts, per-frame boxes (NPT),
interleaved waters (len_wat), host / device input, with and without per-water q, shell cuts, both proposal kernels --
per-frame mean / var and the two histograms against the oracle's tet_order_traj (histograms bit for bit).
    python tools/soak_traj.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from waterorderlib_amd import _lib, engine
import wol_oracle
wol_oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    big = seed >= 100                                     # seeds from 100: larger boxes, fewer frames
    T = int(rng.choice([1, 2, 3, 7, 16] if big else [1, 2, 3, 7, 16, 33, 70]))
    box0 = rng.uniform(14.0, 110.0 if big else 70.0, size=3)
    if seed % 5 == 0: box0[rng.integers(3)] = rng.uniform(13.7, 15.0)
    rho = rng.choice([0.0334, 0.0334, 0.025, 0.045, 0.012])
    n = int(min(max(rho * np.prod(box0), 200), 40000 if big else 12000))
    if seed >= 10000:                                     # seeds from 10000: many frames of 25-60 k waters (the single-pass
        T = int(rng.integers(64, 81))                     # first sort level, full batches)
        n = int(rng.integers(25000, 60000)); box0 = box0 * (n / (rho * np.prod(box0))) ** (1.0 / 3.0)
    npt = seed % 3 != 0
    boxes = box0 * (1.0 + (rng.uniform(-0.04, 0.04, size=(T, 3)) if npt else np.zeros((T, 3))))
    len_wat = int(rng.choice([1, 1, 3, 4]))
    frames = np.zeros((T, n * len_wat, 3))
    base = rng.random((n, 3))
    for f in range(T):
        p = (base + rng.normal(0.0, 0.01, size=(n, 3))) * boxes[f]
        if seed % 4 == 1:   # a dense droplet in every frame
            k = n // 4
            p[:k] = boxes[f] * 0.5 + rng.normal(0.0, 3.0, size=(k, 3))
        p += rng.integers(-1, 2, size=p.shape) * boxes[f]
        frames[f, ::len_wat] = p
        if len_wat > 1: frames[f, 1::len_wat] = 1.0e6     # other atoms of the molecule: must never be read as waters
    hi = float(rng.choice([10.0, 10.0, 3.5, 5.0]))
    lo = float(rng.choice([0.0, 0.0, 2.7]))
    want_q = bool(seed % 2)
    f32 = seed % 7 == 3                                   # float32 storage: promoted exactly, same fp64 arithmetic
    if f32: frames = frames.astype(np.float32)
    if seed % 11 == 5: frames = frames + (1000.0 * boxes[:, None, :]).astype(frames.dtype)    # far outside the box
    oxy = np.ascontiguousarray(frames[:, ::len_wat]).astype(np.float64)
    m_o, v_o, qh_o, dh_o = wol_oracle.tet_order_traj(oxy, boxes, lowCut=lo, highCut=hi, cells=True)
    for rows in (1, 0):
        for dev in (False, True):
            c = _lib.Context(0); c.set_option("search_rows", rows)
            src = torch.from_numpy(frames).cuda() if dev else frames
            r = engine.tet_order_traj(src, boxes, lowCut=lo, highCut=hi, len_wat=len_wat, want_q=want_q, ctx=c)
            g = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
            ok = (np.array_equal(g(r.qhist), qh_o) and np.array_equal(g(r.dhist), dh_o) and
                  np.allclose(g(r.mean), m_o, rtol=0, atol=1e-13) and np.allclose(g(r.var), v_o, rtol=0, atol=1e-13))
            if want_q and ok:
                # per-water q agrees with the frame means it was reduced to
                ok = np.allclose(g(r.q).astype(np.float64).mean(axis=1), m_o, rtol=0, atol=1e-6 if f32 else 1e-12)
            if not ok:
                bad += 1
                print("MISMATCH seed %d rows %d dev %d: qhist %d dhist %d mean %.2e" % (
                    seed, rows, dev, int(np.abs(g(r.qhist) - qh_o).sum()), int(np.abs(g(r.dhist) - dh_o).sum()),
                    float(np.abs(g(r.mean) - m_o).max())))
            c.close()
    print("seed %3d T %3d n %5d len_wat %d npt %d f32 %d cuts %.1f-%.1f q %d %s" % (seed, T, n, len_wat, npt, f32, lo, hi, want_q,
                                                                          "ok" if bad == 0 else ""), flush=True)
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
