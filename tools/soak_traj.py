"""Randomized parity sweep of the TRAJECTORY entry (wol_q_traj): frame counts, water counts, per-frame boxes (NPT),
interleaved waters (len_wat), host / device input, with and without per-water q, shell cuts, both proposal kernels --
per-frame mean / var and the two histograms against the oracle's tet_order_traj (histograms bit for bit).
    python tools/soak_traj.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from waterorderlib_amd import _lib, engine
import wol_oracle
import soak_gen
wol_oracle.set_threads(16)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    case = soak_gen.traj_case(seed)
    frames, boxes, len_wat, lo, hi, want_q, f32 = (case[k] for k in ("frames", "boxes", "len_wat", "lo", "hi", "want_q", "f32"))
    T, n, npt = case["T"], case["n"], case["npt"]
    oxy = np.ascontiguousarray(frames[:, ::len_wat]).astype(np.float64)
    m_o, v_o, qh_o, dh_o = wol_oracle.tet_order_traj(oxy, boxes, lowCut=lo, highCut=hi, cells=True)
    for rows in (1, 0):
        for dev in (False, True):
            c = _lib.Context(0); c.set_option("search_rows", rows)
            for kv in os.environ.get("WOL_SOAK_OPTIONS", "").split(","):   # e.g. WOL_SOAK_OPTIONS=idx_sort=1,seg_sort=2
                if "=" in kv: c.set_option(kv.split("=")[0], float(kv.split("=")[1]))
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
