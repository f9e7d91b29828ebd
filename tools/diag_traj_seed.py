"""One seed of tools/soak_traj.py: per-water q of the trajectory mode against the oracle.  python tools/diag_traj_seed.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import _lib, engine, water_properties as wp
import wol_oracle
wol_oracle.set_threads(16)
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
big = seed >= 100
T = int(rng.choice([1, 2, 3, 7, 16] if big else [1, 2, 3, 7, 16, 33, 70]))
box0 = rng.uniform(14.0, 110.0 if big else 70.0, size=3)
if seed % 5 == 0: box0[rng.integers(3)] = rng.uniform(13.7, 15.0)
rho = rng.choice([0.0334, 0.0334, 0.025, 0.045, 0.012])
n = int(min(max(rho * np.prod(box0), 200), 40000 if big else 12000))
npt = seed % 3 != 0
boxes = box0 * (1.0 + (rng.uniform(-0.04, 0.04, size=(T, 3)) if npt else np.zeros((T, 3))))
len_wat = int(rng.choice([1, 1, 3, 4]))
frames = np.zeros((T, n * len_wat, 3))
base = rng.random((n, 3))
for f in range(T):
    p = (base + rng.normal(0.0, 0.01, size=(n, 3))) * boxes[f]
    if seed % 4 == 1:
        k = n // 4
        p[:k] = boxes[f] * 0.5 + rng.normal(0.0, 3.0, size=(k, 3))
    p += rng.integers(-1, 2, size=p.shape) * boxes[f]
    frames[f, ::len_wat] = p
    if len_wat > 1: frames[f, 1::len_wat] = 1.0e6
oxy = np.ascontiguousarray(frames[:, ::len_wat])
m_o, v_o, qh_o, dh_o = wol_oracle.tet_order_traj(oxy, boxes, cells=True)
for rows in (1, 0):
    c = _lib.Context(0); c.set_option("search_rows", rows)
    r = engine.tet_order_traj(frames, boxes, len_wat=len_wat, want_q=True, ctx=c)
    print("rows", rows, "mean diff", np.abs(r.mean - m_o), "var diff", np.abs(r.var - v_o))
    for f in range(T):
        qo, io, do, ao = wol_oracle.getOrderParamq(oxy[f], oxy[f], boxes[f], full=True, cells=True)
        dq = np.abs(r.q[f] - qo)
        w = np.argsort(dq)[::-1][:4]
        print("   frame", f, "max |dq|", dq.max(), "sum dq", (r.q[f] - qo).sum(), "worst waters", w, "dq", dq[w], "q", qo[w], "\n      dist", do[w[0]], "ang", ao[w[0]])
    c.close()
