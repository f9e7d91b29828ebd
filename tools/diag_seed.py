"""One seed of tools/soak_random.py under the microscope: which waters differ from the oracle, and how close their
4th / 5th neighbour distances are.    python tools/diag_seed.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import _lib, water_properties as wp
import wol_oracle
wol_oracle.set_threads(16)
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
box = rng.uniform(13.6, 120.0, size=3)
if seed % 4 == 0: box[rng.integers(3)] = rng.uniform(13.6, 15.0)
rho = rng.choice([0.0334, 0.0334, 0.02, 0.05, 0.01, 0.07])
n = int(min(max(rho * np.prod(box), 300), 20000))
pos = rng.random((n, 3)) * box
if seed % 2 == 1:
    k = n // 3
    centres = rng.random((5, 3)) * box
    pos[:k] = centres[rng.integers(5, size=k)] + rng.normal(0.0, 2.5, size=(k, 3))
pos += rng.integers(-1, 2, size=pos.shape) * box
q_o, idx_o, dist_o, ang_o = wol_oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
for rows, dyn in ((1, 1), (1, 0), (0, 0)):
    c = _lib.Context(0); c.set_option("path", 3); c.set_option("search_rows", rows); c.set_option("dyn_tiles", dyn)
    q, idx, dist, ang = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
    bad = np.where((idx != idx_o).any(axis=1))[0]
    print("rows %d dyn %d: %d waters differ, block %d" % (rows, dyn, len(bad), c.counter("rows_block")))
    for w in bad[:6]:
        d = pos - pos[w]; d -= box * np.round(d / box)
        r = np.sqrt((d ** 2).sum(axis=1)); r[w] = np.inf
        o = np.argsort(r)[:6]
        print("   water", w, "pos", pos[w], "\n     oracle idx", idx_o[w], "dist", dist_o[w], "\n     gpu    idx", idx[w], "dist", dist[w],
              "\n     brute  idx", o, "dist", r[o])
    c.close()
