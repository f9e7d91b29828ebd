"""Per-phase shader-clock totals of k_search_rows (thread 0 of every block; -DWOL_PHASE_STAMPS build, libwol_prof.so) on the
500 000-water cube and on slabs with vacuum: where do empty tiles spend their time?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from waterorderlib_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "waterorderlib_amd", "libwol_prof.so")
import torch
from waterorderlib_amd import engine, synth
n, F = 500000, 32
L = synth.box_length(n)
base, _ = synth.lattice_sites(n, seed=2026, box=np.array([L, L, L]))
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((F, n, 3), generator=g, device="cuda", dtype=torch.float64)
b = torch.tensor([L, L, L], device="cuda", dtype=torch.float64)
xw = (x - b * torch.floor(x / b)).to(torch.float32).contiguous()
names = {0: "tail + barrier + decode", 1: "tables (cell_start, cnt4, scans)", 4: "placement", 5: "sweep"}
for name, zl in (("cube", 1.0), ("slab 1.5 L", 1.5), ("slab 3 L", 3.0)):
    ctx = _lib.Context(0)
    boxes = np.tile(np.array([L, L, zl * L])[None], (F, 1))
    engine.tet_order_traj(xw, boxes, ctx=ctx); torch.cuda.synchronize()
    ctx.counter("stamp0z")
    engine.tet_order_traj(xw, boxes, ctx=ctx); torch.cuda.synchronize()
    vals = {i: ctx.counter("stamp%d" % i) for i in names}
    tot = float(sum(vals.values()))
    print(name, "  ".join("%s %.1f %% (%.2e)" % (names[i], 100.0 * vals[i] / tot, vals[i]) for i in names))
    ctx.close()
