"""Diagnostic: tile geometry the host picks for the proposal kernel on the benchmark box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
for n in (10000, 100000, 500000):
    base, box = synth.lattice_sites(n, seed=2026)
    fr = torch.from_numpy(base[None].repeat(8, 0)).cuda()
    for rb in (512, 768):
        ctx = _lib.Context(0); ctx.set_option("rows_block", rb)
        engine.tet_order_traj(fr, np.tile(box[None], (8, 1)), ctx=ctx); torch.cuda.synchronize()
        sx, ry, nt = ctx.counter("tile_sx"), ctx.counter("tile_ry"), ctx.counter("tiles_per_frame")
        print("n %7d block %d: tile %d x %d cells, %d tiles per frame, %.1f waters per tile (%.0f %% of the lanes)" % (n, rb, sx, ry, nt, n / nt, 100.0 * n / nt / rb))
        ctx.close()
