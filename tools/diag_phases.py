"""Diagnostic: per-phase cycle shares of k_search_fast from a -DWOL_PHASE_STAMPS build."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from waterorderlib_amd import _lib, build
prof = os.environ.get("WOL_PROF_LIB") or os.path.join(ROOT, "waterorderlib_amd", "libwol_prof.so")
_lib.LIB_PATH = prof
import torch
from waterorderlib_amd import engine, synth
import bench
n, F = 100000, 100
dev = torch.device("cuda", 0)
base, box = synth.lattice_sites(n, seed=2026)
boxes = np.tile(box[None, :], (F, 1))
fr = bench.make_frames(torch, torch.from_numpy(base).to(dev), box, F, 1, 0.3, torch.float64)
ctx = _lib.Context(0)
if len(sys.argv) > 1:
    ctx.set_option("cell_width", float(sys.argv[1]))
rows = os.environ.get("WOL_DIAG_ROWS", "1") == "1"
ctx.set_option("search_rows", 1 if rows else 0)
for rep in range(2):
    engine.tet_order_traj(fr, boxes, ctx=ctx)
    torch.cuda.synchronize()
    names = (["tail+decode", "rows/pieces", "pass A count", "scan", "pass B place", "sweep"] if rows else
             ["tail+decode", "phase1 counts", "phase2 scan", "phase3+4 stage", "query sweep"])
    vals = [ctx.counter("stamp%d" % i) for i in range(len(names))]
    ctx.counter("stamp0z")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): engine.tet_order_traj(fr, boxes, ctx=ctx)
torch.cuda.synchronize(); print("step ms: %.3f" % (1e3 * (time.perf_counter() - t0) / 5))
tot = sum(vals)
for nme, v in zip(names, vals):
    print("%-16s %6.1f%%  %.0f cycles/tile" % (nme, 100.0 * v / tot, v / (float(os.environ.get("TILES","2240")) * F)))
