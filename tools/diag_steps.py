"""Diagnostic: per-step GPU time (torch events) of consecutive trajectory steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
import bench
n, F, S = 100000, 256, 10
dev = torch.device("cuda", 0)
base, box = synth.lattice_sites(n, seed=2026)
base_dev = torch.from_numpy(base).to(dev)
boxes = np.tile(box[None, :], (F, 1))
ctx = _lib.Context(0)
batches = [bench.make_frames(torch, base_dev, box, F, s, 0.3, torch.float64) for s in range(S)]
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(S + 1)]
ev[0].record()
for s in range(S):
    engine.tet_order_traj(batches[s], boxes, ctx=ctx)
    ev[s + 1].record()
torch.cuda.synchronize()
print("per-step ms:", ["%.2f" % ev[s].elapsed_time(ev[s + 1]) for s in range(S)])
