"""NPT-like trajectory (a different box every frame) against the constant-volume one: frames/s of the resident-frames
path.  The tile tables depend on the cell counts only and are memoised; per frame the host recomputes a few scalars."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
import bench
n, F, S = 100000, 256, 6
dev = torch.device("cuda", 0)
base, box = synth.lattice_sites(n, seed=2026)
base = base[np.random.default_rng(2026).permutation(n)]
base_dev = torch.from_numpy(base).to(dev)
rng = np.random.default_rng(7)
for name, amp in (("constant box", 0.0), ("box breathing +-0.3 % per frame", 0.003), ("box breathing +-2 % (cell counts change)", 0.02)):
    scale = 1.0 + amp * (2.0 * rng.random((S, F, 1)) - 1.0)
    batches, boxes = [], []
    for s in range(S):
        fr = bench.make_frames(torch, base_dev, box, F, 100 + s, 0.3, torch.float64)
        sc = torch.from_numpy(scale[s]).to(dev).unsqueeze(-1)
        batches.append((fr * sc).contiguous())
        boxes.append(box[None, :] * scale[s])
    ctx = _lib.Context(0)
    engine.tet_order_traj(batches[0], boxes[0], ctx=ctx); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(S):
        res = engine.tet_order_traj(batches[s], boxes[s], ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-45s %8.0f frames/s  (%.3f ms per %d frames), <q> = %.6f" % (name, S * F / dt, 1e3 * dt / S, F, float(res.mean.mean().item())))
    ctx.close()
