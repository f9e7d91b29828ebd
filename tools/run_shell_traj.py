"""Times the trajectory entries of the other two frame loops (wol_cos_angs_traj, wol_lsi_traj) and the populations entry on
the 100 000-water box, frames resident in HBM, next to the per-frame calls they replace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
from waterorderlib_amd import water_properties as wp

n, F = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
base, box = synth.lattice_sites(n, seed=2026)
base = base[np.random.default_rng(2026).permutation(n)]
g = torch.Generator(device="cuda"); g.manual_seed(1)
fr = (torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((F, n, 3), generator=g, device="cuda", dtype=torch.float64)).contiguous()
b = torch.as_tensor(box, device="cuda"); fr = (fr - b * torch.floor(fr / b)).contiguous()
boxes = np.tile(box[None], (F, 1))
ctx = _lib.Context(0); ctx.set_option("time_kernels", 1)
for name, fn in (("cos_angs_traj", lambda: engine.cos_angs_traj(fr, boxes, ctx=ctx)), ("lsi_traj", lambda: engine.lsi_traj(fr, boxes, ctx=ctx)),
                 ("tet_order_traj", lambda: engine.tet_order_traj(fr, boxes, ctx=ctx))):
    fn(); torch.cuda.synchronize(); ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(3): r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ms, k = ctx.kernel_time(reset=True)
    print("%-16s %8.0f frames/s  %.3f ms per %d frames  (timed kernel %.3f ms per launch)" % (name, F / dt, 1e3 * dt, F, ms / max(k, 1)), flush=True)
r = engine.cos_angs_traj(fr, boxes, ctx=ctx)
print("angles per frame %d, tetrahedral %d" % tuple(int(x) for x in r.counts[0, 0].cpu()))
pos = fr[0].cpu().numpy()
for name, fn in (("getCosAngs", lambda: wp.getCosAngs(pos, pos, box)), ("getLSI", lambda: wp.getLSI(pos, pos, box))):
    fn(); t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print("%-16s per-frame call, host in / out: %.2f ms per frame = %.0f frames/s" % (name, 1e3 * dt, 1.0 / dt))
# populations: 2 000 waters of every frame
rng = np.random.default_rng(3)
idx = torch.from_numpy(rng.choice(n, 2000, replace=False)).cuda()
pop = fr[:, idx].reshape(-1, 3).contiguous()
off = np.arange(F + 1, dtype=np.int64) * 2000
engine.tet_order_traj_pops(fr, boxes, pop, off, 1, ctx=ctx); torch.cuda.synchronize()
t0 = time.perf_counter(); engine.tet_order_traj_pops(fr, boxes, pop, off, 1, ctx=ctx); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("tet_order_traj_pops (2000-water population per frame): %.0f frames/s" % (F / dt))
