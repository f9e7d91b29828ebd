"""Where does the proposal kernel's time go on the config-5 slab?  The same 500 000 waters as (a) the slab (box elongated
along z: a third of the cell layers are vacuum) and (b) a periodic cube (no vacuum), float32 storage, 32 frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
n, F = 500000, 32
L = synth.box_length(n)
base, _ = synth.lattice_sites(n, seed=2026, box=np.array([L, L, L]))
base = base[np.random.default_rng(1).permutation(n)]
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((F, n, 3), generator=g, device="cuda", dtype=torch.float64)
b = torch.tensor([L, L, L], device="cuda", dtype=torch.float64)
xw = (x - b * torch.floor(x / b)).to(torch.float32).contiguous()
for name, box, dyn, skip in (("cube (no vacuum)", np.array([L, L, L]), 1, 1), ("slab (box 1.5 L along z)", np.array([L, L, 1.5 * L]), 1, 1),
                             ("slab 1.5 L, empty tiles not skipped", np.array([L, L, 1.5 * L]), 1, 0), ("slab (box 3 L along z)", np.array([L, L, 3.0 * L]), 1, 1),
                             ("slab 3 L, empty tiles not skipped", np.array([L, L, 3.0 * L]), 1, 0), ("slab 3 L, static order, skipped", np.array([L, L, 3.0 * L]), 0, 1)):
    ctx = _lib.Context(0); ctx.set_option("time_kernels", 1); ctx.set_option("dyn_tiles", dyn)
    try: ctx.set_option("skip_empty_tiles", skip)
    except Exception: pass
    boxes = np.tile(box[None], (F, 1))
    engine.tet_order_traj(xw, boxes, ctx=ctx); torch.cuda.synchronize(); ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(3): engine.tet_order_traj(xw, boxes, ctx=ctx)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ms, k = ctx.kernel_time(reset=True)
    print("%-28s step %.3f ms  proposal kernel %.3f ms  level %d  tiles/frame %d (sx %d ry %d)  slow %.2e" % (
        name, 1e3 * dt, ms / max(k, 1), ctx.counter("density_level"), ctx.counter("tiles_per_frame"), ctx.counter("tile_sx"), ctx.counter("tile_ry"),
        ctx.counter("slow_waters") / (F * n)))
    ctx.close()
