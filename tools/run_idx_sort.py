"""A/B of the index-record form of the sort (option idx_sort: 16-byte records, the certification reads fp64 from the source
frames) against the classic 32-byte records, on the bench box with bare O rows and with TIP4P-interleaved frames resident
in HBM; per-kernel times through the context's timer are not split here: run under tools/kstats.sh for those."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
n, F = 100000, 256
base, box = synth.lattice_sites(n, seed=2026)
base = base[np.random.default_rng(2026).permutation(n)]
g = torch.Generator(device="cuda"); g.manual_seed(1)
o = (torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((F, n, 3), generator=g, device="cuda", dtype=torch.float64)).contiguous()
boxes = np.tile(box[None], (F, 1))
tip4p = torch.full((F, n * 4, 3), 1.0e6, device="cuda", dtype=torch.float64)
tip4p[:, ::4] = o
for name, fr, lw in (("bare O rows", o, 1), ("TIP4P-interleaved", tip4p, 4)):
    for mode in (0, 1, 0, 1):
        ctx = _lib.Context(0); ctx.set_option("idx_sort", mode)
        engine.tet_order_traj(fr, boxes, len_wat=lw, ctx=ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): r = engine.tet_order_traj(fr, boxes, len_wat=lw, ctx=ctx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-18s idx_sort %d  record bytes %2d  %.3f ms per %d frames  mean q %.12f" % (name, mode, ctx.counter("record_bytes"), 1e3 * dt, F, float(r.mean.mean())), flush=True)
        ctx.close()
