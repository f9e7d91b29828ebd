"""PCIe-inclusive rate of the trajectory entry point with HOST frames (WOL_MEM_HOST): what a caller that keeps its
trajectory in numpy arrays sees -- TIP4P-interleaved fp64 frames, by packing mode, packing threads and frames per batch.
bench.py's `value` is measured with frames resident in HBM; this is the other number.
    python tools/run_host_traj.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waterorderlib_amd import _lib, engine, synth
n, F, lw = 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4
base, box = synth.lattice_sites(n, seed=2026)
rng = np.random.default_rng(0)
O = base[None, :, :] + 0.3 * rng.standard_normal((F, n, 3))
O -= box * np.floor(O / box)
host = torch.zeros((F, n * lw, 3), dtype=torch.float64).pin_memory()
host[:, 0::lw] = torch.from_numpy(O)
arr = host.numpy()
sp = host.to(torch.float32).to(torch.float64).pin_memory().numpy()
boxes = np.tile(box[None, :], (F, 1))
ctx = _lib.Context(0)
print("host cores visible: %d" % len(os.sched_getaffinity(0)))
def run(a, pack, threads, fpb=0):
    ctx.set_option("host_pack", pack); ctx.set_option("host_threads", threads); ctx.set_option("frames_per_batch", fpb)
    engine.tet_order_traj(a[:64], boxes[:64], len_wat=lw, ctx=ctx)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); res = engine.tet_order_traj(a, boxes, len_wat=lw, ctx=ctx); best = min(best, time.perf_counter() - t0)
    return F / best, ctx.counter("h2d_bytes") / best / 1e9, float(np.mean(res.mean))
print("whole frames over the link          : %8.0f frames/s  %5.1f GB/s" % run(arr, 0, 0)[:2])
for th in (1, 2, 4, 8, 16, 32):
    print("packed O rows, fp64, %2d threads     : %8.0f frames/s  %5.1f GB/s" % ((th,) + run(arr, 1, th)[:2]), flush=True)
for fpb in (16, 32, 64, 128):
    print("packed O rows, fp64, 16 thr, batch %3d: %8.0f frames/s  %5.1f GB/s" % ((fpb,) + run(arr, 1, 16, fpb)[:2]), flush=True)
for th in (8, 16, 32):
    print("single-precision values, %2d threads : %8.0f frames/s  %5.1f GB/s" % ((th,) + run(sp, 2, th)[:2]), flush=True)
