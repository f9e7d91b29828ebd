"""PCIe-inclusive rate of the trajectory entry point with HOST frames (WOL_MEM_HOST): what a caller that keeps its
trajectory in numpy arrays sees.  bench.py's `value` is measured with frames resident in HBM; this is the other number."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waterorderlib_amd import engine, synth
n, F = 100000, 256
base, box = synth.lattice_sites(n, seed=2026)
rng = np.random.default_rng(0)
frames = base[None, :, :] + 0.3 * rng.standard_normal((F, n, 3))
frames -= box * np.floor(frames / box)
boxes = np.tile(box[None, :], (F, 1))
pinned = torch.from_numpy(frames).pin_memory().numpy()
for name, arr in (("pageable numpy", frames), ("pinned host", pinned)):
    engine.tet_order_traj(arr, boxes)
    t0 = time.perf_counter()
    for _ in range(3): res = engine.tet_order_traj(arr, boxes)
    dt = (time.perf_counter() - t0) / 3
    print("%-15s %d frames of %d waters: %.1f ms -> %.0f frames/s (%.1f GB/s over PCIe), <q> = %.6f" %
          (name, F, n, dt * 1e3, F / dt, frames.nbytes / dt / 1e9, float(np.mean(res.mean))))
