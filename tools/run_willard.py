"""SURVEY 8f row f-1 on BASELINE config 5's surface part: Willard-Chandler density field of a 500k-water slab
on the 80^3 grid surface_library.densityGrid builds, one MI355X.  Checks a sample of grid points against the
CPU oracle and times the call (host buffers in, host buffers out) and its kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import _lib, synth, waterlib as wl, surface_library as sl
import wol_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
xyz, boxes = synth.slab_frames(n, 1, seed=5, n_vapour=100)
pos, box = xyz[0], boxes[0]
print("slab: %d waters, box %s" % (len(pos), box))
heavy = pos[:200]
ctx = _lib.default_context()
ctx.set_option("time_kernels", 1)
dens, norms, spans, sp = sl.densityField(heavy, pos, box[None, :])
ctx.kernel_time(reset=True)
for rep in range(3):
    t0 = time.perf_counter()
    dens, norms, spans, sp = sl.densityField(heavy, pos, box[None, :])
    dt = time.perf_counter() - t0
    ms, nk = ctx.kernel_time(reset=True)
    print("densityField 80^3: %.2f ms wall (upload + sort + kernel + download), k_willard %.3f ms" % (dt * 1e3, ms / max(nk, 1)))
rng = np.random.default_rng(0)
ii = rng.integers(0, 80, size=(400, 3))
pts = np.stack([spans[0][0, ii[:, 0]], spans[1][0, ii[:, 1]], spans[2][0, ii[:, 2]]], axis=1)
wol_oracle.set_threads(16)
t0 = time.perf_counter()
ov, on = wol_oracle.willarddensitypoints(pos, pts, box, 2.4)
dto = time.perf_counter() - t0
gv = dens[ii[:, 0], ii[:, 1], ii[:, 2]]; gn = norms[ii[:, 0], ii[:, 1], ii[:, 2]]
print("max rel density error over 400 sampled grid points: %.2e; max normal error %.2e" %
      (np.nanmax(np.abs(gv - ov) / np.maximum(np.abs(ov), 1e-300)), np.nanmax(np.abs(gn - on))))
assert np.allclose(gv, ov, rtol=1e-12, atol=1e-18) and np.allclose(gn, on, rtol=0, atol=1e-9, equal_nan=True)
print("CPU oracle (O(G N) as the reference, 16 threads): %.2f s for 400 points -> %.0f s for 80^3" % (dto, dto * 512000 / 400))
