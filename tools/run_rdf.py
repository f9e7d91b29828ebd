"""O-O g(r) (rdfCalc's call: binwidth 0.1 A, 150 bins -> 15 A) of a 100k-water frame on one MI355X,
checked against the CPU oracle's O(N^2) loop on a subset of query waters and timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from waterorderlib_amd import _lib, synth, waterlib as wl
import wol_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xyz, boxes = synth.water_frames(n, 1, seed=7)
pos, box = xyz[0], boxes[0]
ctx = _lib.default_context(); ctx.set_option("time_kernels", 1)
res = {}
for tiled in (1, 0):
    ctx.set_option("pair_tiled", tiled)
    wl.radialdistsame(pos, 0.1, 150, 1.0, box); ctx.kernel_time(reset=True)
    for rep in range(3):
        t0 = time.perf_counter(); g = wl.radialdistsame(pos, 0.1, 150, 1.0, box); dt = time.perf_counter() - t0
        ms, nk = ctx.kernel_time(reset=True)
        print("radialdistsame %d waters, %s: %.2f ms wall (host buffers), pair kernel %.3f ms" %
              (n, "LDS-tiled kernel" if tiled else "lane-per-water walk", dt * 1e3, ms / max(nk, 1)))
    res[tiled] = g
assert np.array_equal(res[0], res[1])
print("g(r) of the two kernels identical; pairs inside 15 A: %.3e" % (n * 0.033456 * 4.18879 * 15.0 ** 3 / 2))
ctx.set_option("pair_tiled", 1)
wol_oracle.set_threads(16)
sub = pos[:2000]
t0 = time.perf_counter(); ho = wol_oracle.pairdistancehistogram(sub, pos, 0.1, 150, box); dto = time.perf_counter() - t0
hg = wl.pairdistancehistogram(sub, pos, 0.1, 150, box)
assert np.array_equal(ho, hg)
print("2000 x %d pair histogram identical to the oracle; oracle O(N^2) on 16 threads: %.2f s -> %.0f s for the full frame" % (n, dto, dto * n / 2000 / 2))
