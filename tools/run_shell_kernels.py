"""Timing of the per-frame shell kernels (rows a10 / a11: getCosAngs, getLSI) on a 100k-water frame, host buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from waterorderlib_amd import synth, water_properties as wp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xyz, boxes = synth.water_frames(n, 1, seed=3)
pos, box = xyz[0], boxes[0]
for name, fn in (("getCosAngs", lambda: wp.getCosAngs(pos, pos, box)), ("getLSI", lambda: wp.getLSI(pos, pos, box)),
                 ("getOrderParamq", lambda: wp.getOrderParamq(pos, pos, box))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): out = fn()
    dt = (time.perf_counter() - t0) / 5
    size = len(out[0]) if isinstance(out, tuple) else len(out)
    print("%-15s %d waters: %.2f ms per frame (host to host), %d values" % (name, n, dt * 1e3, size))
