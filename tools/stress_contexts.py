"""Context life-cycle stress: many contexts created, used on different paths and destroyed; two contexts interleaved on one
device; device memory must come back (no leaked buffers / streams / events).    python tools/stress_contexts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from waterorderlib_amd import _lib, engine, synth, water_properties as wp

xyz, boxes = synth.water_frames(4000, 80, seed=3)
dev = torch.from_numpy(xyz).cuda()
ref = engine.tet_order_traj(dev, boxes)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for it in range(60):
    c = _lib.Context(0)
    if it % 3 == 0: c.set_option("overlap", 1); c.set_option("stages", 2 + it % 2)
    if it % 4 == 1: c.set_option("seg_sort", 2)
    if it % 5 == 2: c.set_option("frames_per_batch", 16)
    r = engine.tet_order_traj(dev if it % 2 else xyz, boxes, ctx=c)
    qh = r.qhist.cpu().numpy() if it % 2 else r.qhist
    assert np.array_equal(qh, ref.qhist.cpu().numpy()), it
    if it % 6 == 0: wp.getLSI(xyz[0], xyz[0], boxes[0], ctx=c)
    if it % 7 == 0: wp.getCosAngs(xyz[1], xyz[1], boxes[1], ctx=c)
    c.close()
a, b = _lib.Context(0), _lib.Context(0)
for it in range(20):
    ra = engine.tet_order_traj(dev[:40], boxes[:40], ctx=a)
    rb = engine.tet_order_traj(dev[40:], boxes[40:], ctx=b)
    assert torch.equal(ra.qhist + rb.qhist, ref.qhist)
a.close(); b.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("device memory free before / after: %.1f / %.1f MB" % (free0 / 1e6, free1 / 1e6))
assert free0 - free1 < 64e6, "device memory not returned"
print("ok")
