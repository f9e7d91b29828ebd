"""Diagnostic: host-side enqueue cost of one trajectory step vs its GPU time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waterorderlib_amd import _lib, engine, synth
import bench

n, F = 100000, 100
dev = torch.device("cuda", 0)
base, box = synth.lattice_sites(n, seed=2026)
base_dev = torch.from_numpy(base).to(dev)
boxes = np.tile(box[None, :], (F, 1))
ctx = _lib.Context(0)
batches = [bench.make_frames(torch, base_dev, box, F, s, 0.3, torch.float64) for s in range(6)]
torch.cuda.synchronize()
for rep in range(2):
    ts = []
    t_all = time.perf_counter()
    for s in range(6):
        t0 = time.perf_counter()
        r = engine.tet_order_traj(batches[s], boxes, want_q=False, want_dhist=True, ctx=ctx)
        ts.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print("rep", rep, "enqueue per step (ms):", ["%.2f" % (1e3 * t) for t in ts], "enqueue total %.2f ms, with sync %.2f ms" % (1e3 * t_enq, 1e3 * t_tot))
# C-level only (no torch allocs): reuse output tensors
import ctypes as C
mean = torch.zeros(F, dtype=torch.float64, device=dev); var = torch.zeros_like(mean)
qh = torch.zeros(500, dtype=torch.int64, device=dev); dh = torch.zeros(1000, dtype=torch.int64, device=dev)
L = _lib.load(); vp = C.c_void_p
torch.cuda.synchronize()
ts = []
t_all = time.perf_counter()
for s in range(6):
    t0 = time.perf_counter()
    _lib.check(L.wol_q_traj(ctx.handle, vp(batches[s].data_ptr()), 3, 3 * n, F, n, _lib.ptr(boxes), 0.0, 10.0,
                            vp(mean.data_ptr()), vp(var.data_ptr()), vp(qh.data_ptr()), 500, vp(dh.data_ptr()), 1000, 0.01, None, 1))
    ts.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
print("C only: per step", ["%.2f" % (1e3 * t) for t in ts], "enqueue %.2f ms, with sync %.2f ms" % (1e3 * t_enq, 1e3 * (time.perf_counter() - t_all)))
for tk in (0, 1):
    ctx.set_option("time_kernels", tk)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for s in range(6):
        _lib.check(L.wol_q_traj(ctx.handle, vp(batches[s].data_ptr()), 3, 3 * n, F, n, _lib.ptr(boxes), 0.0, 10.0,
                                vp(mean.data_ptr()), vp(var.data_ptr()), vp(qh.data_ptr()), 500, vp(dh.data_ptr()), 1000, 0.01, None, 1))
    torch.cuda.synchronize()
    print("time_kernels=%d: %.2f ms/step" % (tk, 1e3 * (time.perf_counter() - t_all) / 6))
print(ctx.kernel_time())
