"""Trajectory-level entry points over the C ABI (include/wol.h: wol_q_traj / wol_q_traj_f32) and
the frame-sharded multi-GPU reduction.

`tet_order_traj` is the array form of the reference's tetOrderCalc frame loop
(orderParam_lib.py:1458-1495): T frames in, per-frame mean/var of q plus the integer q histogram
(np.histogram(q, 500, [0,1])) and the 4-NN O-O distance histogram out.  Frames may be host numpy
arrays (staged by the library) or torch tensors already resident in HBM (zero-copy: the library
gets `data_ptr()` and enqueues on torch's current stream).
"""
import ctypes as C

import numpy as np

from . import _lib

QBINS = 500           # orderParam_lib.py:1493
DBINS = 1000          # 0.01 A bins to the 10 A cutoff
DBINWIDTH = 0.01


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class TrajResult:
    """mean[T], var[T] (np.mean / np.var of q per frame), qhist[qbins], dhist[dbins] (int64), q[T,N] or None."""

    def __init__(self, mean, var, qhist, dhist, q=None, frame_offset=0):
        self.mean, self.var, self.qhist, self.dhist, self.q = mean, var, qhist, dhist, q
        self.frame_offset = frame_offset


def tet_order_traj(frames, boxes, lowCut=0.0, highCut=10.0, len_wat=1, qbins=QBINS, dbins=DBINS,
                   dbinwidth=DBINWIDTH, want_q=False, want_dhist=True, ctx=None):
    """frames: [T, natoms, 3] float64 (or float32) numpy array or torch CUDA tensor; the oxygen of
    water w is atom w*len_wat (TrajObject.getWatInds, TrajObject.py:35-52).  boxes: [T,3] host.
    Returns a TrajResult; with device input the arrays are torch tensors on the same device
    (nothing is synchronised -- the work is merely enqueued)."""
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 3))
    if _is_torch(frames):
        return _traj_device(frames, boxes, lowCut, highCut, len_wat, qbins, dbins, dbinwidth, want_q, want_dhist, ctx)
    frames = np.asarray(frames)
    if frames.dtype != np.float32:
        frames = np.asarray(frames, dtype=np.float64)
    frames = np.ascontiguousarray(frames)
    if frames.ndim != 3 or frames.shape[2] != 3:
        raise ValueError("frames must have shape (T, natoms, 3)")
    T, natoms, _ = frames.shape
    _check_shapes(T, natoms, boxes, len_wat)
    n = natoms // len_wat
    ctx = ctx or _lib.default_context()
    mean = np.zeros(T); var = np.zeros(T)
    qh = np.zeros(qbins, dtype=np.int64)
    dh = np.zeros(dbins, dtype=np.int64) if want_dhist else None
    f32 = frames.dtype == np.float32
    q = np.empty((T, n), dtype=frames.dtype) if want_q else None
    fn = _lib.load().wol_q_traj_f32 if f32 else _lib.load().wol_q_traj
    if T and n:
        _lib.check(fn(ctx.handle, _lib.ptr(frames), 3 * len_wat, 3 * natoms, T, n, _lib.ptr(boxes), float(lowCut),
                      float(highCut), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(qh), qbins, _lib.ptr(dh), dbins,
                      float(dbinwidth), _lib.ptr(q), _lib.MEM_HOST))
    return TrajResult(mean, var, qh, dh, q)


def _check_shapes(T, natoms, boxes, len_wat):
    # the library reads boxes[3*f] for every frame and waters at multiples of len_wat: both must cover the frames
    if len(boxes) != T:
        raise ValueError("boxes must have one row per frame")
    if len_wat < 1 or natoms % len_wat:
        raise ValueError("natoms is not a multiple of len_wat")


def _traj_device(frames, boxes, lowCut, highCut, len_wat, qbins, dbins, dbinwidth, want_q, want_dhist, ctx):
    import torch
    if not frames.is_cuda:
        raise ValueError("torch frames must live on the GPU (use a numpy array for host data)")
    if frames.dtype not in (torch.float64, torch.float32) or frames.dim() != 3 or frames.shape[2] != 3:
        raise ValueError("frames must be a float64/float32 tensor of shape (T, natoms, 3)")
    frames = frames.contiguous()
    T, natoms, _ = frames.shape
    _check_shapes(T, natoms, boxes, len_wat)
    n = natoms // len_wat
    dev = frames.device
    ctx = ctx or _lib.default_context(dev.index if dev.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    # the library writes mean / var of every frame and ADDS to the histograms: one zero-fill for both histograms
    stats = torch.zeros((2, T), dtype=torch.float64, device=dev) if not (T and n) else torch.empty((2, T), dtype=torch.float64, device=dev)
    mean, var = stats[0], stats[1]
    hists = torch.zeros(qbins + (dbins if want_dhist else 0), dtype=torch.int64, device=dev)
    qh = hists[:qbins]
    dh = hists[qbins:] if want_dhist else None
    q = torch.empty((T, n), dtype=frames.dtype, device=dev) if want_q else None
    fn = _lib.load().wol_q_traj_f32 if frames.dtype == torch.float32 else _lib.load().wol_q_traj
    vp = C.c_void_p
    if T and n:
        _lib.check(fn(ctx.handle, vp(frames.data_ptr()), 3 * len_wat, 3 * natoms, T, n, _lib.ptr(boxes),
                      float(lowCut), float(highCut), vp(mean.data_ptr()), vp(var.data_ptr()), vp(qh.data_ptr()),
                      qbins, vp(dh.data_ptr()) if dh is not None else None, dbins, float(dbinwidth),
                      vp(q.data_ptr()) if q is not None else None, _lib.MEM_DEVICE))
    return TrajResult(mean, var, qh, dh, q)


# ---- frame sharding across ranks (one process per GPU) ------------------------------------------
def shard_bounds(T, rank, world):
    """Frame-contiguous block of rank r: [r*T/R, (r+1)*T/R)."""
    return (rank * T) // world, ((rank + 1) * T) // world


def reduce_traj_result(local, T_total, group=None):
    """The path's one exchange step: all-reduce (sum) of the int64 histograms and an all-gather of the
    per-frame (mean, var) series, through torch.distributed (backend "nccl" is RCCL on ROCm; "gloo"
    on CPU).  Integer sums are independent of reduction order, so the result is bit-identical to the
    single-GPU one.  `local` holds this rank's shard (frames [frame_offset, frame_offset+len(mean)))."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    as_t = lambda a, dt: a if _is_torch(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    qh = as_t(local.qhist, torch.int64).clone()
    hist = qh if local.dhist is None else torch.cat([qh, as_t(local.dhist, torch.int64)])
    dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    # all_gather needs equal shapes: pad each shard's (mean,var) block to ceil(T/world) frames
    per = (T_total + world - 1) // world + 1
    dev = hist.device
    blk = torch.zeros((per, 2), dtype=torch.float64, device=dev)
    m = as_t(local.mean, torch.float64).to(dev); v = as_t(local.var, torch.float64).to(dev)
    blk[: len(m), 0] = m
    blk[: len(v), 1] = v
    out = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(out, blk, group=group)
    mean = torch.empty(T_total, dtype=torch.float64, device=dev)
    var = torch.empty(T_total, dtype=torch.float64, device=dev)
    for r in range(world):
        a, b = shard_bounds(T_total, r, world)
        mean[a:b] = out[r][: b - a, 0]
        var[a:b] = out[r][: b - a, 1]
    nq = len(qh)
    if _is_torch(local.qhist):
        return TrajResult(mean, var, hist[:nq], hist[nq:] if local.dhist is not None else None, None, 0)
    return TrajResult(mean.cpu().numpy(), var.cpu().numpy(), hist[:nq].cpu().numpy(),
                      hist[nq:].cpu().numpy() if local.dhist is not None else None, None, 0)


def tet_order_traj_sharded(frames_local, boxes_local, T_total, frame_offset, group=None, **kw):
    """Each rank passes its own block of frames; returns the job-wide TrajResult on every rank."""
    local = tet_order_traj(frames_local, boxes_local, **kw)
    local.frame_offset = frame_offset
    return reduce_traj_result(local, T_total, group)


def tet_order_traj_sharded_rccl(frames_local, boxes_local, T_total, comm, lowCut=0.0, highCut=10.0, len_wat=1, qbins=QBINS,
                                dbins=DBINS, dbinwidth=DBINWIDTH, want_q=False, want_dhist=True):
    """The same job through the library's own multi-GPU entry (include/wol.h: wol_q_traj_sharded): this rank's block of
    frames (wol_shard_bounds) goes through the one-GPU pipeline and the exchange step runs inside the library on RCCL.
    ``comm`` is a _lib.Comm; frames may be host numpy arrays or torch CUDA tensors on the communicator's device.
    Returns the job-wide TrajResult (mean / var of ALL T_total frames, reduced histograms) on every rank; q is local."""
    boxes_local = np.ascontiguousarray(np.asarray(boxes_local, dtype=np.float64).reshape(-1, 3))
    ctx = comm.ctx
    dev = _is_torch(frames_local)
    if dev:
        import torch
        frames = frames_local.contiguous()
        if not frames.is_cuda or frames.dtype not in (torch.float64, torch.float32) or frames.dim() != 3 or frames.shape[2] != 3:
            raise ValueError("frames must be a float64/float32 CUDA tensor of shape (T, natoms, 3)")
        f32 = frames.dtype == torch.float32
    else:
        frames = np.asarray(frames_local)
        if frames.dtype != np.float32:
            frames = np.asarray(frames, dtype=np.float64)
        frames = np.ascontiguousarray(frames)
        if frames.ndim != 3 or frames.shape[2] != 3:
            raise ValueError("frames must have shape (T, natoms, 3)")
        f32 = frames.dtype == np.float32
    T, natoms, _ = frames.shape
    _check_shapes(T, natoms, boxes_local, len_wat)
    n = natoms // len_wat
    fn = _lib.load().wol_q_traj_sharded_f32 if f32 else _lib.load().wol_q_traj_sharded
    vp = C.c_void_p
    if dev:
        import torch
        d = frames.device
        ctx.set_stream(torch.cuda.current_stream(d).cuda_stream)
        mean = torch.zeros(T_total, dtype=torch.float64, device=d); var = torch.zeros(T_total, dtype=torch.float64, device=d)
        qh = torch.zeros(qbins, dtype=torch.int64, device=d)
        dh = torch.zeros(dbins, dtype=torch.int64, device=d) if want_dhist else None
        q = torch.empty((T, n), dtype=frames.dtype, device=d) if want_q else None
        p = lambda t: vp(t.data_ptr()) if t is not None else None
        _lib.check(fn(ctx.handle, comm.handle, p(frames), 3 * len_wat, 3 * natoms, T, T_total, n, _lib.ptr(boxes_local), float(lowCut),
                      float(highCut), p(mean), p(var), p(qh), qbins, p(dh), dbins, float(dbinwidth), p(q), _lib.MEM_DEVICE))
    else:
        mean = np.zeros(T_total); var = np.zeros(T_total)
        qh = np.zeros(qbins, dtype=np.int64)
        dh = np.zeros(dbins, dtype=np.int64) if want_dhist else None
        q = np.empty((T, n), dtype=frames.dtype) if want_q else None
        _lib.check(fn(ctx.handle, comm.handle, _lib.ptr(frames), 3 * len_wat, 3 * natoms, T, T_total, n, _lib.ptr(boxes_local),
                      float(lowCut), float(highCut), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(qh), qbins, _lib.ptr(dh), dbins,
                      float(dbinwidth), _lib.ptr(q), _lib.MEM_HOST))
    return TrajResult(mean, var, qh, dh, q, comm.rank)
