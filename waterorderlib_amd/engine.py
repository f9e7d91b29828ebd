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


# ---- populations and the other two frame loops (include/wol.h: wol_q_traj_pops, wol_cos_angs_traj, wol_lsi_traj) ----
def pack_populations(xyz_frames, subInds, nPops):
    """The reference's `subPos = [pos[subInds[t][i]] for i in range(nPops)]` (orderParam_lib.py:1466) for a block of
    frames, in the layout the C ABI takes: (pop_pos[total, 3], pop_off[T * nPops + 1]).  xyz_frames[k] is the full
    atom array of frame k, subInds[k][j] the atom indices of population j in that frame."""
    pos, off = [], [0]
    for k, xyz in enumerate(xyz_frames):
        for j in range(nPops):
            p = np.asarray(xyz)[np.asarray(subInds[k][j], dtype=np.int64)].reshape(-1, 3)
            pos.append(p)
            off.append(off[-1] + len(p))
    pop_pos = np.ascontiguousarray(np.concatenate(pos) if pos else np.zeros((0, 3)), dtype=np.float64)
    return pop_pos, np.asarray(off, dtype=np.int64)


class _Frames:
    """frames argument of the trajectory entries -> pointer, dtype code, memory space (numpy host array or torch CUDA tensor)"""

    def __init__(self, frames, boxes, len_wat, ctx):
        self.boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float64).reshape(-1, 3))
        self.torch = _is_torch(frames)
        if self.torch:
            import torch
            if not frames.is_cuda or frames.dtype not in (torch.float64, torch.float32) or frames.dim() != 3 or frames.shape[2] != 3:
                raise ValueError("frames must be a float64/float32 CUDA tensor of shape (T, natoms, 3)")
            self.arr = frames.contiguous()
            self.f32 = self.arr.dtype == torch.float32
            self.dev = self.arr.device
            self.ctx = ctx or _lib.default_context(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
            self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
            self.ptr = C.c_void_p(self.arr.data_ptr())
            self.mem = _lib.MEM_DEVICE
        else:
            a = np.asarray(frames)
            if a.dtype != np.float32:
                a = np.asarray(a, dtype=np.float64)
            self.arr = np.ascontiguousarray(a)
            if self.arr.ndim != 3 or self.arr.shape[2] != 3:
                raise ValueError("frames must have shape (T, natoms, 3)")
            self.f32 = self.arr.dtype == np.float32
            self.ctx = ctx or _lib.default_context()
            self.ptr = _lib.ptr(self.arr)
            self.mem = _lib.MEM_HOST
        self.T, self.natoms = int(self.arr.shape[0]), int(self.arr.shape[1])
        _check_shapes(self.T, self.natoms, self.boxes, len_wat)
        self.len_wat = len_wat
        self.n = self.natoms // len_wat
        self.dtype = _lib.F32 if self.f32 else _lib.F64

    def empty(self, shape, dtype, zero=False):
        """an output array in the frames' memory space; dtype: 'real' (the frames' own), 'f64' or 'i64'"""
        if self.torch:
            import torch
            dt = {"real": self.arr.dtype, "f64": torch.float64, "i64": torch.int64}[dtype]
            return (torch.zeros if zero else torch.empty)(shape, dtype=dt, device=self.dev)
        dt = {"real": self.arr.dtype, "f64": np.float64, "i64": np.int64}[dtype]
        return (np.zeros if zero else np.empty)(shape, dtype=dt)

    def p(self, a):
        if a is None:
            return None
        return C.c_void_p(a.data_ptr()) if self.torch else _lib.ptr(a)

    def points(self, pop_pos):
        """population points in the frames' memory space and storage type"""
        if pop_pos is None:
            return None
        if self.torch:
            import torch
            t = pop_pos if _is_torch(pop_pos) else torch.from_numpy(np.ascontiguousarray(pop_pos))
            return t.to(device=self.dev, dtype=self.arr.dtype).contiguous()
        return np.ascontiguousarray(np.asarray(pop_pos), dtype=self.arr.dtype)


def _pop_args(fr, pop_pos, pop_off, npops):
    if not npops:
        return None, None, 0, 0
    off = np.ascontiguousarray(np.asarray(pop_off, dtype=np.int64))
    if len(off) != fr.T * npops + 1:
        raise ValueError("pop_off must hold T * npops + 1 offsets")
    pts = fr.points(pop_pos)
    if int(off[-1]) != (0 if pts is None else len(pts)):
        raise ValueError("pop_off[-1] must equal the number of population points")
    return pts, off, int(npops), int(off[-1])


class PopsResult(TrajResult):
    """TrajResult of population 0 (all waters) plus pop_mean[T, npops], pop_var[T, npops], pop_qhist[npops, qbins], pop_q[total] or None"""


def tet_order_traj_pops(frames, boxes, pop_pos, pop_off, npops, lowCut=0.0, highCut=10.0, len_wat=1, qbins=QBINS, dbins=DBINS,
                        dbinwidth=DBINWIDTH, want_q=False, want_dhist=False, want_pop_q=False, ctx=None):
    """tetOrderCalc's frame loop with its populations (orderParam_lib.py:1458-1480) as one call: q of all waters of every
    frame (as tet_order_traj) and of every population point against the frame's waters."""
    fr = _Frames(frames, boxes, len_wat, ctx)
    pts, off, npops, total = _pop_args(fr, pop_pos, pop_off, npops)
    mean, var = fr.empty(fr.T, "f64", True), fr.empty(fr.T, "f64", True)
    qh = fr.empty(qbins, "i64", True)
    dh = fr.empty(dbins, "i64", True) if want_dhist else None
    q = fr.empty((fr.T, fr.n), "real") if want_q else None
    pm, pv = fr.empty((fr.T, npops), "f64", True), fr.empty((fr.T, npops), "f64", True)
    pqh = fr.empty((npops, qbins), "i64", True)
    pq = fr.empty(total, "real") if (want_pop_q and total) else None
    if fr.T and fr.n:
        _lib.check(_lib.load().wol_q_traj_pops(fr.ctx.handle, fr.ptr, 3 * len_wat, 3 * fr.natoms, fr.T, fr.n, _lib.ptr(fr.boxes), float(lowCut),
                                               float(highCut), fr.p(mean), fr.p(var), fr.p(qh), qbins, fr.p(dh), dbins, float(dbinwidth),
                                               fr.p(q), fr.p(pts), _lib.ptr(off), npops, fr.p(pm), fr.p(pv), fr.p(pqh), fr.p(pq),
                                               fr.dtype, fr.mem))
    r = PopsResult(mean, var, qh, dh, q)
    r.pop_mean, r.pop_var, r.pop_qhist, r.pop_q = pm, pv, pqh, pq
    return r


class CosAngsTrajResult:
    """hist[T, npops + 1, nbins] (np.histogram(angles, nbins, [0, 180]) per frame and population; population 0 = all
    waters), counts[T, npops + 1, 2] = (angles, angles in [100, 120]), cos_sums[T, npops + 1, 2] = (sum, sum of squares) of
    the cosine over the angles in [100, 120]"""

    def __init__(self, hist, counts, cos_sums):
        self.hist, self.counts, self.cos_sums = hist, counts, cos_sums

    def metrics(self, t, j=0):
        """tetrahedralMetrics (water_properties.py:314-342) of segment (t, j): (angDist, fracTet, avgCos, varCos, entropy)"""
        g = (lambda a: a.cpu().numpy()) if _is_torch(self.hist) else (lambda a: a)
        h = g(self.hist[t, j]).astype(np.float64)
        n_ang, n_tet = (int(x) for x in g(self.counts[t, j]))
        s1, s2 = (float(x) for x in g(self.cos_sums[t, j]))
        if n_ang == 0:
            return h, 0.0, 0.0, 0.0, 0.0
        avg = s1 / n_tet if n_tet else float("nan")
        var = s2 / n_tet - avg * avg if n_tet else float("nan")
        p = h / float(np.sum(h))
        p = p[p != 0]
        return h, float(n_tet) / float(n_ang), avg, var, float(-np.sum(p * np.log(p)))


def cos_angs_traj(frames, boxes, lowCut=0.0, highCut=3.413, len_wat=1, pop_pos=None, pop_off=None, npops=0, nbins=500, ctx=None):
    """threeBodyCalc's frame loop (orderParam_lib.py:1312-1353) on the GPU: the three-body angle statistics of every frame
    and population, reduced on the device (integer histograms and counts, exact fixed-point sums)."""
    fr = _Frames(frames, boxes, len_wat, ctx)
    pts, off, npops, _ = _pop_args(fr, pop_pos, pop_off, npops)
    hist = fr.empty((fr.T, npops + 1, nbins), "i64", True)
    counts = fr.empty((fr.T, npops + 1, 2), "i64", True)
    sums = fr.empty((fr.T, npops + 1, 2), "f64", True)
    if fr.T and fr.n:
        _lib.check(_lib.load().wol_cos_angs_traj(fr.ctx.handle, fr.ptr, 3 * len_wat, 3 * fr.natoms, fr.T, fr.n, _lib.ptr(fr.boxes),
                                                 float(lowCut), float(highCut), fr.p(pts), _lib.ptr(off), npops, nbins, fr.p(hist),
                                                 fr.p(counts), fr.p(sums), fr.dtype, fr.mem))
    return CosAngsTrajResult(hist, counts, sums)


class LsiTrajResult:
    """hist[npops + 1, nbins] (np.histogram(lsiVals, nbins, [0, hist_max]) per population over all frames), counts[T, npops + 1]
    = len(lsiVals), mean / var [T, npops + 1] (np.mean / np.var of them), lsi[T, n] (NaN where the reference skips the water)
    and pop_lsi[total] when asked for"""

    def __init__(self, hist, counts, mean, var, lsi, pop_lsi):
        self.hist, self.counts, self.mean, self.var, self.lsi, self.pop_lsi = hist, counts, mean, var, lsi, pop_lsi


def lsi_traj(frames, boxes, lowCut=0.0, highCut=3.7, len_wat=1, pop_pos=None, pop_off=None, npops=0, nbins=500, hist_max=0.3,
             want_lsi=False, want_pop_lsi=False, ctx=None):
    """lsiCalc's frame loop (orderParam_lib.py:1618-1640) on the GPU"""
    fr = _Frames(frames, boxes, len_wat, ctx)
    pts, off, npops, total = _pop_args(fr, pop_pos, pop_off, npops)
    hist = fr.empty((npops + 1, nbins), "i64", True)
    counts = fr.empty((fr.T, npops + 1), "i64", True)
    mean, var = fr.empty((fr.T, npops + 1), "f64", True), fr.empty((fr.T, npops + 1), "f64", True)
    lsi = fr.empty((fr.T, fr.n), "real") if want_lsi else None
    plsi = fr.empty(total, "real") if (want_pop_lsi and total) else None
    if fr.T and fr.n:
        _lib.check(_lib.load().wol_lsi_traj(fr.ctx.handle, fr.ptr, 3 * len_wat, 3 * fr.natoms, fr.T, fr.n, _lib.ptr(fr.boxes), float(lowCut),
                                            float(highCut), fr.p(pts), _lib.ptr(off), npops, nbins, float(hist_max), fr.p(hist), fr.p(counts),
                                            fr.p(mean), fr.p(var), fr.p(lsi), fr.p(plsi), fr.dtype, fr.mem))
    return LsiTrajResult(hist, counts, mean, var, lsi, plsi)


def radial_dist_same_traj(frames, boxes, binwidth=0.1, totbins=150, bulkdens=1.0, len_wat=1, want_counts=False, ctx=None):
    """rdfCalc's per-frame wl.radialdistsame (orderParam_lib.py:624-630) for a block of frames in one call: rdf[T, totbins]
    (each row what wl.radialdistsame returns for that frame, bit for bit) and, when asked for, the integer pair counts."""
    fr = _Frames(frames, boxes, len_wat, ctx)
    rdf = fr.empty((fr.T, totbins), "f64", True)
    counts = fr.empty((fr.T, totbins), "i64", True) if want_counts else None
    if fr.T and fr.n:
        _lib.check(_lib.load().wol_radial_dist_same_traj(fr.ctx.handle, fr.ptr, 3 * len_wat, 3 * fr.natoms, fr.T, fr.n, _lib.ptr(fr.boxes),
                                                         float(binwidth), int(totbins), float(bulkdens), fr.p(rdf), fr.p(counts), fr.dtype, fr.mem))
    return (rdf, counts) if want_counts else rdf
