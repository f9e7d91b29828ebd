"""Drop-in for the hot-path functions of the reference's structureLibs/water_properties.py:
``getOrderParamq`` (:344-391), ``getCosAngs`` (:210-250), ``getLSI`` (:252-311) and
``tetrahedralMetrics`` (:314-342) -- same names, arguments, defaults and return values.  Instead of
an N x N mask plus a per-water Python loop around f2py calls, each call is ONE trip through the C
ABI into fused HIP kernels (include/wol.h: wol_q_frame / wol_cos_angs / wol_lsi_frame).

    import waterorderlib_amd.water_properties as wp
"""
import ctypes as C

import numpy as np

from . import _lib
from . import waterlib as wl

watdens = 0.033456  # molecules or oxygens per Angstrom ^ 3 near 300 K (water_properties.py:55)


def _pos(a, name):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("%s must have shape (n,3), got %s" % (name, a.shape))
    return np.ascontiguousarray(a)


def _same(subPos, Pos):
    # the reference picks the allnearneighbors branch with np.array_equal(subPos, Pos) (wp:363)
    return subPos is Pos or (subPos.shape == Pos.shape and np.array_equal(subPos, Pos))


def getOrderParamq(subPos, Pos, BoxDims, lowCut=0.0, highCut=10.0, full=False, ctx=None):
    """Tetrahedral order parameter q of Errington & Debenedetti for every row of subPos
    (water_properties.py:344-391).  ``full=True`` additionally returns
    (nn_idx[ns,4] int32, nn_dist[ns,4], ang_deg[ns,6]): the four nearest neighbours (indices into
    Pos, -1 padded), their distances (NaN padded) and the six angles that entered q."""
    Pos = _pos(Pos, "Pos")
    subPos = Pos if subPos is Pos else _pos(subPos, "subPos")
    box = _lib.box3(BoxDims)
    ctx = ctx or _lib.default_context()
    ns = len(subPos)
    q = np.zeros(ns, dtype=np.float64)
    idx = np.full((ns, 4), -1, dtype=np.int32) if full else None
    dist = np.full((ns, 4), np.nan) if full else None
    ang = np.full((ns, 6), np.nan) if full else None
    if ns:
        sub = None if _same(subPos, Pos) else subPos
        _lib.check(_lib.load().wol_q_frame(ctx.handle, _lib.ptr(Pos), 3, len(Pos), _lib.ptr(sub), 3, ns,
                                           _lib.ptr(box), float(lowCut), float(highCut), _lib.ptr(q),
                                           _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(ang), _lib.MEM_HOST))
    return (q, idx, dist, ang) if full else q


def getCosAngs(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.413, ctx=None):
    """All three-body angles (degrees) between the neighbours in the shell lowCut < d <= highCut of
    every row of subPos, flattened in water order, and the number of shell neighbours per water
    (water_properties.py:210-250).  Returns (angVals, numAngs)."""
    Pos = _pos(Pos, "Pos")
    subPos = Pos if subPos is Pos else _pos(subPos, "subPos")
    box = _lib.box3(BoxDims)
    ctx = ctx or _lib.default_context()
    ns = len(subPos)
    num = np.zeros(ns, dtype=np.float64)
    if ns == 0 or len(Pos) == 0:
        return np.array([]), num
    sub = None if _same(subPos, Pos) else subPos
    total = C.c_int64(0)
    L = _lib.load()
    args = (ctx.handle, _lib.ptr(Pos), len(Pos), _lib.ptr(sub), ns, _lib.ptr(box), float(lowCut), float(highCut))
    _lib.check(L.wol_cos_angs(*args, None, _lib.ptr(num), C.byref(total)))
    ang = np.empty(total.value, dtype=np.float64)
    if total.value:
        _lib.check(L.wol_cos_angs(*args, _lib.ptr(ang), _lib.ptr(num), C.byref(total)))
    return ang, num


def getLSI(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.7, ctx=None, full=False):
    """Local structure index of Shiratani & Sasai per water (water_properties.py:252-311).
    Returns (lsiVals, numLSI); waters the reference skips are absent from lsiVals.  full=True (an extension): the LSI of
    EVERY water of subPos, NaN where the reference skips it, instead of the compacted list."""
    full_out = full
    Pos = _pos(Pos, "Pos")
    subPos = Pos if subPos is Pos else _pos(subPos, "subPos")
    box = _lib.box3(BoxDims)
    ctx = ctx or _lib.default_context()
    ns = len(subPos)
    num = np.zeros(ns, dtype=np.float64)
    if ns == 0 or len(Pos) == 0:
        return (np.full(ns, np.nan) if full_out else np.array([])), num
    sub = None if _same(subPos, Pos) else subPos
    full = np.empty(ns, dtype=np.float64)
    nvalid = C.c_int64(0)
    _lib.check(_lib.load().wol_lsi_frame(ctx.handle, _lib.ptr(Pos), len(Pos), _lib.ptr(sub), ns, _lib.ptr(box),
                                         float(lowCut), float(highCut), _lib.ptr(full), _lib.ptr(num),
                                         C.byref(nvalid)))
    if full_out:
        return full, num
    return full[~np.isnan(full)], num


def tetrahedralMetrics(angVals, nBins=500, binRange=[0.0, 180.0]):
    """water_properties.py:314-342: statistics of a three-body angle list (degrees).  Returns the histogram counts over
    binRange and its edges, the fraction of angles inside [100, 120] degrees, the mean and the population variance of
    the cosine over THAT tetrahedral subset, and the Shannon entropy -sum p ln p of the normalised histogram (empty
    bins left out).  Host-side numpy, as in the reference."""
    ang = np.asarray(angVals, dtype=np.float64)
    counts, edges = np.histogram(ang, bins=nBins, range=binRange, density=False)
    tet = ang[(ang >= 100.0) & (ang <= 120.0)]
    frac_tet = float(len(tet)) / float(len(ang))
    cos_tet = np.cos(tet * np.pi / 180.0)
    p = counts / float(np.sum(counts))
    p = p[p != 0]
    return counts, edges, frac_tet, np.mean(cos_tet), np.var(cos_tet), -np.sum(p * np.log(p))


def waterOrientation(Opos, Hpos, boxDim, refVec=[0.0, 0.0, 1.0]):
    """water_properties.py:621-638: (dipAngs, planeAngs) of wl.watorient"""
    return wl.watorient(Opos, Hpos, refVec, boxDim)


def waterOrientationBinZ(Opos, Hpos, boxDim, refVec=[0.0, 0.0, 1.0], refBins=None, angBins=None):
    """water_properties.py:578-618: 2-D histograms (angle x position along refVec) of the plane-normal and dipole
    angles.  The reference's own body no longer runs -- np.histogram2d lost `normed`, and it pairs the per-water
    plane angles with a per-hydrogen position list -- so this follows its evident intent: both histograms are taken
    against the oxygen position along refVec.  Returns (plane2Dhist, dip2Dhist)."""
    refVec = np.asarray(refVec, dtype=np.float64)
    refVec = refVec / np.linalg.norm(refVec)
    Opos = np.asarray(Opos, dtype=np.float64)
    zOpos = np.dot(Opos, refVec)
    angDip, angPlane = wl.watorient(Opos, Hpos, refVec, boxDim)
    if refBins is None:
        refBins = np.arange(np.min(zOpos), np.max(zOpos), 0.2)
    if angBins is None:
        angBins = np.arange(0.0, 180.001, 180.0 / 500.0)
    plane2Dhist, _, _ = np.histogram2d(angPlane, zOpos, bins=[angBins, refBins])
    dip2Dhist, _, _ = np.histogram2d(angDip, zOpos, bins=[angBins, refBins])
    return plane2Dhist, dip2Dhist


def HBondsGeneral(accPos, donPos, donHPos, boxL, accInds, donInds, donHInds, distCut=3.5, angCut=150.0):
    """water_properties.py:681-720: (NumHB, HBlist[NumHB,2] = (acceptor index, donor index), HBloc[NumHB,3] = midpoint
    between the acceptor and the donor hydrogen imaged about it).  The reference scans the (nacc, ndon) mask of
    wl.generalhbonds row by row; the bonds come from the cell list here (wl.generalhbonds_list, same order)."""
    accPos = _pos(accPos, "accPos"); donHPos = _pos(donHPos, "donHPos")
    box = np.asarray(boxL, dtype=np.float64).reshape(-1)[:3]
    ai, di = wl.generalhbonds_list(accPos, donPos, donHPos, box, distCut, angCut)
    NumHB = len(ai)
    HBlist = (-1) * np.ones((NumHB, 2))
    HBlist[:, 0] = np.asarray(accInds)[ai]
    HBlist[:, 1] = np.asarray(donInds)[di]
    a = accPos[ai]
    d = donHPos[di] - a
    ibox = np.where(box >= 0.0, 1.0 / np.where(box == 0.0, np.inf, box), 0.0) if np.any(box <= 0.0) else 1.0 / box
    t = d * ibox
    d = d - box * (np.sign(t) * np.floor(np.abs(t) + 0.5))     # anint: half away from zero (waterlib.f90:44)
    HBloc = 0.5 * ((a + d) + a)
    return NumHB, HBlist, HBloc


# ---- device-resident / float32 forms (include/wol.h: wol_q_frame_t, wol_cos_angs_t, wol_lsi_frame_t) ------------
# Same functions for callers whose coordinates already live in HBM as torch tensors (float64 or float32): nothing
# crosses the host link, results are torch tensors on the same device with the input's dtype (arithmetic is fp64).
def _dev_args(subPos, Pos):
    import torch
    if not (Pos.is_cuda and Pos.dim() == 2 and Pos.shape[1] == 3 and Pos.dtype in (torch.float64, torch.float32)):
        raise ValueError("Pos must be a float64/float32 CUDA tensor of shape (n,3)")
    Pos = Pos.contiguous()
    same = subPos is Pos or subPos is None
    if not same:
        if subPos.dtype != Pos.dtype or subPos.device != Pos.device or subPos.dim() != 2 or subPos.shape[1] != 3:
            raise ValueError("subPos must match Pos in dtype and device and have shape (ns,3)")
        subPos = subPos.contiguous()
    dt = _lib.F64 if Pos.dtype == torch.float64 else _lib.F32
    ctx = _lib.default_context(Pos.device.index if Pos.device.index is not None else torch.cuda.current_device())
    ctx.set_stream(torch.cuda.current_stream(Pos.device).cuda_stream)
    return torch, Pos, (None if same else subPos), dt, ctx


def getOrderParamq_dev(subPos, Pos, BoxDims, lowCut=0.0, highCut=10.0, full=False):
    """getOrderParamq on torch CUDA tensors; returns q (and nn_idx int32, nn_dist, ang_deg with full=True) as tensors.
    Asynchronous: the work is enqueued on torch's current stream."""
    torch, Pos, sub, dt, ctx = _dev_args(subPos, Pos)
    box = _lib.box3(BoxDims)
    ns = len(Pos) if sub is None else len(sub)
    q = torch.zeros(ns, dtype=Pos.dtype, device=Pos.device)
    idx = torch.full((ns, 4), -1, dtype=torch.int32, device=Pos.device) if full else None
    dist = torch.full((ns, 4), float("nan"), dtype=Pos.dtype, device=Pos.device) if full else None
    ang = torch.full((ns, 6), float("nan"), dtype=Pos.dtype, device=Pos.device) if full else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if ns and len(Pos):
        _lib.check(_lib.load().wol_q_frame_t(ctx.handle, p(Pos), 3, len(Pos), p(sub), 3, ns, _lib.ptr(box), float(lowCut),
                                             float(highCut), p(q), p(idx), p(dist), p(ang), dt, _lib.MEM_DEVICE))
    return (q, idx, dist, ang) if full else q


def getCosAngs_dev(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.413):
    """getCosAngs on torch CUDA tensors -> (angVals, numAngs) tensors (the angle count is read back to size the output)."""
    torch, Pos, sub, dt, ctx = _dev_args(subPos, Pos)
    box = _lib.box3(BoxDims)
    ns = len(Pos) if sub is None else len(sub)
    num = torch.zeros(ns, dtype=Pos.dtype, device=Pos.device)
    if ns == 0 or len(Pos) == 0:
        return torch.zeros(0, dtype=Pos.dtype, device=Pos.device), num
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    total = C.c_int64(0)
    L = _lib.load()
    args = (ctx.handle, p(Pos), len(Pos), p(sub), ns, _lib.ptr(box), float(lowCut), float(highCut))
    _lib.check(L.wol_cos_angs_t(*args, None, p(num), C.byref(total), dt, _lib.MEM_DEVICE))
    ang = torch.empty(total.value, dtype=Pos.dtype, device=Pos.device)
    if total.value:
        _lib.check(L.wol_cos_angs_t(*args, p(ang), p(num), C.byref(total), dt, _lib.MEM_DEVICE))
    return ang, num


def getLSI_dev(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.7):
    """getLSI on torch CUDA tensors -> (lsiVals, numLSI) tensors; waters the reference skips are absent from lsiVals."""
    torch, Pos, sub, dt, ctx = _dev_args(subPos, Pos)
    box = _lib.box3(BoxDims)
    ns = len(Pos) if sub is None else len(sub)
    num = torch.zeros(ns, dtype=Pos.dtype, device=Pos.device)
    if ns == 0 or len(Pos) == 0:
        return torch.zeros(0, dtype=Pos.dtype, device=Pos.device), num
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    full = torch.empty(ns, dtype=Pos.dtype, device=Pos.device)
    nvalid = C.c_int64(0)
    _lib.check(_lib.load().wol_lsi_frame_t(ctx.handle, p(Pos), len(Pos), p(sub), ns, _lib.ptr(box), float(lowCut),
                                           float(highCut), p(full), p(num), C.byref(nvalid), dt, _lib.MEM_DEVICE))
    return full[~torch.isnan(full)], num
