"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle (oracle/wol_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (waterorderlib_amd) never does: it fails loudly if its HIP library is
missing instead of falling back to anything here.

Function names mirror the reference's Python/f2py surface (water_properties.py / waterlib) so
the parity tests read like calls into the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwol_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    src = os.path.join(_HERE, "wol_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libwol_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.wolo_cos_angle3.restype = C.c_double
        L.wolo_np_sum.restype = C.c_double
        L.wolo_np_mean.restype = C.c_double
        L.wolo_np_var.restype = C.c_double
        L.wolo_cos_angs.restype = C.c_long
        L.wolo_lsi.restype = C.c_long
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads used by the cell-list form; returns the effective count."""
    return lib().wolo_set_threads(C.c_int(int(n)))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _box(b):
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1)[:3])
    assert b.size == 3
    return b


# ---- waterlib-level -----------------------------------------------------------------------
def reimage(pos, refpos, boxl):
    pos = _d(pos); ref = _d(refpos); box = _box(boxl)
    out = np.empty_like(pos)
    lib().wolo_reimage(_p(pos), C.c_int(len(pos)), _p(ref), _p(box), _p(out))
    return out


def nearneighbors(subpos, pos, boxl, lowcut, highcut):
    sub = _d(subpos); pos = _d(pos); box = _box(boxl)
    mask = np.zeros((len(sub), len(pos)), dtype=np.int32, order="F")
    lib().wolo_near_neighbors(_p(sub), C.c_int(len(sub)), _p(pos), C.c_int(len(pos)), _p(box),
                              C.c_double(lowcut), C.c_double(highcut), mask.ctypes.data_as(_ip))
    return mask


def allnearneighbors(pos, boxl, lowcut, highcut):
    pos = _d(pos); box = _box(boxl)
    mask = np.zeros((len(pos), len(pos)), dtype=np.int32, order="F")
    lib().wolo_all_near_neighbors(_p(pos), C.c_int(len(pos)), _p(box), C.c_double(lowcut),
                                  C.c_double(highcut), mask.ctypes.data_as(_ip))
    return mask


def tetracosang(refpos, neighpos, boxl):
    ref = _d(refpos); nb = _d(neighpos).reshape(-1, 3); box = _box(boxl)
    n = len(nb)
    out = np.empty((n, n), dtype=np.float64)
    lib().wolo_tetra_cos_ang(_p(ref), _p(nb), C.c_int(n), _p(box), _p(out))
    return out


def lsidists(refpos, neighpos, boxl):
    ref = _d(refpos); nb = _d(neighpos).reshape(-1, 3); box = _box(boxl)
    out = np.empty(len(nb), dtype=np.float64)
    lib().wolo_lsi_dists(_p(ref), _p(nb), C.c_int(len(nb)), _p(box), _p(out))
    return out


def cosangle3(p1, p2, p3):
    return lib().wolo_cos_angle3(_p(_d(p1)), _p(_d(p2)), _p(_d(p3)))


# ---- water_properties-level ----------------------------------------------------------------
def getOrderParamq(subPos, Pos, BoxDims, lowCut=0.0, highCut=10.0, full=False, cells=False):
    """water_properties.getOrderParamq; full=True also returns (nn_idx, nn_dist, ang_deg)."""
    sub = _d(subPos); pos = _d(Pos); box = _box(BoxDims)
    ns = len(sub)
    q = np.zeros(ns)
    idx = np.full((ns, 4), -1, dtype=np.int32)
    dist = np.full((ns, 4), np.nan)
    ang = np.full((ns, 6), np.nan)
    fn = lib().wolo_order_param_q_cells if cells else lib().wolo_order_param_q
    rc = fn(_p(sub), C.c_int(ns), _p(pos), C.c_int(len(pos)), _p(box), C.c_double(lowCut),
            C.c_double(highCut), _p(q), idx.ctypes.data_as(_ip), _p(dist), _p(ang))
    if rc != 0:
        raise ValueError("cell-list oracle needs a fully periodic box")
    return (q, idx, dist, ang) if full else q


def getCosAngs(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.413):
    sub = _d(subPos); pos = _d(Pos); box = _box(BoxDims)
    ns = len(sub)
    num = np.zeros(ns)
    args = (_p(sub), C.c_int(ns), _p(pos), C.c_int(len(pos)), _p(box), C.c_double(lowCut), C.c_double(highCut))
    total = lib().wolo_cos_angs(*args, None, _p(num))
    ang = np.empty(total)
    lib().wolo_cos_angs(*args, _p(ang), _p(num))
    return ang, num


def getLSI(subPos, Pos, BoxDims, lowCut=0.0, highCut=3.7):
    sub = _d(subPos); pos = _d(Pos); box = _box(BoxDims)
    ns = len(sub)
    full = np.empty(ns)
    num = np.zeros(ns)
    lib().wolo_lsi(_p(sub), C.c_int(ns), _p(pos), C.c_int(len(pos)), _p(box), C.c_double(lowCut),
                   C.c_double(highCut), _p(full), _p(num))
    return full[~np.isnan(full)], num


# ---- numpy restatements / trajectory reduction ---------------------------------------------
def np_mean(a):
    a = _d(a)
    return lib().wolo_np_mean(_p(a), C.c_long(a.size))


def np_var(a):
    a = _d(a)
    return lib().wolo_np_var(_p(a), C.c_long(a.size))


def np_histogram(a, bins=500, range=(0.0, 1.0)):
    a = _d(a).reshape(-1)
    counts = np.zeros(bins, dtype=np.int64)
    lib().wolo_np_histogram(_p(a), C.c_long(a.size), C.c_int(bins), C.c_double(range[0]),
                            C.c_double(range[1]), counts.ctypes.data_as(_lp))
    return counts


def dist_histogram(d, binwidth=0.01, nbins=1000):
    d = _d(d).reshape(-1)
    counts = np.zeros(nbins, dtype=np.int64)
    lib().wolo_dist_histogram(_p(d), C.c_long(d.size), C.c_double(binwidth), C.c_int(nbins),
                              counts.ctypes.data_as(_lp))
    return counts


def tet_order_traj(frames, boxes, lowCut=0.0, highCut=10.0, cells=True, qbins=500, dbins=1000,
                   dbinwidth=0.01):
    """Per-trajectory reduction of orderParam_lib.tetOrderCalc over frames[T,N,3], boxes[T,3]:
    returns (mean[T], var[T], qhist[qbins], dhist[dbins])."""
    frames = _d(frames); boxes = _d(boxes)
    T, n, _ = frames.shape
    mean = np.empty(T); var = np.empty(T)
    qh = np.zeros(qbins, dtype=np.int64); dh = np.zeros(dbins, dtype=np.int64)
    lib().wolo_tet_order_traj(_p(frames), _p(boxes), C.c_int(T), C.c_int(n), C.c_double(lowCut),
                              C.c_double(highCut), C.c_int(1 if cells else 0), _p(mean), _p(var),
                              qh.ctypes.data_as(_lp), C.c_int(qbins), dh.ctypes.data_as(_lp),
                              C.c_int(dbins), C.c_double(dbinwidth))
    return mean, var, qh, dh


# ---- Willard-Chandler density field --------------------------------------
def willarddensitypoints(pos, denspts, boxl, smoothlen):
    """waterlib.willarddensitypoints -> (densvals[npts], densnorms[npts,3])"""
    pos = _d(pos); pts = _d(denspts).reshape(-1, 3); box = _box(boxl)
    dv = np.empty(len(pts)); dn = np.empty((len(pts), 3))
    lib().wolo_willard_density_points(_p(pos), C.c_int(len(pos)), _p(pts), C.c_int(len(pts)), _p(box),
                                      C.c_double(smoothlen), _p(dv), _p(dn))
    return dv, dn


def willarddensityfield(pos, gridx, gridy, gridz, boxl, smoothlen):
    """waterlib.willarddensityfield -> (densvals[Nx,Ny,Nz], densnorms[Nx,Ny,Nz,3]) (Fortran order)"""
    gx = _d(gridx).reshape(-1); gy = _d(gridy).reshape(-1); gz = _d(gridz).reshape(-1)
    X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    dv, dn = willarddensitypoints(pos, pts, boxl, smoothlen)
    return (np.asfortranarray(dv.reshape(len(gx), len(gy), len(gz))),
            np.asfortranarray(dn.reshape(len(gx), len(gy), len(gz), 3)))


# ---- radial distribution functions -----------------------------------------
def radialdistsame(pos, binwidth, totbins, bulkdens, boxl):
    """waterlib.radialdistsame -> rdf[totbins]"""
    pos = _d(pos); box = _box(boxl); rdf = np.empty(int(totbins))
    lib().wolo_radial_dist_same(_p(pos), C.c_int(len(pos)), C.c_double(binwidth), C.c_int(int(totbins)),
                                C.c_double(bulkdens), _p(box), _p(rdf))
    return rdf


def radialdist(pos1, pos2, binwidth, totbins, bulkdens, boxl):
    """waterlib.radialdist -> rdf[totbins] (atoms of pos2 around atoms of pos1)"""
    p1 = _d(pos1); p2 = _d(pos2); box = _box(boxl); rdf = np.empty(int(totbins))
    lib().wolo_radial_dist(_p(p1), C.c_int(len(p1)), _p(p2), C.c_int(len(p2)), C.c_double(binwidth),
                           C.c_int(int(totbins)), C.c_double(bulkdens), _p(box), _p(rdf))
    return rdf


def pairdistancehistogram(pos1, pos2, binwidth, totbins, boxl):
    """waterlib.pairdistancehistogram (Dim = 3) -> hist[totbins]"""
    p1 = _d(pos1); p2 = _d(pos2); box = _box(boxl); h = np.empty(int(totbins))
    lib().wolo_pair_distance_histogram(_p(p1), C.c_int(len(p1)), _p(p2), C.c_int(len(p2)), C.c_double(binwidth),
                                       C.c_int(int(totbins)), _p(box), _p(h))
    return h


# ---- water orientation -----------------------------------------------------
def watorient(opos, hpos, refvec, boxl):
    """waterlib.watorient -> (angdip[n], angplane[n]) in degrees"""
    o = _d(opos); h = _d(hpos); box = _box(boxl); ref = np.ascontiguousarray(np.asarray(refvec, dtype=np.float64).reshape(-1)[:3])
    if len(h) != 2 * len(o):
        raise ValueError("Number of hydrogens must be two times number of oxygens.")
    ad = np.empty(len(o)); ap = np.empty(len(o))
    lib().wolo_wat_orient(_p(o), _p(h), C.c_int(len(o)), _p(ref), _p(box), _p(ad), _p(ap))
    return ad, ap


def watohvec(opos, hpos, boxl):
    """waterlib.watohvec -> normdip[n,3] (Fortran order, as f2py returns it)"""
    o = _d(opos); h = _d(hpos); box = _box(boxl)
    if len(h) != 2 * len(o):
        raise ValueError("Number of hydrogens must be two times number of oxygens.")
    out = np.empty((len(o), 3))
    lib().wolo_wat_oh_vec(_p(o), _p(h), C.c_int(len(o)), _p(box), _p(out))
    return np.asfortranarray(out)


# ---- hydrogen bonds ---------------------------------------------------------
def generalhbonds(acceptorpos, donorpos, donorhpos, boxl, distcut, angcut):
    """waterlib.generalhbonds -> bondbool (nacc, ndon) int32, Fortran order"""
    a = _d(acceptorpos); d = _d(donorpos); h = _d(donorhpos); box = _box(boxl)
    if len(d) != len(h):
        raise ValueError("Number of donor hydrogens and heavy-atoms do not match.")
    out = np.zeros((len(a), len(d)), dtype=np.int32, order="F")
    lib().wolo_general_hbonds(_p(a), C.c_int(len(a)), _p(d), _p(h), C.c_int(len(d)), _p(box), C.c_double(distcut),
                              C.c_double(angcut), out.ctypes.data_as(C.c_void_p))
    return out
