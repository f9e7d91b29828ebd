"""Drop-in for the reference's f2py module ``waterlib`` (built from fortran/waterlib.f90 by
fortran/buildWrappers.sh:3-5), restricted to the routines on the q / 4-NN / angle path.

Same names (lower-cased by f2py), argument order, optional trailing dimension arguments, dtypes
and memory order as the f2py wrappers: ``logical`` results come back as Fortran-ordered int32,
rank-2 real results as Fortran-ordered float64.  The work is done by libwol.so on the GPU; there is
no CPU path.

    import waterorderlib_amd.waterlib as wl      # instead of: import waterlib as wl
"""
import ctypes as C

import numpy as np

from . import _lib


def _pos(a, name):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("%s must have shape (n,3), got %s" % (name, a.shape))
    return np.ascontiguousarray(a)


def _vec3(a, name):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if a.size != 3:
        raise ValueError("%s must have 3 elements" % name)
    return a


def reimage(pos, refpos, boxl, npos=None, dim=None):
    """reimagedpos = reimage(pos,refpos,boxl,[npos,dim])   -- waterlib.f90:32-47"""
    pos = _pos(pos, "pos")
    if (npos is not None and npos != len(pos)) or (dim is not None and dim != 3):
        raise ValueError("npos/dim do not match pos")
    ref = _vec3(refpos, "refpos"); box = _lib.box3(boxl)
    out = np.empty_like(pos)
    _lib.check(_lib.load().wol_reimage(_lib.default_context().handle, _lib.ptr(pos), len(pos), _lib.ptr(ref),
                                       _lib.ptr(box), _lib.ptr(out)))
    return np.asfortranarray(out)


def nearneighbors(subpos, pos, boxl, lowcut, highcut, nsubpos=None, npos=None):
    """nneighbors = nearneighbors(subpos,pos,boxl,lowcut,highcut,[nsubpos,npos]) -- waterlib.f90:710-743"""
    sub = _pos(subpos, "subpos"); pos = _pos(pos, "pos"); box = _lib.box3(boxl)
    if (nsubpos is not None and nsubpos != len(sub)) or (npos is not None and npos != len(pos)):
        raise ValueError("nsubpos/npos do not match the arrays")
    mask = np.zeros((len(sub), len(pos)), dtype=np.int32, order="F")
    _lib.check(_lib.load().wol_near_neighbors(_lib.default_context().handle, _lib.ptr(sub), len(sub), _lib.ptr(pos),
                                              len(pos), _lib.ptr(box), float(lowcut), float(highcut), _lib.ptr(mask)))
    return mask


def allnearneighbors(pos, boxl, lowcut, highcut, npos=None):
    """nneighbors = allnearneighbors(pos,boxl,lowcut,highcut,[npos]) -- waterlib.f90:830-862"""
    pos = _pos(pos, "pos"); box = _lib.box3(boxl)
    if npos is not None and npos != len(pos):
        raise ValueError("npos does not match pos")
    mask = np.zeros((len(pos), len(pos)), dtype=np.int32, order="F")
    _lib.check(_lib.load().wol_all_near_neighbors(_lib.default_context().handle, _lib.ptr(pos), len(pos),
                                                  _lib.ptr(box), float(lowcut), float(highcut), _lib.ptr(mask)))
    return mask


def tetracosang(refpos, neighpos, boxl, nneighpos=None):
    """allangs = tetracosang(refpos,neighpos,boxl,[nneighpos]) -- waterlib.f90:867-895 (degrees;
    the diagonal, which the reference leaves uninitialised, is 0)"""
    ref = _vec3(refpos, "refpos"); nb = _pos(neighpos, "neighpos"); box = _lib.box3(boxl)
    if nneighpos is not None and nneighpos != len(nb):
        raise ValueError("nneighpos does not match neighpos")
    n = len(nb)
    out = np.zeros((n, n), dtype=np.float64)
    _lib.check(_lib.load().wol_tetra_cos_ang(_lib.default_context().handle, _lib.ptr(ref), _lib.ptr(nb), n,
                                             _lib.ptr(box), _lib.ptr(out)))
    return np.asfortranarray(out)


def lsidists(refpos, neighpos, boxl, nneighpos=None):
    """dist1 = lsidists(refpos,neighpos,boxl,[nneighpos]) -- waterlib.f90:900-918"""
    ref = _vec3(refpos, "refpos"); nb = _pos(neighpos, "neighpos"); box = _lib.box3(boxl)
    if nneighpos is not None and nneighpos != len(nb):
        raise ValueError("nneighpos does not match neighpos")
    out = np.empty(len(nb), dtype=np.float64)
    _lib.check(_lib.load().wol_lsi_dists(_lib.default_context().handle, _lib.ptr(ref), _lib.ptr(nb), len(nb),
                                         _lib.ptr(box), _lib.ptr(out)))
    return out


def willarddensityfield(pos, gridx, gridy, gridz, boxl, smoothlen, npos=None, nx=None, ny=None, nz=None):
    """densvals,densnorms = willarddensityfield(pos,gridx,gridy,gridz,boxl,smoothlen,[npos,nx,ny,nz])
    -- waterlib.f90:1286-1341 (the call of surface_library.densityGrid, :197-199).  Fortran-ordered
    (nx,ny,nz) and (nx,ny,nz,3) arrays; the grid vectors may have shape (n,) or (1,n) as f2py accepts."""
    pos = _pos(pos, "pos"); box = _lib.box3(boxl)
    gx = np.ascontiguousarray(np.asarray(gridx, dtype=np.float64).reshape(-1))
    gy = np.ascontiguousarray(np.asarray(gridy, dtype=np.float64).reshape(-1))
    gz = np.ascontiguousarray(np.asarray(gridz, dtype=np.float64).reshape(-1))
    dv = np.empty((len(gx), len(gy), len(gz)), dtype=np.float64, order="F")
    dn = np.empty((len(gx), len(gy), len(gz), 3), dtype=np.float64, order="F")
    _lib.check(_lib.load().wol_willard_density_field(_lib.default_context().handle, _lib.ptr(pos), len(pos),
                                                     _lib.ptr(gx), len(gx), _lib.ptr(gy), len(gy), _lib.ptr(gz),
                                                     len(gz), _lib.ptr(box), float(smoothlen), _lib.ptr(dv),
                                                     _lib.ptr(dn)))
    return dv, dn


def willarddensitypoints(pos, denspts, boxl, smoothlen, npos=None, npts=None):
    """densvals,densnorms = willarddensitypoints(pos,denspts,boxl,smoothlen,[npos,npts]) -- waterlib.f90:1351-1398"""
    pos = _pos(pos, "pos"); pts = _pos(denspts, "denspts"); box = _lib.box3(boxl)
    dv = np.empty(len(pts), dtype=np.float64)
    dn = np.empty((len(pts), 3), dtype=np.float64)
    _lib.check(_lib.load().wol_willard_density_points(_lib.default_context().handle, _lib.ptr(pos), len(pos),
                                                      _lib.ptr(pts), len(pts), _lib.ptr(box), float(smoothlen),
                                                      _lib.ptr(dv), _lib.ptr(dn)))
    return dv, np.asfortranarray(dn)


def radialdistsame(pos, binwidth, totbins, bulkdens, boxl, npos=None):
    """rdf = radialdistsame(pos,binwidth,totbins,bulkdens,boxl,[npos]) -- waterlib.f90:316-357"""
    pos = _pos(pos, "pos"); box = _lib.box3(boxl)
    rdf = np.empty(int(totbins), dtype=np.float64)
    _lib.check(_lib.load().wol_radial_dist_same(_lib.default_context().handle, _lib.ptr(pos), len(pos), float(binwidth),
                                                int(totbins), float(bulkdens), _lib.ptr(box), _lib.ptr(rdf)))
    return rdf


def radialdist(pos1, pos2, binwidth, totbins, bulkdens, boxl, npos1=None, npos2=None):
    """rdf = radialdist(pos1,pos2,binwidth,totbins,bulkdens,boxl,[npos1,npos2]) -- waterlib.f90:193-231"""
    p1 = _pos(pos1, "pos1"); p2 = _pos(pos2, "pos2"); box = _lib.box3(boxl)
    rdf = np.empty(int(totbins), dtype=np.float64)
    _lib.check(_lib.load().wol_radial_dist(_lib.default_context().handle, _lib.ptr(p1), len(p1), _lib.ptr(p2), len(p2),
                                           float(binwidth), int(totbins), float(bulkdens), _lib.ptr(box), _lib.ptr(rdf)))
    return rdf


def pairdistancehistogram(pos1, pos2, binwidth, totbins, boxl, npos1=None, npos2=None, dim=None):
    """hist = pairdistancehistogram(pos1,pos2,binwidth,totbins,boxl,[npos1,npos2,dim]) -- waterlib.f90:363-389
    (three-dimensional positions only)"""
    p1 = _pos(pos1, "pos1"); p2 = _pos(pos2, "pos2"); box = _lib.box3(boxl)
    hist = np.empty(int(totbins), dtype=np.float64)
    _lib.check(_lib.load().wol_pair_distance_histogram(_lib.default_context().handle, _lib.ptr(p1), len(p1), _lib.ptr(p2),
                                                       len(p2), float(binwidth), int(totbins), _lib.ptr(box),
                                                       _lib.ptr(hist)))
    return hist


def _oh(opos, hpos):
    o = _pos(opos, "opos"); h = _pos(hpos, "hpos")
    if len(h) != 2 * len(o):   # the Fortran prints this and stops the interpreter (waterlib.f90:989-992)
        raise ValueError("Number of hydrogens must be two times number of oxygens.")
    return o, h


def watorient(opos, hpos, refvec, boxl, nopos=None, nhpos=None):
    """angdip,angplane = watorient(opos,hpos,refvec,boxl,[nopos,nhpos]) -- waterlib.f90:973-1011"""
    o, h = _oh(opos, hpos); ref = _vec3(refvec, "refvec"); box = _lib.box3(boxl)
    ad = np.empty(len(o), dtype=np.float64); ap = np.empty(len(o), dtype=np.float64)
    _lib.check(_lib.load().wol_wat_orient(_lib.default_context().handle, _lib.ptr(o), _lib.ptr(h), len(o), _lib.ptr(ref),
                                          _lib.ptr(box), _lib.ptr(ad), _lib.ptr(ap)))
    return ad, ap


def watohvec(opos, hpos, boxl, nopos=None, nhpos=None):
    """normdip = watohvec(opos,hpos,boxl,[nopos,nhpos]) -- waterlib.f90:1019-1044, (n,3) Fortran order"""
    o, h = _oh(opos, hpos); box = _lib.box3(boxl)
    out = np.empty((len(o), 3), dtype=np.float64)
    _lib.check(_lib.load().wol_wat_oh_vec(_lib.default_context().handle, _lib.ptr(o), _lib.ptr(h), len(o), _lib.ptr(box),
                                          _lib.ptr(out)))
    return np.asfortranarray(out)


def _hb_args(acceptorpos, donorpos, donorhpos, boxl):
    a = _pos(acceptorpos, "acceptorpos"); d = _pos(donorpos, "donorpos"); h = _pos(donorhpos, "donorhpos")
    if len(d) != len(h):   # the Fortran prints this and stops the interpreter (waterlib.f90:1173-1176)
        raise ValueError("Number of donor hydrogens and heavy-atoms do not match.")
    return a, d, h, _lib.box3(boxl)


def generalhbonds(acceptorpos, donorpos, donorhpos, boxl, distcut, angcut, nacc=None, ndon=None, ndonh=None):
    """bondbool = generalhbonds(acceptorpos,donorpos,donorhpos,boxl,distcut,angcut,[nacc,ndon,ndonh]) --
    waterlib.f90:1156-1210; (nacc, ndon) Fortran-ordered int32 like f2py's logical arrays"""
    a, d, h, box = _hb_args(acceptorpos, donorpos, donorhpos, boxl)
    out = np.zeros((len(a), len(d)), dtype=np.int32, order="F")
    _lib.check(_lib.load().wol_general_hbonds(_lib.default_context().handle, _lib.ptr(a), len(a), _lib.ptr(d), _lib.ptr(h),
                                              len(d), _lib.ptr(box), float(distcut), float(angcut), _lib.ptr(out)))
    return out


def generalhbonds_list(acceptorpos, donorpos, donorhpos, boxl, distcut, angcut):
    """(acc_idx, don_idx) = np.where(generalhbonds(...)) without ever forming the mask (cell list on the donors)"""
    a, d, h, box = _hb_args(acceptorpos, donorpos, donorhpos, boxl)
    cap = max(1024, 4 * len(a))
    while True:
        ai = np.empty(cap, dtype=np.int32); di = np.empty(cap, dtype=np.int32); cnt = C.c_int64(0)
        _lib.check(_lib.load().wol_general_hbonds_list(_lib.default_context().handle, _lib.ptr(a), len(a), _lib.ptr(d),
                                                       _lib.ptr(h), len(d), _lib.ptr(box), float(distcut), float(angcut),
                                                       cap, _lib.ptr(ai), _lib.ptr(di), C.byref(cnt)))
        if cnt.value <= cap:
            return ai[:cnt.value].astype(np.int64), di[:cnt.value].astype(np.int64)
        cap = int(cnt.value)
