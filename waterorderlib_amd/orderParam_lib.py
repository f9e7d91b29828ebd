"""Drop-in for the hot-path drivers of the reference's structureLibs/orderParam_lib.py:
``tetOrderCalc`` (:1426-1503), ``threeBodyCalc`` (:1269-1424), ``lsiCalc`` (:1586-1663) with
``blockAverage`` (:394-417) / ``getCI`` (:387-391).

Same call: tetOrderCalc(topFile, trajFile, subInds=None, nPops=0, ...) -> (avgQ, varQ) and the same
side effect (qDistribution_<j>.txt, "%.3e").  ``topFile`` may be a TrajObject-like object (anything
with ``.traj`` and ``getWatInds()``: the array-backed waterorderlib_amd.TrajObject, or the
reference's own pytraj-backed class) instead of a path; paths need parmed/pytraj, which the
reference imports at TrajObject.py:10-11 and which are outside this build.

The frame loops themselves move to the GPU: the trajectory is read lazily in chunks, every chunk is one call of
wol_q_traj(_pops) / wol_cos_angs_traj / wol_lsi_traj (frames batched in HBM, the copy of one batch under the kernels of
the previous, nothing per-water on the host); sub-populations ride along as position lists.
"""
import numpy as np

from . import engine
from . import water_properties as wp


def getCI(means):
    """orderParam_lib.py:387-391: half-width of the 95 % interval of a SORTED sample about its median, the larger of
    the two sides (entries at the 2.5 %, 50 % and 97.5 % positions, positions truncated as the reference's int())."""
    n = len(means)
    lo, mid, hi = means[int(0.025 * n)], means[int(0.5 * n)], means[int(0.975 * n)]
    return max(hi - mid, mid - lo)


def blockAverage(vals, nBlocks=20, rng=None):
    """orderParam_lib.py:394-417: the series is cut into nBlocks consecutive blocks (edges at int(i * len / nBlocks)),
    the block means are resampled with replacement 10 000 times, and getCI of the sorted resample means is returned.
    The reference draws from the unseeded global np.random (its CI is not reproducible); pass ``rng``
    (np.random.Generator) for a reproducible one."""
    vals = np.asarray(vals)
    width = len(vals) / nBlocks
    edges = [int(i * width) for i in range(nBlocks + 1)]
    blocks = np.array([np.mean(vals[a:b]) for a, b in zip(edges[:-1], edges[1:])])
    draw = np.random.choice if rng is None else rng.choice
    resampled = draw(blocks, (10000, nBlocks)).mean(axis=1)
    return getCI(np.sort(resampled))


def _as_traj_object(topFile, trajFile, stride, solResName, watResName):
    if hasattr(topFile, "traj") and hasattr(topFile, "getWatInds"):
        return topFile
    try:
        import parmed  # noqa: F401
        import pytraj  # noqa: F401
    except ImportError as e:
        raise ImportError("tetOrderCalc(topFile=<path>) needs parmed/pytraj as the reference does "
                          "(TrajObject.py:10-11); pass a TrajObject-like object instead") from e
    raise NotImplementedError("file-backed trajectories: wrap them in the reference's TrajObject and pass the object")


# host bytes of frames handed to one wol_q_traj call by the streaming drivers (the library cuts them into batches
# and overlaps the copy of one batch with the kernels of the previous one)
CHUNK_BYTES = 1 << 30


def _frame_chunks(traj, select=None, chunk_bytes=None):
    """Lazily yields (t0, xyz[k, m, 3], boxes[k, 3]) over a trajectory: frames are taken from the iterator as they
    are needed (pytraj.iterload reads from disk on demand, orderParam_lib.py:1458) and only one chunk is held.
    ``select`` picks the atoms kept of every frame: None (all), a slice, or an index array."""
    chunk_bytes = CHUNK_BYTES if chunk_bytes is None else chunk_bytes
    sel = slice(None) if select is None else select
    buf, boxes, t0, per, k = None, [], 0, 1, 0
    for t, frame in enumerate(traj):
        x = np.asarray(frame.xyz)[sel]
        if buf is None:
            dt = np.float32 if x.dtype == np.float32 else np.float64
            per = max(1, min(4096, int(chunk_bytes // max(x.shape[0] * 3 * np.dtype(dt).itemsize, 1))))
            buf = np.empty((per,) + x.shape, dtype=dt)
        buf[k] = x
        boxes.append(np.asarray(frame.box.values[:3], dtype=np.float64))
        k += 1
        if k == per:
            yield t0, buf[:k], np.array(boxes).reshape(-1, 3)
            t0, k, boxes = t + 1, 0, []
    if k:
        yield t0, buf[:k], np.array(boxes).reshape(-1, 3)


def tetOrderCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)',
                 stride=1, outPrefix='qDistribution_', rng=None, write=True, chunk_bytes=None):
    """orderParam_lib.py:1426-1503.  Returns (avgQ, varQ) = ([mean over frames, CI], [...]) per
    population (index 0 = all waters) and writes qDistribution_<j>.txt.

    The frame loop (opl:1458-1480) runs on the GPU: the trajectory is consumed lazily in chunks of at most
    ``chunk_bytes`` of host memory (default 1 GiB), each chunk is one wol_q_traj call; the integer histogram and the
    per-frame mean / variance series accumulate across chunks, so memory does not grow with the trajectory."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    traj = obj.traj
    watInds, watHInds, lenWat = obj.getWatInds()
    watInds = np.asarray(watInds)
    nW = len(watInds)
    # the O rows are read on the GPU straight out of the interleaved frame when the waters are a regular stride of
    # the atom list (watInds = first + k*lenWat); otherwise they are gathered on the host (pos[watInds], opl:1462)
    regular = nW > 0 and lenWat > 0 and np.array_equal(watInds, watInds[0] + lenWat * np.arange(nW))
    span = slice(int(watInds[0]), int(watInds[0]) + lenWat * nW) if regular else None
    avgQ = [[] for _ in range(nPops + 1)]
    varQ = [[] for _ in range(nPops + 1)]
    qHist = [np.zeros(engine.QBINS, dtype=np.int64) for _ in range(nPops + 1)]
    if nPops:
        select = None            # sub-populations index the whole frame
    else:
        select = span if regular else watInds
    for t0, xyz, boxes in (_frame_chunks(traj, select, chunk_bytes) if nW else ()):
        if nPops:
            # one call per chunk: q of all waters and of every population's points (pos[subInds[t][j]], opl:1466) against
            # the frame's waters -- wol_q_traj_pops sorts every frame once for both
            waters = np.ascontiguousarray(xyz[:, span] if regular else xyz[:, watInds])
            pop_pos, pop_off = engine.pack_populations(xyz, [subInds[t0 + k] for k in range(len(xyz))], nPops)
            res = engine.tet_order_traj_pops(waters, boxes, pop_pos, pop_off, nPops, len_wat=lenWat if regular else 1)
            for j in range(1, nPops + 1):
                avgQ[j].extend(res.pop_mean[:, j - 1]); varQ[j].extend(res.pop_var[:, j - 1])
                qHist[j] += res.pop_qhist[j - 1]
        else:
            res = engine.tet_order_traj(xyz, boxes, len_wat=lenWat if regular else 1, want_dhist=False)
        avgQ[0].extend(res.mean); varQ[0].extend(res.var)
        qHist[0] += res.qhist
    avgQ = [np.asarray(a, dtype=np.float64) for a in avgQ]
    varQ = [np.asarray(v, dtype=np.float64) for v in varQ]

    avgQ_mean = np.zeros(nPops + 1); avgQ_CI = np.zeros(nPops + 1)
    varQ_mean = np.zeros(nPops + 1); varQ_CI = np.zeros(nPops + 1)
    for j in range(nPops + 1):
        avgQ_CI[j] = blockAverage(avgQ[j][:], rng=rng); avgQ_mean[j] = np.mean(avgQ[j][:])
        varQ_CI[j] = blockAverage(varQ[j][:], rng=rng); varQ_mean[j] = np.mean(varQ[j][:])
    if write:
        bins = np.linspace(0.0, 1.0, engine.QBINS + 1)
        for j in range(nPops + 1):
            np.savetxt(outPrefix + str(j) + '.txt', np.stack([0.5 * (bins[:-1] + bins[1:]), qHist[j]], axis=1),
                       header='qVal    frequency', fmt="%.3e")
    tetOrderCalc.last_series = (avgQ, varQ, qHist)
    return [avgQ_mean, avgQ_CI], [varQ_mean, varQ_CI]


def _frames_of(obj):
    """(xyz list, boxes[T,3], watInds) from a TrajObject-like object."""
    watInds, _, _ = obj.getWatInds()
    xyz, boxes = [], []
    for frame in obj.traj:
        xyz.append(np.asarray(frame.xyz, dtype=np.float64))
        boxes.append(np.asarray(frame.box.values[:3], dtype=np.float64))
    return xyz, np.array(boxes).reshape(-1, 3), np.asarray(watInds)


def _mean_ci(series, rng):
    n = len(series)
    mean = np.zeros(n); ci = np.zeros(n)
    for j in range(n):
        ci[j] = blockAverage(series[j][:], rng=rng)
        mean[j] = np.mean(series[j][:])
    return [mean, ci]


def _shell_chunks(obj, subInds, nPops, chunk_bytes):
    """(t0, waters[k, n or n*lenWat, 3], boxes, len_wat, pop_pos, pop_off) per chunk of a trajectory, for the shell drivers"""
    watInds, _, lenWat = obj.getWatInds()
    watInds = np.asarray(watInds)
    nW = len(watInds)
    regular = nW > 0 and lenWat > 0 and np.array_equal(watInds, watInds[0] + lenWat * np.arange(nW))
    span = slice(int(watInds[0]), int(watInds[0]) + lenWat * nW) if regular else None
    select = None if nPops else (span if regular else watInds)
    for t0, xyz, boxes in (_frame_chunks(obj.traj, select, chunk_bytes) if nW else ()):
        if nPops:
            waters = np.ascontiguousarray(xyz[:, span] if regular else xyz[:, watInds])
            pop_pos, pop_off = engine.pack_populations(xyz, [subInds[t0 + k] for k in range(len(xyz))], nPops)
        else:
            waters, pop_pos, pop_off = xyz, None, None
        yield t0, waters, boxes, (lenWat if regular else 1), pop_pos, pop_off


def threeBodyCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)',
                  nBins=500, stride=1, output2D=False, outPrefix='3bDistribution_', rng=None, write=True, chunk_bytes=None):
    """orderParam_lib.py:1269-1424: three-body angle distribution (wp.getCosAngs, 3.413 A shell) per
    frame and population, tetrahedralMetrics statistics with block-average CIs, and the pooled
    histogram written to 3bDistribution_<j>.txt.  Returns (pTet, avgCos, varCos, entropy, nWats), each
    [mean, CI] over the populations.

    The frame loop (opl:1312-1353) runs on the GPU: every chunk of the trajectory is one wol_cos_angs_traj call, which
    reduces each frame's angles to what tetrahedralMetrics keeps of them (500-bin histogram, count, count and cosine sums
    of the angles in [100, 120]); the angles themselves never leave the device.  The matplotlib (theta, Nc) plot of
    output2D=True is outside this build's scope and is skipped."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    watInds = np.asarray(obj.getWatInds()[0])
    stats = {k: [[] for _ in range(nPops + 1)] for k in ("nWats", "pTet", "avgCos", "varCos", "entropy")}
    pooled = [np.zeros(nBins, dtype=np.int64) for _ in range(nPops + 1)]
    for t0, waters, boxes, len_wat, pop_pos, pop_off in _shell_chunks(obj, subInds, nPops, chunk_bytes):
        res = engine.cos_angs_traj(waters, boxes, len_wat=len_wat, pop_pos=pop_pos, pop_off=pop_off, npops=nPops, nbins=nBins)
        for k in range(len(waters)):
            for j in range(nPops + 1):
                _, a, b, c_, d = res.metrics(k, j)
                stats["nWats"][j].append(len(watInds) if j == 0 else len(subInds[t0 + k][j - 1]))
                stats["pTet"][j].append(a); stats["avgCos"][j].append(b)
                stats["varCos"][j].append(c_); stats["entropy"][j].append(d)
                pooled[j] += res.hist[k, j]
    stats = {k: [np.asarray(x, dtype=np.float64) for x in v] for k, v in stats.items()}
    out = {k: _mean_ci(v, rng) for k, v in stats.items()}
    bins = np.linspace(0.0, 180.0, nBins + 1)
    for j in range(nPops + 1):
        if write and pooled[j].sum():
            np.savetxt(outPrefix + str(j) + '.txt', np.stack([0.5 * (bins[:-1] + bins[1:]), pooled[j]], axis=1),
                       header='3-body angle (deg)    frequency', fmt="%.3e")
    threeBodyCalc.last_hist = pooled
    threeBodyCalc.last_series = stats
    return out["pTet"], out["avgCos"], out["varCos"], out["entropy"], out["nWats"]


def lsiCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)', stride=1,
            outPrefix='lsiDistribution_', rng=None, write=True, chunk_bytes=None):
    """orderParam_lib.py:1586-1663: local structure index (wp.getLSI) per frame and population, mean and
    variance with block-average CIs, pooled 500-bin histogram on [0, 0.3] A^2 written to
    lsiDistribution_<j>.txt.  Returns (avgLSI, varLSI), each [mean, CI].

    The frame loop (opl:1618-1640) runs on the GPU, one wol_lsi_traj call per chunk of the trajectory."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    avgLSI = [[] for _ in range(nPops + 1)]
    varLSI = [[] for _ in range(nPops + 1)]
    pooled = np.zeros((nPops + 1, 500), dtype=np.int64)
    for t0, waters, boxes, len_wat, pop_pos, pop_off in _shell_chunks(obj, subInds, nPops, chunk_bytes):
        res = engine.lsi_traj(waters, boxes, lowCut=0.0, highCut=3.7, len_wat=len_wat, pop_pos=pop_pos, pop_off=pop_off, npops=nPops)
        pooled += res.hist
        for j in range(nPops + 1):
            avgLSI[j].extend(res.mean[:, j]); varLSI[j].extend(res.var[:, j])
    avgLSI = [np.asarray(a, dtype=np.float64) for a in avgLSI]
    varLSI = [np.asarray(v, dtype=np.float64) for v in varLSI]
    bins = np.linspace(0.0, 0.3, 501)
    for j in range(nPops + 1):
        if write:
            np.savetxt(outPrefix + str(j) + '.txt', np.stack([0.5 * (bins[:-1] + bins[1:]), pooled[j]], axis=1),
                       header='lsiVal [A^2]    frequency', fmt="%.3e")
    lsiCalc.last_hist = pooled
    lsiCalc.last_series = (avgLSI, varLSI)
    return _mean_ci(avgLSI, rng), _mean_ci(varLSI, rng)


def rdfCalc(topFile, trajFile=None, solResName='(!:WAT)', watResName='(:WAT)', binwidth=0.1, totbins=150, stride=1,
            write=True):
    """orderParam_lib.py:575-727: Ow-Ow, solute-Ow and solute-solute radial distribution functions (wl.radialdistsame /
    wl.radialdist per frame with bulkdens = 1, averaged over five trajectory chunks), running coordination numbers by
    Simpson integration, first-minimum coordination n1 and the translational parameter; writes rdf.txt and coord.txt.

    Returns ([n1_OwOw, se], [n1_SolOw, se], [tParam, se]) with solute atoms, else (n1_OwOw, tParam) -- the reference
    returns its frame counter `t` in that branch (:727), an obvious slip.  scipy's `simps` of the reference is
    today's `simpson`, whose even-sample rule changed in scipy 1.11: the integrals are parity-unpinned, the g(r)
    arrays (rdfCalc.last_rdfs) are what the parity tests check."""
    from scipy.integrate import simpson
    from scipy.signal import argrelmin
    from . import waterlib as wl
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    xyz, boxes, watInds = _frames_of(obj)
    solInds = np.asarray(obj.getSolInds()[0], dtype=int)
    has_sol = len(solInds) > 0
    nChunks = 5
    chunkSize = int(len(xyz) / nChunks)
    if chunkSize < 1:
        raise ValueError("rdfCalc needs at least %d frames (five chunks, as the reference)" % nChunks)
    dist = np.linspace(0, (totbins - 1) * binwidth, totbins) + binwidth
    tot = {k: [] for k in ("OwOw", "SolOw", "SolSol")}
    totc = {k: [] for k in ("OwOw", "SolOw", "SolSol")}
    tot_n1_OwOw, tot_n1_SolOw, tot_tParam = [], [], []
    # the water-water g(r) of every frame the five chunks use, in one trajectory call (wol_radial_dist_same_traj): row f is what
    # wl.radialdistsame returns for frame f; the sums below run in the reference's order
    nUsed = nChunks * chunkSize
    owow = engine.radial_dist_same_traj(np.stack([xyz[f][watInds] for f in range(nUsed)]), boxes[:nUsed], binwidth, totbins, 1.0)
    for c in range(nChunks):
        rdf = {k: np.zeros(totbins) for k in tot}
        t = 0
        for t, f in enumerate(range(int(c * chunkSize), int((c + 1) * chunkSize))):
            thisWat = np.ascontiguousarray(xyz[f][watInds])
            rdf["OwOw"] = rdf["OwOw"] + owow[f]
            if has_sol:
                thisSol = np.ascontiguousarray(xyz[f][solInds])
                rdf["SolSol"] = rdf["SolSol"] + wl.radialdistsame(thisSol, binwidth, totbins, 1.0, boxes[f])
                rdf["SolOw"] = rdf["SolOw"] + wl.radialdist(thisSol, thisWat, binwidth, totbins, 1.0, boxes[f])
        coord = {k: np.zeros(len(dist) - 2) for k in tot}
        for k in tot:
            rdf[k] = rdf[k] / (1.0 + t)
            tot[k].append(rdf[k])
        for j in range(2, len(dist)):
            coord["OwOw"][j - 2] = 8.0 * np.pi * simpson(rdf["OwOw"][:j] * dist[:j] ** 2.0, x=dist[:j])
            if has_sol:
                coord["SolOw"][j - 2] = 4.0 * np.pi * simpson(rdf["SolOw"][:j] * dist[:j] ** 2.0, x=dist[:j])
                coord["SolSol"][j - 2] = 8.0 * np.pi * simpson(rdf["SolSol"][:j] * dist[:j] ** 2.0, x=dist[:j])
        for k in tot:
            totc[k].append(coord[k])
        if has_sol:
            tot_n1_SolOw.append(coord["SolOw"][argrelmin(rdf["SolOw"])[0][0] - 2])
        m0 = argrelmin(rdf["OwOw"])[0][0]
        tot_n1_OwOw.append(coord["OwOw"][m0 - 2])
        tot_tParam.append(simpson(rdf["OwOw"][:m0] / rdf["OwOw"][-1], x=dist[:m0]) / dist[m0])
    arr = {k: np.array(v) for k, v in tot.items()}
    arrc = {k: np.array(v) for k, v in totc.items()}
    se = lambda a: np.std(a, axis=0, ddof=1) / np.sqrt(nChunks - 1)
    rdfCalc.last_rdfs = arr
    if write:   # the reference writes the LAST chunk's curves next to the across-chunk standard errors (:703-716)
        np.savetxt('rdf.txt', np.stack([dist, arr["OwOw"][-1], se(arr["OwOw"]), arr["SolSol"][-1], se(arr["SolSol"]),
                                        arr["SolOw"][-1], se(arr["SolOw"])], axis=1),
                   header='pair distance (A)     Ow-Ow rdf     err     Sol-Sol rdf     err     Sol-Ow rdf     err', fmt="%.3e")
        np.savetxt('coord.txt', np.stack([dist[2:], arrc["OwOw"][-1], se(arrc["OwOw"]), arrc["SolSol"][-1],
                                          se(arrc["SolSol"]), arrc["SolOw"][-1], se(arrc["SolOw"])], axis=1),
                   header='pair distance (A)     Ow-Ow n1     err     Sol-Sol n1     err     Sol-Ow n1     err', fmt="%.3e")
    n1_OwOw, tParam = np.mean(tot_n1_OwOw), np.mean(tot_tParam)
    if has_sol:
        return ([n1_OwOw, np.std(np.array(tot_n1_OwOw), ddof=1) / np.sqrt(nChunks - 1)],
                [np.mean(tot_n1_SolOw), np.std(np.array(tot_n1_SolOw), ddof=1) / np.sqrt(nChunks - 1)],
                [tParam, np.std(np.array(tot_tParam), ddof=1) / np.sqrt(nChunks - 1)])
    return n1_OwOw, tParam


def getBoundWrap(pos, thisbox, watInds, lenWat, solInds, sol_acceptorOInds, sol_donorOInds, sol_donorHOInds,
                 cutoff=4.0, hbDist=3.0, hbAng=150.0):
    """orderParam_lib.py:419-572: bound / wrap / shell / non-shell water-oxygen indices of one frame.

    The reference derives the hydrogen-bond partner lists from a parmed topology (getHBInds, :46-120); an
    array-backed caller passes them: the solute's acceptor oxygens, and its donor oxygens listed once per bonded
    hydrogen next to those hydrogens' indices.  Waters are O,H1,H2[,EP] blocks of `lenWat` atoms, as getWatInds
    yields them, so a shell water with oxygen index o contributes acceptor o and donors (o, o+1), (o, o+2) -- what
    getHBInds returns for it.  Shell = waters within `cutoff` of any solute heavy atom (wl.nearneighbors, :497);
    bound = shell waters hydrogen-bonded to the solute in either direction (wl.generalhbonds, :519-560)."""
    from . import waterlib as wl
    pos = np.asarray(pos, dtype=np.float64)
    thisbox = np.asarray(thisbox, dtype=np.float64).reshape(-1)[:3]
    watInds = np.asarray(watInds, dtype=int)
    watPos = pos[watInds]
    solPos = pos[np.asarray(solInds, dtype=int)]
    neighbors = wl.nearneighbors(solPos, watPos, thisbox, 0.0, cutoff)
    mask = np.unique(np.where(neighbors == 1)[1])
    shellInds = watInds[mask]
    nonShellInds = np.delete(watInds, mask)
    wat_acceptorOPos = pos[shellInds]
    wat_donorOPos = pos[np.repeat(shellInds, 2)]
    wat_donorHOPos = pos[(np.repeat(shellInds, 2).reshape(-1, 2) + np.array([1, 2])).reshape(-1)] if lenWat >= 3 else np.zeros((0, 3))
    sol_acceptorOPos = pos[np.asarray(sol_acceptorOInds, dtype=int)]
    sol_donorOPos = pos[np.asarray(sol_donorOInds, dtype=int)]
    sol_donorHOPos = pos[np.asarray(sol_donorHOInds, dtype=int)]
    # water accepts from the solute (:541-545), solute accepts from water (:548-556)
    watSolHBs = wl.generalhbonds(wat_acceptorOPos, sol_donorOPos, sol_donorHOPos, thisbox, hbDist, hbAng)
    boundMask_wat = np.unique(np.where(watSolHBs == 1)[0])
    solWatHBs = wl.generalhbonds(sol_acceptorOPos, wat_donorOPos, wat_donorHOPos, thisbox, hbDist, hbAng)
    dummy = np.zeros(len(wat_donorOPos))
    dummy[np.unique(np.where(solWatHBs == 1)[1])] = 1
    boundMask_sol = np.where(np.ceil(0.5 * (dummy[0::2] + dummy[1::2])))[0]
    boundMask = np.sort(np.unique(np.concatenate([boundMask_wat, boundMask_sol]))).astype(int)
    m = np.ones(len(shellInds), dtype=bool)
    m[boundMask] = False
    return shellInds[boundMask], shellInds[m], shellInds, nonShellInds
