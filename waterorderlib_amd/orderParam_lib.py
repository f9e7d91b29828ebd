"""Drop-in for the hot-path drivers of the reference's structureLibs/orderParam_lib.py:
``tetOrderCalc`` (:1426-1503), ``threeBodyCalc`` (:1269-1424), ``lsiCalc`` (:1586-1663) with
``blockAverage`` (:394-417) / ``getCI`` (:387-391).

Same call: tetOrderCalc(topFile, trajFile, subInds=None, nPops=0, ...) -> (avgQ, varQ) and the same
side effect (qDistribution_<j>.txt, "%.3e").  ``topFile`` may be a TrajObject-like object (anything
with ``.traj`` and ``getWatInds()``: the array-backed waterorderlib_amd.TrajObject, or the
reference's own pytraj-backed class) instead of a path; paths need parmed/pytraj, which the
reference imports at TrajObject.py:10-11 and which are outside this build.

The frame loop itself moves to the GPU: all frames go through ONE wol_q_traj call (frames batched
in HBM, nothing per-water on the host); sub-populations use wol_q_frame per frame.
"""
import numpy as np

from . import engine
from . import water_properties as wp


def getCI(means):
    """orderParam_lib.py:387-391"""
    meanCI = means[int(0.5 * len(means))]
    upperCI = means[int(0.975 * len(means))] - meanCI
    lowerCI = meanCI - means[int(0.025 * len(means))]
    return max(upperCI, lowerCI)


def blockAverage(vals, nBlocks=20, rng=None):
    """orderParam_lib.py:394-417: 20 block means, 10 000 bootstrap resamples, CI half-width.
    The reference draws from the unseeded global np.random; pass ``rng`` (np.random.Generator) for
    a reproducible CI."""
    vals = np.asarray(vals)
    obsBlocks = np.zeros(nBlocks)
    lenBlock = len(vals) / nBlocks
    for i in range(nBlocks):
        obsBlocks[i] = np.mean(vals[int(i * lenBlock):int((i + 1) * lenBlock)])
    nResamp = 10000
    if rng is None:
        draws = np.random.choice(obsBlocks, (nResamp, nBlocks))
    else:
        draws = rng.choice(obsBlocks, (nResamp, nBlocks))
    obsMeans = np.sort(draws.mean(axis=1))
    return getCI(obsMeans)


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


def tetOrderCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)',
                 stride=1, outPrefix='qDistribution_', rng=None, write=True):
    """orderParam_lib.py:1426-1503.  Returns (avgQ, varQ) = ([mean over frames, CI], [...]) per
    population (index 0 = all waters) and writes qDistribution_<j>.txt."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    traj = obj.traj
    watInds, watHInds, lenWat = obj.getWatInds()
    watInds = np.asarray(watInds)
    nT = len(traj)
    avgQ = [np.zeros(nT) for _ in range(nPops + 1)]
    varQ = [np.zeros(nT) for _ in range(nPops + 1)]
    qHist = [np.zeros(engine.QBINS, dtype=np.int64) for _ in range(nPops + 1)]

    # gather frames: the O rows are read on the GPU straight out of the interleaved frame when the
    # waters are a regular stride of the atom list (watInds = first + k*lenWat)
    xyz, boxes = [], []
    for frame in traj:
        xyz.append(np.asarray(frame.xyz))
        boxes.append(np.asarray(frame.box.values[:3], dtype=np.float64))
    boxes = np.array(boxes).reshape(-1, 3)
    regular = len(watInds) > 0 and lenWat > 0 and np.array_equal(watInds, watInds[0] + lenWat * np.arange(len(watInds)))
    if nT:
        if regular:
            frames = np.stack([x[watInds[0]: watInds[0] + lenWat * len(watInds)] for x in xyz])
            res = engine.tet_order_traj(frames, boxes, len_wat=lenWat, want_dhist=False)
        else:
            frames = np.stack([x[watInds] for x in xyz])
            res = engine.tet_order_traj(frames, boxes, len_wat=1, want_dhist=False)
        avgQ[0][:] = res.mean
        varQ[0][:] = res.var
        qHist[0] += res.qhist
    for t in range(nT):
        if nPops:
            watPos = np.ascontiguousarray(xyz[t][watInds], dtype=np.float64)
            for j in range(1, nPops + 1):
                subPos = np.asarray(xyz[t], dtype=np.float64)[subInds[t][j - 1]]
                jQ = wp.getOrderParamq(subPos, watPos, boxes[t])
                avgQ[j][t] = np.mean(jQ)
                varQ[j][t] = np.var(jQ)
                h, _ = np.histogram(jQ, bins=engine.QBINS, range=[0.0, 1.0], density=False)
                qHist[j] += h

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


def threeBodyCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)',
                  nBins=500, stride=1, output2D=False, outPrefix='3bDistribution_', rng=None, write=True):
    """orderParam_lib.py:1269-1424: three-body angle distribution (wp.getCosAngs, 3.413 A shell) per
    frame and population, tetrahedralMetrics statistics with block-average CIs, and the pooled
    histogram written to 3bDistribution_<j>.txt.  Returns (pTet, avgCos, varCos, entropy, nWats), each
    [mean, CI] over the populations.  The matplotlib (theta, Nc) plot of output2D=True is outside this
    build's scope and is skipped."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    xyz, boxes, watInds = _frames_of(obj)
    nT = len(xyz)
    stats = {k: [np.zeros(nT) for _ in range(nPops + 1)] for k in ("nWats", "pTet", "avgCos", "varCos", "entropy")}
    angles = [[] for _ in range(nPops + 1)]
    for t in range(nT):
        watPos = np.ascontiguousarray(xyz[t][watInds])
        for j in range(nPops + 1):
            sub = watPos if j == 0 else xyz[t][subInds[t][j - 1]]
            jAngles, _ = wp.getCosAngs(sub, watPos, boxes[t])
            angles[j].append(jAngles)
            stats["nWats"][j][t] = len(watInds) if j == 0 else len(subInds[t][j - 1])
            if len(jAngles):
                _, _, a, b, c_, d = wp.tetrahedralMetrics(jAngles, nBins=nBins)
                stats["pTet"][j][t] = a; stats["avgCos"][j][t] = b
                stats["varCos"][j][t] = c_; stats["entropy"][j][t] = d
    out = {k: _mean_ci(v, rng) for k, v in stats.items()}
    pooled = []
    for j in range(nPops + 1):
        allj = np.concatenate(angles[j]) if angles[j] else np.array([])
        pooled.append(allj)
        if write and len(allj):
            angDist, bins, *_ = wp.tetrahedralMetrics(allj, nBins=nBins)
            np.savetxt(outPrefix + str(j) + '.txt', np.stack([0.5 * (bins[:-1] + bins[1:]), angDist], axis=1),
                       header='3-body angle (deg)    frequency', fmt="%.3e")
    threeBodyCalc.last_angles = pooled
    return out["pTet"], out["avgCos"], out["varCos"], out["entropy"], out["nWats"]


def lsiCalc(topFile, trajFile=None, subInds=None, nPops=0, solResName='(!:WAT)', watResName='(:WAT)', stride=1,
            outPrefix='lsiDistribution_', rng=None, write=True):
    """orderParam_lib.py:1586-1663: local structure index (wp.getLSI) per frame and population, mean and
    variance with block-average CIs, pooled 500-bin histogram on [0, 0.3] A^2 written to
    lsiDistribution_<j>.txt.  Returns (avgLSI, varLSI), each [mean, CI]."""
    obj = _as_traj_object(topFile, trajFile, stride, solResName, watResName)
    xyz, boxes, watInds = _frames_of(obj)
    nT = len(xyz)
    avgLSI = [np.zeros(nT) for _ in range(nPops + 1)]
    varLSI = [np.zeros(nT) for _ in range(nPops + 1)]
    vals = [[] for _ in range(nPops + 1)]
    for t in range(nT):
        watPos = np.ascontiguousarray(xyz[t][watInds])
        for j in range(nPops + 1):
            sub = watPos if j == 0 else xyz[t][subInds[t][j - 1]]
            jLSI, _ = wp.getLSI(sub, watPos, boxes[t], lowCut=0.0, highCut=3.7)
            vals[j].append(jLSI)
            avgLSI[j][t] = np.mean(jLSI); varLSI[j][t] = np.var(jLSI)
    pooled = []
    for j in range(nPops + 1):
        allj = np.concatenate(vals[j]) if vals[j] else np.array([])
        pooled.append(allj)
        if write:
            lsiDist, bins = np.histogram(allj, bins=500, range=[0.0, 0.3], density=False)
            np.savetxt(outPrefix + str(j) + '.txt', np.stack([0.5 * (bins[:-1] + bins[1:]), lsiDist], axis=1),
                       header='lsiVal [A^2]    frequency', fmt="%.3e")
    lsiCalc.last_values = pooled
    return _mean_ci(avgLSI, rng), _mean_ci(varLSI, rng)
