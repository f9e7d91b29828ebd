"""GPU parity tests proper: the HIP path (through the C ABI, via the reference-style Python
surface) against (i) the golden vectors the reference produced and (ii) the CPU oracle on seeded
inputs.  Bars (BASELINE.json north_star): neighbour indices bit-exact; q and angles within 1e-6
(fp64) -- asserted far tighter here (1e-9 deg on angles, 1e-12 on q) because only libm-level
differences (acos/cos) are expected; distances and integer histograms bit-exact."""
import numpy as np
import pytest

from conftest import Q_CASES, case_inputs, golden

pytestmark = pytest.mark.gpu

ANG_TOL = 1e-9   # degrees
Q_TOL = 1e-12


@pytest.fixture(scope="module")
def wp():
    from waterorderlib_amd import water_properties
    return water_properties


@pytest.fixture(scope="module")
def ctxs():
    """one context per search path: 0 auto, 1 all-pairs, 2 exact cell list, 3 fp32-filtered cell list"""
    from waterorderlib_amd import _lib
    out = {}
    for path in (0, 1, 2, 3):
        c = _lib.Context(0)
        c.set_option("path", path)
        out[path] = c
    yield out
    for c in out.values():
        c.close()


def check_full(res, g_q, g_idx, g_dist, g_ang):
    q, idx, dist, ang = res
    assert np.array_equal(idx, g_idx), "neighbour indices differ"
    assert np.array_equal(dist, g_dist, equal_nan=True), "neighbour distances differ"
    np.testing.assert_allclose(ang, g_ang, rtol=0, atol=ANG_TOL, equal_nan=True)
    np.testing.assert_allclose(q, g_q, rtol=0, atol=Q_TOL)


@pytest.mark.parametrize("path", [0, 1, 2, 3])
@pytest.mark.parametrize("name", Q_CASES)
def test_q_golden(wp, ctxs, name, path):
    g = golden(name)
    sub, pos, box, lo, hi = case_inputs(g)
    res = wp.getOrderParamq(sub, pos, box, lo, hi, full=True, ctx=ctxs[path])
    check_full(res, g["q"], g["nn_idx"], g["nn_dist"], g["ang"])
    # plain call returns q only, like the reference
    q = wp.getOrderParamq(sub, pos, box, lo, hi, ctx=ctxs[path])
    assert q.shape == (len(sub),)
    np.testing.assert_allclose(q, res[0], rtol=0, atol=1e-13)  # q-only calls may skip the acos/cos round trip


@pytest.mark.parametrize("path", [1, 2, 3])
@pytest.mark.parametrize("n,seed,cw", [(3000, 21, 5.0), (3000, 22, 3.1), (10000, 23, 5.0), (4097, 24, 7.7),
                                       (20000, 25, 4.2), (1500, 26, 5.0)])
def test_q_vs_oracle(wp, ctxs, oracle, n, seed, cw, path):
    from waterorderlib_amd import synth
    xyz, boxes = synth.water_frames(n, 1, seed=seed)
    pos, box = xyz[0], boxes[0]
    ctxs[path].set_option("cell_width", cw)
    res = wp.getOrderParamq(pos, pos, box, full=True, ctx=ctxs[path])
    ctxs[path].set_option("cell_width", 5.0)
    ref = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    check_full(res, *ref)


def test_q_gas_and_subsets_vs_oracle(wp, ctxs, oracle):
    from waterorderlib_amd import synth
    pos, box = synth.ideal_gas(6000, seed=31)
    box = box * np.array([1.0, 1.3, 0.8])
    pos = pos * np.array([1.0, 1.3, 0.8])
    ref = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    check_full(wp.getOrderParamq(pos, pos, box, full=True, ctx=ctxs[2]), *ref)
    check_full(wp.getOrderParamq(pos, pos, box, full=True, ctx=ctxs[3]), *ref)
    assert 0 < ctxs[3].counter("slow_waters") < len(pos)  # the gas has sparse spots the filter must hand over
    # q-only call (no angle outputs): q comes straight from the cosines, within 2e-14 of the round trip
    np.testing.assert_allclose(wp.getOrderParamq(pos, pos, box, ctx=ctxs[3]), ref[0], rtol=0, atol=1e-13)
    # sub-population branch through the cell path: rows of pos and arbitrary points
    rng = np.random.default_rng(5)
    rows = pos[rng.choice(len(pos), 700, replace=False)]
    check_full(wp.getOrderParamq(rows, pos, box, full=True, ctx=ctxs[2]),
               *oracle.getOrderParamq(rows, pos, box, full=True, cells=True))
    pts = rng.random((333, 3)) * box * 3.0 - box
    check_full(wp.getOrderParamq(pts, pos, box, full=True, ctx=ctxs[2]),
               *oracle.getOrderParamq(pts, pos, box, full=True, cells=True))
    # short cutoffs: most waters have fewer than four neighbours (180-degree padding, wp:379-384)
    for lo, hi in ((0.0, 3.0), (2.5, 3.6), (0.0, 17.0)):
        check_full(wp.getOrderParamq(pos[:2000], pos, box, lo, hi, full=True, ctx=ctxs[2]),
                   *oracle.getOrderParamq(pos[:2000], pos, box, lo, hi, full=True, cells=True))


def test_slab_with_vapour(wp, ctxs, oracle):
    """BASELINE config 5 geometry at test size: vacuum gap >> cutoff, isolated waters (q = 0)"""
    from waterorderlib_amd import synth
    xyz, boxes = synth.slab_frames(4000, 1, seed=41, n_vapour=60)
    pos, box = xyz[0], boxes[0]
    ref = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    assert (ref[1][:, 3] < 0).sum() > 0
    for path in (1, 2, 3):
        check_full(wp.getOrderParamq(pos, pos, box, full=True, ctx=ctxs[path]), *ref)


def test_analytic_and_edge_cases(wp, ctxs):
    tet = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float)
    pos = np.vstack([np.zeros(3), tet * 1.6]) + 15.0
    q, idx, dist, ang = wp.getOrderParamq(pos[:1], pos, np.array([30.0] * 3), full=True)
    assert abs(q[0] - 1.0) < 1e-13
    np.testing.assert_allclose(ang[0], np.degrees(np.arccos(-1.0 / 3.0)), atol=1e-9)
    lone = np.array([[1.0, 2.0, 3.0], [40.0, 40.0, 40.0]])
    q, idx, dist, ang = wp.getOrderParamq(lone, lone, np.array([90.0] * 3), full=True)
    assert np.all(q == 0.0) and np.all(idx == -1) and np.all(np.isnan(dist)) and np.all(np.isnan(ang))
    pair = np.array([[1.0, 2.0, 3.0], [2.0, 2.0, 3.0]])
    q, idx, dist, ang = wp.getOrderParamq(pair, pair, np.array([90.0] * 3), full=True)
    assert np.all(ang == 180.0) and np.all(np.abs(q) < 1e-15) and idx[0, 0] == 1 and idx[1, 0] == 0
    # empty inputs
    assert wp.getOrderParamq(np.zeros((0, 3)), pos, np.array([30.0] * 3)).shape == (0,)
    q = wp.getOrderParamq(pos, np.zeros((0, 3)), np.array([30.0] * 3))
    assert q.shape == (5,) and np.all(q == 0.0)
    # BoxDims of shape (1,3) is accepted like f2py does
    q2 = wp.getOrderParamq(pos[:1], pos, np.array([[30.0, 30.0, 30.0]]))
    assert abs(q2[0] - 1.0) < 1e-13
    with pytest.raises(ValueError):
        wp.getOrderParamq(np.zeros((4, 2)), pos, np.array([30.0] * 3))


def test_waterlib_level1(oracle):
    import waterorderlib_amd.waterlib as wl
    g = golden("waterlib_kat")
    ang = wl.tetracosang(g["tet_pos"][0], g["tet_pos"][1:], g["tet_box"])
    assert ang.flags.f_contiguous
    np.testing.assert_allclose(ang[np.triu_indices(4, k=1)], g["tet_ang"], rtol=0, atol=ANG_TOL)
    assert np.array_equal(wl.reimage(g["reimage_in"], np.zeros(3), np.array([10.0] * 3)), g["reimage_out"])
    assert np.array_equal(wl.lsidists(g["tet_pos"][0], g["tet_pos"][1:] + 40.0, g["tet_box"]), g["lsidists"])
    for name in ("gas512", "tiny_box", "nonperiodic_z", "subpop_points", "cut_2p9_4p0"):
        gg = golden(name)
        sub, pos, box, lo, hi = case_inputs(gg)
        m = wl.nearneighbors(sub, pos, box, lo, hi)
        assert m.dtype == np.int32 and m.flags.f_contiguous and m.shape == (len(sub), len(pos))
        assert np.array_equal(m, oracle.nearneighbors(sub, pos, box, lo, hi))
        assert int(m.sum()) == int(gg["mask_sum"])
        if "sub" not in gg.files:
            assert np.array_equal(wl.allnearneighbors(pos, box, lo, hi), m)


def test_traj_golden_and_oracle(oracle):
    from waterorderlib_amd import engine, synth
    g = golden("traj700x3")
    res = engine.tet_order_traj(g["frames"], g["boxes"], want_q=True)
    np.testing.assert_allclose(res.q, g["q"], rtol=0, atol=Q_TOL)
    np.testing.assert_allclose(res.mean, g["mean"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(res.var, g["var"], rtol=0, atol=1e-13)
    assert np.array_equal(res.qhist, g["qhist"])
    # a bigger batch against the oracle (cell path, several frames per launch, TIP4P-style frames)
    xyz, boxes = synth.water_frames(5000, 12, seed=77, len_wat=4)
    res = engine.tet_order_traj(xyz, boxes, len_wat=4, want_q=True)
    mean, var, qh, dh = oracle.tet_order_traj(np.ascontiguousarray(xyz[:, 0::4]), boxes)
    np.testing.assert_allclose(res.mean, mean, rtol=0, atol=1e-13)
    np.testing.assert_allclose(res.var, var, rtol=0, atol=1e-13)
    assert np.array_equal(res.qhist, qh) and np.array_equal(res.dhist, dh)
    assert res.dhist.sum() == 4 * 5000 * 12


def test_traj_lds_sort_batch(oracle):
    """>= 128 frames per batch switches the cell sort to the one-block-per-frame LDS kernel"""
    from waterorderlib_amd import engine, synth, _lib
    xyz, boxes = synth.water_frames(1200, 130, seed=79)
    a = engine.tet_order_traj(xyz, boxes, want_q=True)
    ctx = _lib.Context(0)
    ctx.set_option("lds_sort", 0)
    b = engine.tet_order_traj(xyz, boxes, want_q=True, ctx=ctx)
    assert np.array_equal(a.q, b.q) and np.array_equal(a.qhist, b.qhist) and np.array_equal(a.dhist, b.dhist)
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a.var, b.var)
    q = oracle.getOrderParamq(xyz[77], xyz[77], boxes[77])
    np.testing.assert_allclose(a.q[77], q, rtol=0, atol=Q_TOL)
    ctx.close()


def test_traj_device_resident_matches_host():
    import torch
    from waterorderlib_amd import engine, synth
    xyz, boxes = synth.water_frames(6000, 16, seed=78)
    host = engine.tet_order_traj(xyz, boxes, want_q=True)
    dev = engine.tet_order_traj(torch.from_numpy(xyz).cuda(), boxes, want_q=True)
    torch.cuda.synchronize()
    assert np.array_equal(dev.q.cpu().numpy(), host.q)
    assert np.array_equal(dev.mean.cpu().numpy(), host.mean) and np.array_equal(dev.var.cpu().numpy(), host.var)
    assert np.array_equal(dev.qhist.cpu().numpy(), host.qhist) and np.array_equal(dev.dhist.cpu().numpy(), host.dhist)
    # float32 storage: coordinates promoted exactly, same arithmetic => equals fp64 run on the rounded frames
    x32 = xyz.astype(np.float32)
    a = engine.tet_order_traj(x32, boxes, want_q=True)
    b = engine.tet_order_traj(x32.astype(np.float64), boxes, want_q=True)
    assert np.array_equal(a.qhist, b.qhist) and np.array_equal(a.dhist, b.dhist)
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a.q, b.q.astype(np.float32))


# ---- getCosAngs / getLSI ------------------------------------------------
@pytest.mark.parametrize("name", ["gas512", "ice512", "ortho2000_unwrapped", "sparse_clusters", "subpop_rows",
                                  "subpop_points", "tiny_box", "nonperiodic_z"])
def test_cos_angs_and_lsi_golden(wp, name):
    g = golden(name)
    sub, pos, box, _, _ = case_inputs(g)
    ca, cn = wp.getCosAngs(sub, pos, box)
    assert ca.shape == g["cos_ang"].shape and np.array_equal(cn, g["cos_num"])
    np.testing.assert_allclose(ca, g["cos_ang"], rtol=0, atol=ANG_TOL)
    h, _ = np.histogram(ca, bins=500, range=[0.0, 180.0])
    assert np.array_equal(h, g["ang_hist"])
    li, ln = wp.getLSI(sub, pos, box)
    assert li.shape == g["lsi"].shape and np.array_equal(ln, g["lsi_num"])
    np.testing.assert_allclose(li, g["lsi"], rtol=0, atol=1e-13)


def test_cos_angs_and_lsi_vs_oracle(wp, oracle):
    from waterorderlib_amd import synth
    xyz, boxes = synth.water_frames(4000, 1, seed=91)
    pos, box = xyz[0], boxes[0] * np.array([1.0, 1.0, 1.0])
    for lo, hi in ((0.0, 3.413), (2.6, 3.8)):
        ca, cn = wp.getCosAngs(pos, pos, box, lo, hi)
        ra, rn = oracle.getCosAngs(pos, pos, box, lo, hi)
        assert np.array_equal(cn, rn) and ca.shape == ra.shape
        np.testing.assert_allclose(ca, ra, rtol=0, atol=ANG_TOL)
    li, ln = wp.getLSI(pos, pos, box)
    ri, rn = oracle.getLSI(pos, pos, box)
    assert np.array_equal(ln, rn) and li.shape == ri.shape
    np.testing.assert_allclose(li, ri, rtol=0, atol=1e-13)
    sub = pos[::7] + 0.01
    li, ln = wp.getLSI(sub, pos, box, 0.0, 3.5)
    ri, rn = oracle.getLSI(sub, pos, box, 0.0, 3.5)
    assert np.array_equal(ln, rn)
    np.testing.assert_allclose(li, ri, rtol=0, atol=1e-13)
    # tetrahedralMetrics on top (host numpy, as in the reference)
    ca, _ = wp.getCosAngs(pos, pos, box)
    dist, bins, frac, avg, var, ent = wp.tetrahedralMetrics(ca)
    assert dist.sum() == len(ca) and 0.0 < frac < 1.0 and ent > 0.0
    # empty shells
    ca, cn = wp.getCosAngs(pos[:50], pos, box, 0.0, 0.5)
    assert len(ca) == 0 and np.all(cn == 0)


@pytest.mark.filterwarnings("ignore:Mean of empty slice", "ignore:invalid value encountered")
def test_drivers(oracle, tmp_path, monkeypatch):
    """tetOrderCalc / threeBodyCalc / lsiCalc over an array-backed TrajObject against the oracle"""
    from waterorderlib_amd import orderParam_lib as opl
    from waterorderlib_amd import synth
    from waterorderlib_amd.TrajObject import TrajObject
    monkeypatch.chdir(tmp_path)
    xyz, boxes = synth.water_frames(1500, 5, seed=95, len_wat=3)
    obj = TrajObject(xyz, boxes, lenWat=3)
    watInds, watH, lenWat = obj.getWatInds()
    assert lenWat == 3 and len(watInds) == 1500 and len(watH) == 3000
    rng = np.random.default_rng(0)
    subInds = [[watInds[rng.choice(1500, 200, replace=False)]] for _ in range(5)]
    avgQ, varQ = opl.tetOrderCalc(obj, subInds=subInds, nPops=1, rng=np.random.default_rng(1))
    series_avg, series_var, hists = opl.tetOrderCalc.last_series
    O = np.ascontiguousarray(xyz[:, 0::3])
    for t in range(5):
        q = oracle.getOrderParamq(O[t], O[t], boxes[t])
        assert abs(series_avg[0][t] - np.mean(q)) < 1e-13 and abs(series_var[0][t] - np.var(q)) < 1e-13
        qs = oracle.getOrderParamq(xyz[t][subInds[t][0]], O[t], boxes[t])
        assert abs(series_avg[1][t] - np.mean(qs)) < 1e-12
    assert abs(avgQ[0][0] - np.mean(series_avg[0])) < 1e-15 and avgQ[0][1] > 0.0
    # (with fewer than 20 frames the reference's 20-block CI is NaN; the mirror keeps that behaviour)
    tab = np.loadtxt("qDistribution_0.txt")
    assert tab.shape == (500, 2) and tab[:, 1].sum() == hists[0].sum()
    pTet, avgCos, varCos, ent, nW = opl.threeBodyCalc(obj, rng=np.random.default_rng(2))
    ang = np.concatenate([oracle.getCosAngs(O[t], O[t], boxes[t])[0] for t in range(5)])
    assert np.array_equal(opl.threeBodyCalc.last_hist[0], np.histogram(ang, bins=500, range=[0.0, 180.0])[0])
    for t in range(5):
        a_t = oracle.getCosAngs(O[t], O[t], boxes[t])[0]
        tet = a_t[(a_t >= 100.0) & (a_t <= 120.0)]
        assert abs(opl.threeBodyCalc.last_series["pTet"][0][t] - len(tet) / len(a_t)) < 1e-15
        assert abs(opl.threeBodyCalc.last_series["avgCos"][0][t] - np.mean(np.cos(tet * np.pi / 180.0))) < 1e-12
    assert nW[0][0] == 1500 and 0.0 < pTet[0][0] < 1.0 and np.loadtxt("3bDistribution_0.txt").shape == (500, 2)
    avgL, varL = opl.lsiCalc(obj, rng=np.random.default_rng(3))
    l0 = [oracle.getLSI(O[t], O[t], boxes[t])[0] for t in range(5)]
    assert abs(avgL[0][0] - np.mean([np.mean(x) for x in l0])) < 1e-13
    assert abs(varL[0][0] - np.mean([np.var(x) for x in l0])) < 1e-13
    assert np.array_equal(opl.lsiCalc.last_hist[0], np.histogram(np.concatenate(l0), bins=500, range=[0.0, 0.3])[0])
    assert np.loadtxt("lsiDistribution_0.txt").shape == (500, 2)


def test_two_stream_overlap_option_gives_identical_results():
    """opt-in: the sort of a batch on a second stream beside the previous batch's kernels"""
    import torch
    from waterorderlib_amd import engine, synth, _lib
    xyz, boxes = synth.water_frames(1500, 192, seed=81)
    fr = torch.from_numpy(xyz).cuda()
    a = engine.tet_order_traj(fr, boxes, want_q=True)
    ctx = _lib.Context(0)
    ctx.set_option("overlap", 1)
    ctx.set_option("inputs_resident", 1)
    ctx.set_option("frames_per_batch", 64)
    outs = [engine.tet_order_traj(fr, boxes, want_q=True, ctx=ctx) for _ in range(3)]
    torch.cuda.synchronize()
    for b in outs:
        assert torch.equal(a.q, b.q) and torch.equal(a.qhist, b.qhist) and torch.equal(a.dhist, b.dhist)
        assert torch.equal(a.mean, b.mean) and torch.equal(a.var, b.var)
    ctx.close()


def test_both_proposal_kernels_agree_and_match_oracle(wp, oracle):
    """the row-merged proposal kernel (default) and the 3-D tile one (fallback for frames above 2^20 waters) feed
    the same certification: identical results, full outputs included; odd box shapes exercise the tile edges"""
    from waterorderlib_amd import _lib, synth
    oracle.set_threads(8)
    cases = [(synth.water_frames(6000, 1, seed=91), None),
             (synth.water_frames(9000, 1, seed=92, box=np.array([93.0, 41.0, 70.5])), None),   # long rows, few rows
             (synth.water_frames(5000, 1, seed=93, box=np.array([14.2, 150.0, 70.0])), None),  # 3 cells along x
             (synth.slab_frames(8000, 1, seed=94, n_vapour=30), None)]
    for (xyz, boxes), _ in cases:
        pos, box = xyz[0], boxes[0]
        res = {}
        for rows in (1, 0):
            c = _lib.Context(0)
            c.set_option("path", 3)
            c.set_option("search_rows", rows)
            res[rows] = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
            c.close()
        q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
        for rows in (1, 0):
            check_full(res[rows], q_o, idx_o, dist_o, ang_o)
        for a, b in zip(res[0], res[1]):
            assert np.array_equal(a, b, equal_nan=True)


def test_npt_like_batch_with_changing_cell_counts(oracle):
    """a constant-pressure trajectory: the box breathes, and across 13.5 A x k the number of cells per axis changes
    from frame to frame inside one batch (different row tables, tile shapes and bin counts per frame)"""
    import torch
    from waterorderlib_amd import engine, synth
    oracle.set_threads(8)
    n, T = 6000, 96
    xyz, boxes0 = synth.water_frames(n, 1, seed=401)
    L0 = boxes0[0]                                      # ~56.4 A: 12 cells of 4.70; 54.0 / 58.5 are the thresholds
    rng = np.random.default_rng(402)
    scale = np.where(np.arange(T) % 3 == 0, 0.955, np.where(np.arange(T) % 3 == 1, 1.0, 1.04))[:, None] * (1.0 + 0.003 * rng.standard_normal((T, 3)))
    boxes = L0[None, :] * scale
    frames = (xyz[0][None, :, :] / L0) * boxes[:, None, :] + 0.05 * rng.standard_normal((T, n, 3))
    assert len({tuple(np.floor(b / 4.5).astype(int)) for b in boxes}) >= 3
    res = engine.tet_order_traj(torch.from_numpy(frames).cuda(), boxes, want_q=True)
    qh = np.zeros(500, dtype=np.int64)
    for t in range(T):
        if t in (0, 1, 2, 50, T - 1):
            q_o = oracle.getOrderParamq(frames[t], frames[t], boxes[t], cells=True)
            np.testing.assert_allclose(res.q[t].cpu().numpy(), q_o, rtol=0, atol=Q_TOL)
            assert abs(float(res.mean[t]) - np.mean(q_o)) < 1e-13
    h = np.histogram(res.q.cpu().numpy().ravel(), bins=500, range=[0.0, 1.0])[0]
    assert np.array_equal(res.qhist.cpu().numpy(), h)
    assert int(res.dhist.sum().item()) == 4 * n * T


def test_batch_mixing_periodic_and_open_frames(oracle):
    """frames with a non-periodic axis (BoxL < 0: all-pairs kernels) inside a batch of periodic ones"""
    import torch
    from waterorderlib_amd import engine, synth
    oracle.set_threads(8)
    n, T = 1500, 70
    xyz, boxes = synth.water_frames(n, T, seed=411)
    boxes = boxes.copy()
    open_frames = [3, 4, 40, 69]
    for t in open_frames:
        boxes[t, t % 3] = -1.0
    res = engine.tet_order_traj(torch.from_numpy(xyz).cuda(), boxes, want_q=True)
    for t in open_frames + [0, 5, 41]:
        q_o = oracle.getOrderParamq(xyz[t], xyz[t], boxes[t], cells=bool(np.all(boxes[t] > 0)))
        np.testing.assert_allclose(res.q[t].cpu().numpy(), q_o, rtol=0, atol=Q_TOL)
        assert abs(float(res.mean[t]) - np.mean(q_o)) < 1e-13 and abs(float(res.var[t]) - np.var(q_o)) < 1e-13
    h = np.histogram(res.q.cpu().numpy().ravel(), bins=500, range=[0.0, 1.0])[0]
    assert np.array_equal(res.qhist.cpu().numpy(), h)


def test_nan_coordinates_do_not_fault(wp, oracle):
    """a water with NaN coordinates (a corrupt frame) gets q = 0 as in the reference (all its comparisons are false);
    nobody can have it as a neighbour; the other waters are unaffected and nothing faults -- single frame and trajectory"""
    import torch
    from waterorderlib_amd import engine, synth
    xyz, boxes = synth.water_frames(2500, 70, seed=421)
    pos, box = xyz[0].copy(), boxes[0]
    pos[17] = np.nan
    pos[1200, 1] = np.inf
    q = wp.getOrderParamq(pos, pos, box)
    q_o = oracle.getOrderParamq(pos, pos, box)            # literal O(N^2) form: NaN distances fail every test
    assert q[17] == 0.0 and q[1200] == 0.0
    np.testing.assert_allclose(q, q_o, rtol=0, atol=Q_TOL)
    xyz[5, 17] = np.nan
    xyz[60, 99, 2] = -np.inf
    res = engine.tet_order_traj(torch.from_numpy(xyz).cuda(), boxes, want_q=True)
    qt = res.q.cpu().numpy()
    assert qt[5, 17] == 0.0 and qt[60, 99] == 0.0 and np.all(np.isfinite(qt))
    np.testing.assert_allclose(qt[5], oracle.getOrderParamq(xyz[5], xyz[5], boxes[5]), rtol=0, atol=Q_TOL)


def test_big_frames_in_small_batches(oracle):
    """a few large frames: the streaming sort splits every frame over several blocks (count / scan / scatter) and the
    long rows of a big box go through the four-slot row sort; fp64 and fp32 storage, sampled waters vs the oracle"""
    import torch
    from waterorderlib_amd import engine, synth
    oracle.set_threads(8)
    n, F = 420000, 3
    xyz, boxes = synth.water_frames(n, 1, seed=301)
    base = torch.from_numpy(xyz[0]).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    frames = (base.unsqueeze(0) + 0.05 * torch.randn((F, n, 3), generator=g, device="cuda", dtype=torch.float64)).contiguous()
    bx = np.tile(boxes[0][None, :], (F, 1))
    res = engine.tet_order_traj(frames, bx, want_q=True)
    res32 = engine.tet_order_traj(frames.to(torch.float32), bx, want_q=True)
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    sample = rng.choice(n, 1500, replace=False)
    for t in (0, F - 1):
        pos = frames[t].cpu().numpy()
        q_o = oracle.getOrderParamq(pos[sample], pos, boxes[0], cells=True)
        np.testing.assert_allclose(res.q[t].cpu().numpy()[sample], q_o, rtol=0, atol=Q_TOL)
        pos32 = frames[t].to(torch.float32).cpu().numpy().astype(np.float64)
        q_o32 = oracle.getOrderParamq(pos32[sample], pos32, boxes[0], cells=True)
        np.testing.assert_allclose(res32.q[t].cpu().numpy().astype(np.float64)[sample], q_o32, rtol=0, atol=2e-7)
    assert int(res.qhist.sum().item()) <= F * n and int(res.dhist.sum().item()) == 4 * F * n


@pytest.mark.parametrize("seed", range(10))
def test_random_boxes_and_densities_vs_oracle(wp, oracle, seed):
    """seeded sweep over box shapes (3 to ~25 cells per axis), densities and clustering -- tiles that wrap around the
    box, rows with few or no waters, cells beyond the tile tables' capacity (forced onto the exact path): indices,
    distances, angles and q of every water against the oracle, for both proposal kernels"""
    from waterorderlib_amd import _lib
    oracle.set_threads(8)
    rng = np.random.default_rng(1000 + seed)
    box = rng.uniform(13.6, 110.0, size=3)
    if seed % 3 == 0:
        box[rng.integers(3)] = rng.uniform(13.6, 15.0)          # three cells along one axis
    rho = [0.0334, 0.0334, 0.02, 0.05, 0.0334, 0.01, 0.0334, 0.07, 0.0334, 0.004][seed]
    n = int(min(max(rho * np.prod(box), 300), 14000))
    pos = rng.random((n, 3)) * box
    if seed % 2 == 1:                                              # droplets: a third of the waters in dense blobs
        k = n // 3
        centres = rng.random((5, 3)) * box
        pos[:k] = centres[rng.integers(5, size=k)] + rng.normal(0.0, 2.5, size=(k, 3))
    if seed == 4:
        pos[:200] = pos[0] + rng.normal(0.0, 0.35, size=(200, 3))  # one overfull cell
    pos += rng.integers(-1, 2, size=pos.shape) * box               # unwrapped molecules
    q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    # the reference leaves exact distance ties to argsort: make sure this input has none among the first five
    for rows in (1, 0):
        c = _lib.Context(0)
        c.set_option("path", 3)
        c.set_option("search_rows", rows)
        check_full(wp.getOrderParamq(pos, pos, box, full=True, ctx=c), q_o, idx_o, dist_o, ang_o)
        np.testing.assert_allclose(wp.getOrderParamq(pos, pos, box, ctx=c), q_o, rtol=0, atol=Q_TOL)
        c.close()


# ---- Willard-Chandler density field ---------------------------------------------
DENS_RTOL = 1e-12   # device exp() vs glibc exp(): a few ulp on each Gaussian term
NORM_ATOL = 1e-10   # unit normals; looser where the gradient nearly cancels


def check_density(dv, dn, g_dv, g_dn):
    assert dv.shape == g_dv.shape and dn.shape == g_dn.shape
    np.testing.assert_allclose(dv, g_dv, rtol=DENS_RTOL, atol=1e-18)
    assert np.array_equal(np.isnan(dn), np.isnan(g_dn)), "NaN-normal pattern (no water within 3 sigma) differs"
    np.testing.assert_allclose(dn, g_dn, rtol=0, atol=NORM_ATOL, equal_nan=True)


def test_willard_golden():
    from waterorderlib_amd import waterlib as wl
    g = golden("willard_slab700")
    s = float(g["smoothlen"])
    dv, dn = wl.willarddensityfield(g["pos"], g["gridx"], g["gridy"], g["gridz"], g["box"], s)
    assert dv.flags.f_contiguous and dn.flags.f_contiguous
    check_density(dv, dn, g["densvals"], g["densnorms"])
    # f2py accepts (1, n) grid vectors, which is what surface_library.densityGrid passes (:184-199)
    dv2, dn2 = wl.willarddensityfield(g["pos"], g["gridx"][None, :], g["gridy"][None, :], g["gridz"][None, :],
                                      g["box"][None, :], s)
    assert np.array_equal(dv2, dv) and np.array_equal(dn2, dn, equal_nan=True)   # reproducible run to run
    pv, pn = wl.willarddensitypoints(g["pos"], g["pts"], g["box"], s)
    check_density(pv, pn, g["pts_densvals"], g["pts_densnorms"])


def test_willard_vs_oracle(oracle):
    from waterorderlib_amd import synth, waterlib as wl
    oracle.set_threads(8)
    # (i) bulk box, unwrapped molecules, grid reaching outside the box; (ii) non-periodic axis (all-pairs walk);
    # (iii) box smaller than 2 * 3 sigma (every image decision through anint); (iv) other smoothing length
    pos, boxes = synth.water_frames(4000, 1, seed=41)
    rng = np.random.default_rng(42)
    p1 = pos[0] + rng.integers(-1, 2, size=(4000, 3)) * boxes[0]
    cases = [(p1, boxes[0], 2.4), (pos[0], boxes[0] * np.array([1, 1, -1.0]), 2.4),
             (pos[0][:60] * 0.25, np.array([13.0, 12.0, 14.5]), 2.4), (pos[0], boxes[0], 1.3)]
    for p, box, s in cases:
        L = np.abs(box)
        gx = np.linspace(-4.0, L[0] + 3.0, 23); gy = np.linspace(0.0, L[1], 17); gz = np.linspace(0.3, L[2] * 1.5, 19)
        dv, dn = wl.willarddensityfield(p, gx, gy, gz, box, s)
        o_dv, o_dn = oracle.willarddensityfield(p, gx, gy, gz, box, s)
        check_density(dv, dn, o_dv, o_dn)
        pts = rng.random((777, 3)) * L * 1.4 - 5.0
        pv, pn = wl.willarddensitypoints(p, pts, box, s)
        o_pv, o_pn = oracle.willarddensitypoints(p, pts, box, s)
        check_density(pv, pn, o_pv, o_pn)
    # bulk water: the field averages to the number density times the integral of the truncated-shifted Gaussian,
    # erf(3/sqrt2) - sqrt(2/pi) 3 e^-4.5 - e^-4.5 (4/3 pi 27) / (2 pi)^1.5 = 0.8909
    dv, _ = wl.willarddensityfield(pos[0], np.linspace(5, 30, 6), np.linspace(5, 30, 6), np.linspace(5, 30, 6), boxes[0], 2.4)
    assert abs(dv.mean() / (0.8909 * 4000 / np.prod(boxes[0])) - 1.0) < 0.03


def test_willard_edge_cases():
    from waterorderlib_amd import waterlib as wl, surface_library as sl, synth
    box = np.array([30.0, 30.0, 30.0])
    # no waters: zero density and 0/0 normals, as the reference's empty loop leaves them
    dv, dn = wl.willarddensityfield(np.zeros((0, 3)), [1.0, 2.0], [1.0], [1.0, 2.0, 3.0], box, 2.4)
    assert dv.shape == (2, 1, 3) and np.all(dv == 0.0) and np.all(np.isnan(dn))
    # one water: analytic value at the water itself and beyond the 3-sigma cut
    s = 2.4
    pv, pn = wl.willarddensitypoints(np.array([[3.0, 4.0, 5.0]]), np.array([[3.0, 4.0, 5.0], [3.0, 4.0, 5.01 + 3 * s],
                                                                          [4.0, 4.0, 5.0], [33.5, 4.0, 5.0]]), box, s)
    pref = (2 * np.pi * s * s) ** 1.5
    assert abs(pv[0] - (1.0 - np.exp(-4.5)) / pref) < 1e-17 and pv[1] == 0.0
    assert np.all(np.isnan(pn[0])) and np.all(np.isnan(pn[1]))       # zero gradient -> 0/0
    np.testing.assert_allclose(pn[2], [-1.0, 0.0, 0.0], atol=1e-15)    # density falls away from the water
    np.testing.assert_allclose(pn[3], [-1.0, 0.0, 0.0], atol=1e-15)    # through the periodic image (+0.5 A)
    assert abs(pv[3] - (np.exp(-0.25 / (2 * s * s)) - np.exp(-4.5)) / pref) < 1e-17
    # zero points
    pv, pn = wl.willarddensitypoints(np.array([[3.0, 4.0, 5.0]]), np.zeros((0, 3)), box, s)
    assert pv.shape == (0,) and pn.shape == (0, 3)
    with pytest.raises(ValueError):
        wl.willarddensitypoints(np.zeros((4, 2)), np.zeros((1, 3)), box, s)
    # surface_library.densityGrid's grid construction (:170-199) on a slab
    xyz, boxes = synth.slab_frames(3000, 1, seed=5)
    heavy = xyz[0][:40]
    dens, norms, spans, spacing = sl.densityField(heavy, xyz[0], boxes[0][None, :])
    assert dens.shape == (80, 80, 80) and norms.shape == (80, 80, 80, 3) and len(spans) == 3
    assert dens.max() > 0.02 and dens.min() >= 0.0


# ---- radial distribution functions ------------------------------------------------
def test_rdf_golden():
    """bit-exact: integer pair counts per bin and the reference's normalisation arithmetic"""
    from waterorderlib_amd import waterlib as wl
    g = golden("rdf700")
    pos, box, sol = g["pos"], g["box"], g["sol"]
    assert np.array_equal(wl.radialdistsame(pos, 0.1, 150, 1.0, box), g["rdf_same"])
    assert np.array_equal(wl.radialdist(sol, pos, 0.1, 150, 0.0334, box), g["rdf_sol_wat"])
    assert np.array_equal(wl.radialdistsame(sol, 0.1, 150, 1.0, box[None, :]), g["rdf_sol_sol"])
    assert np.array_equal(wl.pairdistancehistogram(sol, pos, 0.25, 60, box), g["hist_sol_wat"])
    assert np.array_equal(wl.pairdistancehistogram(pos, pos, 0.05, 300, box), g["hist_wat_wat"])
    assert np.array_equal(wl.radialdistsame(pos, 0.1, 150, 1.0, box * np.array([1, 1, -1.0])), g["rdf_same_nonperiodic_z"])


def test_rdf_vs_oracle_and_properties(oracle):
    from waterorderlib_amd import synth, waterlib as wl
    oracle.set_threads(8)
    xyz, boxes = synth.water_frames(6000, 1, seed=61, box=np.array([52.0, 61.5, 56.1]))
    pos, box = xyz[0], boxes[0]
    rng = np.random.default_rng(62)
    unwrapped = pos + rng.integers(-2, 3, size=pos.shape) * box
    sol = rng.random((300, 3)) * box * 1.3 - 4.0
    assert np.array_equal(wl.radialdistsame(unwrapped, 0.1, 150, 1.0, box), oracle.radialdistsame(unwrapped, 0.1, 150, 1.0, box))
    assert np.array_equal(wl.radialdist(sol, pos, 0.1, 150, 0.0334, box), oracle.radialdist(sol, pos, 0.1, 150, 0.0334, box))
    assert np.array_equal(wl.radialdist(pos, sol, 0.07, 111, 2.0, box), oracle.radialdist(pos, sol, 0.07, 111, 2.0, box))
    # range beyond half the box (every cell visited once, minimum image decides) and a tiny box
    assert np.array_equal(wl.pairdistancehistogram(pos[:1500], pos, 0.4, 120, box), oracle.pairdistancehistogram(pos[:1500], pos, 0.4, 120, box))
    small, sbox = pos[:80] * 0.2, np.array([11.0, 12.5, 10.4])
    assert np.array_equal(wl.radialdistsame(small, 0.1, 150, 1.0, sbox), oracle.radialdistsame(small, 0.1, 150, 1.0, sbox))
    # a histogram that spans the whole box counts every ordered pair once; g(r) of an ideal gas is flat
    h = wl.pairdistancehistogram(pos, pos, 0.5, 120, box)
    assert h.sum() == len(pos) * (len(pos) - 1)
    gas, gbox = synth.ideal_gas(20000, seed=63)
    gr = wl.radialdistsame(gas, 0.5, 20, 20000 / np.prod(gbox), gbox)
    assert np.all(np.abs(2.0 * gr[4:] - 1.0) < 0.05)       # pairs counted once -> g/2
    # edge cases: empty sets, coincident points (zero distance is not counted)
    assert np.all(np.isnan(wl.radialdistsame(np.zeros((0, 3)), 0.1, 10, 1.0, box)))   # 0 / (0 * ...), as the reference
    assert np.all(wl.radialdist(sol, np.zeros((0, 3)), 0.1, 10, 1.0, box) == 0)
    two = np.array([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 1.0, 2.0]])
    assert wl.pairdistancehistogram(two, two, 0.5, 4, box).tolist() == [0.0, 4.0, 0.0, 0.0]
    with pytest.raises(RuntimeError):
        wl.radialdistsame(pos, -0.1, 10, 1.0, box)


def test_rdf_driver(tmp_path, monkeypatch):
    """rdfCalc (orderParam_lib.py:575-727) on an array-backed trajectory with solute atoms; its g(r) arrays equal
    the per-frame library calls averaged the same way"""
    from waterorderlib_amd import synth, orderParam_lib as opl, waterlib as wl
    from waterorderlib_amd.TrajObject import TrajObject
    monkeypatch.chdir(tmp_path)
    xyz, boxes = synth.water_frames(900, 10, seed=71)
    sol = np.random.default_rng(72).random((10, 12, 3)) * boxes[:, None, :]
    allxyz = np.concatenate([xyz, sol], axis=1)
    obj = TrajObject(allxyz, boxes, lenWat=1, nWaters=900, solInds=np.arange(900, 912))
    n1_ow, n1_sol, tpar = opl.rdfCalc(obj, binwidth=0.1, totbins=150)
    # (two noisy frames per chunk: the first relative minimum comes early; only sanity is asserted on the integrals)
    assert 0.0 < n1_ow[0] < 8.0 and np.isfinite(n1_sol[0]) and np.isfinite(tpar[0])
    rd = opl.rdfCalc.last_rdfs["OwOw"]
    assert rd.shape == (5, 150)
    expect = (wl.radialdistsame(xyz[0], 0.1, 150, 1.0, boxes[0]) + wl.radialdistsame(xyz[1], 0.1, 150, 1.0, boxes[1])) / 2.0
    assert np.array_equal(rd[0], expect)
    assert np.loadtxt("rdf.txt").shape == (150, 7) and np.loadtxt("coord.txt").shape == (148, 7)


# ---- water orientation ------------------------------------------------------------
def test_orientation_golden_and_oracle(wp, oracle):
    from waterorderlib_amd import synth, waterlib as wl
    g = golden("orient500")
    ad, ap = wl.watorient(g["opos"], g["hpos"], g["refvec"], g["box"])
    np.testing.assert_allclose(ad, g["angdip"], rtol=0, atol=ANG_TOL)      # device acos vs glibc acos
    np.testing.assert_allclose(ap, g["angplane"], rtol=0, atol=ANG_TOL)
    nd = wl.watohvec(g["opos"], g["hpos"], g["box"])
    assert nd.flags.f_contiguous and np.array_equal(nd, g["normdip"])       # no transcendental: bit-exact
    np.testing.assert_allclose(wp.waterOrientation(g["opos"], g["hpos"], g["box"])[0], g["angdip_z"], rtol=0, atol=ANG_TOL)
    # larger frame, molecules crossing the box faces, a non-periodic axis
    xyz, boxes = synth.water_frames(20000, 1, seed=43, len_wat=3)
    f, box = xyz[0], boxes[0] * np.array([1.0, 1.0, -1.0])
    O = f[0::3]; H = np.empty((40000, 3)); H[0::2] = f[1::3]; H[1::2] = f[2::3]
    H[:, :2] -= box[:2] * np.floor(H[:, :2] / box[:2])
    ad, ap = wl.watorient(O, H, [1.0, 2.0, -0.5], box)
    o_ad, o_ap = oracle.watorient(O, H, [1.0, 2.0, -0.5], box)
    np.testing.assert_allclose(ad, o_ad, rtol=0, atol=ANG_TOL); np.testing.assert_allclose(ap, o_ap, rtol=0, atol=ANG_TOL)
    assert np.array_equal(wl.watohvec(O, H, box), oracle.watohvec(O, H, box))
    # a water lying in the xy plane with its dipole along +x
    o1 = np.array([[5.0, 5.0, 5.0]]); h1 = np.array([[5.6, 5.8, 5.0], [5.6, 4.2, 5.0]])
    ad, ap = wl.watorient(o1, h1, [1.0, 0.0, 0.0], boxes[0])
    assert abs(ad[0]) < 1e-6 and abs(ap[0] - 90.0) < 1e-9
    with pytest.raises(ValueError):
        wl.watorient(O, H[:-1], [0.0, 0.0, 1.0], box)
    with pytest.raises(RuntimeError):
        wl.watorient(O, H, [0.0, 0.0, 0.0], box)
    plane2d, dip2d = wp.waterOrientationBinZ(O, H, box)
    assert plane2d.shape == dip2d.shape and plane2d.sum() <= len(O) and dip2d.sum() > 0.99 * len(O) - plane2d.shape[1]


# ---- hydrogen bonds ------------------------------------------------------------------
def _water_hb_arrays(f):
    O = f[0::3].copy(); H = np.empty((2 * len(O), 3)); H[0::2] = f[1::3]; H[1::2] = f[2::3]
    return O, np.repeat(O, 2, axis=0), H


def test_hbonds_golden_and_oracle(oracle):
    """bit-exact decisions (the angle test uses the device acos: a flip needs an angle within ~1e-12 deg of the cut)"""
    from waterorderlib_amd import synth, waterlib as wl
    g = golden("hbonds600")
    for key, (dc, ac) in (("pairs_3p0_150", (3.0, 150.0)), ("pairs_3p5_120", (3.5, 120.0))):
        b = wl.generalhbonds(g["acc"], g["don"], g["donh"], g["box"], dc, ac)
        assert b.dtype == np.int32 and b.flags.f_contiguous
        assert np.array_equal(np.stack(np.where(b == 1), axis=1), g[key])
        ai, di = wl.generalhbonds_list(g["acc"], g["don"], g["donh"], g["box"], dc, ac)
        assert np.array_equal(np.stack([ai, di], axis=1), g[key])
    # larger frame through the cell list vs the oracle's dense mask; acceptors that are not the donors' atoms;
    # a non-periodic axis (all-pairs walk)
    oracle.set_threads(8)
    xyz, boxes = synth.water_frames(5000, 1, seed=53, len_wat=3)
    O, don, H = _water_hb_arrays(xyz[0])
    for box in (boxes[0], boxes[0] * np.array([1.0, -1.0, 1.0])):
        ref = oracle.generalhbonds(O[:1500], don, H, box, 3.5, 130.0)
        ai, di = wl.generalhbonds_list(O[:1500], don, H, box, 3.5, 130.0)
        assert np.array_equal(np.stack([ai, di], axis=1), np.stack(np.where(ref == 1), axis=1))
        assert np.array_equal(wl.generalhbonds(O[:1500], don, H, box, 3.5, 130.0), ref)
    ai, di = wl.generalhbonds_list(O, don, H, boxes[0], 3.5, 120.0)
    assert len(ai) > len(O) // 2 and not np.any(ai == di // 2)          # a water never bonds to itself
    assert wl.generalhbonds(np.zeros((0, 3)), don, H, boxes[0], 3.0, 150.0).shape == (0, len(don))
    with pytest.raises(ValueError):
        wl.generalhbonds(O, don[:-1], H, boxes[0], 3.0, 150.0)


def test_hbonds_general_wrapper(wp):
    """wp.HBondsGeneral (water_properties.py:681-720) against the reference's own output"""
    g = golden("hbonds600")
    n, hl, loc = wp.HBondsGeneral(g["acc"], g["don"], g["donh"], g["box"], g["acc_inds"], g["don_inds"], g["donh_inds"],
                                  distCut=3.5, angCut=120.0)
    assert n == int(g["hbg_num"]) and np.array_equal(hl, g["hbg_list"]) and np.array_equal(loc, g["hbg_loc"])


def test_bound_wrap_driver(oracle):
    """getBoundWrap (orderParam_lib.py:419-572) on arrays: shell / bound / wrap sets against the same logic run on the
    oracle's masks"""
    from waterorderlib_amd import synth, orderParam_lib as opl
    xyz, boxes = synth.water_frames(1500, 1, seed=57, len_wat=3)
    f, box = xyz[0], boxes[0]
    rng = np.random.default_rng(58)
    # a toy solute: 6 hydroxyl-like sites O-H placed in the box
    so = rng.random((6, 3)) * box
    sh = so + 0.96 * np.array([[1.0, 0.0, 0.0]])
    pos = np.concatenate([f, so, sh])
    nw = len(f)
    watInds = np.arange(0, nw, 3)
    solInds = np.arange(nw, nw + 6)
    solH = np.arange(nw + 6, nw + 12)
    bound, wrap, shell, non = opl.getBoundWrap(pos, box, watInds, 3, solInds, solInds, solInds, solH, cutoff=4.0, hbDist=3.5, hbAng=120.0)
    nb = oracle.nearneighbors(pos[solInds], pos[watInds], box, 0.0, 4.0)
    assert np.array_equal(shell, watInds[np.unique(np.where(nb == 1)[1])])
    assert len(bound) + len(wrap) == len(shell) and len(shell) + len(non) == len(watInds) and len(shell) > 0
    acc_w = oracle.generalhbonds(pos[shell], pos[solInds], pos[solH], box, 3.5, 120.0)
    don_w = oracle.generalhbonds(pos[solInds], pos[np.repeat(shell, 2)], pos[(np.repeat(shell, 2).reshape(-1, 2) + [1, 2]).reshape(-1)], box, 3.5, 120.0)
    expect = np.union1d(np.where(acc_w == 1)[0], np.where(don_w == 1)[1] // 2)
    assert np.array_equal(bound, shell[expect])
