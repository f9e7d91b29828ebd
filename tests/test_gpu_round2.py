"""Round-2 GPU parity tests (all through the C ABI):
  * integer outputs bit-exact BY CONSTRUCTION: the trajectory mode's q histogram on inputs whose q sits within 1e-15 of
    a bin edge, hydrogen-bond decisions with the angle within one ulp of the cut (tests/edge_cases.py builds both with the
    oracle), and lean qhist == np.histogram of the full-chain q on frames of the BASELINE config-3 box;
  * BASELINE config 2 (10 000 waters x 1 000 frames, fp64) at full length on every instantiation of the proposal
    kernel, the instantiation forced and asserted;
  * strided O-row gather against the reference's golden TIP4P frame;
  * breathing boxes (NPT) through the memoised tile tables."""
import numpy as np
import pytest

import edge_cases
from conftest import golden

pytestmark = pytest.mark.gpu

Q_TOL = 1e-12


@pytest.fixture(scope="module")
def wp_mod():
    from waterorderlib_amd import water_properties
    return water_properties


def _np_hist(q, bins=500):
    return np.histogram(q, bins=bins, range=[0.0, 1.0], density=False)[0].astype(np.int64)


def test_q_on_bin_edges_is_binned_as_the_full_chain_bins_it(oracle):
    """Every target water's q is within ~1e-15 of an np.histogram edge (adjacent doubles of one coordinate put it just
    below the edge in one frame, on / above it in the other, in the reference's arithmetic).  The trajectory mode forms q
    without the acos -> degrees -> cos round trip (|dq| < 2e-14): unguarded, about half of these would be binned on the
    wrong side.  By construction the integer histogram is np.histogram of the library's own full-chain q
    (wp.getOrderParamq(full=True): the reference's operation order, device libm); against the reference itself that q can
    differ in the last place (acos / cos of the device vs the host's libm), so only targets further than 5e-16 from
    their edge are required to fall in the reference's bin."""
    from waterorderlib_amd import _lib, engine
    from waterorderlib_amd import water_properties as wp
    frames, boxes, targets, edges = edge_cases.frames_with_q_on_bin_edges(oracle)
    assert len(targets) >= 20
    ctx = _lib.Context(0)
    ctx.set_option("path", 3)          # the fp32-filtered cell path (the one trajectories of real boxes take)
    res = engine.tet_order_traj(frames, boxes, want_q=True, ctx=ctx)
    assert ctx.counter("slow_waters") >= 2 * len(targets)    # the guard handed them to the exact kernel
    expect = np.zeros(500, dtype=np.int64)
    lin = np.linspace(0.0, 1.0, 501)
    for f in range(2):
        q_full = wp.getOrderParamq(frames[f], frames[f], boxes[f], full=True, ctx=ctx)[0]
        expect += _np_hist(q_full)
        assert np.array_equal(res.q[f][targets], q_full[targets])          # the exact kernel's q, bit for bit
        np.testing.assert_allclose(res.q[f], q_full, rtol=0, atol=1e-13)
        q_ref = oracle.getOrderParamq(frames[f], frames[f], boxes[f], cells=True)     # the reference's chain, host libm
        assert np.max(np.abs(q_ref[targets] - edges)) < 5e-15
        np.testing.assert_allclose(q_full, q_ref, rtol=0, atol=Q_TOL)
        clear = np.abs(q_ref[targets] - edges) > 5e-16
        assert np.array_equal(np.digitize(q_full[targets][clear], lin), np.digitize(q_ref[targets][clear], lin))
        assert np.abs(_np_hist(q_full) - oracle.np_histogram(q_ref)).sum() <= 2 * int((~clear).sum())
    assert np.array_equal(res.qhist, expect)
    # the same through the one-block-per-frame sort (>= 128 frames per batch) and without per-water q
    res2 = engine.tet_order_traj(np.repeat(frames, 70, axis=0), np.repeat(boxes, 70, axis=0), ctx=ctx)
    assert np.array_equal(res2.qhist, 70 * expect)
    ctx.close()


def test_lean_qhist_equals_histogram_of_full_chain_q_config3_box():
    """three frames of the 100 000-water box: trajectory-mode qhist == np.histogram of the per-frame call's q (full=True:
    the reference's acos/deg/cos chain); the 4-NN distance histogram == the reference's ceiling(d / w) rule applied to the
    per-frame call's distances"""
    import torch
    from waterorderlib_amd import engine, synth
    from waterorderlib_amd import water_properties as wp
    n, T = 100000, 3
    base, box = synth.lattice_sites(n, seed=2026)
    rng = np.random.default_rng(7)
    frames = base[None] + 0.3 * rng.standard_normal((T, n, 3))
    frames -= box * np.floor(frames / box)
    boxes = np.tile(box[None], (T, 1))
    res = engine.tet_order_traj(torch.from_numpy(frames).cuda(), boxes)
    torch.cuda.synchronize()
    qh = np.zeros(500, dtype=np.int64)
    dh = np.zeros(1000, dtype=np.int64)
    for t in range(T):
        q, idx, dist, ang = wp.getOrderParamq(frames[t], frames[t], box, full=True)
        qh += _np_hist(q)
        k = np.ceil(dist.reshape(-1) / 0.01).astype(np.int64)
        dh += np.bincount(k[(k >= 1) & (k <= 1000)] - 1, minlength=1000)
    assert np.array_equal(res.qhist.cpu().numpy(), qh)
    assert np.array_equal(res.dhist.cpu().numpy(), dh)


def test_hbond_decisions_with_the_angle_on_the_cut(oracle):
    """400 acceptor / donor / hydrogen triples in adjacent-double pairs that straddle the cut-off angle in the reference's
    arithmetic (host libm acos): dense mask and cell-list form must reproduce every decision"""
    from waterorderlib_amd import waterlib as wl
    for angcut in (150.0, 120.0, 133.7):
        acc, don, hyd, box = edge_cases.hbond_triples_on_the_cut(oracle, n=200, angcut=angcut)
        ref = oracle.generalhbonds(acc, don, hyd, box, 3.5, angcut)
        d = np.diag(ref).reshape(-1, 2)
        assert np.all(d.sum(axis=1) == 1)            # each pair of adjacent doubles straddles the cut
        got = wl.generalhbonds(acc, don, hyd, box, 3.5, angcut)
        assert np.array_equal(got, ref)
        ai, di = wl.generalhbonds_list(acc, don, hyd, box, 3.5, angcut)
        assert np.array_equal(np.stack([ai, di], axis=1), np.stack(np.where(ref == 1), axis=1))


@pytest.mark.parametrize("variant", ["rows512", "rows768", "tiles3d"])
def test_config2_full_length_on_every_proposal_instantiation(oracle, variant):
    """BASELINE config 2: 10 000 waters x 1 000 frames, fp64, one MI355X; north-star bar: indices bit-exact, q within
    1e-6 (asserted at 1e-12).  The proposal kernel's instantiation is forced and read back."""
    import torch
    from waterorderlib_amd import _lib, engine, synth
    from waterorderlib_amd import water_properties as wp
    n, T = 10000, 1000
    base, box = synth.lattice_sites(n, seed=404)
    base = torch.from_numpy(base[np.random.default_rng(2).permutation(n)]).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(17)
    x = base.unsqueeze(0) + 0.3 * torch.randn((T, n, 3), generator=g, device="cuda", dtype=torch.float64)
    b = torch.as_tensor(box, device="cuda")
    frames = (x - b * torch.floor(x / b)).contiguous()
    boxes = np.tile(box[None], (T, 1))
    ctx = _lib.Context(0)
    if variant == "tiles3d":
        ctx.set_option("search_rows", 0)
    else:
        ctx.set_option("rows_block", int(variant[4:]))
    res = engine.tet_order_traj(frames, boxes, want_q=True, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.counter("rows_block") == {"rows512": 512, "rows768": 768, "tiles3d": 0}[variant]
    assert ctx.counter("slow_waters") < 0.01 * n * T
    q = res.q.cpu().numpy()
    assert int(res.dhist.sum().item()) == 4 * n * T
    assert int(res.qhist.sum().item()) == int(((q >= 0.0) & (q <= 1.0)).sum())
    np.testing.assert_allclose(res.mean.cpu().numpy(), q.mean(axis=1), rtol=0, atol=1e-13)
    np.testing.assert_allclose(res.var.cpu().numpy(), q.var(axis=1), rtol=0, atol=1e-13)
    oracle.set_threads(8)
    qh = np.zeros(500, dtype=np.int64)
    sampled = (0, 211, 499, 500, 777, 999)
    for t in sampled:
        pos = frames[t].cpu().numpy()
        q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
        np.testing.assert_allclose(q[t], q_o, rtol=0, atol=Q_TOL)
        q_g, idx_g, dist_g, ang_g = wp.getOrderParamq(pos, pos, box, full=True, ctx=ctx)    # same forced instantiation
        assert ctx.counter("rows_block") == {"rows512": 512, "rows768": 768, "tiles3d": 0}[variant]
        assert np.array_equal(idx_g, idx_o) and np.array_equal(dist_g, dist_o)
        np.testing.assert_allclose(ang_g, ang_o, rtol=0, atol=1e-9)
        qh += oracle.np_histogram(q_o)
    # the sampled frames on their own: integer histograms bit-exact against the oracle
    sub = engine.tet_order_traj(frames[list(sampled)].contiguous(), boxes[: len(sampled)], ctx=ctx)
    assert np.array_equal(sub.qhist.cpu().numpy(), qh)
    # every variant gives the same integers (kept across the parametrised runs); the per-frame means agree to rounding
    # only: which waters take the exact kernel (full acos/cos chain) instead of the cosine form depends on the proposal
    # kernel's error bound, and the two forms of q differ by ~1e-14
    key = (res.qhist.cpu().numpy(), res.dhist.cpu().numpy(), res.mean.cpu().numpy())
    prev = test_config2_full_length_on_every_proposal_instantiation.__dict__.setdefault("seen", key)
    assert np.array_equal(prev[0], key[0]) and np.array_equal(prev[1], key[1])
    np.testing.assert_allclose(prev[2], key[2], rtol=0, atol=1e-13)
    ctx.close()


def test_strided_gather_of_the_golden_tip4p_frame():
    """the O rows are read straight out of the interleaved [O, H, H, EP] frame (TrajObject.getWatInds,
    TrajObject.py:35-52; `pos[watInds]`, orderParam_lib.py:1462).  tests/golden/ice512_frame.npz is the 2 048-atom frame
    whose O rows the reference was run on for tests/golden/ice512.npz."""
    import torch
    from waterorderlib_amd import TrajObject, engine
    fr, g = golden("ice512_frame"), golden("ice512")
    xyz, box, lw = fr["xyz"], fr["box"], int(fr["len_wat"])
    assert np.array_equal(xyz[0::lw], g["pos"])
    obj = TrajObject.TrajObject(xyz[None], box[None], lenWat=lw)
    assert np.array_equal(obj.getWatInds()[0], np.arange(512) * lw)
    for frames in (xyz[None], torch.from_numpy(xyz[None].copy()).cuda()):
        res = engine.tet_order_traj(frames, box[None], len_wat=lw, want_q=True)
        q = res.q.cpu().numpy() if hasattr(res.q, "cpu") else res.q
        np.testing.assert_allclose(q[0], g["q"], rtol=0, atol=Q_TOL)
        qh = res.qhist.cpu().numpy() if hasattr(res.qhist, "cpu") else res.qhist
        assert np.array_equal(qh, _np_hist(g["q"]))
    # a batch of copies takes the cell-list path's strided reads too (n <= 1024 alone goes through the all-pairs kernel)
    big = np.tile(xyz[None], (3, 1, 1))
    from waterorderlib_amd import _lib
    ctx = _lib.Context(0)
    ctx.set_option("path", 2)
    res = engine.tet_order_traj(big, np.tile(box[None], (3, 1)), len_wat=lw, want_q=True, ctx=ctx)
    for t in range(3):
        np.testing.assert_allclose(res.q[t], g["q"], rtol=0, atol=Q_TOL)
    ctx.close()


def test_breathing_boxes_reuse_tile_tables(oracle):
    """an NPT-like trajectory: every frame has its own box; the cell counts -- and with them the memoised tile tables --
    change a few dozen times (more distinct (cells, waters) keys than the cache holds), boxes repeat later on"""
    from waterorderlib_amd import _lib, engine, synth
    n, T = 2500, 60
    xyz, boxes = synth.water_frames(n, 1, seed=91)
    rng = np.random.default_rng(91)
    scale = 1.0 + 0.25 * rng.random((T, 3))
    scale[40:] = scale[:20]                      # earlier boxes come back
    frames = xyz[0][None] * scale[:, None, :]
    bx = boxes[0][None] * scale
    ctx = _lib.Context(0)
    ctx.set_option("path", 3)
    res = engine.tet_order_traj(frames, bx, want_q=True, ctx=ctx)
    mean, var, qh, dh = oracle.tet_order_traj(frames, bx)
    assert np.array_equal(res.qhist, qh) and np.array_equal(res.dhist, dh)
    np.testing.assert_allclose(res.mean, mean, rtol=0, atol=1e-13)
    for t in (0, 19, 41, 59):
        np.testing.assert_allclose(res.q[t], oracle.getOrderParamq(frames[t], frames[t], bx[t], cells=True), rtol=0, atol=Q_TOL)
    ctx.close()


def test_host_pipeline_equals_device_resident_bit_for_bit():
    """wol_q_traj with host frames: double-buffered staging (several batches, the copy of one under the kernels of the
    previous), TIP4P-interleaved frames; every wire format of option host_pack -- whole interleaved frames (0), O rows
    packed by host threads (1), packed and narrowed to float32 when every coordinate is exactly one (2, the default) --
    gives the device-resident results bit for bit; per-water q on and off; the link carries what the mode says"""
    import torch
    from waterorderlib_amd import _lib, engine, synth
    xyz, boxes = synth.water_frames(3000, 45, seed=88, len_wat=4)
    xyz_sp = xyz.astype(np.float32).astype(np.float64)          # a trajectory read from a single-precision file
    ctx = _lib.Context(0)
    for frames, sp in ((xyz, False), (xyz_sp, True)):
        dev = engine.tet_order_traj(torch.from_numpy(frames).cuda(), boxes, len_wat=4, want_q=True)
        torch.cuda.synchronize()
        for pack, threads in ((0, 0), (1, 3), (2, 0)):
            ctx.set_option("host_pack", pack)
            ctx.set_option("host_threads", threads)
            for fpb in (8, 16, 0):                       # 6 batches (last one ragged), 3 batches, automatic
                ctx.set_option("frames_per_batch", fpb)
                for want_q in (True, False):
                    res = engine.tet_order_traj(frames, boxes, len_wat=4, want_q=want_q, ctx=ctx)
                    assert np.array_equal(res.qhist, dev.qhist.cpu().numpy()) and np.array_equal(res.dhist, dev.dhist.cpu().numpy())
                    assert np.array_equal(res.mean, dev.mean.cpu().numpy()) and np.array_equal(res.var, dev.var.cpu().numpy())
                    if want_q:
                        assert np.array_equal(res.q, dev.q.cpu().numpy())
                moved = ctx.counter("h2d_bytes")
                o_rows = frames.nbytes // 4                  # the O rows of TIP4P frames
                if pack == 0:
                    assert 0.99 * frames.nbytes <= moved <= frames.nbytes       # whole interleaved frames cross the link, once
                elif pack == 1 or not sp:
                    assert moved == o_rows                                      # the 24-byte O rows only
                else:
                    assert moved == o_rows // 2                                 # ... as float32: exact for these frames
        # float32 frames keep their storage type on the wire and in the results
        f32 = frames.astype(np.float32)
        ctx.set_option("host_pack", 2); ctx.set_option("frames_per_batch", 16)
        r32 = engine.tet_order_traj(f32, boxes, len_wat=4, want_q=True, ctx=ctx)
        d32 = engine.tet_order_traj(torch.from_numpy(f32).cuda(), boxes, len_wat=4, want_q=True)
        assert np.array_equal(r32.qhist, d32.qhist.cpu().numpy()) and np.array_equal(r32.q, d32.q.cpu().numpy())
        assert ctx.counter("h2d_bytes") == f32.nbytes // 4
    ctx.close()


def test_streamed_tetordercalc_equals_one_call(tmp_path, monkeypatch):
    """orderParam_lib.tetOrderCalc consumes the trajectory lazily, chunk by chunk (the reference iterates
    pytraj.iterload, orderParam_lib.py:1458): a one-shot frame generator, chunks of 7 frames, result = one-call result"""
    from waterorderlib_amd import engine, synth
    from waterorderlib_amd import orderParam_lib as opl
    from waterorderlib_amd.TrajObject import Frame, TrajObject
    monkeypatch.chdir(tmp_path)
    xyz, boxes = synth.water_frames(1500, 40, seed=89, len_wat=3)
    alive = {"made": 0}

    class OneShot:
        """frames exist only while they are being read: a generator that can be walked once"""
        def __init__(self):
            self.used = False
        def __len__(self):
            return len(xyz)
        def __iter__(self):
            assert not self.used, "the trajectory was iterated twice"
            self.used = True
            for t in range(len(xyz)):
                alive["made"] += 1
                yield Frame(xyz[t].copy(), boxes[t])

    obj = TrajObject(xyz[:1], boxes[:1], lenWat=3)
    obj.traj = OneShot()
    chunk = 7 * xyz[0].nbytes
    avgQ, varQ = opl.tetOrderCalc(obj, rng=np.random.default_rng(1), chunk_bytes=chunk)
    assert alive["made"] == 40
    series_avg, series_var, hists = opl.tetOrderCalc.last_series
    one = engine.tet_order_traj(xyz, boxes, len_wat=3)
    assert np.array_equal(series_avg[0], one.mean) and np.array_equal(series_var[0], one.var)
    assert np.array_equal(hists[0], one.qhist)
    assert abs(avgQ[0][0] - np.mean(one.mean)) < 1e-15
    assert np.loadtxt("qDistribution_0.txt")[:, 1].sum() == one.qhist.sum()
    # irregular water selection (not a constant stride): host-side gather per frame, same numbers for those waters
    sel = np.sort(np.random.default_rng(3).choice(1500, 1200, replace=False)) * 3
    obj2 = TrajObject(xyz, boxes, lenWat=3)
    obj2.getWatInds = lambda: (sel, np.array([], dtype=int), 3)
    opl.tetOrderCalc(obj2, rng=np.random.default_rng(1), chunk_bytes=chunk, write=False)
    ref = engine.tet_order_traj(np.ascontiguousarray(xyz[:, sel]), boxes)
    assert np.array_equal(opl.tetOrderCalc.last_series[0][0], ref.mean) and np.array_equal(opl.tetOrderCalc.last_series[2][0], ref.qhist)


def test_typed_and_device_forms_of_the_per_frame_calls(oracle):
    """T in {double, float}, host or device memory for wol_q_frame / wol_cos_angs / wol_lsi_frame.
    Device doubles reproduce the host-double call bit for bit; float storage = the double call on the float-rounded
    coordinates with every real output rounded once to float (indices unchanged)."""
    import ctypes as C
    import torch
    from waterorderlib_amd import _lib, synth
    from waterorderlib_amd import water_properties as wp
    xyz, boxes = synth.water_frames(4000, 1, seed=61)
    pos, box = xyz[0], boxes[0]
    sub = pos[100:900:3].copy()
    for s_np in (None, sub):
        s64 = pos if s_np is None else s_np
        q, idx, dist, ang = wp.getOrderParamq(s64, pos, box, full=True)
        ca, cn = wp.getCosAngs(s64, pos, box)
        li, ln = wp.getLSI(s64, pos, box)
        # device doubles
        P = torch.from_numpy(pos).cuda(); S = None if s_np is None else torch.from_numpy(s_np).cuda()
        dq, didx, ddist, dang = wp.getOrderParamq_dev(S, P, box, full=True)
        torch.cuda.synchronize()
        assert np.array_equal(dq.cpu().numpy(), q) and np.array_equal(didx.cpu().numpy(), idx)
        assert np.array_equal(ddist.cpu().numpy(), dist, equal_nan=True) and np.array_equal(dang.cpu().numpy(), ang, equal_nan=True)
        dca, dcn = wp.getCosAngs_dev(S, P, box)
        dli, dln = wp.getLSI_dev(S, P, box)
        assert np.array_equal(dca.cpu().numpy(), ca) and np.array_equal(dcn.cpu().numpy(), cn)
        assert np.array_equal(dli.cpu().numpy(), li) and np.array_equal(dln.cpu().numpy(), ln)
        # float storage, device and host: the double result of the rounded coordinates, rounded once
        p32 = pos.astype(np.float32); s32 = None if s_np is None else s_np.astype(np.float32)
        rq, ridx, rdist, rang = wp.getOrderParamq(p32.astype(np.float64) if s32 is None else s32.astype(np.float64), p32.astype(np.float64), box, full=True)
        P32 = torch.from_numpy(p32).cuda(); S32 = None if s32 is None else torch.from_numpy(s32).cuda()
        fq, fidx, fdist, fang = wp.getOrderParamq_dev(S32, P32, box, full=True)
        torch.cuda.synchronize()
        assert fq.dtype == torch.float32 and np.array_equal(fidx.cpu().numpy(), ridx)
        assert np.array_equal(fq.cpu().numpy(), rq.astype(np.float32)) and np.array_equal(fdist.cpu().numpy(), rdist.astype(np.float32), equal_nan=True)
        assert np.array_equal(fang.cpu().numpy(), rang.astype(np.float32), equal_nan=True)
        rca, rcn = wp.getCosAngs(p32.astype(np.float64) if s32 is None else s32.astype(np.float64), p32.astype(np.float64), box)
        fca, fcn = wp.getCosAngs_dev(S32, P32, box)
        assert np.array_equal(fca.cpu().numpy(), rca.astype(np.float32)) and np.array_equal(fcn.cpu().numpy(), rcn.astype(np.float32))
        rli, rln = wp.getLSI(p32.astype(np.float64) if s32 is None else s32.astype(np.float64), p32.astype(np.float64), box)
        fli, fln = wp.getLSI_dev(S32, P32, box)
        assert np.array_equal(fli.cpu().numpy(), rli.astype(np.float32)) and np.array_equal(fln.cpu().numpy(), rln.astype(np.float32))
        # float storage in HOST memory, straight through ctypes
        ns = len(p32) if s32 is None else len(s32)
        hq = np.zeros(ns, dtype=np.float32); hidx = np.zeros((ns, 4), dtype=np.int32)
        hdist = np.zeros((ns, 4), dtype=np.float32); hang = np.zeros((ns, 6), dtype=np.float32)
        ctx = _lib.default_context()
        _lib.check(_lib.load().wol_q_frame_t(ctx.handle, _lib.ptr(p32), 3, len(p32), _lib.ptr(s32), 3, ns, _lib.ptr(_lib.box3(box)), 0.0, 10.0,
                                             _lib.ptr(hq), _lib.ptr(hidx), _lib.ptr(hdist), _lib.ptr(hang), _lib.F32, _lib.MEM_HOST))
        assert np.array_equal(hq, rq.astype(np.float32)) and np.array_equal(hidx, ridx) and np.array_equal(hang, rang.astype(np.float32), equal_nan=True)
        tot = C.c_int64(0); hnum = np.zeros(ns, dtype=np.float32)
        args = (ctx.handle, _lib.ptr(p32), len(p32), _lib.ptr(s32), ns, _lib.ptr(_lib.box3(box)), 0.0, 3.413)
        _lib.check(_lib.load().wol_cos_angs_t(*args, None, _lib.ptr(hnum), C.byref(tot), _lib.F32, _lib.MEM_HOST))
        hca = np.zeros(tot.value, dtype=np.float32)
        _lib.check(_lib.load().wol_cos_angs_t(*args, _lib.ptr(hca), _lib.ptr(hnum), C.byref(tot), _lib.F32, _lib.MEM_HOST))
        assert np.array_equal(hca, rca.astype(np.float32)) and np.array_equal(hnum, rcn.astype(np.float32))
        hl = np.zeros(ns, dtype=np.float32); hn = np.zeros(ns, dtype=np.float32); nv = C.c_int64(0)
        _lib.check(_lib.load().wol_lsi_frame_t(ctx.handle, _lib.ptr(p32), len(p32), _lib.ptr(s32), ns, _lib.ptr(_lib.box3(box)), 0.0, 3.7,
                                               _lib.ptr(hl), _lib.ptr(hn), C.byref(nv), _lib.F32, _lib.MEM_HOST))
        assert nv.value == len(rli) and np.array_equal(hl[~np.isnan(hl)], rli.astype(np.float32)) and np.array_equal(hn, rln.astype(np.float32))
    # against the oracle too (device doubles)
    q_o = oracle.getOrderParamq(pos, pos, box, cells=True)
    np.testing.assert_allclose(wp.getOrderParamq_dev(None, torch.from_numpy(pos).cuda(), box).cpu().numpy(), q_o, rtol=0, atol=Q_TOL)


def test_sharded_entry_of_the_c_abi_one_rank_over_rccl():
    """wol_q_traj_sharded with a one-rank RCCL communicator (all a one-GPU box can hold): the exchange step runs through
    RCCL inside the library and must leave the one-GPU result untouched; host and device memory, float32 storage"""
    import torch
    from waterorderlib_amd import _lib, engine, synth
    xyz, boxes = synth.water_frames(2500, 24, seed=62, len_wat=3)
    ctx = _lib.Context(0)
    comm = _lib.Comm(ctx, 0, 1, _lib.Comm.unique_id())
    assert _lib.shard_bounds(24, 0, 1) == (0, 24) and _lib.shard_bounds(10, 2, 4) == (5, 7)
    one = engine.tet_order_traj(xyz, boxes, len_wat=3, want_q=True)
    res = engine.tet_order_traj_sharded_rccl(xyz, boxes, 24, comm, len_wat=3, want_q=True)
    assert np.array_equal(res.qhist, one.qhist) and np.array_equal(res.dhist, one.dhist)
    assert np.array_equal(res.mean, one.mean) and np.array_equal(res.var, one.var) and np.array_equal(res.q, one.q)
    dev = engine.tet_order_traj_sharded_rccl(torch.from_numpy(xyz).cuda(), boxes, 24, comm, len_wat=3, want_q=True)
    torch.cuda.synchronize()
    assert np.array_equal(dev.qhist.cpu().numpy(), one.qhist) and np.array_equal(dev.dhist.cpu().numpy(), one.dhist)
    assert np.array_equal(dev.mean.cpu().numpy(), one.mean) and np.array_equal(dev.var.cpu().numpy(), one.var)
    assert np.array_equal(dev.q.cpu().numpy(), one.q)
    x32 = xyz.astype(np.float32)
    a = engine.tet_order_traj_sharded_rccl(x32, boxes, 24, comm, len_wat=3)
    b = engine.tet_order_traj(x32, boxes, len_wat=3)
    assert np.array_equal(a.qhist, b.qhist) and np.array_equal(a.mean, b.mean)
    with pytest.raises(_lib.WolError):        # a block that is not this rank's share of the trajectory
        engine.tet_order_traj_sharded_rccl(xyz[:10], boxes[:10], 24, comm, len_wat=3)
    comm.close(); ctx.close()


def _two_rank_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import numpy as np
    import torch
    import torch.distributed as dist
    from waterorderlib_amd import _lib, engine, synth
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    xyz, boxes = synth.water_frames(2000, 25, seed=63)
    a, b = engine.shard_bounds(25, rank, world)
    ctx = _lib.Context(rank)
    # (i) the torch.distributed exchange (what bench.py uses)
    r1 = engine.tet_order_traj_sharded(torch.from_numpy(xyz[a:b]).cuda(), boxes[a:b], 25, a, ctx=ctx)
    # (ii) the library's own exchange over RCCL
    comm = _lib.Comm.from_torch_distributed(ctx)
    r2 = engine.tet_order_traj_sharded_rccl(xyz[a:b], boxes[a:b], 25, comm)
    torch.cuda.synchronize()
    q.put((rank, r1.qhist.cpu().numpy(), r1.mean.cpu().numpy(), r2.qhist, r2.dhist, r2.mean, r2.var))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_when_two_gpus_are_present():
    """the real multi-GPU path: two ranks, one GPU each, backend nccl (= RCCL); skipped on a one-GPU box"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from waterorderlib_amd import engine, synth
    xyz, boxes = synth.water_frames(2000, 25, seed=63)
    one = engine.tet_order_traj(xyz, boxes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, 29541, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, qh1, m1, qh2, dh2, m2, v2 in got:
        assert np.array_equal(qh1, one.qhist) and np.array_equal(m1, one.mean)
        assert np.array_equal(qh2, one.qhist) and np.array_equal(dh2, one.dhist) and np.array_equal(m2, one.mean) and np.array_equal(v2, one.var)


@pytest.mark.parametrize("rows", [1, 0])
def test_cells_with_hundreds_of_waters(wp_mod, oracle, rows):
    """Blobs far denser than any liquid: cells with more than 255 waters (the 3-D tile kernel enumerates at most 255 per
    cell; waters it does not reach must be finished by the exact kernel, never read from an uninitialised proposal
    buffer -- soak seed 205 faulted the GPU there), fresh contexts, a second call on the same context after a
    different frame"""
    from waterorderlib_amd import _lib
    rng = np.random.default_rng(205)
    box = np.array([110.2, 37.9, 101.2])
    n = 20000
    pos = rng.random((n, 3)) * box
    centres = rng.random((5, 3)) * box
    k = n // 3
    pos[:k] = centres[rng.integers(5, size=k)] + rng.normal(0.0, 2.5, size=(k, 3))
    pos += rng.integers(-1, 2, size=pos.shape) * box
    q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    for attempt in range(2):
        c = _lib.Context(0)
        c.set_option("path", 3)
        c.set_option("search_rows", rows)
        if attempt == 1:   # leave other proposals in the buffers first
            other = rng.random((n, 3)) * box
            wp_mod.getOrderParamq(other, other, box, ctx=c)
        q, idx, dist, ang = wp_mod.getOrderParamq(pos, pos, box, full=True, ctx=c)
        assert np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o, equal_nan=True)
        np.testing.assert_allclose(q, q_o, rtol=0, atol=Q_TOL)
        np.testing.assert_allclose(ang, ang_o, rtol=0, atol=1e-9, equal_nan=True)
        c.close()


def test_tile_order_of_the_proposal_kernel_does_not_show_in_the_results(oracle):
    """the blocks of k_search_rows draw their tiles from per-XCD counters (default) or walk them in a fixed order
    (dyn_tiles = 0): same proposals either way -- every output identical, and equal to the oracle's.  Frames with
    different boxes (different tile counts per frame) and fewer than eight frames (no XCD grouping) included."""
    from waterorderlib_amd import _lib, engine, synth
    n = 6000
    xyz, boxes = synth.water_frames(n, 1, seed=17)
    rng = np.random.default_rng(17)
    for T in (70, 5):
        scale = 1.0 + 0.12 * rng.random((T, 3))
        frames = xyz[0][None] * scale[:, None, :] + rng.normal(0.0, 0.02, size=(T, n, 3))
        bx = boxes[0][None] * scale
        out = {}
        for dyn in (1, 0):
            ctx = _lib.Context(0)
            ctx.set_option("path", 3)
            ctx.set_option("dyn_tiles", dyn)
            out[dyn] = engine.tet_order_traj(frames, bx, want_q=True, ctx=ctx)
            assert ctx.counter("rows_block") in (512, 768)
            ctx.close()
        a, b = out[1], out[0]
        assert np.array_equal(a.q, b.q) and np.array_equal(a.qhist, b.qhist) and np.array_equal(a.dhist, b.dhist)
        assert np.array_equal(a.mean, b.mean) and np.array_equal(a.var, b.var)
        mean, var, qh, dh = oracle.tet_order_traj(frames, bx)
        assert np.array_equal(a.qhist, qh) and np.array_equal(a.dhist, dh)
        np.testing.assert_allclose(a.mean, mean, rtol=0, atol=1e-13)


@pytest.mark.parametrize("stages", [2, 3])
def test_overlapped_batches_equal_the_serial_run(stages):
    """option "overlap": the sort of one batch runs on a second stream beside the proposal kernel of the batch before
    (stages = 3: the certification of the batch before that on a third).  Several batches per call (each batch owns one of
    three sets of intermediate arrays, so the fourth batch reuses the first's), a ragged last batch below the overlap
    threshold, and back-to-back calls that run into each other: everything identical to the serial run."""
    import torch
    from waterorderlib_amd import _lib, engine, synth
    xyz, boxes = synth.water_frames(4000, 300, seed=23)
    dev = torch.from_numpy(xyz).cuda()
    ref_ctx = _lib.Context(0)
    ref = engine.tet_order_traj(dev, boxes, want_q=True, ctx=ref_ctx)
    torch.cuda.synchronize()
    ctx = _lib.Context(0)
    ctx.set_option("overlap", 1)
    ctx.set_option("stages", stages)
    for resident in (0, 1):
        ctx.set_option("inputs_resident", resident)
        for fpb in (64, 72, 0):                  # 5 batches (64 x 4 + 44: the last below the threshold), 72 x 4 + 12, one batch
            ctx.set_option("frames_per_batch", fpb)
            res = engine.tet_order_traj(dev, boxes, want_q=True, ctx=ctx)
            assert torch.equal(res.q, ref.q) and torch.equal(res.qhist, ref.qhist) and torch.equal(res.dhist, ref.dhist)
            assert torch.equal(res.mean, ref.mean) and torch.equal(res.var, ref.var)
    # calls enqueued back to back without a synchronisation in between: the sort of the second starts under the first
    ctx.set_option("frames_per_batch", 64)
    parts = [engine.tet_order_traj(dev[k:k + 100], boxes[k:k + 100], want_q=True, ctx=ctx) for k in (0, 100, 200)]
    assert torch.equal(torch.cat([p.q for p in parts]), ref.q)
    assert torch.equal(sum(p.qhist for p in parts), ref.qhist)
    # ... and a per-frame call right behind them (it takes over the first set of arrays)
    from waterorderlib_amd import water_properties as wp
    q0 = wp.getOrderParamq(xyz[0], xyz[0], boxes[0], ctx=ctx)
    np.testing.assert_allclose(q0, ref.q[0].cpu().numpy(), rtol=0, atol=Q_TOL)
    ctx.close(); ref_ctx.close()


@pytest.mark.parametrize("variant", ["rows512", "rows768", "tiles3d"])
def test_waters_almost_on_top_of_each_other(wp_mod, oracle, variant):
    """two waters a thousandth of an Angstrom apart (overlapping atoms in a broken input, or random test points): the
    squared distance rounds to key 0 in the proposal kernel, like the water's own record -- which must therefore be told
    apart by its position in the tile, not by its distance (found by tools/soak_random.py seed 321: the near twin was
    dropped with the water itself and both got a wrong 4th neighbour).  Exact duplicates (distance 0, which the
    reference's lowCut < d test excludes) included."""
    from waterorderlib_amd import _lib, synth
    n = 6000
    xyz, boxes = synth.water_frames(n, 1, seed=29)
    pos, box = xyz[0].copy(), boxes[0]
    rng = np.random.default_rng(29)
    picks = rng.permutation(n)[:120].reshape(60, 2)
    for k, (w, v) in enumerate(picks):
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        delta = 0.0 if k < 6 else 10.0 ** rng.uniform(-6, -1.7)      # 1e-6 ... 0.02 A
        pos[v] = pos[w] + delta * u + (rng.integers(-1, 2, size=3) * box if k % 3 == 0 else 0.0)
    # three on one spot, and a pair straddling a cell wall
    pos[picks[10, 1]] = pos[picks[11, 0]] + 1e-3
    q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    ctx = _lib.Context(0)
    ctx.set_option("path", 3)
    ctx.set_option("search_rows", 0 if variant == "tiles3d" else 1)
    if variant != "tiles3d":
        ctx.set_option("rows_block", int(variant[4:]))
    q, idx, dist, ang = wp_mod.getOrderParamq(pos, pos, box, full=True, ctx=ctx)
    assert np.array_equal(idx, idx_o) and np.array_equal(dist, dist_o, equal_nan=True)
    np.testing.assert_allclose(q, q_o, rtol=0, atol=Q_TOL)
    frames = np.stack([pos, pos[::-1].copy()])
    from waterorderlib_amd import engine
    res = engine.tet_order_traj(frames, np.stack([box, box]), ctx=ctx)
    mean, var, qh, dh = oracle.tet_order_traj(frames, np.stack([box, box]))
    assert np.array_equal(res.qhist, qh) and np.array_equal(res.dhist, dh)
    ctx.close()


def test_trajectory_mode_q_where_its_shortcut_is_ill_conditioned(oracle):
    """the trajectory mode forms q from the minimum-image vectors d themselves, the reference from (c + d) - c: they differ
    by ~4 ulp(c) / |d|, far below the tolerance for a liquid in its box -- not for frames that sit a hundred thousand box
    lengths from the origin, nor next to a neighbour a few thousandths of an Angstrom away.  Those waters take the exact chain:
    per-water q stays within tolerance, the histograms stay identical (found by tools/soak_traj.py seeds 1017, 2205)."""
    from waterorderlib_amd import _lib, engine, synth
    n = 5000
    xyz, boxes = synth.water_frames(n, 3, seed=31)
    far = xyz + 1.0e5 * boxes[:, None, :] * np.array([1.0, -1.0, 1.0])    # (|dq| reached 1.5e-10 here, 2e-9 at 1e7)
    twins = xyz.copy()
    rng = np.random.default_rng(31)
    for f in range(3):
        for w, v in rng.permutation(n)[:80].reshape(40, 2):
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            twins[f, v] = twins[f, w] + 10.0 ** rng.uniform(-6, -1.5) * u
    ctx = _lib.Context(0)
    for frames in (far, twins):
        res = engine.tet_order_traj(frames, boxes, want_q=True, ctx=ctx)
        mean, var, qh, dh = oracle.tet_order_traj(frames, boxes)
        assert np.array_equal(res.qhist, qh) and np.array_equal(res.dhist, dh)
        np.testing.assert_allclose(res.mean, mean, rtol=0, atol=1e-13)
        for f in range(3):
            np.testing.assert_allclose(res.q[f], oracle.getOrderParamq(frames[f], frames[f], boxes[f], cells=True), rtol=0, atol=Q_TOL)
    ctx.close()


def test_lsi_with_shells_of_hundreds_of_waters(wp_mod, oracle):
    """getLSI sorts the shell's distances in a per-water scratch row sized for a liquid; a droplet of random points with
    hundreds of neighbours inside highCut makes the library size the rows from the kernel's own report and run again
    (it used to refuse such input; tools/soak_shell.py seed 21)"""
    rng = np.random.default_rng(41)
    box = np.array([40.0, 37.0, 43.0])
    pos = rng.random((3000, 3)) * box
    pos[:900] = box * 0.5 + rng.normal(0.0, 2.0, size=(900, 3))
    li, ln = wp_mod.getLSI(pos, pos, box, 0.0, 3.7)
    ri, rn = oracle.getLSI(pos, pos, box, 0.0, 3.7)
    assert rn.max() > 200 and np.array_equal(ln, rn) and li.shape == ri.shape
    np.testing.assert_allclose(li, ri, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [2500, 7000, 40000, 210000])
def test_both_forms_of_the_first_sort_level(oracle, n):
    """the first sort level either sizes row buckets with a counting pass and appends to them (seg_sort = 0) or writes
    every chunk as it lies in LDS with a table of the rows' starts, from which the second level collects its row
    (seg_sort = 2: forced; one chunk, a few, 14, and more than 64 chunks -- two rounds of the per-row piece scan).  The
    order of waters inside a fine bin differs between the two; no result may."""
    from waterorderlib_amd import _lib, engine, synth
    T = 3 if n > 100000 else 70
    xyz, boxes = synth.water_frames(n, 1, seed=37)
    rng = np.random.default_rng(37)
    scale = 1.0 + 0.05 * rng.random((T, 3))
    frames = np.stack([xyz[0][rng.permutation(n)] for _ in range(T)]) * scale[:, None, :]
    bx = boxes[0][None] * scale
    out = {}
    for mode in (0, 2):
        ctx = _lib.Context(0)
        ctx.set_option("seg_sort", mode)
        if T < 64: ctx.set_option("sort_parts", 1)      # (few frames would take the split form of the first level)
        out[mode] = engine.tet_order_traj(frames, bx, want_q=True, ctx=ctx)
        ctx.close()
    a, b = out[0], out[2]
    assert np.array_equal(a.q, b.q) and np.array_equal(a.qhist, b.qhist) and np.array_equal(a.dhist, b.dhist)
    assert np.array_equal(a.mean, b.mean) and np.array_equal(a.var, b.var)
    mean, var, qh, dh = oracle.tet_order_traj(frames, bx)
    assert np.array_equal(b.qhist, qh) and np.array_equal(b.dhist, dh)
    np.testing.assert_allclose(b.mean, mean, rtol=0, atol=1e-13)
