"""Round-3 GPU tests (all through the C ABI):
  * the row-merged proposal kernel writes a key for EVERY water it is given (context option check_proposals: poisoned
    arrays + a counting kernel) on the BASELINE config-2 / 3 / 5 shapes, NPT batches and blobs with hundreds of waters per
    cell -- the invariant that lets the trajectory path skip the per-batch reset of the keys;
  * bounded slices of the three soak generators (tests/soak_gen.py; the scripts that found round 2's three bugs), so the
    driver's own `pytest -m gpu` run exercises them;
  * the frame-sharded entry agrees on errors instead of leaving ranks inside a collective."""
import numpy as np
import pytest

import soak_gen

pytestmark = pytest.mark.gpu

Q_TOL = 1e-12


def _checked_ctx(rows_block=0):
    from waterorderlib_amd import _lib
    c = _lib.Context(0)
    c.set_option("path", 3)
    c.set_option("check_proposals", 1)
    if rows_block:
        c.set_option("rows_block", rows_block)
    return c


def _assert_all_written(c, what):
    exp, wr, bad = c.counter("proposals_expected"), c.counter("proposals_written"), c.counter("proposals_bad")
    assert exp > 0, what
    assert wr == exp, "%s: the proposal kernel wrote %d keys for %d waters" % (what, wr, exp)
    assert bad == 0, "%s: %d proposals name a slot outside their frame" % (what, bad)


def test_row_kernel_writes_a_key_for_every_water(oracle):
    """written == F * n on: the config-2 box (10 000 waters, both instantiations, 70 and 130 frames per batch: the two
    sort forms), the config-3 box (100 000 waters, segment sort), the config-5 slab (fp32 storage, vacuum + vapour), NPT
    batches whose cell counts change from frame to frame, blobs with hundreds of waters per cell (overflowing tiles), a
    context reused across all of them.  The results of the checked runs are the unchecked runs' (same integers)."""
    import torch
    from waterorderlib_amd import _lib, engine, synth
    c = _checked_ctx()
    # config 2 shape, both instantiations
    xyz, boxes = synth.water_frames(10000, 2, seed=3)
    rng = np.random.default_rng(3)
    for blk in (512, 768):
        for T in (70, 130):
            fr = xyz[0][None] + rng.normal(0.0, 0.05, size=(T, 10000, 3))
            c.set_option("rows_block", blk)
            c.set_option("check_proposals", 1)
            r = engine.tet_order_traj(torch.from_numpy(fr).cuda(), np.tile(boxes[0][None], (T, 1)), ctx=c)
            torch.cuda.synchronize()
            assert c.counter("rows_block") == blk
            _assert_all_written(c, "config-2 box, block %d, %d frames" % (blk, T))
            c2 = _lib.Context(0)
            r2 = engine.tet_order_traj(torch.from_numpy(fr).cuda(), np.tile(boxes[0][None], (T, 1)), ctx=c2)
            assert torch.equal(r.qhist, r2.qhist) and torch.equal(r.dhist, r2.dhist) and torch.equal(r.mean, r2.mean)
            c2.close()
    c.set_option("rows_block", 0)
    # config 3 shape: 100 000 waters, 64 frames (segment form of the first sort level), waters in random order
    base, box = synth.lattice_sites(100000, seed=2026)
    base = base[np.random.default_rng(1).permutation(len(base))]
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    fr = torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((64, 100000, 3), generator=g, device="cuda", dtype=torch.float64)
    c.set_option("check_proposals", 1)
    engine.tet_order_traj(fr.contiguous(), np.tile(box[None], (64, 1)), ctx=c)
    torch.cuda.synchronize()
    _assert_all_written(c, "config-3 box")
    del fr
    # config 5 shape: 500 100-water slab, fp32 storage, 8 frames
    xyz5, boxes5 = synth.slab_frames(500000, 1, seed=5, n_vapour=100)
    b5 = torch.from_numpy(xyz5[0]).cuda()
    fr5 = (b5.unsqueeze(0) + 0.05 * torch.randn((8,) + tuple(b5.shape), generator=g, device="cuda", dtype=torch.float64)).to(torch.float32).contiguous()
    c.set_option("check_proposals", 1)
    engine.tet_order_traj(fr5, np.tile(boxes5[0][None], (8, 1)), ctx=c)
    torch.cuda.synchronize()
    if c.counter("rows_block"):     # (frames above 2^20 waters would take the 3-D tile kernel: not this one)
        _assert_all_written(c, "config-5 slab")
    del fr5, b5
    # NPT: a different box (and different cell counts) every frame
    n = 6000
    x1, bx1 = synth.water_frames(n, 1, seed=17)
    for T in (70, 5):
        scale = 1.0 + 0.12 * rng.random((T, 3))
        fr = x1[0][None] * scale[:, None, :] + rng.normal(0.0, 0.02, size=(T, n, 3))
        c.set_option("check_proposals", 1)
        engine.tet_order_traj(fr, bx1[0][None] * scale, ctx=c)
        _assert_all_written(c, "NPT batch of %d frames" % T)
    # blobs: cells with hundreds of waters (tiles that overflow their tables still write -1 for every water)
    pos, bbox = soak_gen.random_case(205)
    from waterorderlib_amd import water_properties as wp
    c.set_option("check_proposals", 1)
    q = wp.getOrderParamq(pos, pos, bbox, ctx=c)
    _assert_all_written(c, "blobs of seed 205")
    np.testing.assert_allclose(q, oracle.getOrderParamq(pos, pos, bbox, cells=True), rtol=0, atol=Q_TOL)
    c.close()


@pytest.mark.parametrize("seed", list(range(100, 112)) + [205, 321, 322, 323, 140, 141, 142, 143])
def test_soak_random_slice(oracle, seed):
    """tools/soak_random.py, 20 seeds (205: the blobs that faulted the GPU in round 2; 321: the twin waters 0.005 A apart):
    every water's indices, distances, angles and q against the oracle, both proposal kernels, proposals checked"""
    from waterorderlib_amd import _lib
    from waterorderlib_amd import water_properties as wp
    pos, box = soak_gen.random_case(seed)
    q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    for rows in (1, 0):
        c = _lib.Context(0)
        c.set_option("path", 3)
        c.set_option("search_rows", rows)
        if rows:
            c.set_option("check_proposals", 1)
        q, idx, dist, ang = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
        if rows and c.counter("proposals_expected"):
            _assert_all_written(c, "seed %d" % seed)
        assert np.array_equal(idx, idx_o), "seed %d rows %d: %d waters differ" % (seed, rows, int((idx != idx_o).any(axis=1).sum()))
        assert np.array_equal(dist, dist_o, equal_nan=True)
        np.testing.assert_allclose(ang, ang_o, rtol=0, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(q, q_o, rtol=0, atol=Q_TOL)
        c.close()


@pytest.mark.parametrize("seed", list(range(0, 14)) + [100, 101, 1017, 2205, 16, 27])
def test_soak_traj_slice(oracle, seed):
    """tools/soak_traj.py, 20 seeds (1017 / 2205: frames a thousand boxes from the origin, round 2's trajectory-mode bug):
    per-frame mean / var and the two histograms against the oracle (histograms bit for bit), host and device input, both
    proposal kernels"""
    import torch
    from waterorderlib_amd import _lib, engine
    case = soak_gen.traj_case(seed, nmax=12000)
    frames, boxes, len_wat, lo, hi, want_q, f32 = (case[k] for k in ("frames", "boxes", "len_wat", "lo", "hi", "want_q", "f32"))
    oxy = np.ascontiguousarray(frames[:, ::len_wat]).astype(np.float64)
    m_o, v_o, qh_o, dh_o = oracle.tet_order_traj(oxy, boxes, lowCut=lo, highCut=hi, cells=True)
    for rows in (1, 0):
        for dev in (False, True):
            c = _lib.Context(0)
            c.set_option("search_rows", rows)
            src = torch.from_numpy(frames).cuda() if dev else frames
            r = engine.tet_order_traj(src, boxes, lowCut=lo, highCut=hi, len_wat=len_wat, want_q=want_q, ctx=c)
            g = (lambda a: a.cpu().numpy()) if dev else (lambda a: a)
            tag = "seed %d rows %d dev %d" % (seed, rows, dev)
            assert np.array_equal(g(r.qhist), qh_o), tag
            assert np.array_equal(g(r.dhist), dh_o), tag
            np.testing.assert_allclose(g(r.mean), m_o, rtol=0, atol=1e-13, err_msg=tag)
            np.testing.assert_allclose(g(r.var), v_o, rtol=0, atol=1e-13, err_msg=tag)
            if want_q:
                np.testing.assert_allclose(g(r.q).astype(np.float64).mean(axis=1), m_o, rtol=0, atol=1e-6 if f32 else 1e-12)
            c.close()


@pytest.mark.parametrize("seed", range(18))
def test_soak_lattice_slice(oracle, seed):
    """tools/soak_lattice.py, 18 seeds: perfect and nearly perfect lattices -- the 4th / 5th neighbour ties send (nearly)
    every water to the exact kernel, whose choice among equals must be the oracle's (lower original index)"""
    from waterorderlib_amd import _lib
    from waterorderlib_amd import water_properties as wp
    pos, box, kind, jitter = soak_gen.lattice_case(seed)
    q_o, idx_o, dist_o, ang_o = oracle.getOrderParamq(pos, pos, box, full=True, cells=True)
    for rows in (1, 0):
        c = _lib.Context(0)
        c.set_option("path", 3)
        c.set_option("search_rows", rows)
        q, idx, dist, ang = wp.getOrderParamq(pos, pos, box, full=True, ctx=c)
        tag = "seed %d %s jitter %g rows %d" % (seed, kind, jitter, rows)
        assert np.array_equal(idx, idx_o), tag
        assert np.array_equal(dist, dist_o, equal_nan=True), tag
        np.testing.assert_allclose(ang, ang_o, rtol=0, atol=1e-9, equal_nan=True, err_msg=tag)
        np.testing.assert_allclose(q, q_o, rtol=0, atol=Q_TOL, err_msg=tag)
        c.close()


def test_sharded_entry_reports_a_local_failure_after_the_exchange():
    """wol_q_traj_sharded with a one-rank RCCL communicator and arguments its local part rejects (a frame count that is not
    this rank's shard; a frame stride smaller than a frame): the call still goes through the exchange step and returns the
    local error -- with more ranks the peers would return WOL_ERR_PEER instead of waiting inside the all-reduce -- and the
    caller's arrays are untouched; the same context then runs a correct call."""
    import ctypes as C
    from waterorderlib_amd import _lib, engine, synth
    L = _lib.load()
    ctx = _lib.Context(0)
    comm = _lib.Comm(ctx, 0, 1, _lib.Comm.unique_id())
    xyz, boxes = synth.water_frames(3000, 4, seed=9)
    mean = np.full(4, -7.0); var = np.full(4, -7.0)
    qh = np.zeros(500, dtype=np.int64); dh = np.zeros(1000, dtype=np.int64)
    args = lambda T_local, fstride: (ctx.handle, comm.handle, _lib.ptr(xyz), 3, fstride, T_local, 4, 3000, _lib.ptr(boxes), 0.0, 10.0,
                                     _lib.ptr(mean), _lib.ptr(var), _lib.ptr(qh), 500, _lib.ptr(dh), 1000, 0.01, None, _lib.MEM_HOST)
    rc = L.wol_q_traj_sharded(*args(3, 9000))          # this rank owns 4 frames, not 3
    assert rc == -1 and b"owns frames" in L.wol_last_error()
    rc = L.wol_q_traj_sharded(*args(4, 100))           # frame stride smaller than one frame
    assert rc == -1 and b"frame_stride" in L.wol_last_error()
    assert np.all(mean == -7.0) and np.all(var == -7.0) and qh.sum() == 0 and dh.sum() == 0
    rc = L.wol_q_traj_sharded(*args(4, 9000))
    assert rc == 0, L.wol_last_error()
    ref = engine.tet_order_traj(xyz, boxes, ctx=ctx)
    assert np.array_equal(qh, ref.qhist) and np.array_equal(dh, ref.dhist) and np.array_equal(mean, ref.mean) and np.array_equal(var, ref.var)
    comm.close()
    ctx.close()


def test_option_ranges_are_checked():
    from waterorderlib_amd import _lib
    c = _lib.Context(0)
    for key, bad in (("stages", 5), ("seg_sort", 7), ("dyn_tiles", 3), ("sort_parts", -1), ("search_margin", 9), ("rows_block", 300)):
        with pytest.raises(_lib.WolError):
            c.set_option(key, bad)
    c.close()
