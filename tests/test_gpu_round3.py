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


# ---- trajectory entries for the other two frame loops and for the populations -------------------------------------------
def _traj_inputs(n=3000, T=5, len_wat=3, seed=31, npops=2):
    from waterorderlib_amd import synth
    xyz, boxes = synth.water_frames(n, T, seed=seed, len_wat=len_wat)
    boxes = boxes * (1.0 + 0.01 * np.arange(T))[:, None]          # a box per frame
    xyz = xyz * (1.0 + 0.01 * np.arange(T))[:, None, None]
    rng = np.random.default_rng(seed)
    O = np.ascontiguousarray(xyz[:, 0::len_wat])
    pos, off = [], [0]
    for t in range(T):
        for j in range(npops):
            if j == 0:
                p = O[t][rng.choice(n, 150 + 10 * t, replace=False)]          # water oxygens (the usual populations)
            elif t == 1:
                p = np.zeros((0, 3))                                           # an empty population in one frame
            else:
                p = rng.random((40, 3)) * boxes[t] * 1.5 - 0.2 * boxes[t]      # arbitrary points, some outside the box
            pos.append(p); off.append(off[-1] + len(p))
    return xyz, boxes, O, np.concatenate(pos), np.asarray(off, dtype=np.int64)


def _hist180(a, bins=500):
    return np.histogram(a, bins=bins, range=[0.0, 180.0])[0].astype(np.int64)


@pytest.mark.parametrize("form", ["host", "device", "f32"])
def test_cos_angs_traj_equals_the_per_frame_chain(oracle, form):
    """wol_cos_angs_traj per (frame, population) against wp.getCosAngs + wp.tetrahedralMetrics on the same frames:
    integer histogram and counts exact (also against the oracle's angles), cosine sums and the derived statistics to
    1e-12; TIP3P-interleaved frames, a box per frame, populations of water oxygens / arbitrary points / none"""
    import torch
    from waterorderlib_amd import engine
    from waterorderlib_amd import water_properties as wp
    xyz, boxes, O, pop_pos, pop_off = _traj_inputs()
    if form == "f32":
        xyz = xyz.astype(np.float32); pop_pos = pop_pos.astype(np.float32)
        O = np.ascontiguousarray(xyz[:, 0::3]).astype(np.float64)
    fr = torch.from_numpy(xyz).cuda() if form == "device" else xyz
    res = engine.cos_angs_traj(fr, boxes, len_wat=3, pop_pos=pop_pos, pop_off=pop_off, npops=2)
    g = (lambda a: a.cpu().numpy()) if form == "device" else (lambda a: a)
    hist, counts, sums = g(res.hist), g(res.counts), g(res.cos_sums)
    T = len(xyz)
    for t in range(T):
        for j in range(3):
            sub = O[t] if j == 0 else pop_pos[pop_off[2 * t + j - 1]:pop_off[2 * t + j]].astype(np.float64)
            ang = wp.getCosAngs(sub, O[t], boxes[t])[0] if len(sub) else np.zeros(0)
            tag = "frame %d population %d" % (t, j)
            assert np.array_equal(hist[t, j], _hist180(ang)), tag
            tet = ang[(ang >= 100.0) & (ang <= 120.0)]
            assert counts[t, j, 0] == len(ang) and counts[t, j, 1] == len(tet), tag
            cs = np.cos(tet * np.pi / 180.0)
            assert abs(sums[t, j, 0] - cs.sum()) < 1e-9 and abs(sums[t, j, 1] - (cs * cs).sum()) < 1e-9, tag
            if len(ang) and form != "f32":
                ang_o = oracle.getCosAngs(sub, O[t], boxes[t])[0]
                assert np.array_equal(hist[t, j], _hist180(ang_o)), tag + " (oracle)"
            if len(tet):
                _, _, frac, avg, var, ent = wp.tetrahedralMetrics(ang)
                h, a, b, c_, d = res.metrics(t, j)
                assert abs(a - frac) < 1e-15 and abs(b - avg) < 1e-12 and abs(c_ - var) < 1e-12 and abs(d - ent) < 1e-12, tag


def test_cos_angs_traj_golden_and_big_batches(oracle):
    """the reference's own golden angles (gas512, ice512) binned; a 64-frame batch of 30 000 waters (streaming sort forms,
    several blocks per frame) against the per-frame call; dense blobs whose shells outgrow the LDS rows"""
    import torch
    from conftest import golden
    from waterorderlib_amd import engine, synth
    from waterorderlib_amd import water_properties as wp
    for name in ("gas512", "ice512"):
        gd = golden(name)
        res = engine.cos_angs_traj(gd["pos"][None], gd["box"][None])
        assert np.array_equal(res.hist[0, 0], _hist180(gd["cos_ang"])), name
        assert np.array_equal(res.hist[0, 0], gd["ang_hist"].astype(np.int64)), name     # the reference's own np.histogram
        assert res.counts[0, 0, 0] == len(gd["cos_ang"])
        lr = engine.lsi_traj(gd["pos"][None], gd["box"][None], want_lsi=True)
        vals = lr.lsi[0][~np.isnan(lr.lsi[0])]
        np.testing.assert_allclose(vals, gd["lsi"], rtol=0, atol=1e-13, err_msg=name)
        assert lr.counts[0, 0] == len(gd["lsi"])
    n, T = 30000, 64
    base, box = synth.lattice_sites(n, seed=4)
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    fr = (torch.from_numpy(base).cuda().unsqueeze(0) + 0.3 * torch.randn((T, n, 3), generator=g, device="cuda", dtype=torch.float64)).contiguous()
    res = engine.cos_angs_traj(fr, np.tile(box[None], (T, 1)))
    for t in (0, 37, 63):
        ang = wp.getCosAngs(fr[t].cpu().numpy(), fr[t].cpu().numpy(), box)[0]
        assert np.array_equal(res.hist[t, 0].cpu().numpy(), _hist180(ang))
    # blobs: shells of hundreds of waters (listed again by the slow kernel), long highCut
    rng = np.random.default_rng(5)
    bx = np.array([40.0, 38.0, 42.0])
    pos = rng.random((3, 2500, 3)) * bx
    pos[:, :300] = bx * 0.5 + rng.normal(0.0, 1.2, size=(3, 300, 3))
    res = engine.cos_angs_traj(pos, np.tile(bx[None], (3, 1)), highCut=3.0)
    for t in range(3):
        ang = wp.getCosAngs(pos[t], pos[t], bx, 0.0, 3.0)[0]
        assert np.array_equal(res.hist[t, 0], _hist180(ang)) and res.counts[t, 0, 0] == len(ang)


@pytest.mark.parametrize("form", ["host", "device", "f32"])
def test_lsi_traj_equals_the_per_frame_chain(oracle, form):
    """wol_lsi_traj against wp.getLSI per frame and population: per-water LSI bit for bit (NaN where the reference skips
    the water), len(lsiVals), np.mean / np.var to 1e-14, the pooled 500-bin histogram over [0, 0.3] exact"""
    import torch
    from waterorderlib_amd import engine
    from waterorderlib_amd import water_properties as wp
    xyz, boxes, O, pop_pos, pop_off = _traj_inputs(seed=32)
    if form == "f32":
        xyz = xyz.astype(np.float32); pop_pos = pop_pos.astype(np.float32)
        O = np.ascontiguousarray(xyz[:, 0::3]).astype(np.float64)
    fr = torch.from_numpy(xyz).cuda() if form == "device" else xyz
    res = engine.lsi_traj(fr, boxes, len_wat=3, pop_pos=pop_pos, pop_off=pop_off, npops=2, want_lsi=True, want_pop_lsi=True)
    g = (lambda a: a.cpu().numpy()) if form == "device" else (lambda a: a)
    hist, counts, mean, var, lsi, plsi = (g(a) for a in (res.hist, res.counts, res.mean, res.var, res.lsi, res.pop_lsi))
    pooled = np.zeros((3, 500), dtype=np.int64)
    for t in range(len(xyz)):
        for j in range(3):
            sub = O[t] if j == 0 else pop_pos[pop_off[2 * t + j - 1]:pop_off[2 * t + j]].astype(np.float64)
            tag = "frame %d population %d" % (t, j)
            if len(sub) == 0:
                assert counts[t, j] == 0 and np.isnan(mean[t, j]) and np.isnan(var[t, j]), tag
                continue
            full = wp.getLSI(sub, O[t], boxes[t], full=True)[0]
            vals = full[~np.isnan(full)]
            mine = lsi[t] if j == 0 else plsi[pop_off[2 * t + j - 1]:pop_off[2 * t + j]]
            if form == "f32":
                np.testing.assert_allclose(mine.astype(np.float64), full, rtol=1e-6, atol=0, equal_nan=True, err_msg=tag)
            else:
                assert np.array_equal(mine, full, equal_nan=True), tag
                if j == 0:
                    np.testing.assert_allclose(vals, oracle.getLSI(sub, O[t], boxes[t])[0], rtol=0, atol=1e-13)
            assert counts[t, j] == len(vals), tag
            assert abs(mean[t, j] - np.mean(vals)) < 1e-14 and abs(var[t, j] - np.var(vals)) < 1e-14, tag
            pooled[j] += np.histogram(vals, bins=500, range=[0.0, 0.3])[0]
    assert np.array_equal(hist, pooled)


def test_lsi_traj_with_shells_of_dozens_and_hundreds_of_waters():
    """shells that outgrow the LDS rows (a long highCut: ~40 waters; a blob: hundreds) take the slow kernel's global rows:
    the same values as wp.getLSI, bit for bit"""
    from waterorderlib_amd import engine, synth
    from waterorderlib_amd import water_properties as wp
    xyz, boxes = synth.water_frames(2000, 3, seed=8)
    res = engine.lsi_traj(xyz, boxes, highCut=6.5, want_lsi=True)
    for t in range(3):
        full = wp.getLSI(xyz[t], xyz[t], boxes[t], highCut=6.5, full=True)[0]
        assert np.array_equal(res.lsi[t], full, equal_nan=True)
        assert res.counts[t, 0] == int((~np.isnan(full)).sum())
    rng = np.random.default_rng(6)
    bx = np.array([40.0, 38.0, 42.0])
    pos = rng.random((2, 2500, 3)) * bx
    pos[:, :300] = bx * 0.5 + rng.normal(0.0, 1.5, size=(2, 300, 3))
    res = engine.lsi_traj(pos, np.tile(bx[None], (2, 1)), want_lsi=True)
    for t in range(2):
        full = wp.getLSI(pos[t], pos[t], bx, full=True)[0]
        assert np.array_equal(res.lsi[t], full, equal_nan=True)
    # a droplet in every frame: more dense shells in one batch (6 x 1000) than any fixed list would hold, population points
    # inside the droplet as well (found by tools/soak_shell_traj.py, seed 49)
    T = 6
    pos = rng.random((T, 4000, 3)) * bx
    pos[:, :1000] = bx * 0.5 + rng.normal(0.0, 3.0, size=(T, 1000, 3))
    pts = np.ascontiguousarray(pos[:, 900:1100]).reshape(-1, 3)
    off = np.arange(T + 1, dtype=np.int64) * 200
    res = engine.lsi_traj(pos, np.tile(bx[None], (T, 1)), pop_pos=pts, pop_off=off, npops=1, want_lsi=True, want_pop_lsi=True)
    for t in range(T):
        full = wp.getLSI(pos[t], pos[t], bx, full=True)[0]
        assert np.array_equal(res.lsi[t], full, equal_nan=True)
        assert np.array_equal(res.pop_lsi[off[t]:off[t + 1]], full[900:1100], equal_nan=True)


@pytest.mark.parametrize("form", ["host", "device", "f32"])
def test_q_traj_pops_equals_the_per_frame_calls(oracle, form):
    """wol_q_traj_pops: population 0 as wol_q_traj; every population point's q bit for bit that of wp.getOrderParamq(subPos,
    watPos) (and the oracle's to 1e-12), np.mean / np.var per frame, the histograms exact; an empty population gives NaN"""
    import torch
    from waterorderlib_amd import engine
    from waterorderlib_amd import water_properties as wp
    xyz, boxes, O, pop_pos, pop_off = _traj_inputs(seed=33)
    if form == "f32":
        xyz = xyz.astype(np.float32); pop_pos = pop_pos.astype(np.float32)
        O = np.ascontiguousarray(xyz[:, 0::3]).astype(np.float64)
    fr = torch.from_numpy(xyz).cuda() if form == "device" else xyz
    res = engine.tet_order_traj_pops(fr, boxes, pop_pos, pop_off, 2, len_wat=3, want_pop_q=True, want_dhist=True)
    ref = engine.tet_order_traj(fr, boxes, len_wat=3)
    g = (lambda a: a.cpu().numpy()) if form == "device" else (lambda a: a)
    assert np.array_equal(g(res.qhist), g(ref.qhist)) and np.array_equal(g(res.dhist), g(ref.dhist))
    assert np.array_equal(g(res.mean), g(ref.mean)) and np.array_equal(g(res.var), g(ref.var))
    pm, pv, pqh, pq = (g(a) for a in (res.pop_mean, res.pop_var, res.pop_qhist, res.pop_q))
    hist = np.zeros((2, 500), dtype=np.int64)
    for t in range(len(xyz)):
        for j in range(2):
            sub = pop_pos[pop_off[2 * t + j]:pop_off[2 * t + j + 1]].astype(np.float64)
            tag = "frame %d population %d" % (t, j + 1)
            if len(sub) == 0:
                assert np.isnan(pm[t, j]) and np.isnan(pv[t, j]), tag
                continue
            q = wp.getOrderParamq(sub, O[t], boxes[t])
            mine = pq[pop_off[2 * t + j]:pop_off[2 * t + j + 1]]
            if form == "f32":
                np.testing.assert_allclose(mine.astype(np.float64), q, rtol=0, atol=1e-6, err_msg=tag)
            else:
                assert np.array_equal(mine, q), tag
                np.testing.assert_allclose(q, oracle.getOrderParamq(sub, O[t], boxes[t]), rtol=0, atol=Q_TOL)
            assert abs(pm[t, j] - np.mean(q)) < 1e-13 and abs(pv[t, j] - np.var(q)) < 1e-13, tag
            hist[j] += np.histogram(q, bins=500, range=[0.0, 1.0])[0]
    assert np.array_equal(pqh, hist)


def test_pops_and_shell_trajectories_on_frames_without_a_grid(oracle):
    """a non-periodic axis (BoxL < 0, waterlib.f90:41) and frames of a few hundred waters (all-pairs path): the trajectory
    entries fall back to all-pairs walks over the frame with the same results as the per-frame calls"""
    from waterorderlib_amd import engine
    from waterorderlib_amd import water_properties as wp
    rng = np.random.default_rng(12)
    box = np.array([22.0, 25.0, -1.0])
    pos = rng.random((3, 700, 3)) * np.array([22.0, 25.0, 30.0])
    boxes = np.tile(box[None], (3, 1))
    sub = [pos[t][rng.choice(700, 60, replace=False)] for t in range(3)]
    pop_pos = np.concatenate(sub); pop_off = np.arange(4, dtype=np.int64) * 60
    r = engine.tet_order_traj_pops(pos, boxes, pop_pos, pop_off, 1, want_pop_q=True)
    ca = engine.cos_angs_traj(pos, boxes, pop_pos=pop_pos, pop_off=pop_off, npops=1)
    ls = engine.lsi_traj(pos, boxes, pop_pos=pop_pos, pop_off=pop_off, npops=1, want_lsi=True)
    for t in range(3):
        assert np.array_equal(r.pop_q[60 * t:60 * t + 60], wp.getOrderParamq(sub[t], pos[t], box))
        ang = wp.getCosAngs(pos[t], pos[t], box)[0]
        assert np.array_equal(ca.hist[t, 0], _hist180(ang))
        assert np.array_equal(ca.hist[t, 1], _hist180(wp.getCosAngs(sub[t], pos[t], box)[0]))
        assert np.array_equal(ls.lsi[t], wp.getLSI(pos[t], pos[t], box, full=True)[0], equal_nan=True)


def test_tiled_pair_histogram_equals_the_walk_and_the_oracle(oracle):
    """RadialDistSame through the LDS-tiled kernel (fp32 proposal of the bin, exact decision near an edge) against the
    lane-per-water walk (exact arithmetic throughout) and the oracle: g(r) bit for bit -- small boxes where a tile sees
    several images, orthorhombic boxes, unwrapped coordinates, bin widths from 0.01 to 0.37 A, a 60 000-water frame"""
    from waterorderlib_amd import _lib, synth
    from waterorderlib_amd import waterlib as wl
    rng = np.random.default_rng(77)
    cases = []
    for n, box in ((700, None), (3000, np.array([31.0, 52.0, 44.0])), (9000, np.array([95.0, 40.0, 71.0]))):
        xyz, boxes = synth.water_frames(n, 1, seed=n, box=box)
        pos = xyz[0] + rng.integers(-1, 2, size=xyz[0].shape) * boxes[0]        # some waters a box away
        cases.append((pos, boxes[0]))
    ctx = _lib.default_context()
    for pos, box in cases:
        for bw, nb in ((0.1, 150), (0.01, 1000), (0.37, 40), (0.05, 90)):
            ctx.set_option("pair_tiled", 1)
            g1 = wl.radialdistsame(pos, bw, nb, 1.0, box)
            ctx.set_option("pair_tiled", 0)
            g0 = wl.radialdistsame(pos, bw, nb, 1.0, box)
            assert np.array_equal(g1, g0), (len(pos), bw)
            if len(pos) <= 3000:
                assert np.array_equal(g1, oracle.radialdistsame(pos, bw, nb, 1.0, box)), (len(pos), bw)
    xyz, boxes = synth.water_frames(60000, 1, seed=5)
    ctx.set_option("pair_tiled", 1)
    g1 = wl.radialdistsame(xyz[0], 0.1, 150, 1.0, boxes[0])
    ctx.set_option("pair_tiled", 0)
    assert np.array_equal(g1, wl.radialdistsame(xyz[0], 0.1, 150, 1.0, boxes[0]))
    ctx.set_option("pair_tiled", 1)


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 5, 9, 13, 101])
def test_soak_shell_trajectories_slice(seed):
    """the trajectory generators of tools/soak_traj.py (NPT boxes, droplets, interleaved waters, float32 storage, frames
    thousands of boxes from the origin) through wol_cos_angs_traj / wol_lsi_traj / wol_q_traj_pops against the per-frame
    calls on the same frames: angle histograms and counts exact, per-water LSI bit for bit, population q bit for bit"""
    from waterorderlib_amd import engine
    from waterorderlib_amd import water_properties as wp
    case = soak_gen.traj_case(seed, nmax=2500)
    frames, boxes, len_wat, f32 = case["frames"][:6], case["boxes"][:6], case["len_wat"], case["f32"]
    T = len(frames)
    O = np.ascontiguousarray(frames[:, ::len_wat]).astype(np.float64)
    rng = np.random.default_rng(seed)
    sub = [O[t][rng.choice(O.shape[1], min(50, O.shape[1]), replace=False)] for t in range(T)]
    pop_pos = np.concatenate(sub).astype(frames.dtype)
    pop_off = np.concatenate([[0], np.cumsum([len(x) for x in sub])]).astype(np.int64)
    ca = engine.cos_angs_traj(frames, boxes, len_wat=len_wat, pop_pos=pop_pos, pop_off=pop_off, npops=1)
    ls = engine.lsi_traj(frames, boxes, len_wat=len_wat, want_lsi=True)
    qp = engine.tet_order_traj_pops(frames, boxes, pop_pos, pop_off, 1, len_wat=len_wat, want_pop_q=True)
    for t in range(T):
        ang = wp.getCosAngs(O[t], O[t], boxes[t])[0]
        assert np.array_equal(ca.hist[t, 0], _hist180(ang)) and ca.counts[t, 0, 0] == len(ang), (seed, t)
        angp = wp.getCosAngs(sub[t].astype(frames.dtype).astype(np.float64), O[t], boxes[t])[0]
        assert np.array_equal(ca.hist[t, 1], _hist180(angp)), (seed, t)
        full = wp.getLSI(O[t], O[t], boxes[t], full=True)[0]
        if f32:
            np.testing.assert_allclose(ls.lsi[t].astype(np.float64), full, rtol=1e-6, atol=0, equal_nan=True)
        else:
            assert np.array_equal(ls.lsi[t], full, equal_nan=True), (seed, t)
        q = wp.getOrderParamq(sub[t].astype(frames.dtype).astype(np.float64), O[t], boxes[t])
        mine = qp.pop_q[pop_off[t]:pop_off[t + 1]]
        if f32:
            np.testing.assert_allclose(mine.astype(np.float64), q, rtol=0, atol=1e-6)
        else:
            assert np.array_equal(mine, q), (seed, t)


def test_trajectory_entries_reject_bad_arguments_and_take_odd_shapes():
    """status codes, not crashes: population offsets that do not start at 0 / decrease / disagree with the points; NULL
    frames; nbins out of range; T = 0; a histogram of 12 000 bins (more LDS than the default allows)"""
    from waterorderlib_amd import _lib, engine, synth
    L = _lib.load()
    xyz, boxes = synth.water_frames(1500, 3, seed=4)
    ctx = _lib.default_context()
    pts = np.ascontiguousarray(xyz[:, :10].reshape(-1, 3))
    good = np.arange(4, dtype=np.int64) * 10
    hist = np.zeros((3, 2, 500), np.int64); cnt = np.zeros((3, 2, 2), np.int64); cs = np.zeros((3, 2, 2))
    call = lambda off, nb=500, fr=xyz: L.wol_cos_angs_traj(ctx.handle, _lib.ptr(fr) if fr is not None else None, 3, 4500, 3, 1500, _lib.ptr(boxes), 0.0, 3.413,
                                                           _lib.ptr(pts), _lib.ptr(off), 1, nb, _lib.ptr(hist), _lib.ptr(cnt), _lib.ptr(cs), 0, 0)
    assert call(good) == 0
    assert call(np.array([1, 10, 20, 30], dtype=np.int64)) == -1 and b"start at 0" in L.wol_last_error()
    assert call(np.array([0, 20, 10, 30], dtype=np.int64)) == -1 and b"decrease" in L.wol_last_error()
    assert call(good, nb=0) == -1 and call(good, nb=20000) == -1
    assert call(good, fr=None) == -1
    with pytest.raises(ValueError):
        engine.cos_angs_traj(xyz, boxes, pop_pos=pts, pop_off=np.array([0, 10, 20, 31]), npops=1)
    r = engine.cos_angs_traj(xyz[:0], boxes[:0])
    assert r.hist.shape == (0, 1, 500)
    big = engine.cos_angs_traj(xyz, boxes, nbins=12000)
    ref = engine.cos_angs_traj(xyz, boxes, nbins=500)
    assert np.array_equal(big.hist.reshape(3, 1, 500, 24).sum(-1), ref.hist) and np.array_equal(big.counts, ref.counts)
    lr = engine.lsi_traj(xyz, boxes, nbins=3000, hist_max=0.6)
    assert lr.hist.sum() <= lr.counts.sum() and lr.hist.shape == (1, 3000)


def test_long_thin_box_through_the_segment_sort(oracle):
    """a slab-like box with hundreds of cell rows per layer and dozens of chunks per frame: the second sort level's LDS copy of
    the row table outgrows the default dynamic-LDS limit (the launch used to fail there) -- the library
    raises the kernel's limit or falls back to the row-bucket form; results equal the row-bucket form's and the oracle's"""
    import torch
    from waterorderlib_amd import _lib, engine
    rng = np.random.default_rng(3)
    box = np.array([120.0, 2600.0, 27.0])       # 577 cell rows per layer x 66 chunks: 76 KB of row table per block
    n, T = 200000, 64
    base = rng.random((n, 3)) * box
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    fr = (torch.from_numpy(base).cuda().unsqueeze(0) + 0.05 * torch.randn((T, n, 3), generator=g, device="cuda", dtype=torch.float64)).contiguous()
    boxes = np.tile(box[None], (T, 1))
    res = {}
    for seg in (2, 0):
        c = _lib.Context(0)
        c.set_option("seg_sort", seg)
        res[seg] = engine.tet_order_traj(fr, boxes, want_q=True, ctx=c)
        torch.cuda.synchronize()
        c.close()
    assert torch.equal(res[2].qhist, res[0].qhist) and torch.equal(res[2].dhist, res[0].dhist)
    assert torch.equal(res[2].mean, res[0].mean) and torch.equal(res[2].q, res[0].q)
    pos = fr[5].cpu().numpy()
    sample = rng.choice(n, 1500, replace=False)
    np.testing.assert_allclose(res[2].q[5].cpu().numpy()[sample], oracle.getOrderParamq(pos[sample], pos, box, cells=True), rtol=0, atol=Q_TOL)


@pytest.mark.parametrize("form", ["host", "device", "f32"])
def test_rdf_trajectory_equals_the_per_frame_calls(oracle, form):
    """wol_radial_dist_same_traj: every frame's g(r) is wl.radialdistsame's for that frame, bit for bit (and the oracle's);
    interleaved waters, a box per frame, a frame without a grid (non-periodic axis) in the middle"""
    import torch
    from waterorderlib_amd import engine, synth
    from waterorderlib_amd import waterlib as wl
    xyz, boxes = synth.water_frames(2500, 6, seed=21, len_wat=3)
    boxes = boxes * (1.0 + 0.02 * np.arange(6))[:, None]
    xyz = xyz * (1.0 + 0.02 * np.arange(6))[:, None, None]
    boxes[3, 2] = -1.0                                   # frame 3: non-periodic along z
    if form == "f32":
        xyz = xyz.astype(np.float32)
    fr = torch.from_numpy(xyz).cuda() if form == "device" else xyz
    rdf, counts = engine.radial_dist_same_traj(fr, boxes, 0.1, 120, 1.0, len_wat=3, want_counts=True)
    if form == "device":
        rdf, counts = rdf.cpu().numpy(), counts.cpu().numpy()
    for t in range(6):
        O = np.ascontiguousarray(xyz[t, 0::3]).astype(np.float64)
        ref = wl.radialdistsame(O, 0.1, 120, 1.0, boxes[t])
        assert np.array_equal(rdf[t], ref), t
        assert np.array_equal(ref, oracle.radialdistsame(O, 0.1, 120, 1.0, boxes[t])), t
        assert counts[t].sum() > 0


@pytest.mark.parametrize("shape", ["many_small", "few_big", "interleaved", "long_rows", "thin_box", "host"])
def test_float32_frames_keep_16_byte_records_bit_for_bit(shape):
    """float32 frames go through both sort levels and the certification kernel's gathers as 16-byte (x, y, z, index)
    records (option rec32, default) -- on every form of the first sort level (one block per frame: segments and row buckets;
    frames split over blocks; tables of thousands of rows; rows of hundreds of waters) the per-water q, the per-frame
    statistics and both integer histograms equal those of the 32-byte records AND those of the same frames promoted to
    float64 on the caller's side, bit for bit."""
    import torch
    from waterorderlib_amd import engine, synth, _lib
    rng = np.random.default_rng(17)
    if shape == "many_small":
        xyz, boxes = synth.water_frames(3000, 96, seed=3)              # one block per frame, row buckets
    elif shape == "few_big":
        xyz, boxes = synth.water_frames(120000, 4, seed=4)             # frames split over blocks (count / scan / scatter)
    elif shape == "interleaved":
        xyz, boxes = synth.water_frames(30000, 64, seed=5)             # one block per frame, segment form
    elif shape == "long_rows":
        bx = np.array([400.0, 21.0, 21.0])                             # 25 rows of ~240 waters: four waters per lane in a row
        xyz = (rng.random((64, 6000, 3)) * bx); boxes = np.tile(bx[None], (64, 1))
    elif shape == "thin_box":
        bx = np.array([21.0, 50.0, 1700.0])                            # 4 444 short rows: the first level's big row tables
        xyz = (rng.random((8, 60000, 3)) * bx); boxes = np.tile(bx[None], (8, 1))
    else:
        xyz, boxes = synth.water_frames(20000, 70, seed=6)
    f32 = np.ascontiguousarray(xyz.astype(np.float32))
    if shape == "interleaved":                                         # TIP4P-like: the oxygens are every fourth atom
        full = np.full((f32.shape[0], f32.shape[1] * 4, 3), 1.0e6, np.float32)
        full[:, ::4] = f32
        f32, len_wat = full, 4
    else:
        len_wat = 1
    ctx = _lib.Context()
    dev = (lambda a: a) if shape == "host" else (lambda a: torch.as_tensor(a, device="cuda"))
    out = {}
    for rec in (1, 0):
        ctx.set_option("rec32", rec)
        r = engine.tet_order_traj(dev(f32), boxes, len_wat=len_wat, want_q=True, ctx=ctx)
        torch.cuda.synchronize()
        assert ctx.counter("record_bytes") == (16 if rec else 32)
        out[rec] = [np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (r.mean, r.var, r.qhist, r.dhist, r.q)]
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)
    wide = np.ascontiguousarray(f32[:, ::len_wat].astype(np.float64))
    r64 = engine.tet_order_traj(dev(wide), boxes, want_q=True, ctx=ctx)
    torch.cuda.synchronize()
    ref = [np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (r64.mean, r64.var, r64.qhist, r64.dhist)]
    for a, b in zip(out[1][:4], ref):
        assert np.array_equal(a, b)
    q64 = np.asarray(r64.q.cpu() if hasattr(r64.q, "cpu") else r64.q)
    assert np.array_equal(out[1][4], q64.astype(np.float32))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_split_single_pass_sort_of_big_frames_in_small_batches(dtype):
    """big frames in small batches: every frame's first sort level is split over P blocks.  Single-pass form (seg_sort = 1,
    default: block p writes the chunks p, p + P, ... as they lie in LDS, the blocks' row totals are summed by
    k_sort_rows_scan) against the counting form (seg_sort = 0: count / scan / scatter): per-water q, statistics and both
    histograms bit for bit, float64 frames and float32 frames (16-byte records); forced block counts too (P = 3: a block
    takes many chunks; P larger than the number of chunks)."""
    import torch
    from waterorderlib_amd import engine, synth, _lib
    xyz, boxes = synth.water_frames(150000, 4, seed=11)
    rng = np.random.default_rng(5)
    xyz = np.stack([xyz[t][rng.permutation(xyz.shape[1])] for t in range(xyz.shape[0])])
    fr = torch.as_tensor(xyz.astype(np.float32) if dtype == "f32" else xyz, device="cuda")
    out = []
    for seg, parts in ((0, 0), (1, 0), (1, 3), (1, 64)):
        ctx = _lib.Context()
        ctx.set_option("seg_sort", seg)
        ctx.set_option("sort_parts", parts)
        r = engine.tet_order_traj(fr, boxes, want_q=True, ctx=ctx)
        torch.cuda.synchronize()
        out.append([np.asarray(x.cpu()) for x in (r.mean, r.var, r.qhist, r.dhist, r.q)])
        ctx.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", ["bench_like", "few_big", "long_rows", "thin_box", "far_away", "host_packed", "host_o_only"])
def test_index_records_equal_the_fp64_records_bit_for_bit(shape):
    """float64 frames of bare O rows go through the sort as 16-byte index records (fp32 position inside the cell + original
    index | cell | fine bin; option idx_sort, default) and the certification / the exact search read the coordinates from
    the caller's frames by original index -- against the classic 32-byte records (idx_sort = 0): per-water q, statistics and
    both integer histograms bit for bit, over both forms of the single-pass first level (one block per frame, split over
    blocks), long rows, thousands of short rows, frames thousands of boxes from the origin (a tenth of the waters then take
    the exact path), host frames packed by the feeder and host frames that already are O rows."""
    import torch
    from waterorderlib_amd import engine, synth, _lib
    rng = np.random.default_rng(23)
    len_wat = 1
    if shape == "bench_like":
        xyz, boxes = synth.water_frames(40000, 70, seed=2)
    elif shape == "few_big":
        xyz, boxes = synth.water_frames(130000, 4, seed=3)
    elif shape == "long_rows":
        bx = np.array([570.0, 21.0, 21.0])                             # 16 rows of 1 900 waters: counted, then scattered
        xyz = rng.random((64, 30000, 3)) * bx; boxes = np.tile(bx[None], (64, 1))
    elif shape == "thin_box":
        bx = np.array([21.0, 50.0, 1700.0])                            # 4 444 rows: the first level takes the counting form,
        xyz = rng.random((8, 60000, 3)) * bx; boxes = np.tile(bx[None], (8, 1))   # i.e. classic records either way
    elif shape == "far_away":
        xyz, boxes = synth.water_frames(30000, 66, seed=4)
        xyz = xyz + 3000.0 * boxes[:, None, :] * rng.integers(-1, 2, size=(xyz.shape[0], 1, 3))
    elif shape == "host_packed":
        xyz, boxes = synth.water_frames(30000, 70, seed=5, len_wat=4)
        len_wat = 4
    else:
        xyz, boxes = synth.water_frames(30000, 70, seed=6)
    xyz = np.ascontiguousarray(np.stack([x.reshape(-1, len_wat, 3)[rng.permutation(x.shape[0] // len_wat)].reshape(-1, 3) for x in xyz]))
    host = shape.startswith("host")
    fr = xyz if host else torch.as_tensor(xyz, device="cuda")
    out = {}
    for mode in (1, 0):
        ctx = _lib.Context()
        ctx.set_option("idx_sort", mode)
        r = engine.tet_order_traj(fr, boxes, len_wat=len_wat, want_q=True, ctx=ctx)
        torch.cuda.synchronize()
        assert ctx.counter("record_bytes") == (4 if mode and shape != "thin_box" else 32)
        out[mode] = [np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (r.mean, r.var, r.qhist, r.dhist, r.q)]
        ctx.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)
