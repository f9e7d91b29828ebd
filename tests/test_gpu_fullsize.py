"""BASELINE.json's full-size configurations on the GPU, checked through properties that do not need the O(N^2)
reference: exact integer histogram totals, invariance of every integer / fixed-point result under a permutation of the
waters, frame sharding (config 4) reproducing the unsharded run bit for bit, idempotence -- plus a sampled comparison
with the CPU oracle.  Config 3: 100 000 waters x 1 000 frames, fp64.  Config 5: 500 000-water slab with vapour, fp32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(torch, base, box, T, seed, sigma=0.3, dtype=None):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = base.unsqueeze(0) + sigma * torch.randn((T,) + tuple(base.shape), generator=g, device="cuda", dtype=torch.float64)
    b = torch.as_tensor(box, device="cuda")
    x = x - b * torch.floor(x / b)
    return x.to(dtype) if dtype is not None else x


def test_config3_full_size_properties(oracle):
    import torch
    from waterorderlib_amd import engine, synth
    n, T = 100000, 1000
    base, box = synth.lattice_sites(n, seed=2026)
    base = torch.from_numpy(base[np.random.default_rng(1).permutation(n)]).cuda()
    frames = _frames(torch, base, box, T, seed=11)
    boxes = np.tile(box[None, :], (T, 1))
    res = engine.tet_order_traj(frames, boxes, want_q=True)
    torch.cuda.synchronize()
    # (i) totals: every water has four neighbours inside 10 A; the q histogram holds exactly the q inside [0, 1]
    assert int(res.dhist.sum().item()) == 4 * n * T
    q = res.q
    assert int(res.qhist.sum().item()) == int(((q >= 0.0) & (q <= 1.0)).sum().item())
    # (ii) per-frame statistics against the q array (np.mean / np.var semantics; the kernel's sums are exact)
    np.testing.assert_allclose(res.mean.cpu().numpy(), q.mean(dim=1).cpu().numpy(), rtol=0, atol=1e-13)
    np.testing.assert_allclose(res.var.cpu().numpy(), q.var(dim=1, unbiased=False).cpu().numpy(), rtol=0, atol=1e-13)
    # (iii) a permutation of the waters changes nothing that is an integer or an exact sum
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    res_p = engine.tet_order_traj(frames[:, perm, :].contiguous(), boxes, want_q=True)
    assert torch.equal(res.qhist, res_p.qhist) and torch.equal(res.dhist, res_p.dhist)
    assert torch.equal(res.mean, res_p.mean) and torch.equal(res.var, res_p.var)
    assert torch.equal(res.q[:, perm], res_p.q)
    # (iv) config 4's sharding: eight blocks of frames processed on their own add up to the whole, bit for bit
    qh = torch.zeros_like(res.qhist); dh = torch.zeros_like(res.dhist); means = []
    for r in range(8):
        a, b = engine.shard_bounds(T, r, 8)
        part = engine.tet_order_traj(frames[a:b], boxes[a:b])
        qh += part.qhist; dh += part.dhist; means.append(part.mean)
    assert torch.equal(qh, res.qhist) and torch.equal(dh, res.dhist) and torch.equal(torch.cat(means), res.mean)
    # (v) sampled waters of two frames against the oracle
    oracle.set_threads(8)
    rng = np.random.default_rng(3)
    sample = rng.choice(n, 2000, replace=False)
    for t in (0, T - 1):
        pos = frames[t].cpu().numpy()
        np.testing.assert_allclose(q[t].cpu().numpy()[sample], oracle.getOrderParamq(pos[sample], pos, box, cells=True), rtol=0, atol=1e-12)


def test_config5_slab_fp32(oracle):
    import torch
    from waterorderlib_amd import _lib, engine, synth
    n, F = 500000, 16
    xyz, boxes = synth.slab_frames(n, 1, seed=5, n_vapour=100)
    nw = xyz.shape[1]
    base = torch.from_numpy(xyz[0]).cuda()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    frames = (base.unsqueeze(0) + 0.05 * torch.randn((F, nw, 3), generator=g, device="cuda", dtype=torch.float64)).to(torch.float32).contiguous()
    bx = np.tile(boxes[0][None, :], (F, 1))
    ctx = _lib.Context(0)
    res = engine.tet_order_traj(frames, bx, want_q=True, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.counter("slow_waters") < 0.01 * F * nw            # the filter certifies the bulk; vacuum does not break the tiles
    assert res.q.dtype == torch.float32
    t = 3
    pos = frames[t].cpu().numpy().astype(np.float64)
    oracle.set_threads(8)
    sample = np.concatenate([np.random.default_rng(0).choice(n, 2000, replace=False), np.arange(n, nw)])   # bulk + all vapour waters
    q_o = oracle.getOrderParamq(pos[sample], pos, boxes[0], cells=True)
    np.testing.assert_allclose(res.q[t].cpu().numpy().astype(np.float64)[sample], q_o, rtol=0, atol=5e-7)   # float32 output
    # stray vapour waters have fewer than four neighbours inside 10 A: distances binned < 4 per water there
    assert int(res.dhist.sum().item()) < 4 * F * nw and int(res.dhist.sum().item()) > 4 * F * n
    ctx.close()
