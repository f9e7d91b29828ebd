"""The N > 1 path on CPU: two gloo ranks, each holding the results of its own frame shard, go through
the path's single exchange step (engine.reduce_traj_result: all-reduce of the int64 histograms,
all-gather of the per-frame mean/var) and must reproduce the unsharded result bit for bit.  The
per-shard results come from the CPU oracle here (test infrastructure); on the GPU they come from
libwol -- the reduction code is the same."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, T, n, ret):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wol_oracle
        from waterorderlib_amd import engine, synth
        xyz, boxes = synth.water_frames(n, T, seed=5)
        a, b = engine.shard_bounds(T, rank, world)
        mean, var, qh, dh = wol_oracle.tet_order_traj(xyz[a:b], boxes[a:b])
        local = engine.TrajResult(mean, var, qh, dh, None, a)
        tot = engine.reduce_traj_result(local, T)
        ret[rank] = (np.asarray(tot.mean), np.asarray(tot.var), np.asarray(tot.qhist), np.asarray(tot.dhist))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [7, 10])
def test_two_rank_reduction_matches_single(oracle, T):
    from waterorderlib_amd import synth
    n, world = 300, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + T
    mp.spawn(_worker, args=(world, port, T, n, ret), nprocs=world, join=True)
    xyz, boxes = synth.water_frames(n, T, seed=5)
    mean, var, qh, dh = oracle.tet_order_traj(xyz, boxes)
    for r in range(world):
        m, v, q, d = ret[r]
        assert np.array_equal(m, mean) and np.array_equal(v, var)
        assert np.array_equal(q, qh) and np.array_equal(d, dh)
        assert q.dtype == np.int64 and d.sum() == 4 * n * T


def _gpu_worker(rank, world, port, T, n, ret):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from waterorderlib_amd import engine, synth
        xyz, boxes = synth.water_frames(n, T, seed=9)
        a, b = engine.shard_bounds(T, rank, world)
        tot = engine.tet_order_traj_sharded(xyz[a:b], boxes[a:b], T_total=T, frame_offset=a)
        out = [np.asarray(tot.mean), np.asarray(tot.var), np.asarray(tot.qhist), np.asarray(tot.dhist)]
        if rank == 0:
            one = engine.tet_order_traj(xyz, boxes)
            out += [np.asarray(one.mean), np.asarray(one.var), np.asarray(one.qhist), np.asarray(one.dhist)]
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_match_the_unsharded_run():
    """the frame-sharded product path end to end: two processes (sharing the one GPU of the test box), libwol on
    each shard, the exchange step over gloo; every rank ends with the unsharded result, bit for bit"""
    n, T, world = 4000, 130, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gpu_worker, args=(world, port, T, n, ret), nprocs=world, join=True)
    ref = ret[0][4:]
    for r in range(world):
        for got, want in zip(ret[r][:4], ref):
            assert np.array_equal(got, want)
    assert ref[3].sum() == 4 * n * T
