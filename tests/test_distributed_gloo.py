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
