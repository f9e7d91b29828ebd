#!/usr/bin/env python3
"""bench.py -- headline benchmark: frames/s (and waters/s) of the tetrahedral-q hot path on a
synthetic 100k-water periodic box (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (neighbour search + 4-NN select + angles + q + q/distance
histograms + per-frame mean/var) over one batch of --frames-per-step frames that are already
resident in HBM.  Every step uses its own frames.  With N > 1 (launched by torch.distributed.run,
one rank per GPU) every rank processes its own frames (weak scaling, frame sharding: no data-path
collective) and the job ends with the path's single exchange step, an RCCL all-reduce of the int64
histograms + all-gather of the per-frame (mean, var) series, inside the timed region.

Rank 0 prints ONE JSON line; `value` is whole-job frames/s.  `roofline` prices the dominant
kernel against HBM using ALGORITHMIC bytes (32 B per water: 24 B of O coordinates in, 8 B of q out;
DESIGN.md) over its HIP-event-measured launch time; `cpu_baseline` is the CPU oracle (a port of
the reference's algorithm) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_WATER = {"f64": 32.0, "f32": 16.0}
HBM_PEAK_GBPS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_frames(torch, base_dev, box, n_frames, seed, sigma, dtype):
    """jittered-lattice frames generated on the GPU: base[N,3] + N(0, sigma), wrapped into the box"""
    g = torch.Generator(device=base_dev.device)
    g.manual_seed(seed)
    jit = torch.randn((n_frames,) + tuple(base_dev.shape), generator=g, device=base_dev.device, dtype=torch.float64)
    x = base_dev.unsqueeze(0) + sigma * jit
    b = torch.as_tensor(box, device=base_dev.device, dtype=torch.float64)
    x = x - b * torch.floor(x / b)
    x = torch.where(x >= b, torch.zeros_like(x), x)
    return x.to(dtype).contiguous()


def cpu_baseline(pos_host, box, sample_waters, threads):
    """oracle (kind "port") on the host: the literal O(Ns*N) form -- the reference's own pair work
    (Fortran mask + numpy selection) without its Python overhead -- on `sample_waters` waters of one
    100k-water frame, 1 thread; plus the oracle's OpenMP cell-list form on the whole frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wol_oracle
    wol_oracle.lib()
    threads = wol_oracle.set_threads(min(threads, 16))  # the GPU box's CPU share for one GPU
    n = len(pos_host)
    sub = np.ascontiguousarray(pos_host[:sample_waters])
    t0 = time.perf_counter()
    wol_oracle.getOrderParamq(sub, pos_host, box)
    t_lit = time.perf_counter() - t0
    frames_per_s = (sample_waters / n) / t_lit
    t0 = time.perf_counter()
    wol_oracle.getOrderParamq(pos_host, pos_host, box, cells=True)
    t_cell = time.perf_counter() - t0
    return {
        "value": frames_per_s, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": "oracle literal O(Ns*N) getOrderParamq on %d of %d waters of one frame (%.1f s), scaled to a frame"
                  % (sample_waters, n, t_lit),
        "waters_per_s": sample_waters / t_lit,
        "cell_list": {"value": 1.0 / t_cell, "unit": "frames/s", "cores": threads,
                      "sample": "oracle cell-list form, one whole frame (%.2f s), OpenMP" % t_cell},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--waters", type=int, default=100000)
    ap.add_argument("--frames-per-step", type=int, default=256)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--cell-width", type=float, default=0.0)
    ap.add_argument("--path", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-waters", type=int, default=10000)
    ap.add_argument("--search-rows", type=int, default=-1, help="proposal kernel: 1 row-merged tiles, 0 3-D tiles (default: library default)")
    ap.add_argument("--overlap", type=int, default=-1, help="1: sort of the next batch on a second stream beside search/finish")
    ap.add_argument("--frames-per-batch", type=int, default=-1, help="library batch size (0 = automatic)")
    ap.add_argument("--sort-parts", type=int, default=-1, help="blocks per frame of the sort's first level (0 = automatic)")
    ap.add_argument("--search-margin", type=int, default=-1, help="blocks per CU the proposal kernel leaves free")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run "
                         "(default: profiles/r01_hbm_traffic_pmc.json when the workload matches it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from waterorderlib_amd import _lib, engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run the exchange step goes through RCCL even with one rank (lets the
    # collective path be rehearsed on a one-GPU box); a plain `python bench.py` has no process group
    distributed = world > 1 or ("LOCAL_RANK" in os.environ and "MASTER_PORT" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, F = args.waters, args.frames_per_step
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    base, box = synth.lattice_sites(n, seed=2026)
    # waters in random order, as in a real trajectory (topology order is unrelated to position);
    # lattice order would make the cell sort's scatter and the neighbour gathers artificially coherent
    base = base[np.random.default_rng(2026).permutation(n)]
    base_dev = torch.from_numpy(base).to(dev)
    boxes = np.tile(box[None, :], (F, 1))

    ctx = _lib.Context(local_rank)
    ctx.set_option("time_kernels", 1)
    if args.cell_width > 0:
        ctx.set_option("cell_width", args.cell_width)
    if args.path:
        ctx.set_option("path", args.path)
    if args.search_rows >= 0:
        ctx.set_option("search_rows", args.search_rows)
    if args.overlap >= 0:
        ctx.set_option("overlap", args.overlap)
        ctx.set_option("inputs_resident", 1)
    if args.frames_per_batch >= 0:
        ctx.set_option("frames_per_batch", args.frames_per_batch)
    if args.sort_parts >= 0:
        ctx.set_option("sort_parts", args.sort_parts)
    if args.search_margin >= 0:
        ctx.set_option("search_margin", args.search_margin)

    total_steps = args.warmup + args.steps
    # every step owns its frames; generated (untimed) up front so the timed region only reads HBM
    batches = [make_frames(torch, base_dev, box, F, 1000003 * rank + s, 0.3, tdtype) for s in range(total_steps)]
    torch.cuda.synchronize()

    def step(s):
        return engine.tet_order_traj(batches[s], boxes, want_q=False, want_dhist=True, ctx=ctx)

    def run_job(first, count):
        """`count` steps starting at batch `first`, then the job-level reduction: sum of the per-step
        integer histograms and the path's one exchange step (RCCL all-reduce / all-gather if N > 1)."""
        results = [step(first + s) for s in range(count)]
        qh = torch.stack([r.qhist for r in results]).sum(0)
        dh = torch.stack([r.dhist for r in results]).sum(0)
        mean = torch.cat([r.mean for r in results])
        var = torch.cat([r.var for r in results])
        local = engine.TrajResult(mean, var, qh, dh)
        return engine.reduce_traj_result(local, world * count * F) if distributed else local

    if args.warmup > 0:
        run_job(0, args.warmup)  # untimed: also loads every kernel the timed region uses
    torch.cuda.synchronize()
    ctx.kernel_time(reset=True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = run_job(args.warmup, args.steps)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_launches = ctx.kernel_time(reset=False)
    slow_last = ctx.counter("slow_waters")
    dens_level = ctx.counter("density_level")
    rows_block = ctx.counter("rows_block")

    if rank == 0:
        traffic = args.traffic_bytes
        if traffic is None:
            # counters cannot be collected inside this run (rocprofv3 --pmc is its own pass); use the
            # committed PMC result when this run is the workload it was measured on
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")))
                cfg = tj["config"]
                if cfg["waters"] == n and cfg["frames_per_step"] == F and cfg["dtype"] == args.dtype:
                    traffic = tj["k_search_rows_hbm_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                traffic = None
        frames_total = world * args.steps * F
        waters_in_hist = int(total.dhist.sum().item()) // 4 if total.dhist is not None else 0
        value = frames_total / elapsed
        k_avg_s = (k_ms / max(k_launches, 1)) * 1e-3
        waters_per_launch = (args.steps * F * n) / max(k_launches, 1)
        alg_bytes = ALG_BYTES_PER_WATER[args.dtype] * waters_per_launch
        achieved = alg_bytes / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        out = {
            "metric": "frames/sec for tetrahedral q (+ 4-NN O-O distance histogram), %dk-water box" % (n // 1000),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "waters_per_s": value * n,
            "config": {"workload": "%d-water cubic periodic box (jittered diamond lattice, rho=0.033456/A^3), "
                                   "q + 4-NN O-O distance histogram + per-frame mean/var, %d frames per step per GPU, "
                                   "frames resident in HBM" % (n, F),
                       "waters": n, "frames_per_step": F, "frames_total": frames_total,
                       "sharding": "frames x%d" % world, "highCut": 10.0},
            "check": {"mean_q": float(total.mean.mean().item()), "waters_binned": waters_in_hist},
            "roofline": {"bound": "hbm", "kernel": "k_search_rows" if args.search_rows != 0 else "k_search_fast", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic,
                         "launches": int(k_launches), "avg_launch_ms": k_avg_s * 1e3,
                         "exact_path_fraction_last_launch": slow_last / float(F * n),
                         "tile_density_level": dens_level, "proposal_block_threads": rows_block,
                         "alg_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 of the one-GPU run only
            pos_host = batches[args.warmup][0].to(torch.float64).cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(pos_host, box, min(args.cpu_sample_waters, n), len(os.sched_getaffinity(0)))
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
