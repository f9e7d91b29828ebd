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
kernel against HBM using ALGORITHMIC bytes (24 B of O coordinates per water in; + 8 B of q out when
per-water q is written, --want-q) over its HIP-event-measured launch time, and carries the
kernel's real limiter as `secondary` (VALU issue: instructions per water from the committed rocprofv3
counter pass over this same command, priced at 4 cycles per wave64 instruction on 1024 SIMDs);
`end_to_end` is the same workload fed from HOST memory (TIP4P-interleaved frames, double-buffered
over PCIe) -- never `value`; `cpu_baseline` is the CPU oracle (a port of the reference's algorithm)
timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_IN = {"f64": 24.0, "f32": 12.0}     # O coordinates read per water ...
ALG_BYTES_Q = {"f64": 8.0, "f32": 4.0}        # ... and q written per water (only when the caller asks for per-water q)
HBM_PEAK_GBPS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 1024                                 # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9                         # MI355X_MICROARCH.md: max clock
VALU_ISSUE_CYCLES = 4                         # cycles one wave64 VALU instruction holds its SIMD (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU, profiles/)
COUNTERS_FILE = os.path.join("profiles", "r03_sq_counters.json")      # config 3; config 5: r03_config5_sq_counters.json


def make_frames(torch, base_dev, box, n_frames, seed, sigma, dtype, vapour=None):
    """jittered-lattice frames generated on the GPU: base[N,3] + N(0, sigma), wrapped into the box.  vapour (slab
    workload): isolated waters appended to every frame; the slab's z is left unwrapped (it does not reach the box)"""
    g = torch.Generator(device=base_dev.device)
    g.manual_seed(seed)
    jit = torch.randn((n_frames,) + tuple(base_dev.shape), generator=g, device=base_dev.device, dtype=torch.float64)
    x = base_dev.unsqueeze(0) + sigma * jit
    b = torch.as_tensor(box, device=base_dev.device, dtype=torch.float64)
    if vapour is None:
        x = x - b * torch.floor(x / b)
        x = torch.where(x >= b, torch.zeros_like(x), x)
    else:
        x[..., :2] = x[..., :2] - b[:2] * torch.floor(x[..., :2] / b[:2])
        x = torch.cat([x, vapour.unsqueeze(0).expand(n_frames, -1, -1)], dim=1)
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
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {
        "value": frames_per_s, "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
        "sample": "oracle literal O(Ns*N) getOrderParamq on %d of %d waters of one frame (%.1f s), scaled to a frame"
                  % (sample_waters, n, t_lit),
        "waters_per_s": sample_waters / t_lit,
        "cell_list": {"value": 1.0 / t_cell, "unit": "frames/s", "cores": threads,
                      "sample": "oracle cell-list form, one whole frame (%.2f s), OpenMP" % t_cell},
    }


def end_to_end(torch, engine, ctx, batch_dev, box, args):
    """The same workload fed from HOST memory, as the reference's caller holds it (orderParam_lib.py:1458-1462):
    TIP4P-interleaved fp64 frames [O, H, H, EP] in pinned memory and in a pageable numpy array, through wol_q_traj's
    WOL_MEM_HOST path.  The library packs the O rows with host threads into pinned staging (option host_pack: what the
    reference's caller does with pos[watInds]) -- as float32 when every coordinate is exactly one, as in frames read from
    single-precision trajectory files -- and double-buffers the copies under the kernels; `whole_frames` is the old form
    (the interleaved frames cross the link as they lie).  Link bound = linear-copy rate of the link / bytes that cross it
    per frame."""
    F0, n, _ = batch_dev.shape
    T = max(F0, args.e2e_frames)
    lw = 4
    o = batch_dev.cpu()
    host = torch.zeros((T, n * lw, 3), dtype=o.dtype).pin_memory()
    for t in range(T):
        host[t, 0::lw] = o[t % F0]
    boxes = np.tile(box[None, :], (T, 1))
    threads = min(16, len(os.sched_getaffinity(0)))
    ctx.set_option("host_threads", threads)
    out = {"frames": T, "atoms_per_water": lw, "frame_bytes": int(host[0].numel() * host.element_size()), "pack_threads": threads}
    # what the link does on a plain linear copy of pinned bytes (the bound of any host-fed path), median of 5
    scratch = torch.empty_like(host[:64], device=batch_dev.device)
    rates = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        scratch.copy_(host[:64], non_blocking=True); torch.cuda.synchronize()
        rates.append(host[:64].numel() * host.element_size() / (time.perf_counter() - t0) / 1e9)
    link = float(np.median(rates[1:]))
    del scratch
    out["link_GBps_linear_copy"] = link

    def leg(arr, len_wat, pack):
        ctx.set_option("host_pack", pack)
        engine.tet_order_traj(arr[:64], boxes[:64], len_wat=len_wat, ctx=ctx)          # warm: staging buffers, page faults
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            res = engine.tet_order_traj(arr, boxes, len_wat=len_wat, ctx=ctx)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, ctx.counter("h2d_bytes"), float(np.mean(res.mean)))
        dt, moved, mq = best
        per_frame = moved / T
        return {"frames_per_s": T / dt, "GBps_pcie": moved / dt / 1e9, "ms": 1e3 * dt, "link_bytes_per_frame": per_frame,
                "link_bound_frames_per_s": link * 1e9 / per_frame, "frac_of_link_bound": (T / dt) / (link * 1e9 / per_frame), "mean_q": mq}

    out["pinned"] = leg(host.numpy(), lw, 2)
    out["pageable"] = leg(host.numpy().copy(), lw, 2)
    out["whole_frames_pinned"] = leg(host.numpy(), lw, 0)
    # a trajectory read from a single-precision file: every coordinate is exactly a float32 (same waters, rounded once)
    sp = host.to(torch.float32).to(torch.float64).pin_memory()
    out["single_precision_values_pinned"] = leg(sp.numpy(), lw, 2)
    del sp
    # O-only host frames (what a caller that slices the oxygens out first would hand over)
    oo = o[:F0].repeat((T + F0 - 1) // F0, 1, 1)[:T].contiguous().pin_memory().numpy()
    out["o_only_pinned"] = leg(oo, 1, 2)
    ctx.set_option("host_pack", 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=[3, 5], default=3,
                    help="BASELINE config: 3 = 100k-water cubic box, fp64 (the headline metric; default); 5 = 500k-water slab with "
                         "vacuum and vapour waters, fp32 storage (16 B per water), 32 frames per step")
    ap.add_argument("--waters", type=int, default=0, help="0 = the config's own (100 000 / 500 000)")
    ap.add_argument("--frames-per-step", type=int, default=0, help="0 = the config's own (256 / 32)")
    ap.add_argument("--dtype", choices=["f64", "f32", "auto"], default="auto")
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
                         "(default: the committed counter pass, %s, when the workload matches it)" % COUNTERS_FILE)
    ap.add_argument("--dyn-tiles", type=int, default=-2, help="proposal kernel tile order: 1 per-XCD counters, 0 static (default: library default)")
    ap.add_argument("--stages", type=int, default=-1, help="pipe depth of the overlapped trajectory path (2 / 3)")
    ap.add_argument("--want-q", action="store_true", help="also write per-water q (8 B per water more per step)")
    ap.add_argument("--rows-block", type=int, default=-1, help="force the proposal kernel's instantiation (512 / 768)")
    ap.add_argument("--rec32", type=int, default=-1, help="float32 frames: 1 = 16-byte sorted records (library default), 0 = 32-byte")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-memory (PCIe-inclusive) leg")
    ap.add_argument("--e2e-frames", type=int, default=192, help="host frames of the end-to-end leg")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON: everything libraries print there (RCCL announces its version on
    # stdout when a communicator comes up) goes to stderr instead, at the file-descriptor level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
        import datetime
        # (a rank that dies leaves its peers in a collective: they give up after five minutes instead of the default thirty)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch N ranks with torch.distributed.run for --gpus N)"
                         % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    slab = args.config == 5
    n_bulk = args.waters or (500000 if slab else 100000)
    F = args.frames_per_step or (32 if slab else 256)
    if args.dtype == "auto":
        args.dtype = "f32" if slab else "f64"
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    n_vapour = 100 if slab else 0
    if slab:
        # BASELINE config 5: a cubic water block in a box elongated along z (vacuum gap 0.5 L >> the 10 A cutoff)
        # plus 100 isolated waters in the gap; periodic in x and y only as far as the water goes
        L = synth.box_length(n_bulk)
        base, _ = synth.lattice_sites(n_bulk, seed=2026, box=np.array([L, L, L]))
        box = np.array([L, L, 1.5 * L])
        vap = np.random.default_rng([2026, 0x51AB]).random((n_vapour, 3)) * np.array([L, L, 0.3 * L]) + np.array([0, 0, 1.1 * L])
    else:
        base, box = synth.lattice_sites(n_bulk, seed=2026)
    n = n_bulk + n_vapour
    # waters in random order, as in a real trajectory (topology order is unrelated to position);
    # lattice order would make the cell sort's scatter and the neighbour gathers artificially coherent
    base = base[np.random.default_rng(2026).permutation(n_bulk)]
    base_dev = torch.from_numpy(base).to(dev)
    vap_dev = torch.from_numpy(vap).to(dev) if slab else None
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
    if args.stages >= 0:
        ctx.set_option("stages", args.stages)
    if args.dyn_tiles >= -1:
        ctx.set_option("dyn_tiles", args.dyn_tiles)
    if args.rows_block >= 0:
        ctx.set_option("rows_block", args.rows_block)
    if args.rec32 >= 0:
        ctx.set_option("rec32", args.rec32)

    total_steps = args.warmup + args.steps
    # every step owns its frames; generated (untimed) up front so the timed region only reads HBM
    batches = [make_frames(torch, base_dev, box, F, 1000003 * rank + s, 0.3, tdtype, vap_dev) for s in range(total_steps)]
    torch.cuda.synchronize()

    def step(s):
        return engine.tet_order_traj(batches[s], boxes, want_q=args.want_q, want_dhist=True, ctx=ctx)

    exchange_ms = [0.0]

    def run_job(first, count):
        """`count` steps starting at batch `first`, then the job-level reduction: sum of the per-step
        integer histograms and the path's one exchange step (RCCL all-reduce / all-gather if N > 1)."""
        results = [step(first + s) for s in range(count)]
        qh = torch.stack([r.qhist for r in results]).sum(0)
        dh = torch.stack([r.dhist for r in results]).sum(0)
        mean = torch.cat([r.mean for r in results])
        var = torch.cat([r.var for r in results])
        local = engine.TrajResult(mean, var, qh, dh)
        if not distributed:
            return local
        torch.cuda.synchronize()      # so the exchange step can be timed on its own (diagnostic; inside the timed region)
        local_done = time.perf_counter()
        out = engine.reduce_traj_result(local, world * count * F)
        torch.cuda.synchronize()
        exchange_ms[0] = 1e3 * (time.perf_counter() - local_done)
        run_job.local_s = local_done
        return out

    def rank_failed(stage, exc):
        """a rank that fails says so itself (its peers are left in a collective until the process group's timeout): one JSON
        line on stderr with what it had measured, so a failed scaling run still shows which rank broke, where, and how far
        it got"""
        done = getattr(run_job, "local_s", None)
        sys.stderr.write(json.dumps({"bench_rank_failed": {"rank": rank, "world": world, "stage": stage, "error": repr(exc)[:400],
                                                            "frames_per_s_before_exchange": (args.steps * F / (done - t0)) if (done and stage == "timed") else None,
                                                            "exchange_step_ms": exchange_ms[0]}}) + "\n")
        sys.stderr.flush()

    t0 = time.perf_counter()
    try:
        if args.warmup > 0:
            run_job(0, args.warmup)  # untimed: also loads every kernel the timed region uses
        torch.cuda.synchronize()
    except Exception as e:
        rank_failed("warmup", e)
        raise
    ctx.kernel_time(reset=True)
    run_job.local_s = None
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        total = run_job(args.warmup, args.steps)
        torch.cuda.synchronize()
    except Exception as e:
        rank_failed("timed", e)
        raise
    t_local = (getattr(run_job, "local_s", None) or time.perf_counter()) - t0   # this rank's own frames, before the exchange
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([args.steps * F / t_local, exchange_ms[0]], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"frames_per_s_before_exchange": [float(a[0].item()) for a in allr],
                    "exchange_step_ms": [float(a[1].item()) for a in allr]}
    k_ms, k_launches = ctx.kernel_time(reset=False)
    slow_last = ctx.counter("slow_waters")
    dens_level = ctx.counter("density_level")
    rows_block = ctx.counter("rows_block")
    try:
        grid_cells = [ctx.counter("cells_x"), ctx.counter("cells_y"), ctx.counter("cells_z")]
    except Exception:   # (a library build from before the counter existed: tools/ab.sh)
        grid_cells = None

    if rank == 0:
        kname = "k_search_rows" if args.search_rows != 0 else "k_search_fast"
        # counters cannot be collected inside this run (rocprofv3 --pmc is its own pass, tools/collect_sq_counters.sh);
        # the committed pass over this same command supplies HBM traffic and instruction counts when the workload matches
        traffic, traffic_source, insts = args.traffic_bytes, ("--traffic-bytes" if args.traffic_bytes else None), None
        counters_file = COUNTERS_FILE.replace("r03_", "r03_config5_") if slab else COUNTERS_FILE
        try:
            tj = json.load(open(os.path.join(ROOT, counters_file)))
            cfg = tj.get("_config", {})
            if cfg.get("waters") == n and cfg.get("frames_per_step") == F and cfg.get("dtype") == args.dtype:
                ent = next(v for k, v in tj.items() if k.startswith(kname))
                cnt = ent["counters_per_dispatch"]
                insts = cnt.get("SQ_INSTS_VALU")
                if traffic is None and "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
                    # KiB per dispatch; FETCH_SIZE counts half the bytes of 16-B-per-lane loads on gfx950 (guide, HBM section)
                    traffic = 1024.0 * (2.0 * cnt["FETCH_SIZE"] + cnt["WRITE_SIZE"])
                    traffic_source = counters_file + " (separate rocprofv3 --pmc passes, FETCH_SIZE doubled)"
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        frames_total = world * args.steps * F
        waters_in_hist = int(total.dhist.sum().item()) // 4 if total.dhist is not None else 0
        value = frames_total / elapsed
        k_avg_s = (k_ms / max(k_launches, 1)) * 1e-3
        waters_per_launch = (args.steps * F * n) / max(k_launches, 1)
        bytes_per_water = ALG_BYTES_IN[args.dtype] + (ALG_BYTES_Q[args.dtype] if args.want_q else 0.0)
        alg_bytes = bytes_per_water * waters_per_launch
        achieved = alg_bytes / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        secondary = None
        if insts and k_avg_s > 0:
            secondary = {"bound": "valu", "kernel": kname, "insts_per_water": insts / waters_per_launch,
                         "issue_cycles_per_inst": VALU_ISSUE_CYCLES,
                         "issue_frac": insts * VALU_ISSUE_CYCLES / (N_SIMD * PEAK_CLOCK_HZ * k_avg_s),
                         "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction",
                         "insts_source": counters_file + " (SQ_INSTS_VALU per launch; launch time measured in this run)"}
        out = {
            "metric": ("frames/sec for tetrahedral q (+ 4-NN O-O distance histogram), %dk-water slab with vacuum, fp32 storage" if slab else
                       "frames/sec for tetrahedral q (+ 4-NN O-O distance histogram), %dk-water box") % (n_bulk // 1000),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "waters_per_s": value * n,
            "config": {"workload": (("BASELINE config 5: %d-water slab (cubic water block + %d vapour waters, box elongated 1.5x along z: "
                                     "vacuum gap >> cutoff), float32 storage, " % (n_bulk, n_vapour)) if slab else
                                    ("BASELINE config 3: %d-water cubic periodic box (jittered diamond lattice, rho=0.033456/A^3), " % n)) +
                                   "q + 4-NN O-O distance histogram + per-frame mean/var, %d frames per step per GPU, "
                                   "frames resident in HBM" % F,
                       "baseline_config": args.config,
                       "waters": n, "frames_per_step": F, "frames_total": frames_total,
                       "sharding": "frames x%d" % world, "highCut": 10.0, "per_water_q_written": bool(args.want_q)},
            "check": {"mean_q": float(total.mean.mean().item()), "waters_binned": waters_in_hist},
            # `achieved` / `frac`: algorithmic bytes of ONE launch of the dominant kernel over that launch's time, against the
            # HBM roof -- a kernel-only figure; `step_*`: the same bytes over the whole step (every kernel of the pipeline).
            # The kernel is not HBM-bound: what limits it is vector-instruction issue (`limited_by`, priced in `secondary`).
            "roofline": {"bound": "hbm", "limited_by": "valu_issue (see secondary)", "kernel": kname, "kernel_only": True,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "step_achieved": bytes_per_water * F * n / (elapsed / args.steps) / 1e9,
                         "step_frac": bytes_per_water * F * n / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launches": int(k_launches), "avg_launch_ms": k_avg_s * 1e3,
                         "alg_bytes_per_water": bytes_per_water,
                         "exact_path_fraction_last_launch": slow_last / float(F * n),
                         "tile_density_level": dens_level, "proposal_block_threads": rows_block,
                         "tile_cells": [ctx.counter("tile_sx"), ctx.counter("tile_ry")],
                         "grid_cells": grid_cells,
                         "alg_bytes_per_launch": alg_bytes,
                         "secondary": secondary},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if not args.no_end_to_end and world == 1 and not slab:
            out["end_to_end"] = end_to_end(torch, engine, ctx, batches[args.warmup], box, args)
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 of the one-GPU run only
            pos_host = batches[args.warmup][0].to(torch.float64).cpu().numpy()
            sample = min(args.cpu_sample_waters, n) if not slab else min(args.cpu_sample_waters, 2000)   # ~10-30 s of CPU either way
            out["cpu_baseline"] = cpu_baseline(pos_host, box, sample, len(os.sched_getaffinity(0)))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
