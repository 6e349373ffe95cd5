#!/usr/bin/env python3
"""bench.py -- 1080p frame-pairs/s of the HIP pair-motion path (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of B consecutive frame pairs of a synthetic
1920x1080 sine-translate stream whose gray frames are already resident in HBM: Farneback flow for
the B pairs (pyramid + PolyExp once per frame, UpdateMatrices, 3 x blur+solve per level, 4 levels),
pass 1 (|div| argmax + mean magnitude), the +-6 centre smoothing on the host and pass 2 (radial
weighted mean).  Batches are software-pipelined (batch s+1 is queued before batch s is finalised)
so the device never waits for the host.

N > 1: one process per GPU (torchrun), every rank streams its own clip (weak scaling, pairs are
independent); the only exchange is the barrier/MAX around the timed region and a host (gloo) gather
of the per-pair scalars -- no RCCL collective in the data path.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DOMINANT = "k_blur_solve"  # per profiles/*_kernel_stats.csv; --profile-all re-derives it live


FUSE_FIRST = 10000  # the library's default threshold; --fuse-first overrides (see ffl_set_option)


def alg_bytes_per_batch(N, B, U, level_sizes):
    """Algorithmic HBM bytes per kernel class for one batch (SURVEY 8d stage graph: every stage reads its inputs
    once and writes its outputs once, f32 planes).  level_sizes = [(lw, lh)] for k = 0 .. levels.

    SURVEY 8(d): "no cross-stage fusion ... a fused implementation that moves fewer real bytes still reports
    against B_alg (state real rocprof bytes alongside)".  Per level: flow init / x2 upsample 10 n_k (zero fill
    8 n_k at the coarsest level), UpdateMatrices_0 68 n_k (R0 20 + R1 20 + flow 8 -> M 20), three iterations
    3 x (M 20 -> flow 8) + 2 x 68 = 220 n_k.  On the levels where the library folds the flow-init and
    UpdateMatrices_0 stages into the first blur+solve launch (ffl_set_option "fuse_first": levels with at least
    that many 64x16 tiles over the batch; their M and flow never reach memory) those stages' bytes are counted on
    the k_blur_solve class; `roofline.traffic` is the measured traffic."""
    s = float(sum(lw * lh for lw, lh in level_sizes))
    um = ks = 0.0
    for k, (lw, lh) in enumerate(level_sizes):
        n_k = float(lw * lh)
        init = (8.0 if k == len(level_sizes) - 1 else 10.0) * n_k + 68.0 * n_k
        tiles = ((lw + 63) // 64) * ((lh + 15) // 16) * B
        if FUSE_FIRST > 0 and tiles >= FUSE_FIRST:
            ks += init
        else:
            um += init
        ks += 220.0 * n_k
    return {
        "k_pyr_level": U * (4 * N + 4 * s),          # u8 full-res read per level + f32 level image write
        "k_polyexp": U * 24 * s,                      # I 4 -> R 20
        "k_update_matrices": B * um,
        "k_blur_solve": B * ks,
        "k_pass1": B * 8 * N,
        "k_radial": B * 8 * N,
    }


def cpu_baseline(frames, threads, pairs):
    """Oracle (C restatement, 'port') timed on the host cores: one pair per thread, like the
    reference's Pool(threads).starmap over pairs (FunscriptFlow.pyw:1190-1191)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor
    import oracle as orc
    orc.lib()
    h, w = frames[0].shape

    def one(j):
        flow, x, y, v, mm = orc.pair_c(frames[j % (len(frames) - 1)], frames[j % (len(frames) - 1) + 1])
        return orc.radial_c(flow, (w / 2.0, h / 2.0), mm > 7, False)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, range(pairs)))
    dt = time.perf_counter() - t0
    return pairs / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=32,
                    help="pairs per ffl_flow_pairs call (a step); 32 fills the device at every pyramid level: "
                         "1080p 4450 / 4700 / 5030 / 4880 pairs/s at B = 8 / 16 / 32 / 64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU sample (default: one per thread)")
    ap.add_argument("--independent", action="store_true", help="2B frames per batch (no frame sharing)")
    ap.add_argument("--zoom", type=float, default=0.0,
                    help="breathing zoom of the synthetic clip (0 = BASELINE's pure sine-translate); a zooming clip has a "
                         "spatially varying flow, which the data-dependent gather fast path sees less often")
    ap.add_argument("--fuse-first", type=int, default=-1,
                    help="minimum tiles x pairs of a level for folding its initial UpdateMatrices into the first blur+solve "
                         "launch (library default 10000; 0: never, 1: always)")
    ap.add_argument("--blur-rows", type=int, default=0, help="tiles a k_blur_solve workgroup walks down (0 = automatic)")
    ap.add_argument("--trace-steps", action="store_true", help="print per-step host wall times to stderr")
    ap.add_argument("--blur-tile-h", type=int, default=0, help="k_blur_solve LDS tile rows (fixed at 16)")
    ap.add_argument("--lanes", type=int, default=0, help="compute lanes (co-scheduled batches) per context, default 1")
    ap.add_argument("--expand", type=int, default=0, choices=[0, 1, 2],
                    help="schedule of the frame-only kernels (pyramid + PolyExp): 0 (default) serial on the lane's "
                         "stream; 2 the 4 levels fork onto side streams and join before the flow chain starts; "
                         "1 run-ahead, the chain waits per level (coarse-level flow launches get co-scheduled and "
                         "stretched: +3 %% pairs/s)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the process group is gloo")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: no per-kernel HIP events (roofline omitted)")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around every kernel class (adds ~0.2 ms/step); default: the dominant kernel only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch  # first: its HIP runtime (same SONAME) is the one libffl_hip.so binds to
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    if args.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
            host_group = dist.group.WORLD
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            host_group = dist.new_group(backend="gloo")

    from funscript_flow_amd import _capi
    from funscript_flow_amd.pipeline import SMOOTH_RADIUS
    from funscript_flow_amd.synth import sine_translate_frames

    global FUSE_FIRST
    if args.fuse_first >= 0:
        _capi.set_option("fuse_first", args.fuse_first)
        FUSE_FIRST = args.fuse_first
    if args.blur_rows:
        _capi.set_option("blur_rows", args.blur_rows)
    if args.blur_tile_h:
        _capi.set_option("blur_tile_h", args.blur_tile_h)
    # One compute lane by default: kernels of consecutive batches then run back to back, so the
    # per-launch HIP-event durations behind `roofline` are those of the kernel alone (and agree with
    # rocprofv3's).  `--lanes 2` (the library's default for production use) co-schedules two batches:
    # ~+11 % pairs/s at 1080p, but every launch is stretched by its co-runner (profiles/README.md).
    _capi.set_option("lanes", args.lanes or 1)
    _capi.set_option("run_ahead", args.expand)
    W, H, B = args.width, args.height, args.batch
    N = W * H
    U = 2 * B if args.independent else B + 1
    frames = sine_translate_frames(U, W, H, seed=1 if world == 1 else 10 + rank, zoom=args.zoom)
    ctx = _capi.Context(W, H, device=local_rank, frame_slots=U + 1, flow_slots=3 * B, max_batch=B)
    level_sizes = [ctx.level_size(k) for k in range(ctx.num_levels() + 1)]
    for i in range(U):
        ctx.upload_frame(i, frames[i])
    ctx.sync()
    if args.independent:
        f0, f1 = [2 * i for i in range(B)], [2 * i + 1 for i in range(B)]
    else:
        f0, f1 = list(range(B)), list(range(1, B + 1))

    results = []

    DEPTH = 2  # batches queued ahead of the one being finalised (flow slots: (DEPTH + 1) * B)
    jj = np.arange(B)
    lo, hi = np.maximum(0, jj - SMOOTH_RADIUS), np.minimum(B, jj + SMOOTH_RADIUS + 1)

    def enqueue(step):
        slots = [(step % (DEPTH + 1)) * B + i for i in range(B)]
        ctx.flow_pairs(f0, f1, slots)
        return slots

    def finalize(slots):
        recs = ctx.pass1_results(slots, 7.0)                       # one call per batch
        psum = np.zeros((B + 1, 2), np.int64)                      # FF:1203-1214 inside the batch: exact integer
        psum[1:] = np.cumsum(np.array([(r[0], r[1]) for r in recs], np.int64), axis=0)   # window sums / counts
        cs = (psum[hi] - psum[lo]) / (hi - lo)[:, None]
        dots = ctx.radial(slots, cs, [r[4] for r in recs], False)
        results.append((recs, dots))

    TRACE, TRACE2 = [], []

    def run(steps):
        pending = []
        for s in range(steps):
            if args.trace_steps:
                TRACE.append(time.perf_counter())
            pending.append(enqueue(s))
            if args.trace_steps:
                TRACE2.append(time.perf_counter())
            if len(pending) > DEPTH:
                finalize(pending.pop(0))
        while pending:
            finalize(pending.pop(0))
        ctx.sync()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    results.clear()
    ctx.profile_enable(False if args.no_events else (True if args.profile_all else [DOMINANT]))
    # A generation-2 collection walks every object torch's import created (40-50 ms: longer than the whole
    # timed region at small frame sizes); park those objects in the permanent generation first.
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if TRACE:
        d = np.diff(np.array(TRACE[-args.steps:])) * 1e3
        print("step ms:", " ".join(f"{v:.2f}" for v in d), file=sys.stderr)
        e = (np.array(TRACE2[-args.steps:]) - np.array(TRACE[-args.steps:])) * 1e3
        print("enqueue ms:", " ".join(f"{v:.2f}" for v in e), file=sys.stderr)
    prof = ctx.profile_read()
    ctx.profile_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # host gather of the per-pair scalars (x, y, cut, dot): the path's only exchange, ~40 B/pair
        mine = np.array([[r[0], r[1], int(r[4]), d] for recs, dots in results for r, d in zip(recs, dots)], np.float64)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0, group=host_group)
        if rank == 0:
            assert sum(len(g) for g in gathered) == world * args.steps * B

    if rank == 0:
        pairs = world * args.steps * B
        alg = alg_bytes_per_batch(N, B, U, level_sizes)
        dom = max((k for k in alg), key=lambda k: prof[k][1])
        n_launch, ms = prof[dom]
        achieved = alg[dom] * args.steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"{W}x{H}" and tj.get("batch") == B and tj.get("kernel") == dom:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "1080p frame-pairs/sec" if (W, H) == (1920, 1080) else f"{W}x{H} frame-pairs/sec",
            "value": pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic sine-translate frame-pair stream, gray frames resident in HBM",
                       "pairs_per_step": B, "frames_per_step": U, "levels": len(level_sizes), "winsize": 15, "iterations": 3,
                       "poly_n": 5, "parallelism": f"pair-shard x{world}", "compute_lanes": args.lanes or 1},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "launches": n_launch, "avg_launch_ms": ms / max(n_launch, 1),
                         "alg_bytes_per_launch": alg[dom] * args.steps / max(n_launch, 1)},
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items() if v[0]},
            "whole_path": {"alg_bytes_per_pair": sum(alg.values()) / B,
                           "achieved_GBps": sum(alg.values()) / B * (pairs / world) / dt / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            npairs = args.cpu_pairs or 2 * threads   # ~20 s of CPU work at 1080p (0.6 s per pair and core)
            v, cdt = cpu_baseline(frames, threads, npairs)
            out["cpu_baseline"] = {"value": v, "unit": "pairs/s", "cores": threads, "kind": "port",
                                   "sample": f"{npairs} pairs of the same {W}x{H} stream, one pair per thread, "
                                             f"C oracle (Farneback + argmax + mean + radial), {cdt:.1f} s wall"}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
