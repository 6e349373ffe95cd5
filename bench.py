#!/usr/bin/env python3
"""bench.py -- 1080p frame-pairs/s of the HIP pair-motion path (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of B consecutive frame pairs of a synthetic
1920x1080 sine-translate stream whose gray frames are already resident in HBM: Farneback flow for
the B pairs (pyramid + PolyExp once per frame, UpdateMatrices, 3 x blur+solve per level, 4 levels),
pass 1 (|div| argmax + mean magnitude), the +-6 centre smoothing on the host and pass 2 (radial
weighted mean).  Batches are software-pipelined (batch s+1 is queued before batch s is finalised)
so the device never waits for the host.

After the timed region the results of the timed steps are compared with golden scalars the CPU oracle
produced in the build container (tests/golden/bench_*.json: argmax index + value bit-exact, reductions to
1e-4, three flow fields by crc32) -> "checked"; then, on rank 0 at N = 1, three short extra passes add
  "kernel_classes"   HIP events around every kernel class (per-class ms, algorithmic bytes, roofline fraction)
  "pcie_inclusive"   host numpy frames (gray and BGR) -> scalars, uploads included (never `value`)
  "small_image"      the reference's own operating point, 256x256 pairs in large batches (FF:1057)
and "cpu_baseline" times the C oracle on the host cores.

N > 1: one process per GPU (torchrun), every rank streams its own clip (weak scaling, pairs are
independent); the only exchange is the barrier/MAX around the timed region and a host (gloo) gather
of the per-pair scalars -- no RCCL collective in the data path.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DOMINANT = "k_blur_solve"  # per profiles/*_kernel_stats.csv; the "kernel_classes" pass re-derives it live
PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)

FUSE_FIRST = 10000  # the library's default threshold; --fuse-first overrides (see ffl_set_option)


def kernel_signature():
    """sha256 over the device sources: profiles/traffic.json records the signature it was measured on, and a
    PMC traffic figure is only reported for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "funscript_flow_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def alg_bytes_per_batch(N, B, U, level_sizes):
    """Algorithmic HBM bytes per kernel class for one batch (SURVEY 8d stage graph: every stage reads its inputs
    once and writes its outputs once, f32 planes).  level_sizes = [(lw, lh)] for k = 0 .. levels.

    SURVEY 8(d): "no cross-stage fusion ... a fused implementation that moves fewer real bytes still reports
    against B_alg (state real rocprof bytes alongside)".  Per level: flow init / x2 upsample 10 n_k (zero fill
    8 n_k at the coarsest level), UpdateMatrices_0 68 n_k (R0 20 + R1 20 + flow 8 -> M 20), three iterations
    3 x (M 20 -> flow 8) + 2 x 68 = 220 n_k.  On the levels where the library folds the flow-init and
    UpdateMatrices_0 stages into the first blur+solve launch (ffl_set_option "fuse_first": levels with at least
    that many 64x16 tiles over the batch; their M and flow never reach memory) those stages' bytes are counted on
    the k_blur_solve class; `roofline.traffic` is the measured traffic and `roofline.frac_measured` its rate."""
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
    """Oracle (C restatement, 'port') timed on the host cores: one pair per worker, like the
    reference's Pool(os.cpu_count()).starmap over pairs (FunscriptFlow.pyw:1190-1191)."""
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


class StepRunner:
    """B-pair steps over U resident gray frames: enqueue / finalize with DEPTH batches queued ahead."""
    DEPTH = 2

    def __init__(self, ctx, B, independent, smooth_radius, depth=None):
        self.ctx, self.B = ctx, B
        if depth is not None:
            self.DEPTH = depth
        if independent:
            self.f0, self.f1 = [2 * i for i in range(B)], [2 * i + 1 for i in range(B)]
        else:
            self.f0, self.f1 = list(range(B)), list(range(1, B + 1))
        jj = np.arange(B)
        self.lo, self.hi = np.maximum(0, jj - smooth_radius), np.minimum(B, jj + smooth_radius + 1)
        self.results, self.trace = [], []

    def enqueue(self, step):
        slots = [(step % (self.DEPTH + 1)) * self.B + i for i in range(self.B)]
        self.ctx.flow_pairs(self.f0, self.f1, slots)
        return slots

    def finalize(self, slots):
        B = self.B
        recs = self.ctx.pass1_results(slots, 7.0)                  # one call per batch
        psum = np.zeros((B + 1, 2), np.int64)                      # FF:1203-1214 inside the batch: exact integer
        psum[1:] = np.cumsum(np.array([(r[0], r[1]) for r in recs], np.int64), axis=0)   # window sums / counts
        cs = (psum[self.hi] - psum[self.lo]) / (self.hi - self.lo)[:, None]
        dots = self.ctx.radial(slots, cs, [r[4] for r in recs], False)
        self.results.append((recs, dots, slots))

    def run(self, steps, trace=False):
        pending = []
        for s in range(steps):
            if trace:
                self.trace.append(time.perf_counter())
            pending.append(self.enqueue(s))
            if len(pending) > self.DEPTH:
                self.finalize(pending.pop(0))
        while pending:
            self.finalize(pending.pop(0))
        self.ctx.sync()


def resident_pass(W, H, B, steps, warmup, device, seed, events, independent=False, zoom=0.0, barrier=None, trace=False,
                  frames=None, depth=None):
    """The timed region of the contract on a fresh context: returns (dt, prof, runner, frames, level_sizes, U, ctx)."""
    from funscript_flow_amd import _capi
    from funscript_flow_amd.pipeline import SMOOTH_RADIUS
    from funscript_flow_amd.synth import sine_translate_frames
    U = 2 * B if independent else B + 1
    if frames is None:
        frames = sine_translate_frames(U, W, H, seed=seed, zoom=zoom)
    ctx = _capi.Context(W, H, device=device, frame_slots=U + 1, flow_slots=3 * B, max_batch=B)
    level_sizes = [ctx.level_size(k) for k in range(ctx.num_levels() + 1)]
    ctx.upload_frames(0, list(frames))
    ctx.sync()
    runner = StepRunner(ctx, B, independent, SMOOTH_RADIUS, depth)
    if warmup > 0:
        runner.run(warmup)
    runner.results.clear()
    ctx.profile_enable(events)
    # A generation-2 collection walks every object torch's import created (40-50 ms: longer than the whole
    # timed region at small frame sizes); park those objects in the permanent generation first.
    gc.collect()
    gc.freeze()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    runner.run(steps, trace)
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    return dt, prof, runner, frames, level_sizes, U, ctx


def verify(runner, frames, W, H, B, seed, ctx):
    """Timed steps vs the oracle goldens: every step must give the same records and scalars (same inputs), the
    last one is compared number by number, its flow slots are still resident for the crc check."""
    from funscript_flow_amd import golden_check
    gold = golden_check.load_golden(W, H, B, seed)
    if gold is None or not runner.results:
        return None, "no golden file for this workload (oracle/gen_bench_golden.py W H B seed)"
    recs, dots, slots = runner.results[-1]
    for r2, d2, _ in runner.results[:-1]:
        if [tuple(r) for r in r2] != [tuple(r) for r in recs] or list(d2) != list(dots):
            return False, "two timed steps over the same frames gave different results"
    return golden_check.check_batch(gold, frames, recs, dots, lambda j: ctx.download_flow(slots[j]))


def pcie_inclusive(W, H, B, device, seed, n_frames, bgr, base=None):
    """Host numpy frames -> per-pair scalars through pipeline.PairEngine: every frame crosses PCIe once (pinned
    staging copy + hipMemcpyAsync on the copy stream, overlapped with the previous batches' kernels)."""
    from funscript_flow_amd import _capi, pipeline
    from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames
    if base is None:
        base = sine_translate_frames(17, W, H, seed=seed)
    base = base[:17]                                  # one period of the clip, cycled
    if bgr:
        base = gray_to_bgr(base)
    frames = [base[i % len(base)] for i in range(n_frames)]
    with _capi.Context(W, H, device=device, max_batch=B, frame_slots=2 * B + 2, flow_slots=pipeline.min_flow_slots(B)) as ctx:
        eng = pipeline.PairEngine(ctx)
        eng.process_chunk(frames[:2 * B + 1])  # warm-up
        t0 = time.perf_counter()
        eng.process_chunk(frames)
        dt = time.perf_counter() - t0
    n = n_frames - 1
    return {"value": n / dt, "unit": "pairs/s", "pairs": n, "input": "BGR uint8 ndarrays" if bgr else "gray uint8 ndarrays",
            "h2d_GBps": n / dt * W * H * (3 if bgr else 1) / 1e9, "pairs_per_batch": B}


_REAL_STDOUT = None


def _quiet_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (gloo prints a connection banner per
    process group, from C++): point file descriptor 1 at stderr for the whole run and keep the real stdout for the line."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(line):
    out = _REAL_STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    _quiet_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=32,
                    help="pairs per ffl_flow_pairs call (a step); 32 fills the device at every pyramid level: "
                         "1080p 4450 / 4700 / 5030 / 4880 pairs/s at B = 8 / 16 / 32 / 64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernel_classes / pcie_inclusive / small_image passes")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU sample (default: one per thread, >= 32)")
    ap.add_argument("--independent", action="store_true", help="2B frames per batch (no frame sharing)")
    ap.add_argument("--zoom", type=float, default=0.0,
                    help="breathing zoom of the synthetic clip (0 = BASELINE's pure sine-translate); a zooming clip has a "
                         "spatially varying flow, which the data-dependent gather fast path sees less often")
    ap.add_argument("--fuse-first", type=int, default=-1,
                    help="minimum tiles x pairs of a level for folding its initial UpdateMatrices into the first blur+solve "
                         "launch (library default 10000; 0: never, 1: always)")
    ap.add_argument("--blur-rows", type=int, default=0, help="tiles a k_blur_solve workgroup walks down (0 = automatic)")
    ap.add_argument("--tile-order", type=int, default=-1, help="k_blur_solve / k_update_matrices tile order (ffl_set_option)")
    ap.add_argument("--trace-steps", action="store_true", help="print per-step host wall times to stderr")
    ap.add_argument("--lanes", type=int, default=0, help="compute lanes (co-scheduled batches) per context, default 1")
    ap.add_argument("--expand", type=int, default=0, choices=[0, 1, 2],
                    help="schedule of the frame-only kernels (pyramid + PolyExp): 0 (default) serial on the lane's "
                         "stream; 2 the 4 levels fork onto side streams and join before the flow chain starts; "
                         "1 run-ahead, the chain waits per level")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the process group is gloo")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: no per-kernel HIP events (roofline omitted)")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around every kernel class in the TIMED region (adds ~0.2 ms/step); default: the "
                         "dominant kernel only, the other classes are timed in the separate kernel_classes pass")
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

    global FUSE_FIRST
    if args.fuse_first >= 0:
        _capi.set_option("fuse_first", args.fuse_first)
        FUSE_FIRST = args.fuse_first
    if args.blur_rows:
        _capi.set_option("blur_rows", args.blur_rows)
    if args.tile_order >= 0:
        _capi.set_option("tile_order", args.tile_order)
    # One compute lane by default: kernels of consecutive batches then run back to back, so the
    # per-launch HIP-event durations behind `roofline` are those of the kernel alone (and agree with
    # rocprofv3's).  `--lanes 2` (the library's default for production use) co-schedules two batches.
    _capi.set_option("lanes", args.lanes or 1)
    _capi.set_option("run_ahead", args.expand)
    W, H, B = args.width, args.height, args.batch
    N = W * H
    seed = 1 if world == 1 else 10 + rank

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    events = False if args.no_events else (True if args.profile_all else [DOMINANT])
    dt, prof, runner, frames, level_sizes, U, ctx = resident_pass(
        W, H, B, args.steps, args.warmup, local_rank, seed, events, args.independent, args.zoom, barrier, args.trace_steps)
    if runner.trace:
        d = np.diff(np.array(runner.trace[-args.steps:])) * 1e3
        print("step ms:", " ".join(f"{v:.2f}" for v in d), file=sys.stderr)
    plain = not args.independent and args.zoom == 0.0
    checked, check_detail = verify(runner, frames, W, H, B, seed, ctx) if plain else (None, "non-default clip")
    results = runner.results
    ctx.close()

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # host gather of the per-pair scalars (x, y, cut, dot): the path's only exchange, ~40 B/pair
        mine = np.array([[r[0], r[1], int(r[4]), d] for recs, dots, _ in results for r, d in zip(recs, dots)], np.float64)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0, group=host_group)
        flags = [None] * world if rank == 0 else None
        dist.gather_object((checked, check_detail), flags, dst=0, group=host_group)
        if rank == 0:
            assert sum(len(g) for g in gathered) == world * args.steps * B
            bad = [f for f in flags if f[0] is False]
            checked = False if bad else (True if all(f[0] is True for f in flags) else None)
            check_detail = bad[0][1] if bad else f"{sum(f[0] is True for f in flags)} of {world} ranks have goldens for their clip; " + flags[0][1]

    if rank == 0:
        pairs = world * args.steps * B
        alg = alg_bytes_per_batch(N, B, U, level_sizes)
        dom = max((k for k in alg), key=lambda k: prof[k][1])
        n_launch, ms = prof[dom]
        achieved = alg[dom] * args.steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        avg_ms = ms / max(n_launch, 1)
        traffic, traffic_note = None, "profiles/traffic.json absent"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") != f"{W}x{H}" or tj.get("batch") != B or tj.get("kernel") != dom:
                    traffic_note = "profiles/traffic.json was measured on another workload"
                elif tj.get("kernel_signature") != kernel_signature():
                    traffic_note = (f"profiles/traffic.json is stale: measured on kernels {tj.get('kernel_signature')}, "
                                    f"this build is {kernel_signature()} (re-run profiles/tools/capture_round.sh)")
                elif tj.get("fuse_first", 10000) != FUSE_FIRST:
                    traffic_note = "profiles/traffic.json was measured with another fuse_first"
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on kernels "
                                    f"{tj.get('kernel_signature')} ({tj.get('captured', '?')}); 2 x FETCH + WRITE, KiB")
            except Exception as e:  # noqa: BLE001
                traffic_note = f"profiles/traffic.json unreadable: {e}"
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
            "checked": checked,
            "check_detail": check_detail,
            "config": {"workload": f"{W}x{H} synthetic sine-translate frame-pair stream, gray frames resident in HBM",
                       "pairs_per_step": B, "frames_per_step": U, "levels": len(level_sizes), "winsize": 15, "iterations": 3,
                       "poly_n": 5, "parallelism": f"pair-shard x{world}", "compute_lanes": args.lanes or 1,
                       "kernel_signature": kernel_signature()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / PEAK_GBPS, "traffic": traffic,
                         "frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_GBPS) if traffic and avg_ms > 0 else None,
                         "traffic_note": traffic_note,
                         "launches": n_launch, "avg_launch_ms": avg_ms,
                         "alg_bytes_per_launch": alg[dom] * args.steps / max(n_launch, 1)},
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items() if v[0]},
            "whole_path": {"alg_bytes_per_pair": sum(alg.values()) / B,
                           "achieved_GBps": sum(alg.values()) / B * (pairs / world) / dt / 1e9},
        }
        if world == 1 and not args.no_extras:
            # (1) every kernel class under HIP events, in a pass of its own (the events cost ~0.2 ms per step), one batch
            # at a time (depth 0) so that pass 2 is not co-scheduled with the next batch's kernels: every class alone
            ksteps = max(3, min(args.steps, 6))
            kdt, kprof, _, _, _, _, kctx = resident_pass(W, H, B, ksteps, 1, local_rank, seed, True, args.independent, args.zoom,
                                                          frames=frames, depth=0)
            kctx.close()
            out["kernel_classes"] = {
                k: {"ms_per_step": v[1] / ksteps, "launches_per_step": v[0] / ksteps,
                    "alg_bytes_per_step": alg.get(k), "frac": (alg[k] / (v[1] / ksteps * 1e-3) / 1e9 / PEAK_GBPS) if k in alg and v[1] > 0 else None}
                for k, v in kprof.items() if v[0]}
            out["kernel_classes"]["_pass"] = {"steps": ksteps, "ms_per_step_unpipelined_with_events": kdt / ksteps * 1e3,
                                              "note": "one batch at a time, host waits between batches: class times are those of each kernel running alone"}
            # (2) PCIe-inclusive: host frames -> scalars (never `value`)
            _capi.set_option("lanes", 2)
            nfr = 8 * B + 1
            pbase = frames if (len(frames) >= 17 and not args.independent and args.zoom == 0.0) else None
            out["pcie_inclusive"] = {"gray": pcie_inclusive(W, H, B, local_rank, seed, nfr, False, pbase),
                                     "bgr": pcie_inclusive(W, H, B, local_rank, seed, nfr, True, pbase),
                                     "note": "pipeline.PairEngine, 2 compute lanes, pageable ndarrays copied into pinned staging"}
            _capi.set_option("lanes", args.lanes or 1)
            # (3) the reference's own operating point (FF:1057: every frame is resized to 256x256 first)
            if (W, H) != (256, 256):
                SB = min(_capi.FFL_MAX_BATCH, 256)
                # two compute lanes (the library's default): small levels are latency-bound, a second batch in flight
                # fills the device (+7 % over one lane at this size); no per-kernel events are taken in this pass
                _capi.set_option("lanes", 2)
                sdt, sprof, srun, sfr, slv, sU, sctx = resident_pass(256, 256, SB, 30, 5, local_rank, 1, False)
                schk = verify(srun, sfr, 256, 256, SB, 1, sctx)
                sctx.close()
                salg = alg_bytes_per_batch(256 * 256, SB, sU, slv)
                out["small_image"] = {"workload": "256x256 pairs (FF:1057), gray frames resident", "pairs_per_step": SB,
                                      "value": 30 * SB / sdt, "unit": "pairs/s", "ms_per_step": sdt / 30 * 1e3,
                                      "whole_path_GBps": sum(salg.values()) * 30 / sdt / 1e9,
                                      "whole_path_frac": sum(salg.values()) * 30 / sdt / 1e9 / PEAK_GBPS,
                                      "launch": "captured hipGraph replay per batch, 2 compute lanes (no per-kernel events in this pass)",
                                      "compute_lanes": 2,
                                      "checked": schk[0], "check_detail": schk[1]}
                _capi.set_option("lanes", 2)
                out["small_image"]["pcie_inclusive_gray"] = pcie_inclusive(256, 256, SB, local_rank, 1, 8 * SB + 1, False, sfr)
                _capi.set_option("lanes", args.lanes or 1)
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            threads = cores                              # the reference: Pool(os.cpu_count()) over pairs, FF:1190
            npairs = args.cpu_pairs or max(threads, 32)  # one pair per worker: ~0.6 s per 1080p pair and core
            v, cdt = cpu_baseline(frames, threads, npairs)
            try:
                usable = len(os.sched_getaffinity(0))
            except AttributeError:
                usable = cores
            out["cpu_baseline"] = {"value": v, "unit": "pairs/s", "cores": cores, "threads": threads, "usable_cores": usable,
                                   "kind": "port",
                                   "sample": f"{npairs} pairs of the same {W}x{H} stream, one pair per worker thread, "
                                             f"{threads} threads = os.cpu_count() as the reference's Pool (FF:1190), "
                                             f"C oracle (Farneback + argmax + mean + radial), {cdt:.1f} s wall"}
        _emit(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
