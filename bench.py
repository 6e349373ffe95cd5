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


def _cpu_worker(frames, w, h, jobs, barrier, q, fast):
    """One worker PROCESS of the cpu_baseline pool (forked before any GPU call): its own pre-faulted workspace, so the
    timed loop neither mallocs nor first-touches a page; starts on a barrier shared by the whole pool."""
    try:
        import oracle as orc
        ws = orc.PairWorkspace(w, h, fast)
        n = len(frames) - 1
        barrier.wait(timeout=300)
        t0 = time.perf_counter()              # CLOCK_MONOTONIC: comparable across the pool's processes
        for j in jobs:
            flow, x, y, v, mm = ws.pair(frames[j % n], frames[j % n + 1])
            ws.radial((w / 2.0, h / 2.0), mm > 7, False)
        q.put((t0, time.perf_counter(), len(jobs)))
    except BaseException as e:  # noqa: BLE001
        q.put(("error", repr(e), 0))


def _cpu_pool_rate(frames, workers, pairs_per_worker, fast=False):
    """pairs/s of `workers` processes x `pairs_per_worker` pairs each: all pairs / (last end - first start)."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    h, w = frames[0].shape
    barrier, q = ctx.Barrier(workers), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, daemon=True,
                         args=(frames, w, h, [i * pairs_per_worker + k for k in range(pairs_per_worker)], barrier, q, fast))
             for i in range(workers)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in procs:
            r = q.get(timeout=600)
            if r[0] == "error":
                raise RuntimeError(f"cpu_baseline worker failed: {r[1]}")
            got.append(r)
    finally:
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
    t0, t1 = min(g[0] for g in got), max(g[1] for g in got)
    return sum(g[2] for g in got) / (t1 - t0), t1 - t0


def cpu_baseline(frames, max_workers=None, pairs_per_worker=2):
    """The reference's CPU path as a stated baseline: a pool of PROCESSES over pairs (FunscriptFlow.pyw:1190-1191,
    Pool(processes=threads).starmap), each running the C oracle ('port': Farneback + argmax + mean magnitude + radial) on
    the same synthetic stream.  MUST run before the first HIP call of this process (fork is only safe then): main() calls
    it before `import torch`.  Measures 1 worker, then 16 / 64 / os.cpu_count() workers (those that fit) so that the knee
    is visible; `value` is the best point of the sweep.

    Two builds of the same C source are timed.  `value` (and the sweep) come from liboracle_fast.so: -O3 -march=native of
    the host it runs on, FMA contraction and vectorisation allowed -- the reference's CPU path is cv2's SIMD wheel
    (FF:878-879), so the stated baseline should not be handicapped by the flags bit parity needs.  `value_parity_build`
    is liboracle.so (-O2 -ffp-contract=off, the build every parity test checks against) at the same worker count, and
    `max_abs_dflow_fast_vs_parity` says how far the fast build's flow is from it on the first timed pairs.  The fast
    build is never used as a checker."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.lib()                                   # build + load once, the workers inherit it
    fast = orc.lib_fast() is not None
    h, w = frames[0].shape
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    per_worker = orc.lib().orc_ws_bytes(w, h) + 8 * w * h
    try:
        import psutil
        fit = max(1, int(psutil.virtual_memory().available * 0.5 // per_worker))
    except Exception:  # noqa: BLE001
        fit = cores
    top = min(max_workers or cores, cores, fit)
    single, sdt = _cpu_pool_rate(frames, 1, max(pairs_per_worker, 2), fast)
    sweep = {"1": {"pairs_per_s": single, "per_worker": single, "wall_s": sdt}}
    stopped = None
    for n in sorted({min(16, top), min(64, top), top}):
        if n <= 1:
            continue
        v, dt = _cpu_pool_rate(frames, n, pairs_per_worker, fast)
        sweep[str(n)] = {"pairs_per_s": v, "per_worker": v / n, "wall_s": dt}
        if v / n < single / 3.0 and n < top:
            # the pool is past the knee (a worker gets less than a third of a core: the box's CPU quota or memory system is
            # exhausted); wider pools only get slower and take minutes -- not measured
            stopped = (f"sweep stopped at {n} workers: {v / n:.2f} pairs/s per worker is below a third of the single-worker "
                       f"rate {single:.2f}; {top} workers not measured")
            break
    best = max(sweep, key=lambda k: sweep[k]["pairs_per_s"])
    widest = max(sweep, key=int)
    wall = sum(v["wall_s"] for v in sweep.values())
    parity = {}
    if fast:
        # the parity build at the two points that matter (one worker, the best worker count), and the distance of the two
        # builds' results on the first two timed pairs (computed here, in the parent, which has not touched the GPU)
        ps, pdt = _cpu_pool_rate(frames, 1, max(pairs_per_worker, 2), False)
        pb, pbdt = (ps, 0.0) if int(best) == 1 else _cpu_pool_rate(frames, int(best), pairs_per_worker, False)
        wall += pdt + pbdt
        a, b = orc.PairWorkspace(w, h, True), orc.PairWorkspace(w, h, False)
        dmax, same_argmax = 0.0, True
        for j in range(min(2, len(frames) - 1)):
            fa, xa, ya, _, _ = a.pair(frames[j], frames[j + 1])
            fb, xb, yb, _, _ = b.pair(frames[j], frames[j + 1])
            dmax = max(dmax, float(np.max(np.abs(fa - fb))))
            same_argmax = same_argmax and (xa, ya) == (xb, yb)
        parity = {"value_parity_build": pb, "single_thread_parity_build": ps, "fast_over_parity": sweep[best]["pairs_per_s"] / pb,
                  "max_abs_dflow_fast_vs_parity": dmax, "argmax_equal_fast_vs_parity": same_argmax}
    quota = None
    try:                                         # cgroup v2 CPU quota of the box, if any: "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001
        pass
    out = {"value": sweep[best]["pairs_per_s"], "unit": "pairs/s", "cores": int(best), "kind": "port",
           "build": ("liboracle_fast.so: gcc -O3 -march=native -ffp-contract=fast of the oracle source, built on this host (timing only)"
                     if fast else "liboracle.so: gcc -O2 -ffp-contract=off (the parity build; the -O3 -march=native build could not be made here)"),
           "host_cores": cores, "usable_cores": usable, "cpu_quota_cores": quota, "single_thread": single,
           "per_worker_vs_single_at_value": sweep[best]["per_worker"] / single,
           "widest": {"workers": int(widest), "pairs_per_s": sweep[widest]["pairs_per_s"], "per_worker": sweep[widest]["per_worker"],
                      "per_worker_vs_single": sweep[widest]["per_worker"] / single},
           "sweep": sweep, "sweep_note": stopped,
           "sample": f"{w}x{h} pairs of the same synthetic stream; a pool of worker PROCESSES as the reference's "
                     f"Pool(processes=threads).starmap (FF:1190-1191), forked before any GPU call, one pre-faulted workspace "
                     f"per worker (every page of scratch and output touched before the barrier), {pairs_per_worker} pairs per "
                     f"worker, barrier start; C oracle (Farneback + argmax + mean magnitude + radial); `value` = best point of "
                     f"the sweep ({best} workers), `cores` = its worker count; total {wall:.1f} s wall"}
    out.update(parity)
    return out


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
    ctx = _capi.Context(W, H, device=device, frame_slots=U + 1, flow_slots=((depth if depth else StepRunner.DEPTH) + 1) * B, max_batch=B)
    level_sizes = [ctx.level_size(k) for k in range(ctx.num_levels() + 1)]
    ctx.upload_frames(0, list(frames))
    ctx.sync()
    runner = StepRunner(ctx, B, independent, SMOOTH_RADIUS, depth)
    # A generation-2 collection walks every object torch's import created (40-50 ms: longer than the whole
    # timed region at small frame sizes); park those objects in the permanent generation first -- BEFORE the warm-up: a
    # 50 ms pause between the warm-up and the timed region let the idle device drop its clocks, and the first timed
    # steps then ran 1.5-2 % slow (k_blur_solve 346 us per launch over 20 timed steps against 340 over 100, whatever W).
    gc.collect()
    gc.freeze()
    ctx.profile_enable(events)          # the warm-up steps are launched exactly as the timed ones
    if warmup > 0:
        runner.run(warmup)
    runner.results.clear()
    ctx.profile_read()                  # drops the warm-up's event records
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


def _golden_check():
    """tests/golden_check.py: the comparison with the committed oracle goldens (test infrastructure, kept beside them)."""
    tdir = os.path.join(ROOT, "tests")
    if tdir not in sys.path:
        sys.path.insert(0, tdir)
    import golden_check
    return golden_check


def verify(runner, frames, W, H, B, seed, ctx):
    """Timed steps vs the oracle goldens: every step must give the same records and scalars (same inputs), the
    last one is compared number by number, its flow slots are still resident for the crc check."""
    golden_check = _golden_check()
    gold = golden_check.load_golden(W, H, B, seed)
    if gold is None or not runner.results:
        return None, "no golden file for this workload (oracle/gen_bench_golden.py W H B seed)"
    recs, dots, slots = runner.results[-1]
    for r2, d2, _ in runner.results[:-1]:
        if [tuple(r) for r in r2] != [tuple(r) for r in recs] or list(d2) != list(dots):
            return False, "two timed steps over the same frames gave different results"
    return golden_check.check_batch(gold, frames, recs, dots, lambda j: ctx.download_flow(slots[j]))


CHUNK_FRAMES = 3000   # the reference's default chunk: params["batch_size"] frames are decoded, paired and processed together (FF:2647, FF:1145-1153)


def pcie_inclusive(W, H, B, device, seed, n_frames, bgr, base=None, pinned=False):
    """Host frames -> per-pair scalars through pipeline.PairEngine: every frame crosses PCIe once, on the copy stream,
    overlapped with the previous batches' kernels.  pinned = False: pageable ndarrays (what cv2.VideoCapture.read hands
    Python, FF:178), copied into the context's pinned staging first; pinned = True: the frames already lie in page-locked
    memory of the context (Context.pinned_frames = ffl_host_alloc, where prefetch.PrefetchRing makes the decoder write
    them), so the H2D transfer starts straight out of them -- no staging copy.

    The timed unit is ONE chunk of n_frames frames from a cold pipeline, as process_video runs it (FF:1187-1242): the upload
    of the chunk's first batch has nothing to hide behind (3.7 ms for 33 BGR frames at 1080p), so a short chunk reads low --
    8 batches: 0.87x the resident rate from BGR, 94 batches (the reference's 3000-frame chunk): 0.98x
    (profiles/r04_pcie_chunk_length.txt)."""
    from funscript_flow_amd import _capi, pipeline
    from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames
    if base is None:
        base = sine_translate_frames(17, W, H, seed=seed)
    base = base[:17]                                  # one period of the clip, cycled
    if bgr:
        base = gray_to_bgr(base)
    with _capi.Context(W, H, device=device, max_batch=B, frame_slots=pipeline.min_frame_slots(B, 2),
                       flow_slots=pipeline.min_flow_slots(B, 2)) as ctx:      # room for two batches queued ahead
        if pinned:
            # a ring of page-locked frames, a multiple of the clip's period long (so that entry i % S holds frame i's
            # pixels): the frames of a batch lie back to back in it except where the ring wraps (one staged call per S frames)
            S = min(n_frames, len(base) * max(1, (8 * B + len(base)) // len(base)))
            store = ctx.pinned_frames(S, 3 if bgr else 1)
            for i in range(S):
                store[i] = base[i % len(base)]          # the "decoder" writes into page-locked memory, outside the timed region
            frames = [store[i % S] for i in range(n_frames)]
        else:
            frames = [base[i % len(base)] for i in range(n_frames)]
        eng = pipeline.PairEngine(ctx)
        eng.process_chunk(frames[:2 * B + 1])  # warm-up
        t0 = time.perf_counter()
        eng.process_chunk(frames)
        dt = time.perf_counter() - t0
        frames = store = None
    n = n_frames - 1
    return {"value": n / dt, "unit": "pairs/s", "pairs": n, "chunk_frames": n_frames,
            "input": ("BGR" if bgr else "gray") + (" uint8 frames in page-locked host memory (ffl_host_alloc), no staging copy" if pinned
                                                   else " uint8 pageable ndarrays, staged through pinned memory"),
            "h2d_GBps": n / dt * W * H * (3 if bgr else 1) / 1e9, "pairs_per_batch": B}


def load_traffic(W, H, B, kernel):
    """PMC traffic of `kernel` for this workload from profiles/traffic.json (one entry per workload, each tied to the
    signature of the device sources it was measured on): (bytes per launch or None, note)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "profiles/traffic.json absent"
    try:
        tj = json.load(open(tpath))
        e = tj.get("workloads", {}).get(f"{W}x{H}_b{B}")
        if e is None or e.get("kernel") != kernel:
            return None, f"profiles/traffic.json has no {kernel} entry for {W}x{H} B={B}"
        if e.get("kernel_signature") != kernel_signature():
            return None, (f"profiles/traffic.json entry is stale: measured on kernels {e.get('kernel_signature')}, this build is "
                          f"{kernel_signature()} (re-run profiles/tools/capture_round.sh)")
        if e.get("fuse_first", 10000) != FUSE_FIRST:
            return None, "profiles/traffic.json entry was measured with another fuse_first"
        return e.get("hbm_bytes_per_launch"), (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on kernels "
                                               f"{e.get('kernel_signature')} ({e.get('captured', '?')}); 2 x FETCH + WRITE, KiB")
    except Exception as ex:  # noqa: BLE001
        return None, f"profiles/traffic.json unreadable: {ex}"


def roofline_block(prof, alg, steps, W, H, B, kernel=None):
    """The `roofline` object for the dominant kernel class of a pass timed with HIP events."""
    dom = kernel or max((k for k in alg), key=lambda k: prof[k][1])
    n_launch, ms = prof[dom]
    achieved = alg[dom] * steps / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    avg_ms = ms / max(n_launch, 1)
    traffic, note = load_traffic(W, H, B, dom)
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / PEAK_GBPS, "traffic": traffic,
            "frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_GBPS) if traffic and avg_ms > 0 else None,
            "traffic_note": note, "launches": n_launch, "avg_launch_ms": avg_ms,
            "alg_bytes_per_launch": alg[dom] * steps / max(n_launch, 1)}


def large_image(W, H, B, steps, warmup, device, seed, what):
    """A second resident workload in the same line (configs[2] 3840x2160, the configs[4] eye 2880x2880): pairs/s, the
    dominant kernel's roofline from HIP events in the timed region, `checked` against the oracle golden."""
    dt, prof, run, fr, lv, U, ctx = resident_pass(W, H, B, steps, warmup, device, seed, [DOMINANT])
    chk = verify(run, fr, W, H, B, seed, ctx)
    ctx.close()
    alg = alg_bytes_per_batch(W * H, B, U, lv)
    return {"workload": what, "pairs_per_step": B, "steps": steps, "value": steps * B / dt, "unit": "pairs/s",
            "ms_per_step": dt / steps * 1e3, "roofline": roofline_block(prof, alg, steps, W, H, B, DOMINANT),
            "whole_path_frac": sum(alg.values()) * steps / dt / 1e9 / PEAK_GBPS,
            "checked": chk[0], "check_detail": chk[1]}


def one_clip(args, rank, world, local_rank, host_group, barrier, max_over_ranks, dist):
    """north_star's wording: the pairs of ONE clip dealt round-robin over the GPUs (strong scaling of a chunk).

    The clip is steps x batch x N pairs of the seed-1 stream (host ndarrays: 17 distinct frames, cycled); every rank runs
    pipeline.process_chunk_sharded -- pass 1 on its blocks of --rr-block pairs (a block of b pairs uploads and expands
    b + 1 frames), host all-gather of the 32-byte pass-1 records (gloo), centres smoothed over the whole clip, pass 2 on
    the owning rank, host all-gather of the scalars.  Frames cross PCIe inside the timed region, so this is a
    PCIe-inclusive figure; `value` = pairs of the clip / max-over-ranks time."""
    from funscript_flow_amd import _capi, pipeline
    from funscript_flow_amd.synth import sine_translate_frames
    W, H, B = args.width, args.height, args.batch
    n_pairs = args.steps * B * world
    base = sine_translate_frames(17, W, H, seed=1)
    frames = [base[i % 16] for i in range(n_pairs + 1)]        # period 16: frame 16 == frame 0 (the same array object)
    mine = pipeline.shard_pairs(n_pairs, world, rank, args.assign, args.rr_block)
    _capi.set_option("lanes", args.lanes or 2)

    def allgather(obj):
        if world == 1:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj, group=host_group)
        return out

    def chunk(fr):
        if args.assign == "contiguous":      # streaming form: only the 6 + 6 halo records cross between the passes
            return pipeline.process_chunk_sharded_halo(eng, fr, rank, world, allgather)
        return pipeline.process_chunk_sharded(eng, fr, rank, world, allgather, assign=args.assign, block=args.rr_block)

    with _capi.Context(W, H, device=local_rank, max_batch=B, frame_slots=2 * B + 2, flow_slots=max(len(mine), 1)) as ctx:
        eng = pipeline.HipShardEngine(ctx)
        chunk(frames[:min(n_pairs, 2 * B * world) + 1])    # warm-up
        barrier()
        t0 = time.perf_counter()
        dots, recs = chunk(frames)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks(dt)
    if rank == 0:
        # (a) the stream has period 16: pair j and pair j + 16 are the same images, and (away from the clip's ends, where the
        # smoothing window is clipped) have the same centre -> the same scalar, whichever rank computed them
        period_ok = all(tuple(recs[j]) == tuple(recs[j + 16]) for j in range(n_pairs - 16)) and \
            all(dots[j] == dots[j + 16] for j in range(6, n_pairs - 22))
        # (b) correctness, not only determinism: the clip's first pairs are the first pairs of the seed-1 stream the oracle
        # golden of the default workload was made from (a frame depends on t only): records (x, y, cut) of pairs 0..31 and
        # the scalars of pairs 0..25 (whose +-6 window lies inside both the golden's 32-pair batch and this clip)
        gold = _golden_check().load_golden(W, H, 32, 1)
        ok, detail = None, "no oracle golden for this frame size: only the period-16 property across ranks was checked"
        if not period_ok:
            ok, detail = False, "pairs 16 apart (same images) gave different records / scalars on different ranks"
        elif gold is not None:
            import zlib
            if zlib.crc32(np.ascontiguousarray(sine_translate_frames(33, W, H, seed=1)).tobytes()) != gold["frames_crc32"]:
                detail = "synthetic frames differ from the golden's (numpy/libm rounding): only the period-16 property was checked"
            else:
                m = min(n_pairs, 32)
                bad = [j for j in range(m) if (int(recs[j][0]), int(recs[j][1]), bool(recs[j][2])) !=
                       (gold["x"][j], gold["y"][j], gold["cut"][j])]
                g = np.asarray(gold["dots"], np.float64)
                md = min(n_pairs - 6, 26) if n_pairs >= 32 else 0
                scale = float(np.mean(np.abs(g)))
                bad_d = [j for j in range(md) if abs(dots[j] - g[j]) > 1e-4 * max(abs(g[j]), scale)]
                ok = not bad and not bad_d
                detail = (f"{m} pass-1 records and {md} scalars equal the oracle golden bench_{W}x{H}_b32_s1.json; period-16 property "
                          f"holds over all {n_pairs} pairs across ranks" if ok else
                          f"differs from the oracle golden: records of pairs {bad[:4]}, scalars of pairs {bad_d[:4]}")
        _emit(json.dumps({
            "metric": "1080p frame-pairs/sec" if (W, H) == (1920, 1080) else f"{W}x{H} frame-pairs/sec",
            "mode": "one_clip", "value": n_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": 1,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "checked": ok, "check_detail": detail,
            "config": {"workload": f"ONE {W}x{H} clip of {n_pairs} pairs from host ndarrays (uploads inside the timed region), "
                                   f"pairs dealt {args.assign} in blocks of {args.rr_block} over {world} GPU(s)",
                       "pairs_per_step": B, "parallelism": f"pair-shard x{world} ({args.assign}, block {args.rr_block})",
                       "compute_lanes": args.lanes or 2,
                       "exchange": ("host all-gather (gloo) of the 6 + 6 halo records per rank (32 B each) between the passes, 40 B per pair of results at the end"
                                    if args.assign == "contiguous" else "host all-gather (gloo) of 32 B per pair after pass 1, 16 B after pass 2"),
                       "kernel_signature": kernel_signature()}}))


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
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs per worker process of the CPU sample (default 2)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="widest pool of the cpu_baseline sweep (default os.cpu_count())")
    ap.add_argument("--independent", action="store_true", help="2B frames per batch (no frame sharing)")
    ap.add_argument("--zoom", type=float, default=0.0,
                    help="breathing zoom of the synthetic clip (0 = BASELINE's pure sine-translate); a zooming clip has a "
                         "spatially varying flow, which the data-dependent gather fast path sees less often")
    ap.add_argument("--fuse-first", type=int, default=-1,
                    help="minimum tiles x pairs of a level for folding its initial UpdateMatrices into the first blur+solve "
                         "launch (library default 10000; 0: never, 1: always)")
    ap.add_argument("--blur-rows", type=int, default=0, help="tiles a k_blur_solve workgroup walks down (0 = automatic)")
    ap.add_argument("--blur-min-wgs", type=int, default=0, help="automatic strip length of k_blur_solve: longest strips that still give this many workgroups (library default 3500)")
    ap.add_argument("--tile-order", type=int, default=-1, help="k_blur_solve / k_update_matrices tile order (ffl_set_option)")
    ap.add_argument("--trace-steps", action="store_true", help="print per-step host wall times to stderr")
    ap.add_argument("--lanes", type=int, default=0, help="compute lanes (co-scheduled batches) per context, default 1")
    ap.add_argument("--expand", type=int, default=0, choices=[0, 1, 2],
                    help="schedule of the frame-only kernels (pyramid + PolyExp): 0 (default) serial on the lane's "
                         "stream; 2 the 4 levels fork onto side streams and join before the flow chain starts; "
                         "1 run-ahead, the chain waits per level")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the process group is gloo")
    ap.add_argument("--one-clip", action="store_true",
                    help="strong-scaling form (north_star: 'frame-pairs sharded round-robin across GPUs'): ONE clip of "
                         "steps x batch x N pairs, dealt to the ranks in blocks of --rr-block pairs by "
                         "pipeline.process_chunk_sharded (host frames, uploads included); beside, never instead of, the default")
    ap.add_argument("--rr-block", type=int, default=8, help="--one-clip: pairs per round-robin block (1 = pair by pair)")
    ap.add_argument("--assign", default="round_robin", choices=["round_robin", "contiguous"], help="--one-clip: pair assignment")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: no per-kernel HIP events (roofline omitted)")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around every kernel class in the TIMED region (adds ~0.2 ms/step); default: the "
                         "dominant kernel only, the other classes are timed in the separate kernel_classes pass")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started from a plain shell (`python bench.py --gpus N ...`, the driver's command): this process becomes the PARENT
        # that spawns one fresh rank process per GPU (FF:1190-1191: the reference's parent creates its pool) and nothing
        # else -- no torch import, no HIP call, no CPU baseline here.  Rank 0's one JSON line is relayed to stdout,
        # everything else goes to stderr, the exit status is the first failing rank's (the others are stopped).
        from funscript_flow_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    _quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    W, H, B = args.width, args.height, args.batch
    N = W * H
    seed = 1 if world == 1 else 10 + rank

    # The CPU baseline is a pool of forked worker processes (the reference's Pool, FF:1190-1191): it runs FIRST, before
    # torch / HIP are loaded -- never fork a process that has initialised the GPU.  Under rocprofv3 the profiler's preloaded
    # library has already done that, so the leg is skipped there (the capture scripts pass --no-cpu-baseline anyway).
    from funscript_flow_amd.synth import sine_translate_frames
    frames0 = None
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
            cpu_base = {"value": None, "skipped": "running under rocprofv3: the GPU is already initialised, no fork"}
        else:
            frames0 = sine_translate_frames((2 * B if args.independent else B + 1), W, H, seed=seed, zoom=args.zoom)
            cpu_base = cpu_baseline(frames0[:17], max_workers=args.cpu_workers or None, pairs_per_worker=args.cpu_pairs or 2)

    import torch  # first: its HIP runtime (same SONAME) is the one libffl_hip.so binds to
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # Which device this rank drives: cuda:LOCAL_RANK when the rank sees the whole node, cuda:0 when a visibility mask
    # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) leaves it exactly one device, cuda:0 for every rank of a
    # --rehearse-gloo run.  That two ranks do not share a card is checked below on the devices' PCI identities.
    ndev = torch.cuda.device_count()
    if args.rehearse_gloo or (world > 1 and ndev == 1):
        local_rank = 0
    if ndev <= local_rank:
        print(f"bench.py: FATAL rank {rank}/{world}: LOCAL_RANK {local_rank} but only {ndev} visible GPU(s); "
              f"one rank per GPU is required (use --rehearse-gloo only for single-GPU rehearsals)", file=sys.stderr, flush=True)
        os._exit(3)
    torch.cuda.set_device(local_rank)
    host_group = dev_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # The default group is gloo (host): it carries the identity check, the gather of the per-pair scalars and, in a
        # rehearsal, the barrier / MAX.  Ranks of one launch import torch at different speeds on a fresh box: 600 s.
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        host_group = dist.group.WORLD
        pr = torch.cuda.get_device_properties(local_rank)
        ident = (f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', -1):02x}",
                 str(getattr(pr, "uuid", "")))
        idents = [None] * world
        dist.all_gather_object(idents, ident, group=host_group)
        if not args.rehearse_gloo:
            clash = [(a, b) for a in range(world) for b in range(a + 1, world) if idents[a] == idents[b]]
            if clash:
                if rank == 0:
                    print(f"bench.py: FATAL: ranks {clash[0][0]} and {clash[0][1]} drive the same GPU ({idents[clash[0][0]][0]}); "
                          f"{ndev} device(s) visible to rank 0 -- one rank per GPU is required, no number is reported "
                          f"(use --rehearse-gloo only for single-GPU rehearsals)", file=sys.stderr, flush=True)
                os._exit(3)
            # RCCL carries only the contract's barrier / MAX (no data-path collective).  Bring the communicator up NOW, with
            # a real collective on the device, so that a broken RCCL / xGMI setup ends the run with one clear message and a
            # non-zero exit code before anything is timed -- never a silent switch to another backend.
            try:
                dev_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300),
                                           device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda") * (rank + 1)
                dist.all_reduce(probe, op=dist.ReduceOp.SUM, group=dev_group)
                torch.cuda.synchronize()
                if int(probe.item()) != world * (world + 1) // 2:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: FATAL rank {rank}/{world}: the RCCL ('nccl') process group did not come up on cuda:{local_rank}: "
                      f"{type(e).__name__}: {e}\nbench.py: no number is reported; there is no fallback backend for N > 1 "
                      f"(use --rehearse-gloo only for single-GPU rehearsals)", file=sys.stderr, flush=True)
                os._exit(3)

    from funscript_flow_amd import _capi

    global FUSE_FIRST
    if args.fuse_first >= 0:
        _capi.set_option("fuse_first", args.fuse_first)
        FUSE_FIRST = args.fuse_first
    if args.blur_rows:
        _capi.set_option("blur_rows", args.blur_rows)
    if args.blur_min_wgs:
        _capi.set_option("blur_min_wgs", args.blur_min_wgs)
    if args.tile_order >= 0:
        _capi.set_option("tile_order", args.tile_order)
    # One compute lane by default: kernels of consecutive batches then run back to back, so the
    # per-launch HIP-event durations behind `roofline` are those of the kernel alone (and agree with
    # rocprofv3's).  `--lanes 2` (the library's default for production use) co-schedules two batches.
    _capi.set_option("lanes", args.lanes or 1)
    _capi.set_option("run_ahead", args.expand)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if dev_group is not None:
                dist.barrier(group=dev_group, device_ids=[local_rank])   # RCCL
            else:
                dist.barrier(group=host_group)                           # --rehearse-gloo
        torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if dev_group is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=dev_group if dev_group is not None else host_group)
        return float(t.item())

    if args.one_clip:
        one_clip(args, rank, world, local_rank, host_group, barrier, max_over_ranks, dist)
        if world > 1:
            dist.barrier(group=host_group)
            dist.destroy_process_group()
        return

    events = False if args.no_events else (True if args.profile_all else [DOMINANT])
    dt, prof, runner, frames, level_sizes, U, ctx = resident_pass(
        W, H, B, args.steps, args.warmup, local_rank, seed, events, args.independent, args.zoom, barrier, args.trace_steps,
        frames=frames0)
    if runner.trace:
        d = np.diff(np.array(runner.trace[-args.steps:])) * 1e3
        print("step ms:", " ".join(f"{v:.2f}" for v in d), file=sys.stderr)
    plain = not args.independent and args.zoom == 0.0
    checked, check_detail = verify(runner, frames, W, H, B, seed, ctx) if plain else (None, "non-default clip")
    results = runner.results
    graphs = ctx.graph_stats()
    ctx.close()

    if world > 1:
        dt = max_over_ranks(dt)
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
        roof = roofline_block(prof, alg, args.steps, W, H, B)
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
                       "kernel_signature": kernel_signature(),
                       # hipGraph bookkeeping of the timed context (ffl_graph_stats): per-kernel HIP events (the roofline's
                       # clock) force one-by-one launches, so the timed region replays nothing unless --no-events is given;
                       # a non-zero capture_failures would mean a silently slower path and is reported, never hidden
                       "graphs": graphs},
            "roofline": roof,
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
            # k_gray (BGR -> gray at upload, 3N in + 1N out per frame: the 8N per pair of SURVEY's B_alg) never runs on resident
            # gray frames: time it here on one BGR upload of the batch's frames (events bracket the kernel, not the H2D copy)
            from funscript_flow_amd.synth import gray_to_bgr
            kctx.profile_enable(["k_gray"])
            kctx.upload_frames(0, list(gray_to_bgr(frames[:U])))
            gprof = kctx.profile_read()["k_gray"]
            kctx.close()
            alg = dict(alg, k_gray=4.0 * N * U)
            kprof["k_gray"] = (gprof[0] * ksteps, gprof[1] * ksteps)   # per-step figures below divide by ksteps
            out["kernel_classes"] = {
                k: {"ms_per_step": v[1] / ksteps, "launches_per_step": v[0] / ksteps,
                    "alg_bytes_per_step": alg.get(k), "frac": (alg[k] / (v[1] / ksteps * 1e-3) / 1e9 / PEAK_GBPS) if k in alg and v[1] > 0 else None}
                for k, v in kprof.items() if v[0]}
            out["kernel_classes"]["_pass"] = {"steps": ksteps, "ms_per_step_unpipelined_with_events": kdt / ksteps * 1e3,
                                              "note": "one batch at a time, host waits between batches: class times are those of each kernel running alone"}
            # (2) PCIe-inclusive: host frames -> scalars (never `value`)
            _capi.set_option("lanes", 2)
            nfr = CHUNK_FRAMES
            pbase = frames if (len(frames) >= 17 and not args.independent and args.zoom == 0.0) else None
            out["pcie_inclusive"] = {"gray": pcie_inclusive(W, H, B, local_rank, seed, nfr, False, pbase),
                                     "bgr": pcie_inclusive(W, H, B, local_rank, seed, nfr, True, pbase),
                                     "bgr_pinned": pcie_inclusive(W, H, B, local_rank, seed, nfr, True, pbase, pinned=True),
                                     "note": "one chunk of 3000 frames (the reference's default chunk, FF:2647) from a cold pipeline; "
                                             "pipeline.PairEngine, 2 compute lanes; gray / bgr: pageable ndarrays copied into "
                                             "pinned staging by the library; bgr_pinned: frames already in ffl_host_alloc memory "
                                             "(the prefetch ring's zero-copy path)"}
            for k in ("gray", "bgr", "bgr_pinned"):
                out["pcie_inclusive"][k]["vs_resident"] = out["pcie_inclusive"][k]["value"] / out["value"]
            _capi.set_option("lanes", args.lanes or 1)
            # (3) the reference's own operating point (FF:1057: every frame is resized to 256x256 first)
            if (W, H) != (256, 256):
                SB = min(_capi.FFL_MAX_BATCH, 256)
                # two compute lanes (the library's default): small levels are latency-bound, a second batch in flight
                # fills the device (+7 % over one lane at this size); no per-kernel events are taken in this pass
                _capi.set_option("lanes", 2)
                SSTEPS = 100   # 0.14 s: a 30-step pass (41 ms) still carried 3 % of pipeline fill and drain
                sdt, sprof, srun, sfr, slv, sU, sctx = resident_pass(256, 256, SB, SSTEPS, 5, local_rank, 1, False)
                schk = verify(srun, sfr, 256, 256, SB, 1, sctx)
                sgraphs = sctx.graph_stats()
                sctx.close()
                salg = alg_bytes_per_batch(256 * 256, SB, sU, slv)
                out["small_image"] = {"workload": "256x256 pairs (FF:1057), gray frames resident", "pairs_per_step": SB,
                                      "value": SSTEPS * SB / sdt, "unit": "pairs/s", "ms_per_step": sdt / SSTEPS * 1e3, "steps": SSTEPS,
                                      "whole_path_GBps": sum(salg.values()) * SSTEPS / sdt / 1e9,
                                      "whole_path_frac": sum(salg.values()) * SSTEPS / sdt / 1e9 / PEAK_GBPS,
                                      "launch": "captured hipGraph replay per batch, 2 compute lanes (no per-kernel events in this pass)",
                                      "compute_lanes": 2, "graphs": sgraphs,
                                      "checked": schk[0], "check_detail": schk[1]}
                _capi.set_option("lanes", 2)
                out["small_image"]["pcie_inclusive_gray"] = pcie_inclusive(256, 256, SB, local_rank, 1, CHUNK_FRAMES, False, sfr)
                _capi.set_option("lanes", 1)
                # the dominant kernel's own roofline at this size: a short pass with HIP events (eager launches, one lane)
                edt, eprof, _, _, _, _, ectx = resident_pass(256, 256, SB, 10, 2, local_rank, 1, [DOMINANT], frames=sfr)
                ectx.close()
                out["small_image"]["roofline"] = roofline_block(eprof, salg, 10, 256, 256, SB, DOMINANT)
                _capi.set_option("lanes", args.lanes or 1)
            # (4) the large configurations: configs[2] (3840x2160) and one eye of configs[4] (2880x2880), resident, one lane
            if (W, H) == (1920, 1080) and not args.independent and args.zoom == 0.0:
                _capi.set_option("lanes", 1)
                out["large_image"] = {
                    "3840x2160": large_image(3840, 2160, 32, 10, 2, local_rank, 1, "BASELINE configs[2]: 3840x2160 stream, gray frames resident"),
                    "2880x2880_eye": large_image(2880, 2880, 32, 10, 2, local_rank, 2,
                                                 "BASELINE configs[4] unit: one 2880x2880 eye of a 5760x2880 stereo stream, gray frames resident")}
                _capi.set_option("lanes", args.lanes or 1)
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        _emit(json.dumps(out))
    if world > 1:
        dist.barrier(group=host_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
