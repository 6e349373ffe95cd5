"""Host-side schedule around the HIP pair kernels: the counterpart of the per-chunk section of
process_video (FunscriptFlow.pyw:1187-1242) for the "HIP" backend.

    pairs = zip(frames[:-1], frames[1:])                                   FF:1188
    pass 1  (flow, |div| argmax, mean magnitude, cut) for every pair       FF:1190-1191
    c_j = mean of pos_center over pairs j-6..j+6 inside the chunk          FF:1203-1214
    pass 2  radial_motion_weighted(flow_j, c_j, cut_j, pov_mode)            FF:1232-1236

Everything runs in the calling process (HIP state must not cross fork); flows never leave HBM.
Multi-GPU: pairs are sharded over the ranks either in contiguous blocks or round-robin in blocks of
`block` pairs (`shard_pairs`); the only exchange is a host all-gather of the 32-byte pass-1 records (no
RCCL collective) -- of ALL records under round-robin (`process_chunk_sharded`), of the <= 12 halo records per
rank under contiguous blocks (`process_chunk_sharded_halo`, the streaming form).
"""
import numpy as np

SMOOTH_RADIUS = 6  # FF:1206 range(1, 7)


def smooth_centers(pos_centers, radius=SMOOTH_RADIUS):
    """FF:1203-1214: c_j = mean({p_i : |i-j| <= 6, 0 <= i < n}) as float64[2].

    The reference gathers the window's positions into a list and takes np.mean(..., axis=0) of the int64 pairs: an exact
    integer sum (far below 2^53) converted to float64 and divided by the count.  Window sums from a prefix sum give the
    same integers, hence the same float64 -- bit for bit (pinned by the chain captured from the real process_video,
    tests/test_host_side.py) -- without a Python loop over the chunk's 3000 pairs (13 ms -> 0.1 ms)."""
    p = np.asarray(pos_centers, np.int64).reshape(-1, 2)
    n = len(p)
    psum = np.zeros((n + 1, 2), np.int64)
    np.cumsum(p, axis=0, out=psum[1:])
    j = np.arange(n)
    lo, hi = np.maximum(0, j - radius), np.minimum(n, j + radius + 1)
    return (psum[hi] - psum[lo]) / (hi - lo)[:, None]


def min_flow_slots(max_batch, depth=1):
    """Flow slots a streaming two-pass schedule needs with `depth` + 1 batches in flight (depth = 1: the bound
    include/ffl.h documents for ffl_create)."""
    return (depth + 1) * max_batch + 2 * SMOOTH_RADIUS + 1


def min_frame_slots(max_batch, depth=1):
    """Frame slots for `depth` batches in flight + the one being staged, B + 1 frames each."""
    return (depth + 1) * (max_batch + 1)


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pairs(n_items, world, rank, assign="contiguous", block=1):
    """Sorted pair indices owned by `rank`.

    "contiguous"   one block per rank (shard_range): consecutive pairs share a frame, so a rank uploads and
                   expands ~1 frame per pair, and the +-6 window of FF:1203-1214 stays local except at the two
                   block edges.  Needs the number of pairs up front.
    "round_robin"  blocks of `block` consecutive pairs dealt to ranks in turn (block k -> rank k % world):
                   BASELINE's "frame-pairs sharded round-robin" (block = 1), usable on a stream whose length is
                   not known when the first pairs are dispatched.  A rank expands (block + 1) / block frames
                   per pair (2 at block = 1, where no two of its pairs share a frame)."""
    if assign == "contiguous":
        lo, hi = shard_range(n_items, world, rank)
        return np.arange(lo, hi)
    if assign != "round_robin" or block < 1:
        raise ValueError(f"unknown pair assignment {assign!r} / block {block}")
    idx = np.arange(n_items)
    return idx[(idx // block) % world == rank]


class PairEngine:
    """Streams one chunk of frames through a device context in batches of `max_batch` pairs.

    Consecutive pairs share a frame, so every frame is uploaded and expanded once per batch it
    appears in.  Two batches are kept in flight; pass 2 for pair j is issued once the pass-1 records
    of pairs <= j+6 are known (or the chunk has ended), after which its flow slot is recycled.
    """

    def __init__(self, ctx, upload=None, depth=None):
        """`upload(first_slot, frames)` puts a run of frames into consecutive frame slots; the default takes
        gray (or same-size BGR) operands, frontend.DecodedUploader takes frames as decoded (any size).
        `depth`: batches queued on the device before the oldest one's results are collected (default: 2 when the
        context has the slots for it -- 3B + 3 frame slots, 3B + 13 flow slots -- else 1).  With depth 2 the upload of
        batch s + 2 is already queued while batch s computes, so a slow transfer or a host hiccup does not idle the device;
        a third batch queued ahead measures within 1 % of two at 1080p and at 256x256 (profiles/r04_pcie_chunk_length.txt).
        Results do not depend on it."""
        self.ctx = ctx
        self.upload = upload or ctx.upload_frames
        self.B = ctx.max_batch
        if depth is None:
            depth = 2 if (ctx.frame_slots >= min_frame_slots(self.B, 2) and ctx.flow_slots >= min_flow_slots(self.B, 2)) else 1
        self.depth = int(depth)
        if self.depth > 1 and (ctx.frame_slots < min_frame_slots(self.B, self.depth) or ctx.flow_slots < min_flow_slots(self.B, self.depth)):
            raise ValueError(f"context too small for depth {self.depth}: need frame_slots >= {min_frame_slots(self.B, self.depth)} "
                             f"and flow_slots >= {min_flow_slots(self.B, self.depth)}")
        # frame slots: a batch's <= B+1 (stream) / 2B (arbitrary pairs) frames + the next batch's new ones;
        # flow slots: two batches in flight + the <= 6 pairs still waiting for their +-6 window (2B + 6 live
        # at most; 2B + 13 is the bound ffl.h documents and Context defaults to)
        if ctx.frame_slots < 2 * self.B + 2 or ctx.flow_slots < min_flow_slots(self.B):
            raise ValueError(f"context too small: need frame_slots >= 2B+2 = {2 * self.B + 2} and "
                             f"flow_slots >= 2B+13 = {min_flow_slots(self.B)}")

    def pass1(self, frames, pair_lo, pair_hi, pov_mode=False, cut_threshold=7.0, on_batch=None):
        """Run pass 1 for pairs [pair_lo, pair_hi) of `frames`; flows stay resident in slot
        (j - pair_lo) % flow_slots.  Returns the list of (x, y, val, mean_mag, cut)."""
        fs = self.ctx.flow_slots
        return self.pass1_pairs(frames, range(pair_lo, pair_hi), lambda l: l % fs, pov_mode, cut_threshold,
                                on_batch=(lambda ls, js, got: on_batch(js, got)) if on_batch else None)

    def pass1_pairs(self, frames, pairs, slot_of, pov_mode=False, cut_threshold=7.0, on_batch=None):
        """Pass 1 for an arbitrary ascending list of pair indices (pair j = frames[j], frames[j+1]) in batches of
        max_batch; the flow of the l-th listed pair stays resident in flow slot slot_of(l).  Frames go to the
        device once per run of batches that needs them: a ring over the frame slots, frames of the batch being
        assembled are never evicted, and runs of consecutive frames landing in consecutive slots go up with one
        H2D transfer.  (The library orders an upload into a recycled slot behind the batches that still read
        it.)  on_batch(local_indices, pair_indices, records) is called per finished batch."""
        ctx, B, S = self.ctx, self.B, self.ctx.frame_slots
        pairs = [int(j) for j in pairs]
        recs = [None] * len(pairs)
        resident, owner, state = {}, [None] * S, {"next": 0}

        def stage(ids):
            pinned = {resident[i] for i in ids if i in resident}
            placed = []
            for i in ids:
                if i in resident:
                    continue
                while state["next"] % S in pinned:
                    state["next"] += 1
                s = state["next"] % S
                state["next"] += 1
                if owner[s] is not None:
                    del resident[owner[s]]
                owner[s], resident[i] = i, s
                pinned.add(s)
                placed.append((i, s))
            k = 0
            while k < len(placed):  # runs of consecutive frames in consecutive slots: one transfer each
                n = 1
                while (k + n < len(placed) and placed[k + n][0] == placed[k + n - 1][0] + 1
                       and placed[k + n][1] == placed[k + n - 1][1] + 1):
                    n += 1
                self.upload(placed[k][1], [frames[i] for i, _ in placed[k:k + n]])
                k += n

        def enqueue(l0):
            ls = list(range(l0, min(l0 + B, len(pairs))))
            js = [pairs[l] for l in ls]
            stage(sorted({j for j in js} | {j + 1 for j in js}))
            ctx.flow_pairs([resident[j] for j in js], [resident[j + 1] for j in js], [slot_of(l) for l in ls], pov_mode)
            return ls, js

        release = getattr(frames, "release", None)  # prefetch.PrefetchRing views: frames may be recycled once consumed

        def collect(ls, js):
            got = ctx.pass1_results([slot_of(l) for l in ls], cut_threshold)  # one call per batch
            recs[ls[0]:ls[-1] + 1] = got
            if release:
                # this batch's kernels have run, so every transfer they waited for has left the host: frames up to
                # the batch's last one will not be read from host memory again (ascending pair lists only)
                release(js[-1] + 2)
            if on_batch:
                on_batch(ls, js, got)

        pending, depth = [], getattr(self, "depth", 1)
        for l0 in range(0, len(pairs), B):
            pending.append(enqueue(l0))
            if len(pending) > depth:
                collect(*pending.pop(0))
        while pending:
            collect(*pending.pop(0))
        return recs

    def process_chunk(self, frames, pov_mode=False, cut_threshold=7.0):
        """One whole chunk on one GPU: returns (dots float64[n], records) with n = len(frames)-1."""
        ctx, B = self.ctx, self.B
        n = len(frames) - 1
        if n < 1:
            return np.zeros(0), []
        dots = np.zeros(n, np.float64)
        psum = np.zeros((n + 1, 2), np.int64)  # prefix sums of pos_center: window means are exact integer
        cuts = np.zeros(n, bool)               # sums / counts, bit-identical to np.mean over the window
        state = {"known": 0, "done": 0}
        recs_all = [None] * n

        def finalize(limit):
            while state["done"] < limit:
                j0, j1 = state["done"], min(state["done"] + B, limit)
                js = np.arange(j0, j1)
                lo, hi = np.maximum(0, js - SMOOTH_RADIUS), np.minimum(n, js + SMOOTH_RADIUS + 1)
                cs = (psum[hi] - psum[lo]) / (hi - lo)[:, None]
                out = ctx.radial(list(js % ctx.flow_slots), cs, cuts[j0:j1], pov_mode)
                dots[j0:j1] = out
                state["done"] = j1

        def on_batch(js, got):
            j0 = js[0]
            recs_all[j0:js[-1] + 1] = got
            p = np.array([(r[0], r[1]) for r in got], np.int64)
            psum[j0 + 1:js[-1] + 2] = psum[j0] + np.cumsum(p, axis=0)
            cuts[j0:js[-1] + 1] = [r[4] for r in got]
            state["known"] = js[-1] + 1
            finalize(n if state["known"] == n else max(0, state["known"] - SMOOTH_RADIUS))

        self.pass1(frames, 0, n, pov_mode, cut_threshold, on_batch)
        finalize(n)
        return dots, recs_all


def frames_to_actions(engine, frames, fps, params):
    """Gray (or BGR) frames of one video -> .funscript actions: the `process_video` body from frame
    sampling to keyframes (FF:1127-1385) with the HIP pair engine in the middle.  `frames` holds every
    decoded frame (any sequence); chunking follows FF:1145-1153 (pairs never span chunks, F10)."""
    from . import postchain
    step, _, indices = postchain.sampling(fps, len(frames))
    bracket = int(params.get("batch_size", 3000.0))
    dots, cuts, frame_idx = [], [], []
    for cs in range(0, len(indices), bracket):
        chunk = indices[cs:cs + bracket]
        if len(chunk) < 2:
            continue
        d, recs = engine.process_chunk([frames[i] for i in chunk], bool(params.get("pov_mode", False)),
                                       float(params.get("cut_threshold", 7)))
        dots += [float(v) for v in d]
        cuts += [bool(r[4]) for r in recs]
        frame_idx += chunk[:-1]
    return postchain.actions_from_scalars(dots, cuts, frame_idx, fps, params)


def _shard_pass1(engine, frames, mine, pov_mode, cut_threshold):
    recs = engine.pass1(frames, mine, pov_mode, cut_threshold) if len(mine) else []
    return np.array([[j, r[0], r[1], int(r[4])] for j, r in zip(mine, recs)], np.int64).reshape(-1, 4)


def _merge_records(parts, n):
    """every rank's (pair index, x, y, cut) rows -> (n, 3) array in pair order"""
    rows = np.concatenate([np.asarray(a, np.int64).reshape(-1, 4) for a in parts], axis=0)
    assert len(rows) == n and np.array_equal(np.sort(rows[:, 0]), np.arange(n)), "pairs lost or duplicated in the shard"
    allrecs = np.empty((n, 3), np.int64)
    allrecs[rows[:, 0]] = rows[:, 1:]
    return allrecs


def _shard_pass2(engine, mine, centers, allrecs, pov_mode):
    if not len(mine):
        return np.zeros((0, 2))
    local = engine.radial(list(range(len(mine))), centers[mine], allrecs[mine, 2].astype(bool), pov_mode)
    return np.stack([np.asarray(mine, np.float64), np.asarray(local, np.float64)], axis=1)


def _merge_dots(parts, n):
    rows = np.concatenate([np.asarray(p, np.float64).reshape(-1, 2) for p in parts], axis=0)
    dots = np.empty(n, np.float64)
    dots[rows[:, 0].astype(np.int64)] = rows[:, 1]
    return dots


def process_chunk_sharded(engine, frames, rank, world, allgather, pov_mode=False, cut_threshold=7.0,
                          assign="contiguous", block=1):
    """Multi-GPU form of one chunk: rank r owns the pairs shard_pairs(n, world, r, assign, block).

    `engine` provides pass1(frames, pair_indices, pov_mode, cut_threshold) -> records (its l-th listed pair is
    local index l) and radial(local_indices, centers, cuts, pov_mode) -> floats on its own device;
    `allgather(obj)` returns the list of every rank's object (a host gather of ~32 B per pair: pair index, x, y,
    cut -- the only exchange).  Centres are smoothed over the WHOLE chunk (FF:1203-1214 needs pairs j+-6, which
    cross shard edges under either assignment).  Returns the full dots array and the (n, 3) records on every rank."""
    n = len(frames) - 1
    mine = shard_pairs(n, world, rank, assign, block)
    allrecs = _merge_records(allgather(_shard_pass1(engine, frames, mine, pov_mode, cut_threshold)), n)
    centers = smooth_centers(allrecs[:, :2])
    return _merge_dots(allgather(_shard_pass2(engine, mine, centers, allrecs, pov_mode)), n), allrecs


def halo_rows(rows, radius=SMOOTH_RADIUS):
    """The part of a contiguous shard's pass-1 rows (pair index, x, y, cut; ascending) another rank can ever need: its
    first and last `radius` pairs.  A pair within `radius` of a foreign block lies within the first / last `radius` pairs
    of its own block, however short the blocks in between are."""
    rows = np.asarray(rows, np.int64).reshape(-1, 4)
    return rows if len(rows) <= 2 * radius else np.concatenate([rows[:radius], rows[-radius:]], axis=0)


def process_chunk_sharded_halo(engine, frames, rank, world, allgather, pov_mode=False, cut_threshold=7.0):
    """Streaming multi-GPU form of one chunk for CONTIGUOUS blocks: the only exchange between the passes is the halo.

    process_chunk_sharded gathers every pass-1 record of the chunk before any pass 2 starts, so all ranks idle until the
    slowest has finished pass 1.  FF:1203-1214 only needs pairs j-6..j+6: with contiguous blocks a rank's interior pairs
    (window inside its own block, or clipped by the chunk's ends) depend on nobody else, and its <= 12 edge pairs on the
    neighbouring blocks' first / last 6 records ("block edges need a 6-pair halo of pass-1 scalars only").
    Schedule per rank:
        pass 1 over its block in batches; after every batch, pass 2 for all its pairs whose window is already known
        (runs on the device beside the next batch's pass 1, exactly as PairEngine.process_chunk does on one GPU);
        ONE all-gather of <= 12 halo rows (32 B each) per rank;
        pass 2 for the remaining edge pairs;
        ONE all-gather of the results (pair index, scalar, x, y, cut) -- the output, not a barrier between the passes.
    `engine.pass1` is called with on_batch= when it accepts one (HipShardEngine does); an engine without it still gives
    the same numbers, only without the overlap.  Window means are exact integer sums / counts, so the result is
    bit-identical to process_chunk_sharded's and to a single-GPU process_chunk.  Returns (dots, (n, 3) records) on every
    rank."""
    n = len(frames) - 1
    R = SMOOTH_RADIUS
    lo, hi = shard_range(n, world, rank)
    mine = np.arange(lo, hi)
    pos = np.zeros((n, 2), np.int64)
    cuts = np.zeros(n, bool)
    known = np.zeros(n, bool)
    dots = np.zeros(hi - lo, np.float64)
    todo = list(range(lo, hi))                     # own pairs still waiting for pass 2, ascending

    def flush(final=False):
        ready, rest = [], []
        for j in todo:
            w0, w1 = max(0, j - R), min(n, j + R + 1)
            (ready if known[w0:w1].all() else rest).append(j)
        if final and rest:
            raise RuntimeError(f"rank {rank}: halo exchange left pairs {rest[:4]} without their +-{R} window")
        todo[:] = rest
        if ready:
            js = np.asarray(ready)
            w0, w1 = np.maximum(0, js - R), np.minimum(n, js + R + 1)
            psum = np.zeros((n + 1, 2), np.int64)
            psum[1:] = np.cumsum(pos, axis=0)
            centers = (psum[w1] - psum[w0]) / (w1 - w0)[:, None]
            dots[js - lo] = engine.radial(list(js - lo), centers, cuts[js], pov_mode)

    def take(js, got):
        js = np.asarray(js, np.int64)
        pos[js] = [(r[0], r[1]) for r in got]
        cuts[js] = [bool(r[4]) for r in got]
        known[js] = True

    def on_batch(ls, js, got):
        take(js, got)
        flush()

    if len(mine):
        import inspect
        if "on_batch" in inspect.signature(engine.pass1).parameters:
            recs = engine.pass1(frames, mine, pov_mode, cut_threshold, on_batch=on_batch)
        else:                                      # an engine without the streaming hook: same numbers, no overlap
            recs = engine.pass1(frames, mine, pov_mode, cut_threshold)
        take(mine, recs)
    own = np.concatenate([mine[:, None], pos[lo:hi], cuts[lo:hi, None].astype(np.int64)], axis=1) if len(mine) else np.zeros((0, 4), np.int64)
    for part in allgather(halo_rows(own)):         # the only exchange between pass 1 and pass 2
        part = np.asarray(part, np.int64).reshape(-1, 4)
        if len(part):
            pos[part[:, 0]] = part[:, 1:3]
            cuts[part[:, 0]] = part[:, 3].astype(bool)
            known[part[:, 0]] = True
    flush(final=True)
    out = np.concatenate([own.astype(np.float64), dots[:, None]], axis=1)          # (j, x, y, cut, dot)
    rows = np.concatenate([np.asarray(p, np.float64).reshape(-1, 5) for p in allgather(out)], axis=0)
    idx = rows[:, 0].astype(np.int64)
    assert len(rows) == n and np.array_equal(np.sort(idx), np.arange(n)), "pairs lost or duplicated in the shard"
    all_dots = np.empty(n, np.float64)
    allrecs = np.empty((n, 3), np.int64)
    all_dots[idx] = rows[:, 4]
    allrecs[idx] = rows[:, 1:4].astype(np.int64)
    return all_dots, allrecs


def process_chunk_local_ranks(engines, frames, pov_mode=False, cut_threshold=7.0, assign="contiguous", block=1):
    """The same schedule with every rank driven from THIS process (one engine / context per device, or several
    contexts on one device): phase by phase, no process group.  Equivalent to world = len(engines) processes
    running process_chunk_sharded."""
    n, world = len(frames) - 1, len(engines)
    mine = [shard_pairs(n, world, r, assign, block) for r in range(world)]
    allrecs = _merge_records([_shard_pass1(e, frames, m, pov_mode, cut_threshold) for e, m in zip(engines, mine)], n)
    centers = smooth_centers(allrecs[:, :2])
    return _merge_dots([_shard_pass2(e, m, centers, allrecs, pov_mode) for e, m in zip(engines, mine)], n), allrecs


class HipShardEngine:
    """engine for process_chunk_sharded on one device: keeps the shard's flows resident (local pair l in flow
    slot l), so the context needs flow_slots >= the shard's pair count and frame_slots >= 2B + 2."""

    def __init__(self, ctx, upload=None):
        self.ctx = ctx
        B = ctx.max_batch
        if ctx.frame_slots < 2 * B + 2:
            # two frames of one batch in the same slot would silently compute the wrong flow
            raise ValueError(f"context too small for a shard engine: need frame_slots >= 2B+2 = {2 * B + 2}")
        self.inner = PairEngine.__new__(PairEngine)  # a shard's flow slots are bounded by its size, checked in pass1
        self.inner.ctx, self.inner.B, self.inner.upload = ctx, B, upload or ctx.upload_frames

    def pass1(self, frames, pair_indices, pov_mode, cut_threshold, on_batch=None):
        """on_batch(local_indices, pair_indices, records) after every finished batch (process_chunk_sharded_halo issues the
        pass 2 of pairs whose window is complete from it, beside the next batch's kernels)."""
        if len(pair_indices) > self.ctx.flow_slots:
            raise ValueError(f"shard of {len(pair_indices)} pairs does not fit the context's {self.ctx.flow_slots} flow slots")
        return self.inner.pass1_pairs(frames, pair_indices, lambda l: l, pov_mode, cut_threshold, on_batch=on_batch)

    def radial(self, local_indices, centers, cuts, pov_mode):
        out, B = [], self.ctx.max_batch
        for s in range(0, len(local_indices), B):
            sl = slice(s, s + B)
            out += self.ctx.radial(list(local_indices[sl]), list(centers[sl]), list(cuts[sl]), pov_mode)
        return out
