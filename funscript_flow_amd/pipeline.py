"""Host-side schedule around the HIP pair kernels: the counterpart of the per-chunk section of
process_video (FunscriptFlow.pyw:1187-1242) for the "HIP" backend.

    pairs = zip(frames[:-1], frames[1:])                                   FF:1188
    pass 1  (flow, |div| argmax, mean magnitude, cut) for every pair       FF:1190-1191
    c_j = mean of pos_center over pairs j-6..j+6 inside the chunk          FF:1203-1214
    pass 2  radial_motion_weighted(flow_j, c_j, cut_j, pov_mode)            FF:1232-1236

Everything runs in the calling process (HIP state must not cross fork); flows never leave HBM.
Multi-GPU: pairs are sharded in contiguous blocks; the only exchange is a host all-gather of the
16-byte pass-1 records (no RCCL collective).
"""
import numpy as np

SMOOTH_RADIUS = 6  # FF:1206 range(1, 7)


def smooth_centers(pos_centers, radius=SMOOTH_RADIUS):
    """FF:1203-1214: c_j = mean({p_i : |i-j| <= 6, 0 <= i < n}) as float64[2]."""
    p = np.asarray(pos_centers, np.int64).reshape(-1, 2)
    n = len(p)
    out = np.empty((n, 2), np.float64)
    for j in range(n):
        idx = [j]
        for i in range(1, radius + 1):
            if j - i >= 0:
                idx.append(j - i)
            if j + i < n:
                idx.append(j + i)
        out[j] = np.mean(p[idx], axis=0)
    return out


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PairEngine:
    """Streams one chunk of frames through a device context in batches of `max_batch` pairs.

    Consecutive pairs share a frame, so every frame is uploaded and expanded once per batch it
    appears in.  Two batches are kept in flight; pass 2 for pair j is issued once the pass-1 records
    of pairs <= j+6 are known (or the chunk has ended), after which its flow slot is recycled.
    """

    def __init__(self, ctx, upload=None):
        """`upload(first_slot, frames)` puts a run of frames into consecutive frame slots; the default takes
        gray (or same-size BGR) operands, frontend.DecodedUploader takes frames as decoded (any size)."""
        self.ctx = ctx
        self.upload = upload or ctx.upload_frames
        self.B = ctx.max_batch
        if ctx.frame_slots < 2 * self.B + 2 or ctx.flow_slots < 3 * self.B + 2 * SMOOTH_RADIUS + 1:
            raise ValueError("context too small: need frame_slots >= 2B+2 and flow_slots >= 3B+13")

    def pass1(self, frames, pair_lo, pair_hi, pov_mode=False, cut_threshold=7.0, on_batch=None):
        """Run pass 1 for pairs [pair_lo, pair_hi) of `frames`; flows stay resident in slot
        (j - pair_lo) % flow_slots.  Returns the list of (x, y, val, mean_mag, cut)."""
        ctx, B = self.ctx, self.B
        recs = [None] * (pair_hi - pair_lo)
        uploaded = {}

        def fslot(i):
            return i % ctx.frame_slots

        def enqueue(k):
            js = list(range(k, min(k + B, pair_hi)))
            new = [i for i in range(js[0], js[-1] + 2) if uploaded.get(fslot(i)) != i]
            while new:  # runs of consecutive frame slots go up with one H2D transfer each
                run = [new[0]]
                while len(run) < len(new) and new[len(run)] == run[-1] + 1 and fslot(new[len(run)]) == fslot(run[-1]) + 1:
                    run.append(new[len(run)])
                self.upload(fslot(run[0]), [frames[i] for i in run])
                for i in run:
                    uploaded[fslot(i)] = i
                new = new[len(run):]
            ctx.flow_pairs([fslot(j) for j in js], [fslot(j + 1) for j in js],
                           [(j - pair_lo) % ctx.flow_slots for j in js], pov_mode)
            return js

        def collect(js):
            got = ctx.pass1_results([(j - pair_lo) % ctx.flow_slots for j in js], cut_threshold)  # one call per batch
            recs[js[0] - pair_lo:js[-1] + 1 - pair_lo] = got
            if on_batch:
                on_batch(js, got)

        pending = None
        for k in range(pair_lo, pair_hi, B):
            js = enqueue(k)
            if pending:
                collect(pending)
            pending = js
        if pending:
            collect(pending)
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


def process_chunk_sharded(engine, frames, rank, world, allgather, pov_mode=False, cut_threshold=7.0):
    """Multi-GPU form of one chunk: rank r owns the contiguous pair block shard_range(n, world, r).

    `engine` provides pass1(frames, lo, hi, pov_mode, cut_threshold) -> records and
    radial(local_indices, centers, cuts, pov_mode) -> floats on its own device; `allgather(obj)`
    returns the list of every rank's object (a host gather of ~16 B per pair -- the only exchange).
    Returns the full dots array on every rank."""
    n = len(frames) - 1
    lo, hi = shard_range(n, world, rank)
    recs = engine.pass1(frames, lo, hi, pov_mode, cut_threshold) if hi > lo else []
    mine = np.array([[r[0], r[1], int(r[4])] for r in recs], np.int64).reshape(-1, 3)
    allrecs = np.concatenate([np.asarray(a, np.int64).reshape(-1, 3) for a in allgather(mine)], axis=0)
    assert len(allrecs) == n
    centers = smooth_centers(allrecs[:, :2])
    local = engine.radial(list(range(hi - lo)), centers[lo:hi], allrecs[lo:hi, 2].astype(bool), pov_mode) if hi > lo else []
    parts = allgather(np.asarray(local, np.float64))
    return np.concatenate([np.asarray(p, np.float64).reshape(-1) for p in parts]), allrecs


class HipShardEngine:
    """engine for process_chunk_sharded on one device: keeps the shard's flows resident."""

    def __init__(self, ctx, upload=None):
        self.ctx = ctx
        self.inner = PairEngine.__new__(PairEngine)  # without PairEngine's slot-count check: a shard is bounded below
        self.inner.ctx, self.inner.B, self.inner.upload = ctx, ctx.max_batch, upload or ctx.upload_frames

    def pass1(self, frames, lo, hi, pov_mode, cut_threshold):
        if hi - lo > self.ctx.flow_slots:
            raise ValueError("shard does not fit the context's flow slots")
        return self.inner.pass1(frames, lo, hi, pov_mode, cut_threshold)

    def radial(self, local_indices, centers, cuts, pov_mode):
        out, B = [], self.ctx.max_batch
        for s in range(0, len(local_indices), B):
            sl = slice(s, s + B)
            out += self.ctx.radial([i % self.ctx.flow_slots for i in local_indices[sl]], list(centers[sl]),
                                   list(cuts[sl]), pov_mode)
        return out
