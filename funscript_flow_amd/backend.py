"""The "HIP" backend behind the reference's backend switch (FunscriptFlow.pyw:854-873).

Same names, argument meaning and result shape as the reference's pair functions:

    precompute_flow_info(p0, p1, config)   FF:843-907   -> dict(flow, pos_center, neg_center, val_pos,
                                                           val_neg, cut, cut_center, mean_mag)
    precompute_wrapper(p, params)          FF:1019-1021
    radial_motion_weighted(flow, center, is_cut, pov_mode=False)   FF:761-785  -> float
    precompute_all(pairs, params)          the whole `pool.starmap(precompute_wrapper, ...)` of FF:1190-1191
    radial_all(precomputed, centers, pov_mode)   the whole ProcessPoolExecutor loop of FF:1232-1236
    get_available_backends()               FF:32-63  (reports "HIP" when a device is usable)

Differences, all deliberate:
  * "flow" is a DeviceFlow handle: the field stays in HBM for pass 2; np.asarray(handle) downloads it.
  * runs in the calling process -- never hand these functions to multiprocessing.Pool (HIP state
    must not cross fork); use pipeline.PairEngine for whole chunks.
  * pov_mode is honoured (the reference's CUDA variant ignores it, FF:995) and failures raise
    instead of silently falling back to the CPU path (FF:856-873).
"""
import numpy as np

from . import _capi

_contexts = {}
_next_slot = {}
RING = 32  # one-pair-at-a-time API: flow slots are recycled round-robin after this many pairs


class DeviceFlow:
    """Handle to a flow field resident on the device; behaves like the reference's ndarray on demand."""

    def __init__(self, ctx, slot, serial):
        self.ctx, self.slot, self.serial = ctx, slot, serial
        self.shape = (ctx.height, ctx.width, 2)
        self.dtype = np.dtype(np.float32)

    def _check(self):
        if self.ctx._slot_serial.get(self.slot) != self.serial:
            raise _capi.FFLError("DeviceFlow handle is stale: its device slot has been reused "
                                 f"(only the last {RING} pairs stay resident in the one-pair API)")

    def numpy(self):
        self._check()
        return self.ctx.download_flow(self.slot)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


def _context(width, height, device=0):
    key = (width, height, device)
    if key not in _contexts:
        ctx = _capi.Context(width, height, device=device, frame_slots=2 * RING, flow_slots=RING + 1, max_batch=1)
        ctx._slot_serial = {}
        _contexts[key] = ctx
        _next_slot[key] = 0
    return _contexts[key]


def release_contexts():
    for c in list(_contexts.values()) + list(_chunk_ctx.values()):
        c.close()
    _contexts.clear()
    _next_slot.clear()
    _chunk_ctx.clear()


def get_available_backends():
    """FF:32-63 counterpart for this backend only."""
    try:
        return ["HIP"] if _capi.device_count() > 0 else []
    except _capi.FFLError:
        return []


def precompute_flow_info(p0, p1, config):
    """FF:843-907 for config["backend"] == "HIP".  p0/p1: uint8 (H,W) gray (what the reference feeds
    Farneback, FF:1082) or (H,W,3) BGR as cv2.VideoCapture.read returns."""
    backend = config.get("backend", "CPU")
    if backend != "HIP":
        raise ValueError(f"funscript_flow_amd implements backend 'HIP' only, got {backend!r}")
    cut_threshold = config.get("cut_threshold", 7)
    h, w = p0.shape[:2]
    if p1.shape != p0.shape:
        raise ValueError("frame shapes differ")
    key = (w, h, int(config.get("device", 0)))
    ctx = _context(*key)
    slot = _next_slot[key] % RING
    _next_slot[key] += 1
    serial = _next_slot[key]
    ctx._slot_serial[slot] = serial
    ctx.submit_pair(slot, p0, p1, bool(config.get("pov_mode")))
    x, y, val, mean_mag, cut = ctx.pass1_result(slot, float(cut_threshold))
    if config.get("pov_mode"):
        pos_center, val_pos = (np.int64(x), np.int64(y)), 0          # FF:880-882
    else:
        pos_center, val_pos = (np.int64(x), np.int64(y)), val       # FF:884-886
    return {
        "flow": DeviceFlow(ctx, slot, serial),
        "pos_center": pos_center,
        "neg_center": pos_center,
        "val_pos": val_pos,
        "val_neg": val_pos,
        "cut": cut,
        "cut_center": pos_center[0],
        "mean_mag": mean_mag,
    }


def precompute_wrapper(p, params):
    """FF:1019-1021."""
    return precompute_flow_info(p[0], p[1], params)


def radial_motion_weighted(flow, center, is_cut, pov_mode=False, device=0):
    """FF:761-785.  `flow` is a DeviceFlow (stays on the device) or an (H,W,2) float32 ndarray
    (uploaded to a scratch slot first)."""
    if is_cut:
        return 0.0
    if isinstance(flow, DeviceFlow):
        flow._check()
        return np.float64(flow.ctx.radial([flow.slot], [center], [False], pov_mode)[0])
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    ctx = _context(w, h, device)
    ctx.upload_flow(RING, flow, pov_mode)
    return np.float64(ctx.radial([RING], [center], [False], pov_mode)[0])


# ---- the reference's two pool calls as two batched calls ------------------------------------------------
_chunk_ctx = {}


def _fit_chunk(width, height, n_pairs, device, max_batch, reclaim=0):
    """Largest batch size <= max_batch whose context (the chunk's flows resident + the lanes' work buffers, which
    scale with the batch) fits the device's FREE memory (hipMemGetInfo through the C ABI; `reclaim` = bytes a context
    about to be closed gives back).  Raises with the numbers when even B = 1 does not fit.

    ffl_estimate_bytes counts what ffl_create allocates.  Not in it, and covered by the 5 % margin plus a fixed reserve:
    the raw-frame ring of ffl_upload_frames_raw (4 device buffers of one decoded source frame, grown on first use: 100 MB
    for 4K sources), captured hipGraphs (KBs per batch shape) and the HIP runtime's own pools.  The page-locked bytes
    (`pinned`: staging areas + result records) are host memory and do not count against the device."""
    free, total = _capi.device_mem_info(device)
    reserve = max(256 << 20, 4 * 3 * 4 * width * height)    # raw ring for sources up to 2x the context's size per axis
    budget = (free + reclaim) * 0.95 - reserve
    B = max_batch
    while True:
        need, pinned = _capi.estimate_bytes(width, height, 3 * (B + 1), max(n_pairs, 1), B)
        if need <= budget:
            return B, need
        if B == 1:
            flows = 8.0 * width * height * n_pairs
            raise _capi.FFLError(
                f"a chunk of {n_pairs} pairs at {width}x{height} needs {need / 1e9:.1f} GB on the device ({flows / 1e9:.1f} GB of "
                f"resident flow fields), {budget / 1e9:.1f} GB are free: lower batch_size (FF:2647) or use "
                "pipeline.PairEngine, which recycles flow slots (2B + 13 resident fields)")
        B = max(1, B // 2)


def _chunk_context(width, height, n_pairs, device, max_batch):
    """A context whose flow slots hold a whole chunk (the reference keeps `precomputed`, flows included, for the
    chunk: FF:1191-1236) -- grown on demand, reused across chunks, sized against the device's free memory."""
    key = (width, height, device)
    ctx = _chunk_ctx.get(key)
    if ctx is None or ctx.flow_slots < n_pairs or ctx._asked_batch != max_batch:
        serial = getattr(ctx, "_chunk_serial", 0)
        reclaim = _capi.estimate_bytes(width, height, ctx.frame_slots, ctx.flow_slots, ctx.max_batch)[0] if ctx is not None else 0
        B, _ = _fit_chunk(width, height, n_pairs, device, max_batch, reclaim)   # raises before the old context is given up
        if ctx is not None:
            ctx.close()
            _chunk_ctx.pop(key, None)
        # frame slots: two batches queued ahead of the one being collected (pipeline.min_frame_slots(B, 2))
        ctx = _capi.Context(width, height, device=device, frame_slots=3 * (B + 1), flow_slots=max(n_pairs, 1), max_batch=B)
        ctx._chunk_serial, ctx._asked_batch = serial, max_batch
        _chunk_ctx[key] = ctx
    ctx._chunk_serial += 1
    return ctx


class _ChunkFlow(DeviceFlow):
    def _check(self):
        if getattr(self.ctx, "_h", None) is None or self.ctx._chunk_serial != self.serial:
            raise _capi.FFLError("DeviceFlow handle is stale: a later precompute_all() call reused the chunk's flow slots")


def default_batch(width, height):
    """Pairs per device batch when params has no "hip_batch": the batch that fills the device at every pyramid level --
    32 at 1920x1080 and above, growing as the frame shrinks, 256 (the API's limit) at 640x360 and below, i.e. at the
    reference's own 256x256 operating point (FF:1057), where a 32-pair batch runs at 0.7x the rate of a 256-pair one."""
    return int(min(_capi.FFL_MAX_BATCH, max(32, (32 * 1920 * 1080) // max(1, width * height))))


def precompute_all(pairs, params):
    """Drop-in for `pool.starmap(precompute_wrapper, [(p, params) for p in pairs])` (FF:1190-1191): the same list of
    result dicts (FF:898-907), computed in batches of params.get("hip_batch", default_batch(w, h)) pairs with consecutive pairs
    sharing their frame (pairs = zip(frames[:-1], frames[1:]), FF:1188, is recognised by object identity); every
    flow field stays resident until the next precompute_all() call, like the reference's `precomputed`."""
    from . import pipeline
    if params.get("backend", "HIP") != "HIP":
        raise ValueError("funscript_flow_amd implements backend 'HIP' only")
    pairs = list(pairs)
    if not pairs:
        return []
    h, w = pairs[0][0].shape[:2]
    B = max(1, min(int(params.get("hip_batch", default_batch(w, h))), _capi.FFL_MAX_BATCH))
    ctx = _chunk_context(w, h, len(pairs), int(params.get("device", 0)), B)
    B = ctx.max_batch                       # possibly smaller than asked for: what fits beside the chunk's flows
    # frames of the chunk in order of first use; a pair's two operands become frame indices
    frames, index = [], {}
    idx = []
    for p0, p1 in pairs:
        ij = []
        for f in (p0, p1):
            k = id(f)
            if k not in index:
                index[k] = len(frames)
                frames.append(f)
            ij.append(index[k])
        idx.append(tuple(ij))
    stream = all(b == a + 1 for a, b in idx) and all(idx[j + 1][0] == idx[j][1] for j in range(len(idx) - 1))
    pov, thr = bool(params.get("pov_mode")), float(params.get("cut_threshold", 7))
    eng = pipeline.PairEngine.__new__(pipeline.PairEngine)   # slots are sized for the chunk here, not for a stream
    eng.ctx, eng.B, eng.upload = ctx, B, ctx.upload_frames
    eng.depth = 2 if ctx.frame_slots >= pipeline.min_frame_slots(B, 2) else 1
    out = [None] * len(pairs)
    serial = ctx._chunk_serial

    def on_batch(ls, js, got):   # the dicts of a finished batch are built while the device runs the next ones
        for l, (x, y, val, mean_mag, cut) in zip(ls, got):
            pos_center = (np.int64(x), np.int64(y))
            val_pos = 0 if pov else val                                   # FF:880-886
            out[l] = {"flow": _ChunkFlow(ctx, l, serial), "pos_center": pos_center, "neg_center": pos_center,
                      "val_pos": val_pos, "val_neg": val_pos, "cut": cut, "cut_center": pos_center[0], "mean_mag": mean_mag}

    if stream:
        eng.pass1_pairs(frames, range(len(pairs)), lambda l: l, pov, thr, on_batch=on_batch)
    else:  # arbitrary pairs: every pair brings its own two frames
        flat = [f for p in pairs for f in p]
        eng.pass1_pairs(flat, range(0, 2 * len(pairs), 2), lambda l: l, pov, thr, on_batch=on_batch)
    return out


def radial_all(precomputed, centers, pov_mode=False):
    """Drop-in for the ProcessPoolExecutor loop of FF:1232-1236: radial_motion_weighted(info["flow"], centers[j],
    info["cut"], pov_mode) for every j, in batched device calls.  `precomputed` comes from precompute_all()."""
    out = [0.0] * len(precomputed)
    todo = [j for j, info in enumerate(precomputed) if not info["cut"]]
    if not todo:
        return out
    ctx = precomputed[todo[0]]["flow"].ctx
    for info in precomputed:
        info["flow"]._check()
    for s0 in range(0, len(todo), _capi.FFL_MAX_BATCH):
        js = todo[s0:s0 + _capi.FFL_MAX_BATCH]
        vals = ctx.radial([precomputed[j]["flow"].slot for j in js], [centers[j] for j in js], [False] * len(js), pov_mode)
        for j, v in zip(js, vals):
            out[j] = np.float64(v)
    return out
