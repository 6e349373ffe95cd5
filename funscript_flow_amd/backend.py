"""The "HIP" backend behind the reference's backend switch (FunscriptFlow.pyw:854-873).

Same names, argument meaning and result shape as the reference's pair functions:

    precompute_flow_info(p0, p1, config)   FF:843-907   -> dict(flow, pos_center, neg_center, val_pos,
                                                           val_neg, cut, cut_center, mean_mag)
    precompute_wrapper(p, params)          FF:1019-1021
    radial_motion_weighted(flow, center, is_cut, pov_mode=False)   FF:761-785  -> float
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
    for c in _contexts.values():
        c.close()
    _contexts.clear()
    _next_slot.clear()


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
