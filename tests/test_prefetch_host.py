"""Host-side decode / prefetch ring (FF:103-291, FF:1051-1091, FF:1145-1185) against a fake
capture: sequential reads without seeks, delivery order, back-pressure, slot recycling only after the consuming
batch has returned, chunk boundaries (F10).  No codec and no device here: the capture, the context and the
device side of the engine are stand-ins that record what they are asked."""
import threading
import time

import numpy as np
import pytest

from funscript_flow_amd import pipeline, prefetch


class FakeCapture:
    """cv2.VideoCapture look-alike over synthetic frames: frame i is filled with the byte pattern of i."""

    def __init__(self, n_frames, fps=30.0, size=(8, 6), fail_at=None, into=True):
        self.n, self.fps, self.size, self.fail_at, self.into = n_frames, fps, size, fail_at, into
        self.pos, self.seeks, self.grabs, self.reads = 0, 0, 0, 0

    def isOpened(self):
        return True

    def get(self, prop):
        return {prefetch.CAP_PROP_FRAME_COUNT: self.n, prefetch.CAP_PROP_FPS: self.fps,
                prefetch.CAP_PROP_FRAME_WIDTH: self.size[0], prefetch.CAP_PROP_FRAME_HEIGHT: self.size[1]}[prop]

    def set(self, prop, value):
        self.seeks += 1
        self.pos = int(value)
        return True

    def grab(self):
        if self.pos >= self.n or self.pos == self.fail_at:
            return False
        self.pos += 1
        self.grabs += 1
        return True

    def read(self, image=None):
        if not self.into and image is not None:
            raise TypeError("this capture does not decode into a caller's array")
        if self.pos >= self.n or self.pos == self.fail_at:
            return False, None
        w, h = self.size
        frame = image if image is not None else np.empty((h, w, 3), np.uint8)
        frame[...] = self.pos % 251
        frame[0, 0, :] = [self.pos & 255, (self.pos >> 8) & 255, 7]
        self.pos += 1
        self.reads += 1
        return True, frame

    def release(self):
        pass


def frame_id(a):
    return int(a[0, 0, 0]) | (int(a[0, 0, 1]) << 8)


class FakeCtx:
    """Stands in for the device.  An upload only REMEMBERS the host array; its pixels are read ("the H2D transfer
    runs") at the latest moment the real one may still be reading pinned memory: when the first batch that uses the
    frame slot returns its results.  A ring that recycles a host frame before that shows up as a wrong pair."""

    def __init__(self, max_batch, frame_slots, flow_slots, delay=0.0):
        self.max_batch, self.frame_slots, self.flow_slots, self.delay = max_batch, frame_slots, flow_slots, delay
        self.slot_upload, self.pending, self.pairs_seen = {}, [], []

    def pinned_frames(self, n, channels=1, size=None):
        return np.zeros((n, size[1], size[0], 3), np.uint8)

    def upload_frames(self, first, frames):          # the engine's default uploader (gray path): same bookkeeping
        for k, f in enumerate(frames):
            self.slot_upload[first + k] = {"host": f, "device": None}      # queued, not yet transferred

    def flow_pairs(self, f0, f1, slots, pov):
        # the library orders a batch behind the uploads of its frames: fix WHICH upload each operand refers to
        self.pending.append([(self.slot_upload[a], self.slot_upload[b], s) for a, b, s in zip(f0, f1, slots)])

    def pass1_results(self, slots, thr):
        time.sleep(self.delay)
        out = []
        for (a, b, s), want in zip(self.pending.pop(0), slots):
            assert s == want
            for up in (a, b):                         # the transfers this batch waited for complete now, at the latest
                if up["device"] is None:
                    up["device"] = frame_id(up["host"])
            self.pairs_seen.append((a["device"], b["device"]))
            out.append((a["device"], b["device"], np.float32(0), np.float32(0), False))
        return out

    def radial(self, slots, centers, cuts, pov):
        return [0.0] * len(slots)


def run_ring(n_frames, fps, bracket, B, ring_frames, delay=0.0, **cap_kw):
    cap = FakeCapture(n_frames, fps, **cap_kw)
    from funscript_flow_amd import postchain
    step, _, indices = postchain.sampling(fps, n_frames)
    ctx = FakeCtx(B, 2 * B + 2, pipeline.min_flow_slots(B), delay)
    ring = prefetch.PrefetchRing(ctx, cap, indices, bracket, ring_frames)
    eng = pipeline.PairEngine(ctx)
    chunks = []
    try:
        for view, fidx in ring.chunks():
            dots, recs = eng.process_chunk(view)
            chunks.append((fidx, [(r[0], r[1]) for r in recs]))
    finally:
        ring.close()
    return cap, ring, indices, chunks


def test_sequential_decode_without_seeks_and_pairs_inside_chunks():
    # 60 fps -> every second frame (FF:1127-1129); 71 frames -> 36 sampled; chunks of 10 -> 10, 10, 10, 6
    cap, ring, indices, chunks = run_ring(71, 60.0, 10, 3, 3 * 3 + 2)
    assert indices == list(range(0, 71, 2))
    assert cap.seeks == 0 and cap.reads == 36 and cap.grabs == 35      # skipped frames are grabbed, never seeked
    assert [len(f) for f, _ in chunks] == [9, 9, 9, 5]                  # pairs never span chunks (F10)
    for c, (fidx, pairs) in enumerate(chunks):
        want = indices[c * 10:(c + 1) * 10]
        assert fidx == want[:-1]                                         # frame_indices = chunk[:-1]  (FF:1152)
        assert pairs == list(zip(want[:-1], want[1:]))                   # the device saw exactly these frames
    assert ring.max_outstanding <= ring.ring_frames


def test_single_frame_tail_chunk_is_skipped():
    cap, ring, indices, chunks = run_ring(31, 30.0, 10, 4, 14)           # 31 sampled frames: 10, 10, 10, 1
    assert [len(f) for f, _ in chunks] == [9, 9, 9]
    assert cap.reads == 31 and ring.released == 31


def test_back_pressure_and_recycling_only_after_results():
    """A slow consumer and the smallest legal ring: the decoder must stall rather than overwrite a frame whose batch
    has not returned (FakeCtx reads the uploaded arrays only at collection time, so an early overwrite would show
    up as a wrong pair), and it must never be more than ring_frames ahead."""
    B = 4
    cap, ring, indices, chunks = run_ring(90, 30.0, 45, B, 3 * B + 2, delay=0.003, into=False)
    got = [p for _, pairs in chunks for p in pairs]
    assert got == [(i, i + 1) for i in range(0, 44)] + [(i, i + 1) for i in range(45, 89)]
    assert ring.max_outstanding <= 3 * B + 2
    assert cap.reads == 90


def test_ring_too_small_is_an_error_not_a_deadlock():
    cap = FakeCapture(40)
    ctx = FakeCtx(4, 10, pipeline.min_flow_slots(4))
    ring = prefetch.PrefetchRing(ctx, cap, list(range(40)), 40, ring_frames=6)       # < 3B + 1
    try:
        view, _ = next(ring.chunks())
        with pytest.raises(prefetch.DecodeError, match="too small"):
            pipeline.PairEngine(ctx).process_chunk(view)
    finally:
        ring.close()


def test_decoder_failure_reaches_the_consumer():
    cap = FakeCapture(40, fail_at=17)
    ctx = FakeCtx(3, 8, pipeline.min_flow_slots(3))
    ring = prefetch.PrefetchRing(ctx, cap, list(range(40)), 40, ring_frames=12)
    try:
        view, _ = next(ring.chunks())
        with pytest.raises(prefetch.DecodeError):
            pipeline.PairEngine(ctx).process_chunk(view)
    finally:
        ring.close()
    assert not ring.thread.is_alive()


def test_released_frames_cannot_be_requested_again():
    cap = FakeCapture(20)
    ctx = FakeCtx(2, 6, pipeline.min_flow_slots(2))
    ring = prefetch.PrefetchRing(ctx, cap, list(range(20)), 20, ring_frames=8)
    try:
        view, _ = next(ring.chunks())
        assert frame_id(view[0]) == 0 and frame_id(view[5]) == 5 and len(view) == 20
        view.release(4)
        with pytest.raises(prefetch.DecodeError, match="released"):
            view[3]
        assert frame_id(view[4]) == 4
        with pytest.raises(IndexError):
            view[20]
    finally:
        ring.close()


def test_process_video_outcomes_without_a_device(tmp_path):
    """The reference's process_video contract (FF:1094-1404) for the outcomes that need no device: existing output is
    skipped unless overwrite, a capture that does not open or has no frames is an error that is LOGGED and RETURNED,
    never raised; a failure further down (here: no device context) likewise."""
    logs = []
    video = str(tmp_path / "clip.mp4")
    out = str(tmp_path / "clip.funscript")
    open(out, "w").write("{}")
    assert prefetch.process_video(video, {"overwrite": False}, logs.append, lambda p: FakeCapture(10), lambda c: None) is False
    assert logs[-1].startswith("Skipping: output file exists") and open(out).read() == "{}"

    class Closed(FakeCapture):
        def isOpened(self):
            return False

    logs.clear()
    assert prefetch.process_video(video, {"overwrite": True}, logs.append, lambda p: Closed(10), lambda c: None) is True
    assert any(m.startswith("ERROR: Unable to open video") for m in logs)
    logs.clear()
    assert prefetch.process_video(video, {"overwrite": True}, logs.append, lambda p: FakeCapture(0), lambda c: None) is True
    assert any(m.startswith("ERROR: Unable to read video properties") for m in logs)

    def no_device(cap):
        raise RuntimeError("no HIP device")

    logs.clear()
    assert prefetch.process_video(video, {"overwrite": True}, logs.append, lambda p: FakeCapture(20), no_device) is True
    assert any(m == "ERROR: no HIP device" for m in logs) and logs[-1].startswith("Processing time:")
    assert open(out).read() == "{}"                     # nothing was written over the old file
