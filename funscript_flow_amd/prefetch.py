"""Host-side decode / prefetch path of the HIP backend: the counterpart of

    AsyncVideoReader            FF:103-291   (<= 4 cv2.VideoCapture decoders, one SEEK per frame FF:173-189)
    fetch_frames_optimized      FF:1051-1091 (sub-batches of 50 frames, host resize + gray conversion)
    the chunk prefetch thread   FF:1156-1185 (one whole chunk of gray frames ahead, Queue(maxsize=1))

re-designed for a consumer that is ~100x faster than the reference's pool:

  * ONE capture, read strictly sequentially (grab() for the frames the 30-fps sampling skips, FF:1127-1129): no
    per-frame seeks, no decoder pool to keep in step;
  * decoded frames land directly in a ring of page-locked frames owned by the device context
    (Context.pinned_frames -> ffl_host_alloc): the H2D transfer reads them in place, resize / crop / gray happen
    on the device (frontend.DecodedUploader -> k_frontend), so the host never touches a pixel after the decoder;
  * back-pressure instead of whole-chunk buffering: the decoder thread blocks when the ring is full, the consumer
    releases a slot once the batch that uploaded it has returned its results -- memory is ring_frames decoded
    frames, not a chunk of 3000 (the reference's queue can also hand a stale chunk to a later iteration when a
    prefetch thread outlives its chunk, FF:1156-1185; there is no such queue here);
  * chunking as the reference does it: chunk c = indices[c*bracket : (c+1)*bracket], pairs never span chunks
    (FF:1145-1153, F10), chunks shorter than 2 frames are skipped.

Codec work itself stays with whatever `capture` is (cv2.VideoCapture in production); this module only needs its
read() / grab() / get() / release() methods, which is what the tests' fake capture provides.
"""
import threading

import numpy as np

from . import postchain

CAP_PROP_FRAME_COUNT, CAP_PROP_FPS, CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT = 7, 5, 3, 4  # cv2.CAP_PROP_*


class DecodeError(RuntimeError):
    pass


class SequentialDecoder:
    """Decodes the frames `indices` (ascending) of `capture` in order, without seeking: frames between two wanted
    indices are skipped with grab() (demux + decode, no colour conversion / copy)."""

    def __init__(self, capture, indices):
        self.cap, self.indices = capture, list(indices)
        if any(b <= a for a, b in zip(self.indices, self.indices[1:])):
            raise ValueError("frame indices must be strictly ascending")
        self.pos = 0      # index of the frame the next read() / grab() returns
        self.k = 0        # next entry of `indices`

    def __len__(self):
        return len(self.indices)

    def read_into(self, dst):
        """Decode the next wanted frame into `dst` ((h, w, 3) uint8); returns its frame index, or None at the end."""
        if self.k >= len(self.indices):
            return None
        want = self.indices[self.k]
        while self.pos < want:
            if not self.cap.grab():
                raise DecodeError(f"capture ended at frame {self.pos}, frame {want} wanted")
            self.pos += 1
        try:
            ok, frame = self.cap.read(dst)            # cv2 decodes straight into a matching array
        except TypeError:
            ok, frame = self.cap.read()
        if not ok or frame is None:
            raise DecodeError(f"capture could not read frame {want}")
        if frame is not dst:
            if frame.shape != dst.shape:
                raise DecodeError(f"frame {want} is {frame.shape}, ring slots are {dst.shape}")
            np.copyto(dst, frame)
        self.pos += 1
        self.k += 1
        return want


class _ChunkView:
    """What PairEngine sees of one chunk: a sequence whose items appear as the decoder delivers them.  Item i of the
    chunk is frame `first + i` of the stream; release(n) tells the ring that items < n have left the host."""

    def __init__(self, ring, first, length):
        self.ring, self.first, self.length = ring, first, length

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if isinstance(i, slice):
            raise TypeError("chunk views are indexed frame by frame")
        if i < 0:
            i += self.length
        if not 0 <= i < self.length:
            raise IndexError(i)
        return self.ring._wait_frame(self.first + i)

    def release(self, n_items):
        self.ring._release(self.first + min(n_items, self.length))


class PrefetchRing:
    """A decoder thread filling a ring of page-locked decoded frames, consumed chunk by chunk.

        ring = PrefetchRing(ctx, capture, indices, bracket, ring_frames=4 * B + 2)
        for view, frame_indices in ring.chunks():
            dots, recs = engine.process_chunk(view, ...)      # engine built with frontend.DecodedUploader(ctx)
        ring.close()

    `ctx` provides pinned_frames(n, channels=3, size=(w, h)) (a real Context, or anything with that method).
    ring_frames must cover what the engine keeps in flight: two batches + the one being staged = 3 * B + 1."""

    def __init__(self, ctx, capture, indices, bracket, ring_frames, frame_size=None):
        self.decoder = SequentialDecoder(capture, indices)
        if frame_size is None:
            frame_size = (int(capture.get(CAP_PROP_FRAME_WIDTH)), int(capture.get(CAP_PROP_FRAME_HEIGHT)))
        self.bracket, self.n = int(bracket), len(self.decoder)
        if self.bracket < 2:
            raise ValueError("bracket (frames per chunk) must be >= 2")
        self.ring_frames = int(ring_frames)
        self.slots = ctx.pinned_frames(self.ring_frames, channels=3, size=frame_size)
        self.cv = threading.Condition()
        self.decoded = 0       # stream positions [0, decoded) have been decoded (position = rank in `indices`)
        self.released = 0      # positions [0, released) may be overwritten
        self.error = None
        self.stop = False
        self.max_outstanding = 0
        self.thread = threading.Thread(target=self._run, name="ffl-decode", daemon=True)
        self.thread.start()

    # ---- decoder thread ------------------------------------------------------------------------------------
    def _run(self):
        try:
            while True:
                with self.cv:
                    while not self.stop and self.decoded - self.released >= self.ring_frames:
                        self.cv.wait()                      # back-pressure: every slot holds an unreleased frame
                    if self.stop or self.decoded >= self.n:
                        return
                    pos = self.decoded
                if self.decoder.read_into(self.slots[pos % self.ring_frames]) is None:
                    return
                with self.cv:
                    self.decoded = pos + 1
                    self.max_outstanding = max(self.max_outstanding, self.decoded - self.released)
                    self.cv.notify_all()
        except Exception as e:  # noqa: BLE001 -- handed to the consumer, which re-raises it
            with self.cv:
                self.error = e
                self.cv.notify_all()

    # ---- consumer side -------------------------------------------------------------------------------------
    def _wait_frame(self, pos):
        with self.cv:
            if pos < self.released:
                raise DecodeError(f"frame at stream position {pos} was already released (ring of {self.ring_frames})")
            if pos >= self.released + self.ring_frames:
                raise DecodeError(f"stream position {pos} is {pos - self.released} frames ahead of the oldest unreleased "
                                  f"one: the ring of {self.ring_frames} frames is too small for this consumer")
            while self.decoded <= pos and self.error is None:
                self.cv.wait()
            if self.decoded <= pos:
                raise DecodeError(f"decoder failed before stream position {pos}") from self.error
        return self.slots[pos % self.ring_frames]

    def _release(self, upto):
        with self.cv:
            if upto > self.released:
                self.released = upto
                self.cv.notify_all()

    def chunks(self):
        """Yields (chunk view, frame indices of the chunk's pairs = chunk[:-1]) per chunk with >= 2 frames, in stream
        order; a chunk's slots are released when the next one is requested."""
        idx = self.decoder.indices
        for s in range(0, self.n, self.bracket):
            length = min(self.bracket, self.n - s)
            if length < 2:
                self._release(s + length)
                continue
            yield _ChunkView(self, s, length), idx[s:s + length - 1]
            self._release(s + length)

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()
        self.thread.join()


def video_to_actions(ctx, capture, params, engine=None, ring_frames=None):
    """process_video's body from frame sampling to the action list (FF:1119-1385) on the HIP path, decoding
    included: sequential reads into the pinned ring, device front-end, two-pass pair engine, host post-chain."""
    from . import frontend, pipeline
    total, fps = int(capture.get(CAP_PROP_FRAME_COUNT)), float(capture.get(CAP_PROP_FPS))
    _, _, indices = postchain.sampling(fps, total)
    bracket = int(params.get("batch_size", 3000.0))
    engine = engine or pipeline.PairEngine(ctx, frontend.DecodedUploader(ctx, bool(params.get("vr_mode")), False))
    ring = PrefetchRing(ctx, capture, indices, bracket, ring_frames or 4 * ctx.max_batch + 2)
    dots, cuts, frame_idx = [], [], []
    try:
        for view, fidx in ring.chunks():
            d, recs = engine.process_chunk(view, bool(params.get("pov_mode", False)), float(params.get("cut_threshold", 7)))
            dots += [float(v) for v in d]
            cuts += [bool(r[4]) for r in recs]
            frame_idx += fidx
    finally:
        ring.close()
    return postchain.actions_from_scalars(dots, cuts, frame_idx, fps, params)


def process_video(video_path, params, log_func, open_capture, make_context, progress_callback=None, cancel_flag=None):
    """The call shape and outcomes of the reference's process_video (FF:1094-1404) on the HIP path: returns
    `error_occurred`, writes `<video>.funscript` next to the video, never lets an exception escape (errors are logged as
    "ERROR: ..." and reported through the return value, FF:1115-1117, FF:1396-1398), skips existing outputs unless
    params["overwrite"] (FF:1107-1109).  The two things the reference hard-wires are arguments here:
        open_capture(video_path) -> a cv2.VideoCapture-like object (FF:1114 opens one through VideoReaderCV)
        make_context(capture)    -> the device Context for the operand size (256x256 in the reference, FF:1057)
    cancel_flag() is polled per chunk like FF:1146."""
    import os
    import time
    start = time.time()
    base, _ = os.path.splitext(video_path)
    output_path = base + ".funscript"
    if os.path.exists(output_path) and not params.get("overwrite", False):
        log_func(f"Skipping: output file exists ({output_path})")
        return False
    try:
        log_func(f"Processing video: {video_path}")
        cap = open_capture(video_path)
        if hasattr(cap, "isOpened") and not cap.isOpened():
            raise DecodeError("capture did not open")
    except Exception as e:  # noqa: BLE001
        log_func(f"ERROR: Unable to open video at {video_path}: {e}")
        return True
    try:
        total, fps = int(cap.get(CAP_PROP_FRAME_COUNT)), float(cap.get(CAP_PROP_FPS))
        if total < 1 or fps <= 0:
            raise DecodeError(f"{total} frames at {fps} fps")
    except Exception as e:  # noqa: BLE001
        log_func(f"ERROR: Unable to read video properties: {e}")
        return True
    from . import frontend, pipeline
    step, effective_fps, indices = postchain.sampling(fps, total)
    log_func(f"FPS: {fps:.2f}; downsampled to ~{effective_fps:.2f} fps; {len(indices)} frames selected.")
    log_func("Using backend: HIP")
    error_occurred = False
    ctx = ring = None
    try:
        ctx = make_context(cap)
        engine = pipeline.PairEngine(ctx, frontend.DecodedUploader(ctx, bool(params.get("vr_mode")), False))
        ring = PrefetchRing(ctx, cap, indices, int(params.get("batch_size", 3000.0)), 4 * ctx.max_batch + 2)
        dots, cuts, frame_idx, done = [], [], [], 0
        for view, fidx in ring.chunks():
            if cancel_flag and cancel_flag():
                log_func("User bailed.")
                return error_occurred
            d, recs = engine.process_chunk(view, bool(params.get("pov_mode", False)), float(params.get("cut_threshold", 7)))
            dots += [float(v) for v in d]
            cuts += [bool(r[4]) for r in recs]
            frame_idx += fidx
            done += len(view)
            if progress_callback:
                progress_callback(min(100, int(100 * done / max(len(indices), 1))))
        actions = postchain.actions_from_scalars(dots, cuts, frame_idx, fps, params)
        log_func(f"Keyframe reduction: {len(actions)} actions computed.")
        postchain.write_funscript(output_path, actions)
        log_func(f"Funscript saved: {output_path}")
    except Exception as e:  # noqa: BLE001
        log_func(f"ERROR: {e}")
        error_occurred = True
    finally:
        if ring is not None:
            ring.close()
        if hasattr(cap, "release"):
            cap.release()
    log_func(f"Processing time: {time.time() - start:.2f} seconds")
    return error_occurred
