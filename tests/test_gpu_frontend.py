"""GPU parity of the input front-end (k_frontend behind ffl_upload_frames_raw) against the CPU oracle's
step-by-step restatement of FF:182-186 / FF:1076-1082.  Integer work: bit-exact.  (The oracle itself is
parity-unpinned against cv2 -- see oracle/frontend_oracle.c.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as orc
from funscript_flow_amd import _capi, frontend, pipeline


def rnd(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("sw,sh,vr", [(1920, 1080, False), (3840, 1920, True), (640, 360, False), (517, 333, False),
                                      (517, 333, True), (120, 100, False), (512, 512, False), (1024, 1024, True),
                                      (256, 256, False), (512, 512, True), (3840, 2160, False)])
def test_frontend_bit_exact_reference_sizes(sw, sh, vr):
    f = [rnd(sh, sw, 7 + i) for i in range(2)]
    with _capi.Context(256, 256, max_batch=1) as ctx:
        frontend.upload_decoded(ctx, 0, f, vr_mode=vr)
        for i in range(2):
            assert np.array_equal(ctx.download_frame(i), orc.frontend(f[i], vr_mode=vr))


def test_frontend_other_context_sizes_strides_and_channel_order():
    big = rnd(800, 1400, 3)
    view = big[40:760, 60:1340]                      # 1280x720 window of a larger buffer: row pitch != 3 * width
    with _capi.Context(320, 180, max_batch=1, frame_slots=6) as ctx:
        frontend.upload_decoded(ctx, 0, [view])
        frontend.upload_decoded(ctx, 1, [np.ascontiguousarray(view[:, :, ::-1])], rgb_order=True)
        frontend.upload_decoded(ctx, 2, [view], vr_mode=True)
        ctx.upload_frames_raw(3, [view], (400, 300), (37, 51))      # arbitrary resize + crop window
        want = orc.frontend(view, size=(320, 180))
        assert np.array_equal(ctx.download_frame(0), want)
        assert np.array_equal(ctx.download_frame(1), want)
        assert np.array_equal(ctx.download_frame(2), orc.frontend(view, vr_mode=True, size=(320, 180)))
        r = orc.resize_linear_u8c3(orc.swap_rb(view), 400, 300)
        assert np.array_equal(ctx.download_frame(3), orc.rgb2gray(r[51:51 + 180, 37:37 + 320]))


def test_frontend_ring_reuse_and_growing_sources():
    """More frames than ring buffers, and a later, larger source: every slot still holds its own frame."""
    with _capi.Context(256, 256, max_batch=1, frame_slots=12) as ctx:
        small = [rnd(90, 160, 20 + i) for i in range(7)]
        frontend.upload_decoded(ctx, 0, small)
        large = [rnd(720, 1280, 40 + i) for i in range(5)]
        frontend.upload_decoded(ctx, 7, large)
        for i, f in enumerate(small + large):
            assert np.array_equal(ctx.download_frame(i), orc.frontend(f))


def test_chunk_from_decoded_frames_equals_chunk_from_gray_operands():
    """PairEngine fed decoded frames through k_frontend == PairEngine fed the oracle's gray operands."""
    from funscript_flow_amd.synth import sine_translate_frames
    g = sine_translate_frames(12, 640, 360, seed=9, amp=(5.0, 3.0), period=7)
    dec = [np.ascontiguousarray(np.stack([f, (f.astype(int) * 3 // 4).astype(np.uint8), 255 - f], -1)) for f in g]
    for vr in (False, True):
        with _capi.Context(256, 256, max_batch=4, frame_slots=10, flow_slots=25) as ctx:
            d_raw, r_raw = pipeline.PairEngine(ctx, frontend.DecodedUploader(ctx, vr_mode=vr)).process_chunk(dec)
        with _capi.Context(256, 256, max_batch=4, frame_slots=10, flow_slots=25) as ctx:
            d_ref, r_ref = pipeline.PairEngine(ctx).process_chunk([orc.frontend(f, vr_mode=vr) for f in dec])
        assert np.array_equal(d_raw, d_ref) and [tuple(r) for r in r_raw] == [tuple(r) for r in r_ref]


def test_frontend_errors_are_loud():
    f = rnd(100, 200, 1)
    with _capi.Context(256, 256, max_batch=1) as ctx:
        with pytest.raises(_capi.FFLError):
            ctx.upload_frames_raw(0, [f], (200, 300))                 # crop window wider than the resized frame
        with pytest.raises(_capi.FFLError):
            ctx.upload_frames_raw(0, [f], (256, 256), (1, 0))
        with pytest.raises(_capi.FFLError):
            ctx.upload_frames_raw(5, [f], (256, 256))                 # slot out of range
        with pytest.raises(_capi.FFLError):
            ctx.upload_frames_raw(0, [f[:, :, 0]], (256, 256))        # not 3-channel
        with pytest.raises(_capi.FFLError):
            ctx.download_frame(1)                                      # never uploaded


def test_video_to_actions_through_the_prefetch_ring():
    """SURVEY 8(f) rank 4 end to end on the device: a (fake) capture is read sequentially into the context's
    page-locked ring, frames go up in place (zero-copy path of ffl_upload_frames_raw), resize + gray on the device,
    two-pass engine, post-chain -- and gives exactly the actions of the same engine fed a plain list of the decoded
    frames.  60 fps source (every second frame is skipped with grab()), 3 chunks incl. a ragged one."""
    from funscript_flow_amd import frontend, pipeline, prefetch
    from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames
    sw, sh, n, fps = 320, 240, 101, 60.0
    src = gray_to_bgr(sine_translate_frames(n, sw, sh, seed=6, amp=(3.0, 2.0), period=24, zoom=0.03), gains=(0.9, 1.0, 0.8))

    class Cap:
        def __init__(self):
            self.pos, self.seeks = 0, 0

        def get(self, prop):
            return {prefetch.CAP_PROP_FRAME_COUNT: n, prefetch.CAP_PROP_FPS: fps, prefetch.CAP_PROP_FRAME_WIDTH: sw,
                    prefetch.CAP_PROP_FRAME_HEIGHT: sh}[prop]

        def set(self, *a):
            self.seeks += 1
            return True

        def grab(self):
            self.pos += 1
            return self.pos <= n

        def read(self, image=None):
            if self.pos >= n:
                return False, None
            np.copyto(image, src[self.pos])
            self.pos += 1
            return True, image

    params = {"detrend_window": 1.0, "norm_window": 1.0, "batch_size": 20, "keyframe_reduction": False, "pov_mode": False}
    cap = Cap()
    with _capi.Context(128, 96, max_batch=4, frame_slots=10, flow_slots=pipeline.min_flow_slots(4)) as ctx:
        got = prefetch.video_to_actions(ctx, cap, params)
        eng = pipeline.PairEngine(ctx, frontend.DecodedUploader(ctx))
        want = pipeline.frames_to_actions(eng, [src[i] for i in range(n)], fps, params)
    assert cap.seeks == 0
    assert got == want and len(got) == 48          # 51 sampled frames -> chunks of 20, 20, 11 -> 19 + 19 + 10 pairs


def test_process_video_writes_the_funscript(tmp_path):
    """prefetch.process_video: the reference's process_video call shape (FF:1094-1404) on the HIP path -- logs, progress,
    `<video>.funscript` written, returns False; the file equals what video_to_actions computes for the same capture."""
    import json
    from funscript_flow_amd import pipeline, prefetch
    from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames
    sw, sh, n, fps = 200, 120, 45, 30.0
    src = gray_to_bgr(sine_translate_frames(n, sw, sh, seed=9, amp=(3.0, 2.0), period=12, zoom=0.04))

    class Cap:
        def __init__(self):
            self.pos, self.released = 0, False

        def isOpened(self):
            return True

        def get(self, prop):
            return {prefetch.CAP_PROP_FRAME_COUNT: n, prefetch.CAP_PROP_FPS: fps, prefetch.CAP_PROP_FRAME_WIDTH: sw,
                    prefetch.CAP_PROP_FRAME_HEIGHT: sh}[prop]

        def grab(self):
            self.pos += 1
            return self.pos <= n

        def read(self, image=None):
            if self.pos >= n:
                return False, None
            f = src[self.pos].copy()                 # a capture that returns its own array (no decode-into)
            self.pos += 1
            return True, f

        def release(self):
            self.released = True

    params = {"detrend_window": 1.0, "norm_window": 1.0, "batch_size": 30, "keyframe_reduction": True, "overwrite": True}
    video = str(tmp_path / "clip.mp4")
    logs, progress, caps = [], [], []

    def open_capture(path):
        caps.append(Cap())
        return caps[-1]

    def make_context(cap):
        return _capi.Context(96, 64, max_batch=4, frame_slots=10, flow_slots=pipeline.min_flow_slots(4))

    assert prefetch.process_video(video, params, logs.append, open_capture, make_context, progress.append) is False
    assert caps[0].released and progress and progress[-1] == 100
    assert any(m.startswith("FPS: 30.00; downsampled to ~30.00 fps; 45 frames selected.") for m in logs)
    got = json.load(open(str(tmp_path / "clip.funscript")))
    with _capi.Context(96, 64, max_batch=4, frame_slots=10, flow_slots=pipeline.min_flow_slots(4)) as ctx:
        want = prefetch.video_to_actions(ctx, Cap(), params)
    assert got == {"version": "1.0", "actions": want} and len(want) >= 3
