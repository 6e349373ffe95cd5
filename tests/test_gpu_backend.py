"""GPU tests of the reference-shaped host API (funscript_flow_amd.backend / pipeline) and of the
full-size configurations of BASELINE.json, all through the C ABI.

Full-size cases (1080p, 4K) are checked (a) bit for bit against the C oracle on ONE pair -- the
oracle needs ~1 s (1080p) / ~4 s (4K) per pair -- and (b) through size-independent properties:
the device's pass-1/pass-2 reductions equal the numpy restatement applied to the device's own flow,
results do not depend on batch size, slot placement or frame sharing, and a known translation is
recovered."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as orc
from funscript_flow_amd import _capi, backend, pipeline
from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames


def test_precompute_flow_info_dict_matches_reference_shape():
    """Same keys / value kinds as FF:898-907; values equal the oracle's."""
    w, h = 256, 256
    fr = sine_translate_frames(2, w, h, seed=3, amp=(3.0, 2.0), zoom=0.02)
    info = backend.precompute_flow_info(fr[0], fr[1], {"backend": "HIP"})
    assert set(info) == {"flow", "pos_center", "neg_center", "val_pos", "val_neg", "cut", "cut_center", "mean_mag"}
    ref = orc.farneback(fr[0], fr[1])
    ox, oy, ov = orc.max_divergence_np(ref)
    assert tuple(int(v) for v in info["pos_center"]) == (ox, oy) == tuple(int(v) for v in info["neg_center"])
    assert np.float32(info["val_pos"]) == np.float32(ov) == np.float32(info["val_neg"])
    assert info["cut_center"] == info["pos_center"][0]
    assert info["cut"] is False or info["cut"] is True
    assert abs(float(info["mean_mag"]) - float(orc.mean_mag_np(ref))) <= 1e-4 * float(orc.mean_mag_np(ref))
    assert np.array_equal(np.asarray(info["flow"]), ref)            # lazy download of the device handle
    c = np.array([130.5, 120.25])
    for pov in (False, True):
        got = backend.radial_motion_weighted(info["flow"], c, info["cut"], pov)
        want = orc.radial_np(ref, c, False, pov)
        assert abs(got - want) <= 1e-4 * max(abs(want), 1e-3)
        got2 = backend.radial_motion_weighted(ref, c, False, pov)   # ndarray input path
        assert abs(got2 - want) <= 1e-4 * max(abs(want), 1e-3)
    assert backend.radial_motion_weighted(info["flow"], c, True) == 0.0
    # wrapper + pov_mode (FF:1019-1021, FF:880-882)
    info2 = backend.precompute_wrapper((fr[0], fr[1]), {"backend": "HIP", "pov_mode": True})
    assert tuple(int(v) for v in info2["pos_center"]) == (w // 2, h - 1) and info2["val_pos"] == 0
    # BGR frames as cv2.VideoCapture.read returns them
    bgr = gray_to_bgr(fr)
    info3 = backend.precompute_flow_info(bgr[0], bgr[1], {"backend": "HIP", "cut_threshold": 0.5})
    assert np.array_equal(np.asarray(info3["flow"]), ref) and info3["cut"] is True
    backend.release_contexts()


def test_stale_handle_is_an_error():
    fr = sine_translate_frames(2, 64, 64, seed=1)
    first = backend.precompute_flow_info(fr[0], fr[1], {"backend": "HIP"})
    for _ in range(backend.RING):
        backend.precompute_flow_info(fr[0], fr[1], {"backend": "HIP"})
    with pytest.raises(_capi.FFLError, match="stale"):
        np.asarray(first["flow"])
    backend.release_contexts()


def test_chunk_pipeline_matches_reference_chain(golden_dir):
    """PairEngine.process_chunk vs the chain captured from the REAL process_video (FF:1187-1242)."""
    meta = json.load(open(os.path.join(golden_dir, "chain_golden.json")))
    d = np.load(os.path.join(golden_dir, "chain_golden.npz"))
    s = meta["synth"]
    frames = sine_translate_frames(meta["n_frames"], meta["size"], meta["size"], seed=s["seed"], amp=tuple(s["amp"]),
                                   period=s["period"], zoom=s["zoom"])
    if zlib.crc32(frames.tobytes()) != meta["frames_crc32"]:
        pytest.skip("synthetic frames differ from the ones the golden was captured on (libm/numpy difference)")
    bs = meta["settings"]["batch_size"]
    with _capi.Context(meta["size"], meta["size"], max_batch=5, frame_slots=12, flow_slots=3 * 5 + 13) as ctx:
        eng = pipeline.PairEngine(ctx)
        dots, recs, start = [], [], 0
        for cs in range(0, meta["n_frames"], bs):           # pairs never span chunks (F10)
            chunk = frames[cs:cs + bs]
            if len(chunk) < 2:
                continue
            dd, rr = eng.process_chunk(chunk)
            dots += list(dd)
            recs += rr
    assert np.array_equal(np.array([[r[0], r[1]] for r in recs]), d["pos_center"])
    assert np.array_equal(np.array([np.float32(r[2]) for r in recs]), d["val_pos"])
    assert np.array_equal(np.array([r[4] for r in recs]), d["cut"])
    assert np.allclose(np.array([r[3] for r in recs], np.float64), d["mean_mag"], rtol=1e-4, atol=0)
    scale = np.mean(np.abs(d["dots"]))
    assert np.all(np.abs(np.array(dots) - d["dots"]) <= 1e-4 * np.maximum(np.abs(d["dots"]), scale))


def test_end_to_end_funscript_matches_reference(golden_dir):
    """BASELINE configs[0] shape, end to end on the HIP path: frames -> actions equals the .funscript
    the real process_video wrote (with the oracle as its cv2.calcOpticalFlowFarneback)."""
    meta = json.load(open(os.path.join(golden_dir, "chain_golden.json")))
    s = meta["synth"]
    frames = sine_translate_frames(meta["n_frames"], meta["size"], meta["size"], seed=s["seed"], amp=tuple(s["amp"]),
                                   period=s["period"], zoom=s["zoom"])
    if zlib.crc32(frames.tobytes()) != meta["frames_crc32"]:
        pytest.skip("synthetic frames differ from the ones the golden was captured on (libm/numpy difference)")
    with _capi.Context(meta["size"], meta["size"], max_batch=8, frame_slots=18, flow_slots=3 * 8 + 13) as ctx:
        actions = pipeline.frames_to_actions(pipeline.PairEngine(ctx), frames, meta["fps"], meta["settings"])
    assert actions == meta["funscript"]["actions"]


def test_config0_clip_640x360_to_funscript(tmp_path):
    """BASELINE configs[0] at native resolution: a 64-frame 640x360 sine-translate clip -> .funscript on the
    HIP path equals the same host chain fed by the CPU oracle (flow, argmax, cut, radial), action for action."""
    from funscript_flow_amd import postchain
    w, h, n, fps = 640, 360, 64, 30.0
    frames = sine_translate_frames(n, w, h, seed=0, amp=(4.0, 4.0), period=16, zoom=0.03)
    params = {"detrend_window": 2.0, "norm_window": 3.0, "batch_size": 3000, "keyframe_reduction": True,
              "pov_mode": False}
    with _capi.Context(w, h, max_batch=8, frame_slots=18, flow_slots=37) as ctx:
        got = pipeline.frames_to_actions(pipeline.PairEngine(ctx), frames, fps, params)
    flows = [orc.farneback(frames[j], frames[j + 1]) for j in range(n - 1)]
    pos = [orc.max_divergence_np(f)[:2] for f in flows]
    cuts = [bool(orc.mean_mag_np(f) > 7) for f in flows]
    dots = [float(orc.radial_np(f, c, k)) for f, c, k in zip(flows, orc.smooth_centers(pos), cuts)]
    want = postchain.actions_from_scalars(dots, cuts, list(range(n - 1)), fps, params)
    assert got == want and len(got) >= 4
    out = tmp_path / "clip.funscript"
    postchain.write_funscript(str(out), got)
    assert json.load(open(out)) == {"version": "1.0", "actions": want}


def test_sharded_engine_single_rank_equals_chunk_engine():
    w, h = 160, 120
    frames = sine_translate_frames(14, w, h, seed=9, amp=(2.5, 1.0), period=7)
    with _capi.Context(w, h, max_batch=4, frame_slots=10, flow_slots=25) as ctx:
        dots, recs = pipeline.PairEngine(ctx).process_chunk(frames)
        sharded, allrecs = pipeline.process_chunk_sharded(pipeline.HipShardEngine(ctx), frames, 0, 1, lambda o: [o])
    assert np.array_equal(dots, sharded)
    assert np.array_equal(allrecs[:, :2], np.array([[r[0], r[1]] for r in recs]))


def test_results_do_not_depend_on_batching_or_slots():
    w, h = 320, 180
    fr = sine_translate_frames(6, w, h, seed=12)
    with _capi.Context(w, h, max_batch=5, frame_slots=12, flow_slots=12) as ctx:
        for i in range(6):
            ctx.upload_frame(i, fr[i])
        ctx.flow_pairs([0, 1, 2, 3, 4], [1, 2, 3, 4, 5], [0, 1, 2, 3, 4])          # streamed, shared frames
        a = [ctx.download_flow(j) for j in range(5)]
        ra = [ctx.pass1_result(j) for j in range(5)]
        for i in range(6):
            ctx.upload_frame(11 - i, fr[i])                                        # other slots, reversed
        for j in (3, 0, 4, 2, 1):                                                  # one pair per call
            ctx.flow_pairs([11 - j], [11 - j - 1], [5 + j])
        b = [ctx.download_flow(5 + j) for j in range(5)]
        rb = [ctx.pass1_result(5 + j) for j in range(5)]
    for j in range(5):
        assert np.array_equal(a[j], b[j]) and ra[j] == rb[j]
        assert np.array_equal(a[j], orc.farneback(fr[j], fr[j + 1]))


def test_results_do_not_depend_on_schedule_options():
    """Compute lanes (co-scheduled batches) and the frame-expansion schedules are speed knobs only: many small batches in flight on recycled frame/flow slots give identical bits."""
    w, h = 192, 144
    n = 21
    fr = sine_translate_frames(n + 1, w, h, seed=33, amp=(2.0, 1.0), period=9)
    want = None
    try:
        for lanes, run_ahead, tile in [(1, 0, 16), (2, 0, 16), (2, 1, 16), (3, 2, 16), (1, 1, 16)]:
            _capi.set_option("lanes", lanes)
            _capi.set_option("run_ahead", run_ahead)
            _capi.set_option("blur_tile_h", tile)
            with _capi.Context(w, h, max_batch=3, frame_slots=8, flow_slots=3 * 3 + 13) as ctx:
                dots, recs = pipeline.PairEngine(ctx).process_chunk(fr)
                last = ctx.download_flow((n - 1) % ctx.flow_slots)
            got = (np.array(dots), [tuple(r) for r in recs], last)
            if want is None:
                want = got
                assert np.array_equal(last, orc.farneback(fr[n - 1], fr[n]))
            else:
                assert np.array_equal(got[0], want[0]) and got[1] == want[1] and np.array_equal(got[2], want[2])
    finally:
        _capi.set_option("lanes", 2)
        _capi.set_option("run_ahead", 0)
        _capi.set_option("blur_tile_h", 16)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_full_size_pair_bit_exact_and_properties(w, h):
    fr = sine_translate_frames(3, w, h, seed=1)
    with _capi.Context(w, h, max_batch=2, frame_slots=4, flow_slots=4) as ctx:
        for i in range(3):
            ctx.upload_frame(i, fr[i])
        ctx.flow_pairs([0, 1], [1, 2], [0, 1])
        flows = [ctx.download_flow(0), ctx.download_flow(1)]
        assert np.array_equal(flows[0], orc.farneback(fr[0], fr[1]))       # full-size bit-exact (one pair)
        for j in range(2):
            x, y, v, mm, cut = ctx.pass1_result(j)
            ox, oy, ov = orc.max_divergence_np(flows[j])                   # argmax: bit-exact index and value
            assert (x, y) == (ox, oy) and np.float32(v).tobytes() == np.float32(ov).tobytes()
            rm = float(orc.mean_mag_np(flows[j]))
            assert abs(float(mm) - rm) <= 1e-4 * rm
            c = (0.4 * w, 0.55 * h)
            for pov in (False, True):
                got = ctx.radial([j], [c], [False], pov)[0]
                want = float(orc.radial_np(flows[j], c, False, pov))
                assert abs(got - want) <= 1e-4 * max(abs(want), 1e-6 * w)
        # translation property: interior median flow ~ the synthetic shift between frames 0 and 1
        t = np.arange(2)
        dx = 4.0 * np.sin(2 * np.pi * t / 16)
        dy = 4.0 * np.sin(2 * np.pi * t / 16 + np.pi / 3)
        med = np.median(flows[0][h // 4:-h // 4, w // 4:-w // 4].reshape(-1, 2), axis=0)
        # (Farneback under-estimates by up to ~10 % on this texture)
        for got, want in ((med[0], dx[1] - dx[0]), (med[1], dy[1] - dy[0])):
            assert abs(got - want) <= 0.15 * abs(want) + 0.05


def test_stereo_frame_split_per_eye_via_strides():
    """BASELINE configs[4] shape (VR side-by-side stereo, each eye an independent image): the
    eyes of a (H, 2H) frame are uploaded straight out of the full frame through the row stride of
    ffl_upload_frame -- no host-side copy -- and each eye's pair matches the oracle."""
    eh = 160
    full = sine_translate_frames(2, 2 * eh, eh, seed=17, amp=(3.0, 1.0), period=5)      # (2, eh, 2*eh)
    full_bgr = gray_to_bgr(full)
    with _capi.Context(eh, eh, max_batch=2, frame_slots=4, flow_slots=4) as ctx:
        for eye in (0, 1):
            for t in (0, 1):
                view = full[t][:, eye * eh:(eye + 1) * eh]                           # non-contiguous view
                assert not view.flags["C_CONTIGUOUS"]
                ctx.upload_frame(2 * eye + t, view)
        ctx.flow_pairs([0, 2], [1, 3], [0, 1])
        for eye in (0, 1):
            ref = orc.farneback(np.ascontiguousarray(full[0][:, eye * eh:(eye + 1) * eh]),
                                np.ascontiguousarray(full[1][:, eye * eh:(eye + 1) * eh]))
            assert np.array_equal(ctx.download_flow(eye), ref)
        # same through the 3-channel path
        for t in (0, 1):
            ctx.upload_frame(t, full_bgr[t][:, eh:2 * eh])
        ctx.flow_pairs([0], [1], [2])
        assert np.array_equal(ctx.download_flow(2), ctx.download_flow(1))


def test_ragged_and_minimum_sizes():
    """Odd sizes (no 4-pixel alignment), tiles cut by every border, fewer pyramid levels."""
    for (w, h) in [(67, 45), (131, 70), (257, 255), (64, 33)]:
        fr = sine_translate_frames(2, w, h, seed=w)
        with _capi.Context(w, h, max_batch=1) as ctx:
            assert ctx.num_levels() == orc.num_levels(w, h)
            ctx.submit_pair(0, fr[0], fr[1])
            assert np.array_equal(ctx.download_flow(0), orc.farneback(fr[0], fr[1])), (w, h)


def test_long_stream_recycles_every_ring_and_slot():
    """400 pairs in batches of 4 with two compute lanes: every event ring (16 / 32 entries), frame slot and flow
    slot is recycled many times while batches are in flight; the scalars must equal a one-lane, one-batch-at-a-
    time run of the same clip (which the shorter tests pin against the oracle)."""
    w, h, n = 128, 96, 400
    fr = sine_translate_frames(n + 1, w, h, seed=77, amp=(3.0, 2.0), period=23, zoom=0.02)
    out = []
    try:
        for lanes, run_ahead, merge, fuse in [(1, 0, 1, 0), (2, 0, 1, 1), (2, 1, 0, 1), (3, 2, 1, 10000)]:
            _capi.set_option("lanes", lanes)
            _capi.set_option("run_ahead", run_ahead)
            _capi.set_option("merge_expand", merge)
            _capi.set_option("fuse_first", fuse)
            with _capi.Context(w, h, max_batch=4, frame_slots=10, flow_slots=3 * 4 + 13) as ctx:
                dots, recs = pipeline.PairEngine(ctx).process_chunk(fr)
            out.append((np.array(dots), [tuple(r) for r in recs]))
    finally:
        _capi.set_option("lanes", 2)
        _capi.set_option("run_ahead", 0)
        _capi.set_option("merge_expand", 1)
        _capi.set_option("fuse_first", 10000)
    for d, r in out[1:]:
        assert np.array_equal(d, out[0][0]) and r == out[0][1]
    j = 123
    with _capi.Context(w, h, max_batch=1) as ctx:
        ctx.submit_pair(0, fr[j], fr[j + 1])
        assert tuple(ctx.pass1_result(0)) == out[0][1][j]


def test_uploads_out_of_pinned_context_memory():
    """Frames written into ffl_host_alloc memory (Context.pinned_frames) take the zero-copy upload path (runs of
    consecutive entries, gray and BGR); results equal those of ordinary ndarrays."""
    w, h, n = 160, 96, 30
    fr = sine_translate_frames(n + 1, w, h, seed=5, amp=(2.0, 2.5), period=11)
    with _capi.Context(w, h, max_batch=4, frame_slots=10, flow_slots=25) as ctx:
        want = pipeline.PairEngine(ctx).process_chunk(list(fr))
        pin = ctx.pinned_frames(n + 1)
        pin[:] = fr
        got = pipeline.PairEngine(ctx).process_chunk([pin[i] for i in range(n + 1)])
        assert np.array_equal(got[0], want[0]) and [tuple(r) for r in got[1]] == [tuple(r) for r in want[1]]
        pin3 = ctx.pinned_frames(3, channels=3)
        pin3[:] = gray_to_bgr(fr[:3], gains=(0.9, 1.0, 0.8))
        ctx.upload_frames(0, [pin3[0], pin3[1], pin3[2]])
        ctx.flow_pairs([0, 1], [1, 2], [0, 1])
        a = ctx.download_flow(1)
        ctx.upload_frames(0, [np.array(pin3[i]) for i in range(3)])     # same pixels from pageable memory
        ctx.flow_pairs([0, 1], [1, 2], [0, 1])
        assert np.array_equal(a, ctx.download_flow(1))


def test_bgr_upload_with_odd_pixel_count_into_later_slots():
    """k_gray works on 4-pixel groups with 32-bit loads / stores at bgr + slot*3N and gray + slot*N, which are
    only byte-aligned when N is odd: odd w*h, slots >= 1, single and batched uploads."""
    w, h = 67, 45                                     # N = 3015, odd
    fr = sine_translate_frames(4, w, h, seed=3)
    bgr = gray_to_bgr(fr, gains=(0.9, 1.0, 0.8))
    want = [orc.bgr2gray(b) for b in bgr]
    with _capi.Context(w, h, max_batch=2, frame_slots=6, flow_slots=4) as ctx:
        ctx.upload_frame(1, bgr[0])
        ctx.upload_frame(3, bgr[1])
        ctx.upload_frames(4, [bgr[2], bgr[3]])
        ctx.sync()
        assert np.array_equal(ctx.download_frame(1), want[0])
        assert np.array_equal(ctx.download_frame(3), want[1])
        assert np.array_equal(ctx.download_frame(4), want[2]) and np.array_equal(ctx.download_frame(5), want[3])
        ctx.flow_pairs([1, 4], [3, 5], [0, 1])
        assert np.array_equal(ctx.download_flow(0), orc.farneback(want[0], want[1]))
        assert np.array_equal(ctx.download_flow(1), orc.farneback(want[2], want[3]))


def test_context_calls_from_several_host_threads():
    """submit / pass1 / radial on distinct slots may come from different host threads.  One thread
    streams a chunk through PairEngine (uploads, batches, pass 2) on flow slots 0..20 while a second one keeps
    reading records, radial scalars and the flow of a pair parked in slot 30 of the SAME context; both see exactly
    what a single-threaded run gives."""
    import threading
    w, h, n = 192, 128, 60
    fr = sine_translate_frames(n + 1, w, h, seed=8, amp=(3.0, 2.0), period=13, zoom=0.02)
    with _capi.Context(w, h, max_batch=4, frame_slots=12, flow_slots=32) as ctx:
        want_dots, want_recs = pipeline.PairEngine(ctx).process_chunk(fr)
        ctx.upload_frames(10, [fr[7], fr[8]])
        ctx.flow_pairs([10], [11], [30])
        park = (ctx.pass1_result(30), ctx.radial([30], [(90.5, 60.25)], [False], False)[0], ctx.download_flow(30))
        assert np.array_equal(park[2], orc.farneback(fr[7], fr[8]))
        errors, stop = [], threading.Event()

        def reader():
            try:
                while not stop.is_set():
                    got = (ctx.pass1_result(30), ctx.radial([30], [(90.5, 60.25)], [False], False)[0])
                    if got != park[:2] or not np.array_equal(ctx.download_flow(30), park[2]):
                        errors.append("parked slot changed")
                        return
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        t = threading.Thread(target=reader)
        t.start()
        try:
            sub = _SubCtx(ctx, frame_slots=10, flow_slots=21)
            for _ in range(3):
                dots, recs = pipeline.PairEngine(sub).process_chunk(fr)
                assert np.array_equal(dots, want_dots) and [tuple(r) for r in recs] == [tuple(r) for r in want_recs]
        finally:
            stop.set()
            t.join()
        assert not errors, errors


class _SubCtx:
    """a view of a context that advertises fewer slots (so that an engine leaves the others alone)"""

    def __init__(self, ctx, frame_slots, flow_slots):
        self._c, self.frame_slots, self.flow_slots, self.max_batch = ctx, frame_slots, flow_slots, ctx.max_batch

    def __getattr__(self, k):
        return getattr(self._c, k)


def test_empty_and_minimal_chunks():
    """Chunks of 0 / 1 / 2 frames (FF:1150: a chunk with fewer than 2 frames is skipped) and a chunk one pair longer
    than a batch; the +-6 window is clipped to what exists."""
    w, h = 96, 64
    fr = sine_translate_frames(6, w, h, seed=2, amp=(2.0, 1.0), period=5)
    with _capi.Context(w, h, max_batch=4, frame_slots=10, flow_slots=pipeline.min_flow_slots(4)) as ctx:
        eng = pipeline.PairEngine(ctx)
        for n_frames in (0, 1):
            dots, recs = eng.process_chunk(fr[:n_frames])
            assert len(dots) == 0 and recs == []
        dots, recs = eng.process_chunk(fr[:2])
        ref = orc.farneback(fr[0], fr[1])
        ox, oy, _ = orc.max_divergence_np(ref)
        assert (recs[0][0], recs[0][1]) == (ox, oy)
        want = float(orc.radial_np(ref, (float(ox), float(oy)), recs[0][4]))       # a single pair is its own centre
        assert abs(dots[0] - want) <= 1e-4 * max(abs(want), 1e-3)
        dots6, recs6 = eng.process_chunk(fr)                                         # 5 pairs: one full batch + 1
        assert len(dots6) == 5 and tuple(recs6[0]) == tuple(recs[0])
        assert pipeline.frames_to_actions(eng, fr[:1], 30.0, {"batch_size": 10}) == []


def test_precompute_all_and_radial_all_replace_the_two_pools(golden_dir):
    """backend.precompute_all / radial_all = the reference's `pool.starmap(precompute_wrapper, ...)` (FF:1190-1191) and
    its ProcessPoolExecutor loop (FF:1232-1236) as two batched calls, with the reference's own chunk code in between:
    the per-pair dicts and scalars equal the chain captured from the real process_video."""
    meta = json.load(open(os.path.join(golden_dir, "chain_golden.json")))
    d = np.load(os.path.join(golden_dir, "chain_golden.npz"))
    s = meta["synth"]
    frames = sine_translate_frames(meta["n_frames"], meta["size"], meta["size"], seed=s["seed"], amp=tuple(s["amp"]),
                                   period=s["period"], zoom=s["zoom"])
    if zlib.crc32(frames.tobytes()) != meta["frames_crc32"]:
        pytest.skip("synthetic frames differ from the ones the golden was captured on (libm/numpy difference)")
    bs = meta["settings"]["batch_size"]
    params = {"backend": "HIP", "hip_batch": 5}
    pos, vals, cuts, mm, dots = [], [], [], [], []
    for cs in range(0, meta["n_frames"], bs):
        frames_gray = list(frames[cs:cs + bs])
        if len(frames_gray) < 2:
            continue
        pairs = list(zip(frames_gray[:-1], frames_gray[1:]))                          # FF:1188
        precomputed = backend.precompute_all(pairs, params)                           # FF:1190-1191
        assert set(precomputed[0]) == {"flow", "pos_center", "neg_center", "val_pos", "val_neg", "cut", "cut_center", "mean_mag"}
        centers = pipeline.smooth_centers([info["pos_center"] for info in precomputed])   # FF:1203-1214
        dots += list(backend.radial_all(precomputed, centers, False))                 # FF:1232-1236
        pos += [tuple(int(v) for v in info["pos_center"]) for info in precomputed]
        vals += [np.float32(info["val_pos"]) for info in precomputed]
        cuts += [info["cut"] for info in precomputed]
        mm += [float(info["mean_mag"]) for info in precomputed]
        j = len(pairs) // 2
        assert np.array_equal(np.asarray(precomputed[j]["flow"]), orc.farneback(pairs[j][0], pairs[j][1]))
    assert np.array_equal(np.array(pos), d["pos_center"]) and np.array_equal(np.array(vals), d["val_pos"])
    assert np.array_equal(np.array(cuts), d["cut"]) and np.allclose(mm, d["mean_mag"], rtol=1e-4, atol=0)
    scale = np.mean(np.abs(d["dots"]))
    assert np.all(np.abs(np.array(dots) - d["dots"]) <= 1e-4 * np.maximum(np.abs(d["dots"]), scale))
    # arbitrary (non-stream) pairs, BGR operands, POV, a cut, and a stale handle
    w = h = meta["size"]
    bgr = gray_to_bgr(frames[:6])
    odd = [(bgr[4], bgr[1]), (bgr[0], bgr[5]), (bgr[2], bgr[2])]
    pre = backend.precompute_all(odd, {"backend": "HIP", "pov_mode": True, "cut_threshold": 0.0})
    for (a, b), info in zip(odd, pre):
        one = backend.precompute_flow_info(a, b, {"backend": "HIP", "pov_mode": True, "cut_threshold": 0.0})
        assert np.array_equal(np.asarray(info["flow"]), np.asarray(one["flow"]))
        assert tuple(info["pos_center"]) == (w // 2, h - 1) and info["val_pos"] == 0
    assert pre[0]["cut"] is True and backend.radial_all(pre, [(1.0, 2.0)] * 3, True) == [0.0, 0.0, 0.0]
    backend.precompute_all(odd[:1], {"backend": "HIP"})
    with pytest.raises(_capi.FFLError, match="stale"):
        np.asarray(pre[1]["flow"])
    assert backend.precompute_all([], params) == []
    backend.release_contexts()
