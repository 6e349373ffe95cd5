"""GPU parity: every HIP kernel, through the C ABI (include/ffl.h), against the CPU oracle.

Bar (north_star): flow bit-identical to the oracle (same operations in the same order, no FMA
contraction), argmax pixel index bit-exact, float reductions within 1e-4 relative.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as orc
from funscript_flow_amd import _capi
from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames


def frames(n, w, h, seed, **kw):
    return sine_translate_frames(n, w, h, seed=seed, **kw)


@pytest.fixture(scope="module")
def post(golden_dir):
    return np.load(os.path.join(golden_dir, "post_goldens.npz"))


# ------------------------------------------------------------------ per-kernel, per-level parity
@pytest.mark.parametrize("w,h", [(96, 80), (320, 180), (250, 131), (256, 256), (384, 264), (640, 360)])
def test_farneback_stages_bit_exact(w, h):
    fr = frames(2, w, h, seed=21, amp=(3.0, 2.0), period=6)
    with _capi.Context(w, h, max_batch=1) as ctx:
        ctx.upload_frame(0, fr[0])
        ctx.upload_frame(1, fr[1])
        assert ctx.num_levels() == orc.num_levels(w, h)
        for level in range(ctx.num_levels(), -1, -1):
            assert ctx.level_size(level) == orc.level_params(w, h, level)[:2]
            for it in (0, 2, 3):
                g = ctx.debug_pair(0, 1, level, it)
                o = orc.farneback_dbg(fr[0], fr[1], level, it)
                for key in ("I0", "I1", "R0", "R1", "M", "flow", "out"):
                    assert np.array_equal(g[key], o[key]), f"{key} differs at level {level} iter {it}: " \
                        f"max abs {np.abs(g[key] - o[key]).max()}"


@pytest.mark.parametrize("w,h,seed", [(640, 360, 0), (256, 256, 3), (200, 120, 8)])
def test_pair_end_to_end(w, h, seed):
    fr = frames(3, w, h, seed=seed)
    with _capi.Context(w, h, max_batch=2) as ctx:
        for i in range(3):
            ctx.upload_frame(i, fr[i])
        ctx.flow_pairs([0, 1], [1, 2], [0, 1])
        for j in range(2):
            flow = ctx.download_flow(j)
            ref = orc.farneback(fr[j], fr[j + 1])
            assert np.array_equal(flow, ref)
            x, y, v, mm, cut = ctx.pass1_result(j)
            ox, oy, ov = orc.max_divergence_np(ref)
            assert (x, y) == (ox, oy)
            assert np.float32(v).tobytes() == np.float32(ov).tobytes()
            assert abs(float(mm) - float(orc.mean_mag_np(ref))) <= 1e-4 * float(orc.mean_mag_np(ref))
            assert cut == bool(orc.mean_mag_np(ref) > 7)
            c = (w / 2.0 + 0.3, h / 2.0 - 0.7)
            for pov in (False, True):
                got = ctx.radial([j], [c], [False], pov)[0]
                want = float(orc.radial_np(ref, c, False, pov))
                scale = max(abs(want), float(np.mean(np.abs(ref))) * max(w, h) * 1e-2)
                assert abs(got - want) <= 1e-4 * scale
            assert ctx.radial([j], [c], [True], False)[0] == 0.0


@pytest.mark.parametrize("fuse", [10000, 1])
@pytest.mark.parametrize("kind", ["constant", "noise", "checker", "big_shift"])
def test_degenerate_and_hostile_inputs(kind, fuse):
    """Flat frames (all-zero M, denormal-range sums), uncorrelated noise (large erratic flow, many
    out-of-bounds warps), a 1-px checkerboard and a 40-px jump (coarse levels dominate) -- on the separate-launch path
    and (fuse = 1) on the folded first iteration, whose pipelined phase U gathers branch-free: lanes that land outside
    the image read pixel (0, 0) and must drop it."""
    w, h = 192, 136
    rng = np.random.default_rng(42)
    if kind == "constant":
        a, b = np.full((h, w), 77, np.uint8), np.full((h, w), 79, np.uint8)
    elif kind == "noise":
        a, b = (rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(2))
    elif kind == "checker":
        y, x = np.mgrid[0:h, 0:w]
        a = (((x + y) & 1) * 255).astype(np.uint8)
        b = np.roll(a, 1, axis=1)
    else:
        big = sine_translate_frames(1, w + 80, h, seed=6)[0]
        a, b = np.ascontiguousarray(big[:, 40:40 + w]), np.ascontiguousarray(big[:, :w])
    try:
        _capi.set_option("fuse_first", fuse)
        with _capi.Context(w, h, max_batch=1) as ctx:
            ctx.submit_pair(0, a, b)
            flow = ctx.download_flow(0)
            x, y, v, mm, cut = ctx.pass1_result(0)
    finally:
        _capi.set_option("fuse_first", 10000)
    ref = orc.farneback(a, b)
    assert np.array_equal(flow, ref) and np.isfinite(flow).all()
    ox, oy, ov = orc.max_divergence_np(ref)
    assert (x, y) == (ox, oy) and np.float32(v).tobytes() == np.float32(ov).tobytes()


def test_bgr_upload_matches_gray_path():
    w, h = 320, 180
    fr = frames(2, w, h, seed=5)
    bgr = gray_to_bgr(fr, gains=(0.9, 1.0, 0.8))
    with _capi.Context(w, h, max_batch=1) as ctx:
        ctx.submit_pair(0, bgr[0], bgr[1])
        flow = ctx.download_flow(0)
    assert np.array_equal(flow, orc.farneback(orc.bgr2gray(bgr[0]), orc.bgr2gray(bgr[1])))


# ------------------------------------------------------------------ post path vs REFERENCE goldens
@pytest.mark.parametrize("name", ["noise_36x64", "smooth_90x160", "noise_256x256", "ties_40x72", "negfirst_24x40",
                                  "farneback_180x320", "edge_32x48"])
def test_post_path_matches_reference_goldens(post, name):
    flow = post[f"{name}.flow"]
    h, w, _ = flow.shape
    with _capi.Context(w, h, max_batch=1, flow_slots=2) as ctx:
        ctx.upload_flow(0, flow)
        x, y, v, mm, _ = ctx.pass1_result(0)
        assert (x, y) == tuple(post[f"{name}.maxdiv"])
        assert np.float32(v).tobytes() == np.float32(post[f"{name}.maxdiv_val"]).tobytes()
        ref_mm = float(orc.mean_mag_np(flow))
        assert abs(float(mm) - ref_mm) <= 1e-4 * max(ref_mm, 1e-30)
        scale = float(np.mean(np.abs(flow))) * max(h, w)
        for c, (gw, gp, gc) in zip(post[f"{name}.centers"], post[f"{name}.radial"]):
            assert abs(ctx.radial([0], [c], [False], False)[0] - gw) <= 1e-4 * max(abs(gw), 1e-6 * scale)
            assert abs(ctx.radial([0], [c], [False], True)[0] - gp) <= 1e-4 * max(abs(gp), 1e-6 * scale)
            assert ctx.radial([0], [c], [True], False)[0] == gc == 0.0
        # POV pass 1: centre of the bottom edge, value 0 (FF:880-882)
        ctx.upload_flow(1, flow, pov_mode=True)
        x, y, v, _, _ = ctx.pass1_result(1)
        assert (x, y, float(v)) == (w // 2, h - 1, 0.0)


def test_denormals_and_cut_threshold():
    """Tiny flow values must not be flushed (the CPU path keeps denormals); cut uses mean_mag > thr."""
    h, w = 32, 48
    flow = np.full((h, w, 2), 1e-39, np.float32)
    flow[10, 20, 0] = 3e-39
    with _capi.Context(w, h, max_batch=1) as ctx:
        ctx.upload_flow(0, flow)
        x, y, v, mm, cut = ctx.pass1_result(0)
        ox, oy, ov = orc.max_divergence_np(flow)
        assert (x, y) == (ox, oy) and np.float32(v).tobytes() == np.float32(ov).tobytes()
        big = np.full((h, w, 2), 6.0, np.float32)   # |flow| = 8.49 > 7 -> cut
        ctx.upload_flow(0, big)
        assert ctx.pass1_result(0)[4] is True
        assert ctx.pass1_result(0, cut_threshold=9.0)[4] is False


def test_cut_flag_known_answers_at_the_threshold():
    """FF:893 `is_cut = mean_mag > 7` is a discontinuous function of a float (the cut flip zone): the
    threshold itself is pinned by known answers.  A flow of constant magnitude exactly 7 has mean_mag == 7.0 in every
    summation order (f64 sum of identical exactly representable terms; numpy's pairwise f32 mean likewise) -> NOT cut;
    the next float above 7 in every pixel -> cut; and a field whose mean sits one f32 step above 7 only through ONE
    pixel of 1920x1080 is beyond f32 resolution of the mean and must agree with numpy's own answer for it."""
    h, w = 1080, 1920
    up = np.nextafter(np.float32(7.0), np.float32(8.0))
    flow = np.zeros((h, w, 2), np.float32)
    with _capi.Context(w, h, max_batch=1) as ctx:
        for vec, want in (((7.0, 0.0), False), ((0.0, -7.0), False), ((up, 0.0), True), ((0.0, up), True)):
            flow[...] = vec
            assert np.float32(np.sqrt(np.float32(vec[0]) ** 2 + np.float32(vec[1]) ** 2)) == np.float32(max(abs(vec[0]), abs(vec[1])))
            ctx.upload_flow(0, flow)
            x, y, v, mm, cut = ctx.pass1_result(0)
            assert cut is want and (float(mm) == 7.0) is (not want), (vec, mm, cut)
            assert bool(orc.mean_mag_np(flow) > 7) is want          # the reference's own expression on the same field
        # 3-4-5 triangle scaled: |(4.2, 5.6)| is 7 only up to f32 rounding of the squares -- whatever sqrtf gives per pixel,
        # device and numpy must agree on the flag because every pixel carries the same magnitude
        flow[...] = (np.float32(4.2), np.float32(5.6))
        ctx.upload_flow(0, flow)
        mag = np.sqrt(np.float32(4.2) ** 2 + np.float32(5.6) ** 2, dtype=np.float32)
        assert ctx.pass1_result(0)[4] is bool(mag > np.float32(7.0)) is bool(orc.mean_mag_np(flow) > 7)


def test_errors_are_loud():
    with _capi.Context(64, 64, max_batch=1) as ctx:
        with pytest.raises(_capi.FFLError):
            ctx.flow_pairs([0], [1], [0])            # frames never uploaded
        with pytest.raises(_capi.FFLError):
            ctx.upload_frame(0, np.zeros((32, 32), np.uint8))  # wrong size
        with pytest.raises(_capi.FFLError):
            ctx.pass1_result(0)                       # nothing computed yet
    with pytest.raises(_capi.FFLError):
        _capi.Context(8, 8)


@pytest.mark.parametrize("w,h", [(16, 16), (17, 19), (31, 64), (64, 33), (65, 17), (127, 129), (130, 66), (20, 300), (300, 20)])
def test_small_and_awkward_sizes_bit_exact(w, h):
    """Sizes below one 64x16 tile, not multiples of 4 / 16 / 64, and extreme aspect ratios: every kernel's edge
    paths (clamped rows, partial tiles and strips, byte-wise pyramid taps, 1-strip reductions)."""
    fr = frames(3, w, h, seed=w * 7 + h, amp=(1.5, 1.0), period=5)
    with _capi.Context(w, h, max_batch=2, frame_slots=4, flow_slots=4) as ctx:
        for i in range(3):
            ctx.upload_frame(i, fr[i])
        ctx.flow_pairs([0, 1], [1, 2], [0, 1])
        for j in range(2):
            ref = orc.farneback(fr[j], fr[j + 1])
            assert np.array_equal(ctx.download_flow(j), ref)
            x, y, v, mm, _ = ctx.pass1_result(j)
            ox, oy, ov = orc.max_divergence_np(ref)
            assert (x, y) == (ox, oy) and np.float32(v).tobytes() == np.float32(ov).tobytes()
            rm = float(orc.mean_mag_np(ref))
            assert abs(float(mm) - rm) <= 1e-4 * max(rm, 1e-30)
            c = (0.37 * w, 0.61 * h)
            got = ctx.radial([j], [c], [False], False)[0]
            want = float(orc.radial_np(ref, c, False, False))
            assert abs(got - want) <= 1e-4 * max(abs(want), float(np.mean(np.abs(ref))) * max(w, h) * 1e-2)


@pytest.mark.parametrize("w,h", [(320, 180), (250, 131), (96, 80), (17, 19), (640, 360), (1920, 1080)])
def test_folded_first_iteration_bit_exact(w, h):
    """fuse_first = 1: every level's flow init + first UpdateMatrices run inside the first blur+solve launch
    (phase U into LDS).  By default only levels with >= 10000 tiles over the batch take that path, which no other
    test reaches -- here it is forced at every level and size, against the oracle."""
    fr = frames(3, w, h, seed=w + h, amp=(3.0, 2.0), period=6, zoom=0.03)
    try:
        _capi.set_option("fuse_first", 1)
        with _capi.Context(w, h, max_batch=2, frame_slots=4, flow_slots=4) as ctx:
            for i in range(3):
                ctx.upload_frame(i, fr[i])
            ctx.flow_pairs([0, 1], [1, 2], [0, 1])
            got = [ctx.download_flow(0), ctx.download_flow(1)]
    finally:
        _capi.set_option("fuse_first", 10000)
    for j in range(2):
        assert np.array_equal(got[j], orc.farneback(fr[j], fr[j + 1]))


@pytest.mark.parametrize("rows,fuse", [(2, 0), (3, 1), (8, 0), (8, 1), (17, 1), (23, 0)])
def test_strip_walk_bit_exact(rows, fuse):
    """blur_rows forces how many vertically adjacent tiles a k_blur_solve workgroup walks down (carrying 14 rows in
    registers); automatic selection only exceeds 1 on large grids.  Heights that are / are not multiples of
    16 * rows, with and without the folded first iteration."""
    try:
        _capi.set_option("blur_rows", rows)
        _capi.set_option("fuse_first", fuse)
        for (w, h) in [(250, 131), (320, 180), (96, 80), (64, 256), (80, 560)]:       # 35 tile rows: 17 + 17 + 1, 23 + 12
            fr = frames(2, w, h, seed=rows * 10 + fuse, amp=(2.0, 3.0), period=5)
            with _capi.Context(w, h, max_batch=1) as ctx:
                ctx.submit_pair(0, fr[0], fr[1])
                assert np.array_equal(ctx.download_flow(0), orc.farneback(fr[0], fr[1])), (w, h)
    finally:
        _capi.set_option("blur_rows", 0)
        _capi.set_option("fuse_first", 10000)


@pytest.mark.parametrize("w,h", [(256, 256), (392, 264), (640, 360), (1920, 1080)])
def test_one_pass_coarse_pyramid_levels_bit_exact(w, h):
    """k_pyr_coarse (x1/4 and x1/8 levels from one LDS-staged pass over the frame; sizes with w, h multiples of 8):
    level images of both frames against the oracle's pyr_level, incl. tiles cut by the right / bottom border and
    the REFLECT_101 halo of every edge tile; the H + V kernel pairs (pyr_coarse = 0) give the same bits."""
    fr = frames(2, w, h, seed=w + 3 * h, amp=(3.0, 2.0), period=6)
    got = {}
    try:
        for on in (1, 0):
            _capi.set_option("pyr_coarse", on)
            with _capi.Context(w, h, max_batch=1) as ctx:
                ctx.upload_frame(0, fr[0])
                ctx.upload_frame(1, fr[1])
                got[on] = {lvl: ctx.debug_pair(0, 1, lvl, 0) for lvl in (3, 2)}
    finally:
        _capi.set_option("pyr_coarse", 1)
    for lvl in (3, 2):
        for key, f in (("I0", fr[0]), ("I1", fr[1])):
            want = orc.pyr_level(f, lvl)
            assert np.array_equal(got[1][lvl][key], want), (lvl, key, np.abs(got[1][lvl][key] - want).max())
            assert np.array_equal(got[0][lvl][key], want), (lvl, key)
