"""Known-answer tests pinning the C restatement of cv2.calcOpticalFlowFarneback (FunscriptFlow.pyw:878-879).

The arithmetic lives in opencv-python 4.11.0.86 (uv.lock:238-239), which is absent from
/root/reference and from this image, and the reference holds no fixture for it: PARITY UNPINNED
against cv2.  These tests pin the restatement against analytic answers only."""
import numpy as np
import pytest

import oracle as orc
from funscript_flow_amd.synth import sine_translate_frames


def test_level_logic():
    # all BASELINE sizes use 3 extra scales; sigma/ksize = 3.5/19, 1.5/9, 0.5/3, 0/3
    for (w, h), sizes in {(256, 256): [(32, 32), (64, 64), (128, 128), (256, 256)],
                          (640, 360): [(80, 45), (160, 90), (320, 180), (640, 360)],
                          (1920, 1080): [(240, 135), (480, 270), (960, 540), (1920, 1080)],
                          (3840, 2160): [(480, 270), (960, 540), (1920, 1080), (3840, 2160)]}.items():
        assert orc.num_levels(w, h) == 3
        got = [orc.level_params(w, h, k) for k in (3, 2, 1, 0)]
        assert [(g[0], g[1]) for g in got] == sizes
        assert [(g[2], g[3]) for g in got] == [(3.5, 19), (1.5, 9), (0.5, 3), (0.0, 3)]
    # min_size = 32 stops the pyramid early for small images
    assert orc.num_levels(100, 60) == 0      # 60*0.5 = 30 < 32
    assert orc.num_levels(130, 64) == 1
    assert orc.num_levels(255, 255) == 2     # 255/8 = 31.9 < 32


def test_gaussian_kernels():
    assert np.array_equal(orc.gaussian_kernel(3, 0.0), np.float32([0.25, 0.5, 0.25]))
    k = orc.gaussian_kernel(3, 0.5)
    assert np.allclose(k, [0.10650698, 0.78698604, 0.10650698], atol=1e-8)
    for n, s in ((9, 1.5), (19, 3.5)):
        k = orc.gaussian_kernel(n, s)
        assert abs(k.sum() - 1) < 1e-6 and np.array_equal(k, k[::-1]) and k.argmax() == n // 2


def test_polyexp_constants():
    g, xg, xxg, ig = orc.polyexp_constants()
    assert np.allclose(ig, [0.6944863967, -0.3474535354, 0.2413017476, 0.4823113438], rtol=0, atol=2e-8)  # survey value came from a float64-product transcription
    assert abs(g[0] + 2 * g[1:].sum() - 1) < 1e-6
    assert np.array_equal(xg, (np.arange(6) * g).astype(np.float32))


def test_polyexp_quadratic_known_answer():
    """I = a + bx + cy + dx^2 + ey^2 + fxy  =>  interior R = (c', b', e, d, f) with first-order
    coefficients taken about the probe pixel (channel order [dy, dx, yy, xx, xy])."""
    h, w = 40, 48
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a, b, c, d, e, f = 3.0, 0.5, -0.25, 0.02, -0.03, 0.015
    I = (a + b * x + c * y + d * x * x + e * y * y + f * x * y).astype(np.float32)
    R = orc.polyexp(I)
    ys, xs = slice(8, h - 8), slice(8, w - 8)
    yy, xx = y[ys, xs], x[ys, xs]
    assert np.allclose(R[0][ys, xs], c + 2 * e * yy + f * xx, atol=2e-4)
    assert np.allclose(R[1][ys, xs], b + 2 * d * xx + f * yy, atol=2e-4)
    assert np.allclose(R[2][ys, xs], e, atol=2e-4)
    assert np.allclose(R[3][ys, xs], d, atol=2e-4)
    assert np.allclose(R[4][ys, xs], f, atol=2e-4)


def test_resize_rules():
    import ctypes as C
    L = orc.lib()

    def table(src, dst):
        i0, i1, f = np.empty(dst, np.int32), np.empty(dst, np.int32), np.empty(dst, np.float32)
        L.orc_resize_table(src, dst, i0.ctypes.data_as(C.c_void_p), i1.ctypes.data_as(C.c_void_p),
                           f.ctypes.data_as(C.c_void_p))
        return i0, i1, f

    i0, i1, f = table(64, 32)          # down by 2: mean of 2d, 2d+1
    assert np.array_equal(i0, 2 * np.arange(32)) and np.array_equal(i1, i0 + 1) and np.all(f == 0.5)
    i0, i1, f = table(64, 16)          # down by 4: 4d+1, 4d+2
    assert np.array_equal(i0, 4 * np.arange(16) + 1) and np.all(f == 0.5)
    i0, i1, f = table(64, 8)           # down by 8: 8d+3, 8d+4
    assert np.array_equal(i0, 8 * np.arange(8) + 3) and np.all(f == 0.5)
    i0, i1, f = table(32, 32)          # same size: copy
    assert np.array_equal(i0, np.arange(32)) and np.all(f == 0)
    i0, i1, f = table(16, 32)          # up by 2: .25/.75 with clamped edges
    assert i0[0] == 0 and f[0] == 0 and f[1] == 0.25 and f[2] == 0.75 and i0[31] == 15 and f[31] == 0


def test_pyr_level0_is_3tap_blur():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    I = orc.pyr_level(img, 0)
    p = np.pad(img.astype(np.float32), 1, mode="reflect")
    hb = 0.5 * p[:, 1:-1] + 0.25 * (p[:, :-2] + p[:, 2:])
    vb = 0.5 * hb[1:-1] + 0.25 * (hb[:-2] + hb[2:])
    assert np.array_equal(I, vb.astype(np.float32))


def test_identical_frames_h_vanishes_in_bounds():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    R = orc.polyexp(orc.pyr_level(img, 0))
    M = orc.update_matrices(R, R, np.zeros((48, 64, 2), np.float32))
    assert np.all(M[3][:-1, :-1] == 0) and np.all(M[4][:-1, :-1] == 0)
    assert np.any(M[3][-1, :] != 0)  # last row takes the out-of-bounds branch (SURVEY A.7 caveat)


@pytest.mark.parametrize("shift", [(2.3, -1.1), (-6.0, 3.5), (0.4, 0.0)])
def test_translation_recovered(shift):
    """interior median flow ~ the true translation (Farneback under-estimates slightly)."""
    w, h = 320, 180
    rng = np.random.default_rng(11)
    a, fx, fy, ph = rng.uniform(5, 25, 12), rng.uniform(0.02, 0.25, 12), rng.uniform(0.02, 0.25, 12), rng.uniform(0, 6.28, 12)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)

    def frame(dx, dy):
        f = 128 + sum(a[i] * np.sin(fx[i] * (x - dx) + fy[i] * (y - dy) + ph[i]) for i in range(12))
        return np.clip(np.rint(f), 0, 255).astype(np.uint8)

    flow = orc.farneback(frame(0, 0), frame(*shift))
    med = np.median(flow[40:-40, 40:-40].reshape(-1, 2), axis=0)
    assert abs(med[0] - shift[0]) <= 0.06 * abs(shift[0]) + 0.05
    assert abs(med[1] - shift[1]) <= 0.06 * abs(shift[1]) + 0.05


def test_steps_compose_to_driver():
    """The step functions the GPU parity tests use reproduce orc_farneback exactly."""
    fr = sine_translate_frames(2, 96, 80, seed=4, amp=(2.0, 1.5), period=7)
    w, h = 96, 80
    levels = orc.num_levels(w, h)
    flow = None
    for k in range(levels, -1, -1):
        lw, lh, _, _ = orc.level_params(w, h, k)
        flow = np.zeros((lh, lw, 2), np.float32) if flow is None else orc.flow_upsample(flow, lw, lh)
        R0, R1 = orc.polyexp(orc.pyr_level(fr[0], k)), orc.polyexp(orc.pyr_level(fr[1], k))
        M = orc.update_matrices(R0, R1, flow)
        for it in range(3):
            flow = orc.blur_solve(M)
            if it < 2:
                M = orc.update_matrices(R0, R1, flow)
    assert np.array_equal(flow, orc.farneback(fr[0], fr[1]))
    d = orc.farneback_dbg(fr[0], fr[1], 0, 3)
    assert np.array_equal(d["out"], flow) and np.array_equal(d["flow"], flow)


def test_bgr2gray_fixed_point():
    rng = np.random.default_rng(9)
    bgr = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    g = orc.bgr2gray(bgr)
    b, gg, r = (bgr[..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(g, ((b * 3735 + gg * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8))
    assert np.array_equal(orc.bgr2gray(np.full((4, 4, 3), 255, np.uint8)), np.full((4, 4), 255, np.uint8))


@pytest.mark.parametrize("w,h", [(40, 33), (16, 16), (97, 50), (15, 70)])
def test_blur_solve_against_exact_window_sums(w, h):
    """FarnebackUpdateFlow_Blur: the oracle's position-anchored summation order (blocks of 16, suffix +
    prefix) is only an ORDER -- every 15x15 REPLICATE-border window sum must equal the plain numpy sum to
    double rounding, for sizes that are / are not multiples of the block and smaller than one block."""
    rng = np.random.default_rng(w * 100 + h)
    M = (rng.standard_normal((5, h, w)) * np.array([4, 1, 4, 2, 2])[:, None, None]).astype(np.float32)
    M[0] = np.abs(M[0]) + 1
    M[2] = np.abs(M[2]) + 1
    P = np.pad(M.astype(np.float64), ((0, 0), (7, 7), (7, 7)), mode="edge")
    col = sum(P[:, j:j + h, :] for j in range(15))
    box = sum(col[:, :, i:i + w] for i in range(15)) / 225.0
    g11, g12, g22, h1, h2 = box
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    want = np.stack([(g11 * h2 - g12 * h1) * idet, (g22 * h1 - g12 * h2) * idet], -1)
    got = orc.blur_solve(M)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7)


def _quadratic(h, w, coef, shift=(0.0, 0.0)):
    a, bx, by, cxx, cyy, cxy = coef
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    x, y = x - shift[0], y - shift[1]
    return (a + bx * x + by * y + cxx * x * x + cyy * y * y + cxy * x * y).astype(np.float32)


@pytest.mark.parametrize("d", [(1.5, -0.75), (-2.25, 0.5), (0.0, 3.0), (0.4, 0.0)])
def test_exact_quadratic_pair_gives_the_displacement_after_one_iteration(d):
    """I1(x) = I0(x - d) for an exact quadratic I0: PolyExp recovers the coefficients exactly in the interior, the
    first-order coefficients differ by 2 A d everywhere, so UpdateMatrices (zero initial flow) yields a CONSTANT
    field M = (G, G d) with G = A^T A, the box blur leaves it unchanged and the regularised 2x2 solve returns
        flow = d * det / (det + 1e-3),   det = (cxx cyy - (cxy/2)^2)^2
    at every interior pixel -- one level, one iteration, no under-estimate.  This pins the signs, the channel
    order ([dy, dx, yy, xx, xy] -> r2..r6), the 0.5 / 0.25 factors of UpdateMatrices, the (g11, g12, g22, h1, h2)
    layout of M, the 1/225 scale and the solve's output order (x-flow first) far tighter than translation
    recovery on a texture (~6 % under-estimate)."""
    h, w = 72, 88
    coef = (40.0, 0.7, -0.4, 0.11, 0.07, 0.05)          # a, bx, by, cxx, cyy, cxy
    s = 6.0                                               # curvatures scaled so that det = 0.077 >> the 1e-3 regulariser
    coef = coef[:3] + tuple(c * s for c in coef[3:])
    I0, I1 = _quadratic(h, w, coef), _quadratic(h, w, coef, shift=d)
    R0, R1 = orc.polyexp(I0), orc.polyexp(I1)
    M = orc.update_matrices(R0, R1, np.zeros((h, w, 2), np.float32))
    flow = orc.blur_solve(M)
    cxx, cyy, r6 = coef[3], coef[4], coef[5] / 2
    det = (cxx * cyy - r6 * r6) ** 2
    want = np.array(d) * det / (det + 1e-3)
    inner = flow[20:-20, 20:-20]                          # 5 (PolyExp) + 5 (border scaling) + 7 (box) pixels from every edge
    assert np.allclose(inner[..., 0], want[0], rtol=2e-4, atol=2e-4), (inner[..., 0].mean(), want[0])
    assert np.allclose(inner[..., 1], want[1], rtol=2e-4, atol=2e-4), (inner[..., 1].mean(), want[1])
    # M itself: channels (g11, g12, g22, h1, h2) = (r4^2 + r6^2, (r4 + r5) r6, r5^2 + r6^2, r4 r2 + r6 r3, r6 r2 + r5 r3)
    r2, r3 = cyy * d[1] + r6 * d[0], cxx * d[0] + r6 * d[1]
    mid = M[:, 30, 40]
    assert np.allclose(mid, [cyy * cyy + r6 * r6, (cyy + cxx) * r6, cxx * cxx + r6 * r6, cyy * r2 + r6 * r3, r6 * r2 + cxx * r3],
                       rtol=3e-4)
    # The exact displacement is a fixed point of the iteration: warping R1 by d cancels the first-order difference
    # (r2 = r3 = 0 before the "+= r4 dy + r6 dx" terms put the whole displacement back), so UpdateMatrices at
    # flow = d reproduces the same M -- h encodes the absolute displacement, not an increment.
    M2 = orc.update_matrices(R0, R1, np.broadcast_to(np.float32(d), (h, w, 2)).copy())
    assert np.allclose(M2[:, 24:-24, 24:-24], M[:, 24:-24, 24:-24], rtol=2e-3, atol=1e-5)


def test_border_rules_reflect101_vs_replicate():
    """Which border each stage uses decides the outermost pixels of every level: GaussianBlur REFLECT_101
    (...cb|abc...), PolyExp and the 15x15 box REPLICATE (...aa|abc...).  Known answers on ramps."""
    h, w = 64, 64
    ramp = np.tile(np.arange(w, dtype=np.uint8)[None, :] * 2, (h, 1))      # I(x) = 2x
    I = orc.pyr_level(ramp, 0)                                             # 3-tap [1/4, 1/2, 1/4], no resampling
    assert np.all(I[:, 1:-1] == ramp[:, 1:-1])                             # a ramp is a fixed point inside
    assert np.all(I[:, 0] == 1.0)        # REFLECT_101: (2 + 0 + 2) / 4 = 1   (REPLICATE would give 0.5)
    assert np.all(I[:, -1] == ramp[0, -1] - 1.0)
    rampy = np.ascontiguousarray(ramp.T)
    Iy = orc.pyr_level(rampy, 0)
    assert np.all(Iy[0, :] == 1.0) and np.all(Iy[1:-1] == rampy[1:-1])
    # box filter: constant G = identity, h1 = ramp in x, h2 = 0  ->  flow_y = mean over the replicated window / (1 + 1e-3)
    M = np.zeros((5, h, w), np.float32)
    M[0] = M[2] = 1.0
    M[3] = np.arange(w, dtype=np.float32)[None, :]
    fy = orc.blur_solve(M)[..., 1]
    x = np.arange(w)
    win = np.clip(x[:, None] + np.arange(-7, 8)[None, :], 0, w - 1).mean(axis=1)    # REPLICATE
    assert np.allclose(fy[10], win / (1.0 + 1e-3), rtol=1e-6)
    assert abs(fy[10, 0] - (28 / 15) / 1.001) < 1e-6                       # 8 copies of 0, then 1..7
    # PolyExp on a ramp: the x-derivative channel is exact inside; with REPLICATE columns it is under-estimated at
    # the edge by a known amount: sum_k xg[k] * (min(k, .) - max(-k, .)) ...
    R = orc.polyexp(np.tile(np.arange(w, dtype=np.float32)[None, :], (h, 1)))
    g, xg, xxg, ig = orc.polyexp_constants()
    assert np.allclose(R[1][20, 8:-8], 1.0, atol=1e-5)
    b2_edge = sum(float(xg[k]) * (k - 0) for k in range(1, 6))            # row(p) - row(m) with m clamped to column 0
    assert abs(R[1][20, 0] - b2_edge * ig[0]) < 1e-5


@pytest.mark.parametrize("w,h,seed", [(96, 80, 3), (160, 128, 4), (320, 256, 5), (131, 97, 6)])
def test_second_restatement_agrees(w, h, seed):
    """The C oracle against tests/np_farneback.py, an independent whole-array numpy transcription of SURVEY Appendix A
    (written from the appendix, not from the C file): every stage and the final flow, on 2-, 3- and 4-scale sizes and
    an odd size.  Element-wise float stages must agree to the last bit or an ulp; the box filter sums in another
    order (double), so the flow gets an absolute tolerance of 2e-5 px.  Parity with cv2 itself stays unpinned."""
    import np_farneback as npf
    fr = sine_translate_frames(2, w, h, seed=seed, amp=(2.5, 1.5), period=5, zoom=0.02)
    # stages first (a failure then names the stage): every pyramid level, PolyExp, UpdateMatrices, blur + solve
    nl = orc.num_levels(w, h)
    for k in range(nl + 1):
        lw, lh, sigma, ksize = orc.level_params(w, h, k)
        I_np = npf.resize_linear(npf.gaussian_blur(fr[0].astype(np.float32), int(ksize), float(sigma)), int(lw), int(lh))
        assert np.max(np.abs(orc.pyr_level(fr[0], k) - I_np)) <= 2e-4, ("pyramid level", k)
    I0, I1 = orc.pyr_level(fr[0], 0), orc.pyr_level(fr[1], 0)
    R0c, R1c = orc.polyexp(I0), orc.polyexp(I1)                       # (5, h, w)
    R0n, R1n = npf.polyexp(I0), npf.polyexp(I1)                       # (h, w, 5)
    assert np.max(np.abs(np.moveaxis(R0c, 0, -1) - R0n)) <= 1e-5 * max(1.0, float(np.max(np.abs(R0n)))), "polyexp"
    rng = np.random.default_rng(seed)
    flow = rng.uniform(-3, 3, (h, w, 2)).astype(np.float32)
    Mn = npf.update_matrices(R0n, R1n, flow)
    Mc = orc.update_matrices(R0c, R1c, flow)
    Mc = np.moveaxis(Mc, 0, -1) if Mc.shape[0] == 5 else Mc
    assert np.max(np.abs(Mc - Mn)) <= 1e-5 * max(1.0, float(np.max(np.abs(Mn)))), "update_matrices"
    want = npf.farneback(fr[0], fr[1])
    got = orc.farneback(fr[0], fr[1])
    assert got.shape == want.shape == (h, w, 2)
    assert float(np.max(np.abs(got - want))) <= 2e-5, float(np.max(np.abs(got - want)))
