"""The input front-end oracle (oracle/frontend_oracle.c): cv2.resize INTER_LINEAR u8 + cvtColor luma.

cv2 is not available here and the reference holds no image fixture, so this restatement is **parity
unpinned** against OpenCV; these tests pin it against an independent numpy statement of the same
published rules and against analytic known answers.
"""
import numpy as np
import pytest

import oracle as orc


def np_resize_linear(img, dw, dh):
    """Independent restatement (vectorised numpy) of resize INTER_LINEAR for uint8, 3 channels."""
    sh, sw, _ = img.shape
    if (sw, sh) == (dw, dh):
        return img.copy()
    if (sw, sh) == (2 * dw, 2 * dh):
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def table(src, dst, clamp_frac):
        scale = 1.0 / (np.float64(dst) / np.float64(src))
        f = ((np.arange(dst) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_frac:
            lo, hi = s < 0, s >= src - 1
            f[lo | hi] = 0
            s[lo] = 0
            s[hi] = src - 1
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return s, w0, w1

    sx, a0, a1 = table(sw, dw, True)
    sy, b0, b1 = table(sh, dh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    s = img.astype(np.int64)
    h0 = s[y0][:, sx] * a0[None, :, None] + s[y0][:, sx1] * a1[None, :, None]
    h1 = s[y1][:, sx] * a0[None, :, None] + s[y1][:, sx1] * a1[None, :, None]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1920, 1080, 256, 256), (640, 360, 256, 256), (517, 333, 256, 256),
                                         (120, 100, 256, 256), (3840, 1920, 512, 512), (1024, 1024, 512, 512),
                                         (512, 512, 256, 256), (256, 256, 256, 256), (1280, 720, 320, 180),
                                         (37, 23, 64, 48), (2, 2, 16, 16), (1, 5, 16, 16)])
def test_resize_matches_independent_numpy_statement(sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 31 + sh)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear_u8c3(img, dw, dh), np_resize_linear(img, dw, dh))


def test_resize_known_answers():
    c = np.full((333, 517, 3), 200, np.uint8)
    assert np.unique(orc.resize_linear_u8c3(c, 256, 256)).tolist() == [200]        # weights sum to 2048
    img = np.random.default_rng(1).integers(0, 256, (64, 48, 3), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear_u8c3(img, 48, 64), img)                  # same size: untouched
    s = img.astype(int)
    mean2 = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(orc.resize_linear_u8c3(img, 24, 32), mean2)               # exact 2x: 2x2 mean, half up
    # 4x down-scale of a horizontal ramp samples midway between source pixels 4d+1 and 4d+2
    ramp = np.repeat((np.arange(64, dtype=np.uint8) * 2)[None, :, None], 16, 0).repeat(3, 2)
    out = orc.resize_linear_u8c3(ramp, 16, 4)
    assert np.array_equal(out[0, :, 0], ((ramp[0, 1::4, 0].astype(int) + ramp[0, 2::4, 0] + 1) >> 1))
    # a strided view (crop of a larger frame) is read through its row pitch
    big = np.random.default_rng(2).integers(0, 256, (90, 160, 3), dtype=np.uint8)
    view = big[10:70, 20:140]
    assert np.array_equal(orc.resize_linear_u8c3(view, 32, 32), orc.resize_linear_u8c3(np.ascontiguousarray(view), 32, 32))


def test_luma_known_answers():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 20, 30]]], np.uint8)
    assert orc.rgb2gray(px)[0].tolist() == [76, 150, 29, 255, 0, (10 * 9798 + 20 * 19235 + 30 * 3735 + 16384) >> 15]
    # RGB2GRAY of the swapped frame == BGR2GRAY of the frame (the existing k_gray rule)
    f = np.random.default_rng(3).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    assert np.array_equal(orc.rgb2gray(orc.swap_rb(f)), orc.bgr2gray(f))


def test_frontend_composition_follows_the_reference_steps():
    f = np.random.default_rng(4).integers(0, 256, (270, 480, 3), dtype=np.uint8)
    rgb = f[:, :, ::-1]
    assert np.array_equal(orc.frontend(f), orc.rgb2gray(np_resize_linear(np.ascontiguousarray(rgb), 256, 256)))
    vr = np_resize_linear(np.ascontiguousarray(rgb), 512, 512)[256:, :256]
    assert np.array_equal(orc.frontend(f, vr_mode=True), orc.rgb2gray(np.ascontiguousarray(vr)))
    same = np.random.default_rng(5).integers(0, 256, (256, 256, 3), dtype=np.uint8)
    assert np.array_equal(orc.frontend(same), orc.bgr2gray(same))                    # FF:185: no resize needed
