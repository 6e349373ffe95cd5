"""Deterministic synthetic inputs: a 12-sinusoid texture translated along a
sine path ("sine-translate"), optionally with a breathing zoom about the image centre.

    f(x, y)  = 128 + sum_{i<12} a_i sin(fx_i x + fy_i y + phi_i)
    frame_t  = clip(rint(f(x - dx_t, y - dy_t)), 0, 255).astype(uint8)
    dx_t     = Ax sin(2 pi t / T),   dy_t = Ay sin(2 pi t / T + pi / 3)
"""
import numpy as np


def sine_translate_frames(n_frames, width, height, seed=0, amp=(4.0, 4.0), period=16, zoom=0.0, t0=0):
    """Return (n_frames, height, width) uint8 gray frames."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(5.0, 25.0, 12)
    fx = rng.uniform(0.02, 0.25, 12)
    fy = rng.uniform(0.02, 0.25, 12)
    ph = rng.uniform(0.0, 2.0 * np.pi, 12)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    out = np.empty((n_frames, height, width), np.uint8)
    for i in range(n_frames):
        t = t0 + i
        dx = amp[0] * np.sin(2.0 * np.pi * t / period)
        dy = amp[1] * np.sin(2.0 * np.pi * t / period + np.pi / 3.0)
        s = 1.0 + zoom * np.sin(2.0 * np.pi * t / period)
        xs = (x - cx) / s + cx - dx
        ys = (y - cy) / s + cy - dy
        f = np.full((height, width), 128.0)
        for k in range(12):
            f += a[k] * np.sin(fx[k] * xs + fy[k] * ys + ph[k])
        out[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def gray_to_bgr(gray, gains=(1.0, 1.0, 1.0)):
    """Replicate gray frames to BGR uint8 (..., 3) with optional per-channel gains."""
    g = gray.astype(np.float32)[..., None] * np.asarray(gains, np.float32)
    return np.clip(np.rint(g), 0, 255).astype(np.uint8)
