"""Deterministic synthetic inputs: a 12-sinusoid texture translated along a
sine path ("sine-translate"), optionally with a breathing zoom about the image centre.

    f(x, y)  = 128 + sum_{i<12} a_i sin(fx_i x + fy_i y + phi_i)
    frame_t  = clip(rint(f(x - dx_t, y - dy_t)), 0, 255).astype(uint8)
    dx_t     = Ax sin(2 pi t / T),   dy_t = Ay sin(2 pi t / T + pi / 3)
"""
import numpy as np


def sine_translate_frames(n_frames, width, height, seed=0, amp=(4.0, 4.0), period=16, zoom=0.0, t0=0, workers=None):
    """Return (n_frames, height, width) uint8 gray frames.  Frames are independent of each other, so large clips are
    synthesised by a few threads (numpy releases the GIL); the result does not depend on `workers`."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(5.0, 25.0, 12)
    fx = rng.uniform(0.02, 0.25, 12)
    fy = rng.uniform(0.02, 0.25, 12)
    ph = rng.uniform(0.0, 2.0 * np.pi, 12)
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    out = np.empty((n_frames, height, width), np.uint8)
    ROWS = max(32, 131072 // width)   # rows per block: the float64 temporaries of a block stay in cache (same elementwise arithmetic, same bits)

    def one(i):
        t = t0 + i
        dx = amp[0] * np.sin(2.0 * np.pi * t / period)
        dy = amp[1] * np.sin(2.0 * np.pi * t / period + np.pi / 3.0)
        s = 1.0 + zoom * np.sin(2.0 * np.pi * t / period)
        xrow = (np.arange(width, dtype=np.float64) - cx) / s + cx - dx            # xs[y, x] depends on x only
        for r0 in range(0, height, ROWS):
            r1 = min(r0 + ROWS, height)
            ycol = ((np.arange(r0, r1, dtype=np.float64) - cy) / s + cy - dy)[:, None]   # ys[y, x] depends on y only
            f = np.full((r1 - r0, width), 128.0)
            arg = np.empty((r1 - r0, width))
            for k in range(12):
                np.add(fx[k] * xrow, fy[k] * ycol, out=arg)                      # fx*xs + fy*ys (each product rounded as before)
                arg += ph[k]
                np.sin(arg, out=arg)
                arg *= a[k]
                f += arg
            np.rint(f, out=f)
            np.clip(f, 0, 255, out=f)
            out[i, r0:r1] = f.astype(np.uint8)

    if workers is None:
        import os
        workers = min(8, os.cpu_count() or 1) if (width * height >= (1 << 19) and n_frames > 1) else 1
    if workers > 1 and n_frames > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, range(n_frames)))
    else:
        for i in range(n_frames):
            one(i)
    return out


def gray_to_bgr(gray, gains=(1.0, 1.0, 1.0)):
    """Replicate gray frames to BGR uint8 (..., 3) with optional per-channel gains."""
    g = gray.astype(np.float32)[..., None] * np.asarray(gains, np.float32)
    return np.clip(np.rint(g), 0, 255).astype(np.uint8)
