"""Host post-chain: per-pair scalars -> .funscript actions.

Counterpart of the tail of process_video (FunscriptFlow.pyw:1266-1394), kept as plain numpy on the
host: it touches one float per pair, so there is nothing here for the GPU.  Every stage reproduces
the reference's arithmetic order and its quirks (noted inline); the whole chain is pinned by
tests/golden/chain_golden.* captured from the real process_video.

    dots, cuts, frame_idx --integrate--> cum --detrend--> --smooth5--> --rolling_normalise--> 0..100
                          --keyframes--> indices --to_actions--> [{"at": ms, "pos": 0..100}]
"""
import json
import math

import numpy as np

DISCONTINUITY = 1000.0  # FF:1289 (hard-coded jump threshold between detrend segments)
SMOOTH_TAPS = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])  # FF:1333


def sampling(fps, total_frames):
    """FF:1127-1129: frame step so that the effective rate is <= 30 fps."""
    step = max(1, int(math.ceil(fps / 30.0)))
    return step, fps / step, list(range(0, total_frames, step))


def integrate(dots, cuts):
    """FF:1267-1284.  Midpoint (trapezoid) integration restarted at cuts, then the half-step shift
    that averages each sample with its *unshifted* predecessor."""
    n = len(dots)
    cum = [0.0] * n
    for i in range(1, n):
        cum[i] = 0.0 if cuts[i] else cum[i - 1] + (dots[i - 1] + dots[i]) / 2
    return np.array([cum[0]] + [(cum[i] + cum[i - 1]) / 2 for i in range(1, n)], dtype=np.float64)


def _linear_detrend(seg):
    x = np.arange(len(seg))
    return seg - np.polyval(np.polyfit(x, seg, 1), x)


def detrend(cum, window):
    """FF:1287-1331.  Hann-weighted overlap-add of linearly detrended windows inside each segment
    between jumps > 1000.  Reference quirks kept: segments shorter than 5 samples are mean-centred and
    written with weight 0, so the final division by max(weight, 1e-6) scales them by 1e6; samples
    covered only by Hann end points (weight 0, numerator 0) come out as 0."""
    cum = np.asarray(cum, np.float64)
    out = np.zeros_like(cum)
    wsum = np.zeros_like(cum)
    jumps = np.where(np.abs(np.diff(cum)) > DISCONTINUITY)[0] + 1
    bounds = [0] + list(jumps) + [len(cum)]
    overlap = window // 2
    for a, b in zip(bounds[:-1], bounds[1:]):
        n = b - a
        if n < 5:
            out[a:b] = cum[a:b] - np.mean(cum[a:b])
        elif n <= window:
            w = np.hanning(n)
            out[a:b] += _linear_detrend(cum[a:b]) * w
            wsum[a:b] += w
        else:
            for s in range(a, b - overlap, overlap):
                e = min(s + window, b)
                w = np.hanning(e - s)
                out[s:e] += _linear_detrend(cum[s:e]) * w
                wsum[s:e] += w
    return out / np.maximum(wsum, 1e-6)


def smooth5(x):
    """FF:1333: binomial 5-tap, 'same' length."""
    return np.convolve(x, SMOOTH_TAPS, mode="same")


def rolling_normalise(x, window):
    """FF:1335-1349: min-max to 0..100 inside a centred window (made odd); 50 where the window is flat."""
    if window % 2 == 0:
        window += 1
    half = window // 2
    out = np.empty_like(x)
    for i in range(len(x)):
        win = x[max(0, i - half):min(len(x), i + half + 1)]
        lo, hi = win.min(), win.max()
        out[i] = 50 if hi - lo == 0 else (x[i] - lo) / (hi - lo) * 100
    return out


def keyframes(x, reduce=True):
    """FF:1366-1376: first, last and every slope-sign inversion; every index when reduction is off."""
    if not reduce:
        return list(range(len(x)))
    keep = [0]
    for i in range(1, len(x) - 1):
        if ((x[i] - x[i - 1]) < 0) != ((x[i + 1] - x[i]) < 0):
            keep.append(i)
    keep.append(len(x) - 1)
    return keep


def to_actions(values, frame_idx, fps, key_indices):
    """FF:1377-1385: at = int(frame / fps * 1000) ms, pos = 100 - round(value)."""
    return [{"at": int((frame_idx[k] / fps) * 1000), "pos": 100 - int(round(values[k]))} for k in key_indices]


def actions_from_scalars(dots, cuts, frame_idx, fps, params):
    """The whole chain for one video: params carries detrend_window / norm_window (seconds) and
    keyframe_reduction, exactly like the reference's settings dict (FF:2644-2662)."""
    if len(dots) == 0:
        return []  # no pair at all (a video of fewer than 2 sampled frames): the reference fails at FF:1268; nothing to write
    step = max(1, int(math.ceil(fps / 30.0)))
    effective_fps = fps / step
    cum = integrate(dots, cuts)
    flat = detrend(cum, int(params["detrend_window"] * effective_fps))
    norm = rolling_normalise(smooth5(flat), int(params["norm_window"] * effective_fps))
    return to_actions(norm, frame_idx, fps, keyframes(norm, params.get("keyframe_reduction", True)))


def write_funscript(path, actions):
    """FF:1391-1394."""
    with open(path, "w") as f:
        json.dump({"version": "1.0", "actions": actions}, f, indent=2)
