"""oracle/gen_bench_golden.py -- TEST INFRASTRUCTURE (build container only).

Golden scalars for the EXACT workload bench.py times (and the `-m gpu` tests of the shipped configuration):
one step = B consecutive pairs of the synthetic sine-translate stream (funscript_flow_amd.synth, seed 1 at
N = 1), pass 1 per pair, the +-6 centre smoothing inside the batch, pass 2 -- all computed here by the CPU
oracle (C restatement of cv2.calcOpticalFlowFarneback FF:878-879 + numpy restatement of FF:748-785,
FF:1203-1214).  bench.py compares its device results with this file after the timed region and reports
`"checked": true`; the oracle itself never runs on the GPU box for that.

    python oracle/gen_bench_golden.py [W H B [seed]]      default: every golden the repo commits

Writes tests/golden/bench_{W}x{H}_b{B}_s{seed}.json: crc32 of the input frames (the check is skipped, not
failed, if numpy/libm on another machine rounds the synthetic texture differently), per pair (x, y, the f32
bits of div[y, x], mean magnitude), the smoothed centres, the per-pair scalar, and crc32 of the raw flow
field of the first / middle / last pair (bit-exactness of the whole field in 4 bytes).
Data only: inputs are regenerated from the seed, outputs are numbers.
"""
import json
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import oracle as orc  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
RADIUS = 6  # FF:1206


def golden_name(W, H, B, seed):
    return f"bench_{W}x{H}_b{B}_s{seed}.json"


def generate(W, H, B, seed, threads=8):
    frames = sine_translate_frames(B + 1, W, H, seed=seed)
    # the stream has period 16: pair j and pair j + 16 are the same two images -> compute each distinct pair once
    keys, uniq = [], {}
    for j in range(B):
        k = (zlib.crc32(frames[j].tobytes()), zlib.crc32(frames[j + 1].tobytes()))
        keys.append(k)
        uniq.setdefault(k, j)
    orc.lib()

    def one(j):
        flow = orc.farneback(frames[j], frames[j + 1])
        x, y, v = orc.max_divergence_np(flow)
        return flow, int(x), int(y), np.float32(v), np.float32(orc.mean_mag_np(flow))

    with ThreadPoolExecutor(threads) as ex:
        done = dict(zip(uniq.values(), ex.map(one, uniq.values())))
    per = [done[uniq[k]] for k in keys]
    pos = np.array([(p[1], p[2]) for p in per], np.int64)
    jj = np.arange(B)
    lo, hi = np.maximum(0, jj - RADIUS), np.minimum(B, jj + RADIUS + 1)
    psum = np.zeros((B + 1, 2), np.int64)
    psum[1:] = np.cumsum(pos, axis=0)
    centers = (psum[hi] - psum[lo]) / (hi - lo)[:, None]
    assert np.array_equal(centers, np.array(orc.smooth_centers([tuple(p) for p in pos])))
    cuts = [bool(p[4] > 7) for p in per]

    def dot(j):
        return float(orc.radial_np(per[j][0], centers[j], cuts[j], False))

    with ThreadPoolExecutor(threads) as ex:
        dots = list(ex.map(dot, range(B)))
    picks = sorted({0, B // 2 - 1 if B > 1 else 0, B - 1})
    out = {
        "workload": f"{W}x{H}", "pairs_per_step": B, "seed": seed,
        "frames_crc32": zlib.crc32(frames.tobytes()),
        "x": [p[1] for p in per], "y": [p[2] for p in per],
        "val_bits": [int(np.float32(p[3]).view(np.uint32)) for p in per],
        "mean_mag": [float(p[4]) for p in per],
        "cut": cuts,
        "centers": centers.tolist(),
        "dots": dots,
        "flow_crc32": {str(j): zlib.crc32(per[j][0].tobytes()) for j in picks},
        "tolerance": {"argmax": "bit-exact index and value", "mean_mag": 1e-4, "dots": 1e-4, "flow": "bit-exact (crc32)"},
        "made_by": "oracle/gen_bench_golden.py (C oracle farneback + numpy restatement of FF:748-785, FF:1203-1214)",
    }
    path = os.path.join(GOLD, golden_name(W, H, B, seed))
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print(path, "dots[0..2] =", dots[:3])


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        generate(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    else:  # every golden the repo commits (about 8 minutes on 8 cores)
        for seed in (1, 10, 11, 12, 13, 14, 15, 16, 17):     # N = 1 and the clips of ranks 0..7 of bench.py --gpus N
            generate(1920, 1080, 32, seed)
        generate(3840, 2160, 32, 1)
        generate(2880, 2880, 32, 2)                           # one eye of configs[4] (bench.py's large_image extra)
        generate(256, 256, 64, 1)
        generate(256, 256, 256, 1)
