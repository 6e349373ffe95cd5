#!/usr/bin/env python3
"""PCIe-inclusive rate of the HIP path (never bench.py's `value`): host numpy frames -> per-pair scalars
through pipeline.PairEngine (uploads on the copy stream, two batches in flight, pass 2 lagging by the
+-6 smoothing window).  Usage: python profiles/tools/stream_rate.py [--bgr] [--width W --height H]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from funscript_flow_amd import _capi, pipeline  # noqa: E402
from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=129)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--bgr", action="store_true", help="upload 3-channel BGR frames (gray conversion on the device)")
ap.add_argument("--pinned", action="store_true", help="frames live in ffl_host_alloc memory (zero-copy uploads)")
a = ap.parse_args()

base = sine_translate_frames(17, a.width, a.height, seed=1)
if a.bgr:
    base = gray_to_bgr(base)
frames = [base[i % 17] for i in range(a.frames)]
B = a.batch
with _capi.Context(a.width, a.height, max_batch=B, frame_slots=2 * B + 2, flow_slots=3 * B + 13) as ctx:
    if a.pinned:  # as if a decoder had written the clip into page-locked memory
        pin = ctx.pinned_frames(a.frames, channels=3 if a.bgr else 1)
        for i in range(a.frames):
            pin[i] = frames[i]
        frames = [pin[i] for i in range(a.frames)]
    eng = pipeline.PairEngine(ctx)
    eng.process_chunk(frames[:2 * B + 1])  # warm-up
    t0 = time.perf_counter()
    dots, recs = eng.process_chunk(frames)
    dt = time.perf_counter() - t0
n = len(frames) - 1
print(json.dumps({"pairs": n, "pairs_per_s": n / dt, "seconds": dt, "input": "BGR" if a.bgr else "gray",
                  "h2d_GBps": n / dt * a.width * a.height * (3 if a.bgr else 1) / 1e9,
                  "size": f"{a.width}x{a.height}", "batch": B, "pinned": a.pinned}))
