#!/usr/bin/env python3
"""Rate of the reference-shaped one-pair API (backend.precompute_flow_info + radial_motion_weighted per pair, as the
reference's call sites use it) against pipeline.PairEngine on the same frames.  Quoted in INTEGRATION.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from funscript_flow_amd import _capi, backend, pipeline  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

for (w, h, n, B) in ((256, 256, 257, 256), (1920, 1080, 65, 32)):
    fr = sine_translate_frames(17, w, h, seed=1)
    frames = [fr[i % 17] for i in range(n)]
    cfg = {"backend": "HIP"}
    for j in range(3):
        backend.precompute_flow_info(frames[j], frames[j + 1], cfg)
    m = min(n - 1, 64)
    t0 = time.perf_counter()
    for j in range(m):
        info = backend.precompute_flow_info(frames[j], frames[j + 1], cfg)
        backend.radial_motion_weighted(info["flow"], np.array(info["pos_center"], float), info["cut"])
    per_pair = m / (time.perf_counter() - t0)
    backend.release_contexts()
    with _capi.Context(w, h, max_batch=B, frame_slots=2 * B + 2, flow_slots=pipeline.min_flow_slots(B)) as ctx:
        eng = pipeline.PairEngine(ctx)
        eng.process_chunk(frames)
        t0 = time.perf_counter()
        eng.process_chunk(frames)
        engine = (n - 1) / (time.perf_counter() - t0)
    # the reference's two pool calls as two batched calls (backend.precompute_all / radial_all) with its own chunk code between
    pairs = list(zip(frames[:-1], frames[1:]))
    params = {"backend": "HIP", "hip_batch": B}
    backend.precompute_all(pairs, params)  # warm-up: creates the chunk-sized context
    t0 = time.perf_counter()
    pre = backend.precompute_all(pairs, params)
    centers = pipeline.smooth_centers([i["pos_center"] for i in pre])
    backend.radial_all(pre, centers, False)
    batched = (n - 1) / (time.perf_counter() - t0)
    backend.release_contexts()
    print(json.dumps({"size": f"{w}x{h}", "per_pair_api_pairs_per_s": per_pair, "pair_engine_pairs_per_s": engine,
                      "precompute_all_radial_all_pairs_per_s": batched, "ratio": engine / per_pair, "engine_batch": B}))
