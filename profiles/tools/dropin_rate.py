"""Throughput of the reference-shaped drop-in calls (backend.precompute_all + radial_all, the replacements of the two pools
of process_video, FF:1190-1191 / FF:1232-1236) on one chunk of 3000 frames.   python profiles/tools/dropin_rate.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from funscript_flow_amd import backend, pipeline
from funscript_flow_amd.synth import sine_translate_frames

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
N = int(os.environ.get("FRAMES", "3000"))
base = sine_translate_frames(17, W, H, seed=1)
frames = [base[i % 17][:] for i in range(N)]             # one ndarray object per decoded frame, as cv2 hands them over
params = {"backend": "HIP", "pov_mode": False, "cut_threshold": 7}
if os.environ.get("HIP_BATCH"):
    params["hip_batch"] = int(os.environ["HIP_BATCH"])
for rep in range(3):
    t0 = time.perf_counter()
    pre = backend.precompute_all(list(zip(frames[:-1], frames[1:])), params)
    t1 = time.perf_counter()
    centers = pipeline.smooth_centers([p["pos_center"] for p in pre])
    t2 = time.perf_counter()
    dots = backend.radial_all(pre, centers, False)
    t3 = time.perf_counter()
    n = N - 1
    print(f"{W}x{H} {n} pairs: precompute_all {n / (t1 - t0):.0f} pairs/s ({1e3 * (t1 - t0):.1f} ms), smooth_centers {1e3 * (t2 - t1):.1f} ms, "
          f"radial_all {n / (t3 - t2):.0f} pairs/s ({1e3 * (t3 - t2):.1f} ms); whole chunk {n / (t3 - t0):.0f} pairs/s", flush=True)
