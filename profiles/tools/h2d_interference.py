"""Does a concurrent pinned H2D stream slow the kernels down?  The resident 1080p B = 32 step (2 lanes, graph replay) alone,
then with a second thread that keeps uploading frames into ANOTHER context of the same device out of page-locked memory:
  gray   2 MB frames, no kernel on the copy stream        bgr   6 MB frames + k_gray on the copy stream
python profiles/tools/h2d_interference.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import bench
from funscript_flow_amd import _capi
from funscript_flow_amd.pipeline import SMOOTH_RADIUS
from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames

W, H, B = 1920, 1080, 32
_capi.set_option("lanes", 2)
frames = sine_translate_frames(B + 1, W, H, seed=1)
ctx = _capi.Context(W, H, frame_slots=B + 2, flow_slots=3 * B, max_batch=B)
ctx.upload_frames(0, list(frames)); ctx.sync()
runner = bench.StepRunner(ctx, B, False, SMOOTH_RADIUS)
runner.run(6); runner.results.clear()


def rate(steps=40):
    t0 = time.perf_counter(); runner.run(steps); dt = time.perf_counter() - t0
    runner.results.clear()
    return steps * B / dt


print(f"alone:            {rate():.0f} pairs/s", flush=True)
for ch, name in ((1, "gray"), (3, "bgr")):
    up = _capi.Context(W, H, frame_slots=B + 2, flow_slots=1, max_batch=1)
    store = up.pinned_frames(B + 1, ch)
    src = frames if ch == 1 else gray_to_bgr(frames)
    for i in range(B + 1):
        store[i] = src[i]
    fl = [store[i] for i in range(B + 1)]
    stop = threading.Event()
    sent = [0]

    def feeder():
        while not stop.is_set():
            up.upload_frames(0, fl)
            up.sync()
            sent[0] += 1

    th = threading.Thread(target=feeder); th.start()
    time.sleep(0.2)
    s0, t0 = sent[0], time.perf_counter()
    r = rate()
    dt = time.perf_counter() - t0
    gbps = (sent[0] - s0) * (B + 1) * W * H * ch / dt / 1e9
    stop.set(); th.join(); up.close()
    print(f"with {name:4s} uploads: {r:.0f} pairs/s  (feeder moved {gbps:.1f} GB/s meanwhile)", flush=True)
print(f"alone again:      {rate():.0f} pairs/s")
ctx.close()
