"""Infinity-Cache-resident sub-batches (round-3 verdict item 3): does a batch whose working set fits the 256 MiB MALL,
run on several compute lanes so that the grid still fills 256 CUs, beat the large streaming batches?

Every configuration (frame size, B pairs per batch, lanes, optional blur_min_wgs) gets its own context on the same
resident frames; configurations are visited round-robin (order reversed every other round), graph replay, no per-kernel
events; per configuration the median pairs/s over the rounds.  Records of pair 0 are compared across configurations.

    python profiles/tools/mall_subbatch.py 256 256  "256x2" "16x2" "16x3" "16x4" "32x2" "32x3" "32x4" "48x2" "48x3" "48x4"
    python profiles/tools/mall_subbatch.py 1920 1080 "32x1" "32x2" "1x4:500" "2x4:500" "2x4:1000" "4x4:1000"
config = B x lanes [: blur_min_wgs]          env PAIRS (pairs per timed visit, default 2048 at 256^2 / 256 at 1080p), ROUNDS
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (its HIP runtime first)
import bench  # noqa: E402
from funscript_flow_amd import _capi  # noqa: E402
from funscript_flow_amd.pipeline import SMOOTH_RADIUS  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

W, H = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:]
small = W * H <= 640 * 360
PAIRS = int(os.environ.get("PAIRS", "2048" if small else "256"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))


def parse(c):
    bl, _, mw = c.partition(":")
    b, l = bl.split("x")
    return int(b), int(l), int(mw) if mw else 0


maxB = max(parse(c)[0] for c in cfgs)
frames = sine_translate_frames(maxB + 1, W, H, seed=1)
runs = {}
for c in cfgs:
    B, lanes, mw = parse(c)
    _capi.set_option("lanes", lanes)
    depth = lanes + 1                                    # batches queued ahead: every lane busy + one being finalised
    ctx = _capi.Context(W, H, frame_slots=B + 2, flow_slots=(depth + 1) * B, max_batch=B)
    if mw:
        ctx.set_option("blur_min_wgs", mw)
    ctx.upload_frames(0, list(frames[:B + 1]))
    ctx.sync()
    r = bench.StepRunner(ctx, B, False, SMOOTH_RADIUS, depth)
    r.run(max(3, 2 * depth))
    runs[c] = (ctx, r, B, max(2 * depth, PAIRS // B))
    r.results.clear()
_capi.set_option("lanes", 2)
first = {}
res = {c: [] for c in cfgs}
for k in range(ROUNDS):
    for c in (cfgs if k % 2 == 0 else cfgs[::-1]):
        ctx, r, B, steps = runs[c]
        r.run(2)
        r.results.clear()
        t0 = time.perf_counter()
        r.run(steps)
        dt = time.perf_counter() - t0
        res[c].append(steps * B / dt)
        first[c] = (tuple(r.results[-1][0][0]), r.results[-1][1][0]) if B == maxB or True else None
        r.results.clear()
base = statistics.median(res[cfgs[0]])
print(f"{W}x{H}: pairs/s, median of {ROUNDS} interleaved visits of >= {PAIRS} pairs each (graph replay, no events); config = B x lanes [: blur_min_wgs]")
for c in cfgs:
    m = statistics.median(res[c])
    gs = runs[c][0].graph_stats()
    print(f"{c:12s} {m:10.0f} pairs/s  {100 * (m / base - 1):+6.1f} % vs {cfgs[0]}   visits: " + " ".join(f"{v:.0f}" for v in res[c])
          + f"   graphs {gs['captured']}/{gs['replayed']}/{gs['capture_failures']}", flush=True)
# pair 0 of a batch is the same two frames in every configuration; its window (pairs 0..6) is too whenever B >= 7
ok = len({first[c][0] for c in cfgs}) == 1
print("pair-0 pass-1 record identical across configurations:", ok)
for c in cfgs:
    runs[c][0].close()
