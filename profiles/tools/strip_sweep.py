"""Same-box, interleaved A/B of k_blur_solve launch geometries (box-to-box and run-to-run spread of the bench is 3-5 %,
larger than the effects looked for): ONE context, resident frames, the configurations visited round-robin several times;
per configuration the median over the rounds of (wall ms per step, k_blur_solve ms per step by HIP events).
Usage: python profiles/tools/strip_sweep.py "rows=8" "rows=17" "minwgs=3000" ...   [W H B via env WHB=1920,1080,32]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import bench
from funscript_flow_amd import _capi
from funscript_flow_amd.pipeline import SMOOTH_RADIUS
from funscript_flow_amd.synth import sine_translate_frames

W, H, B = (int(v) for v in os.environ.get("WHB", "1920,1080,32").split(","))
cfgs = sys.argv[1:] or ["rows=8", "rows=17"]
STEPS, ROUNDS = int(os.environ.get("STEPS", "12")), int(os.environ.get("ROUNDS", "5"))
_capi.set_option("lanes", 1)
frames = sine_translate_frames(B + 1, W, H, seed=1)
ctx = _capi.Context(W, H, frame_slots=B + 2, flow_slots=3 * B, max_batch=B)
ctx.upload_frames(0, list(frames)); ctx.sync()
runner = bench.StepRunner(ctx, B, False, SMOOTH_RADIUS)


def apply(cfg):
    ctx.set_option("blur_rows", 0); ctx.set_option("blur_min_wgs", 3500)   # options belong to the context (ffl_ctx_set_option)
    for kv in cfg.split(","):
        k, v = kv.split("=")
        ctx.set_option({"rows": "blur_rows", "minwgs": "blur_min_wgs", "order": "tile_order", "fuse": "fuse_first"}[k], int(v))


res = {c: [] for c in cfgs}
apply(cfgs[0]); runner.run(4)
for r in range(ROUNDS):
    for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
        apply(c)
        runner.run(2); runner.results.clear()
        ctx.profile_enable([bench.DOMINANT])
        t0 = time.perf_counter(); runner.run(STEPS); dt = time.perf_counter() - t0
        n, ms = ctx.profile_read()[bench.DOMINANT]
        ctx.profile_enable(False)
        res[c].append((dt / STEPS * 1e3, ms / STEPS))
        runner.results.clear()
base = statistics.median(v[0] for v in res[cfgs[0]])
for c in cfgs:
    w = statistics.median(v[0] for v in res[c]); k = statistics.median(v[1] for v in res[c])
    print(f"{c:28s} step {w:.3f} ms ({B / w * 1e3:.0f} pairs/s, {100 * (base / w - 1):+.1f} % vs first)  k_blur_solve {k:.3f} ms   rounds: "
          + " ".join(f"{v[0]:.3f}" for v in res[c]), flush=True)
ctx.close()
