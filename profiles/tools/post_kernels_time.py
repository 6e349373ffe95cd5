import sys, time, numpy as np
sys.path.insert(0, '.')
from funscript_flow_amd import _capi
from funscript_flow_amd.synth import sine_translate_frames
W, H, B = 1920, 1080, 32
fr = sine_translate_frames(B + 1, W, H, seed=1)
_capi.set_option("lanes", 1)
with _capi.Context(W, H, frame_slots=B + 2, flow_slots=B, max_batch=B) as ctx:
    ctx.upload_frames(0, list(fr))
    slots = list(range(B))
    ctx.flow_pairs(list(range(B)), list(range(1, B + 1)), slots)
    recs = ctx.pass1_results(slots, 7.0)
    cs = [(900.3, 500.7)] * B
    ctx.profile_enable(["k_radial", "k_pass1"])
    for _ in range(20):
        d = ctx.radial(slots, cs, [False] * B, False)
    ctx.flow_pairs(list(range(B)), list(range(1, B + 1)), slots)
    ctx.sync()
    p = ctx.profile_read()
    print("k_radial ms per call (32 pairs)", p["k_radial"][1] / p["k_radial"][0], "k_pass1", p["k_pass1"][1] / max(p["k_pass1"][0], 1), d[:2])
