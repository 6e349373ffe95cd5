"""Rate of the input front-end (ffl_upload_frames_raw): decoded frames/s, H2D included, and the
k_frontend launch time (HIP events), next to the CPU oracle's frame time on one host core."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
from funscript_flow_amd import _capi, frontend
import oracle as orc

out = []
for (sw, sh, vr) in [(1920, 1080, False), (3840, 2160, False), (3840, 1920, True), (640, 360, False)]:
    n = 16
    frames = [np.random.default_rng(i).integers(0, 256, (sh, sw, 3), dtype=np.uint8) for i in range(n)]
    with _capi.Context(256, 256, max_batch=8, frame_slots=n) as ctx:
        if os.environ.get("PINNED"):  # as if the decoder wrote into page-locked memory of the context
            pin = ctx.pinned_frames(n, channels=3, size=(sw, sh))
            pin[:] = np.stack(frames)
            up = [pin[i] for i in range(n)]
        else:
            up = frames
        frontend.upload_decoded(ctx, 0, up, vr_mode=vr)
        ctx.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            frontend.upload_decoded(ctx, 0, up, vr_mode=vr)
        ctx.sync()
        dt = (time.perf_counter() - t0) / (reps * n)
        ctx.profile_enable(["k_frontend"])
        frontend.upload_decoded(ctx, 0, up, vr_mode=vr)
        ctx.sync()
        launches, ms = ctx.profile_read()["k_frontend"]
    t0 = time.perf_counter()
    for f in frames[:4]:
        orc.frontend(f, vr_mode=vr)
    cpu = (time.perf_counter() - t0) / 4
    rec = {"source": f"{sw}x{sh}", "vr_mode": vr, "frames_per_s_incl_h2d": 1.0 / dt,
           "h2d_GBps": sw * sh * 3 / dt / 1e9, "k_frontend_us": 1e3 * ms / max(launches, 1),
           "cpu_oracle_ms_per_frame": cpu * 1e3, "pinned": bool(os.environ.get("PINNED"))}
    print(json.dumps(rec), flush=True)
