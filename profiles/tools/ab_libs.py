"""Same-box, same-process A/B of two BUILDS of libffl_hip.so (kernel code variants): both libraries are loaded side by
side (two copies of the ctypes binding), each gets its own context on the same resident frames, and the two are stepped
alternately; per build the median over the rounds of (wall ms per step, k_blur_solve ms per step) and a check that both
give identical records.  Rounds repeat to ~0.2 %, but every build has its own context, i.e. its own buffer placement, and
identical builds have been seen 1.75 % apart (profiles/r03_c_ab_noise_floor_and_panel_width.txt): trust differences above
~2 %, load a build twice to see the floor, and prefer strip_sweep.py (one context) for launch-geometry options.
Usage: python profiles/tools/ab_libs.py /path/libA.so /path/libB.so [more.so ...]      env WHB=1920,1080,32 STEPS ROUNDS"""
import importlib.util
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402
from funscript_flow_amd.pipeline import SMOOTH_RADIUS  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402


def binding(path, tag):
    spec = importlib.util.spec_from_file_location(f"_capi_{tag}", os.path.join(ROOT, "funscript_flow_amd", "_capi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.LIB_PATH = os.path.abspath(path)
    m.load()
    return m


W, H, B = (int(v) for v in os.environ.get("WHB", "1920,1080,32").split(","))
STEPS, ROUNDS = int(os.environ.get("STEPS", "12")), int(os.environ.get("ROUNDS", "5"))
libs = sys.argv[1:]
frames = sine_translate_frames(B + 1, W, H, seed=1)
runs = []
for i, p in enumerate(libs):
    m = binding(p, i)
    m.set_option("lanes", 1)
    ctx = m.Context(W, H, frame_slots=B + 2, flow_slots=3 * B, max_batch=B)
    ctx.upload_frames(0, list(frames)); ctx.sync()
    runs.append((p, ctx, bench.StepRunner(ctx, B, False, SMOOTH_RADIUS)))
res = {p: [] for p, _, _ in runs}
first = {}
for p, ctx, r in runs:
    r.run(4)
    first[p] = ([tuple(x) for x in r.results[-1][0]], list(r.results[-1][1]))
    r.results.clear()
for k in range(ROUNDS):
    for p, ctx, r in (runs if k % 2 == 0 else runs[::-1]):
        r.run(2); r.results.clear()
        ctx.profile_enable([bench.DOMINANT])
        t0 = time.perf_counter(); r.run(STEPS); dt = time.perf_counter() - t0
        n, ms = ctx.profile_read()[bench.DOMINANT]
        ctx.profile_enable(False)
        res[p].append((dt / STEPS * 1e3, ms / STEPS)); r.results.clear()
base = statistics.median(v[0] for v in res[libs[0]])
same = all(first[p] == first[libs[0]] for p in libs)
for p in libs:
    w = statistics.median(v[0] for v in res[p]); k = statistics.median(v[1] for v in res[p])
    print(f"{os.path.basename(p):34s} step {w:.3f} ms ({B / w * 1e3:.0f} pairs/s, {100 * (base / w - 1):+.2f} % vs first)  k_blur_solve {k:.3f} ms   rounds: "
          + " ".join(f"{v[0]:.3f}" for v in res[p]), flush=True)
print("identical records and scalars across builds:", same)
for _, ctx, _ in runs:
    ctx.close()
