"""Host-side breakdown of the PCIe-inclusive small-image path (pipeline.PairEngine over host ndarrays): wall time spent
inside each binding call per batch, against the batch's total.  python profiles/tools/host_cost_pcie.py [W H B lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from funscript_flow_amd import _capi, pipeline
if os.environ.get("FFL_LIB"):
    _capi.LIB_PATH = os.path.abspath(os.environ["FFL_LIB"])
from funscript_flow_amd.synth import sine_translate_frames

W, H, B, lanes = (int(v) for v in (sys.argv[1:5] + ["256", "256", "256", "2"][len(sys.argv) - 1:]))
MODE = os.environ.get("MODE", "gray")      # gray | bgr | bgr_pinned
_capi.set_option("lanes", lanes)
base = sine_translate_frames(17, W, H, seed=1)
nfr = int(os.environ.get('BATCHES', '8')) * B + 1
if MODE != "gray":
    from funscript_flow_amd.synth import gray_to_bgr
    base = gray_to_bgr(base)
frames = [base[i % 17] for i in range(nfr)]
acc = {}


def timed(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


DEPTH = int(os.environ.get("DEPTH", "2"))
CTXDEPTH = int(os.environ.get("CTXDEPTH", str(DEPTH)))
FS = int(os.environ.get("FRAME_SLOTS", str(pipeline.min_frame_slots(B, CTXDEPTH))))
with _capi.Context(W, H, max_batch=B, frame_slots=FS, flow_slots=pipeline.min_flow_slots(B, CTXDEPTH)) as ctx:
    for n in ("upload_frames", "flow_pairs", "pass1_results", "radial"):
        timed(ctx, n)
    if MODE == "bgr_pinned":
        store = ctx.pinned_frames(nfr, 3)
        for i in range(nfr):
            store[i] = base[i % 17]
        frames = [store[i] for i in range(nfr)]
    if os.environ.get("USE_SLOTS"):
        ctx.frame_slots = int(os.environ["USE_SLOTS"])   # the engine's ring uses fewer slots than the context owns (placement vs ring-size test)
    eng = pipeline.PairEngine(ctx, depth=DEPTH)
    eng.process_chunk(frames[:2 * B + 1])
    stamps = []
    if os.environ.get("TRACE"):
        orig = ctx.pass1_results

        def traced(*a, **k):
            r = orig(*a, **k)
            stamps.append(time.perf_counter())
            return r
        ctx.pass1_results = traced
    for rep in range(3):
        acc.clear()
        stamps.clear()
        t0 = time.perf_counter()
        eng.process_chunk(frames)
        dt = time.perf_counter() - t0
        if stamps:
            print("batch completion deltas ms:", " ".join(f"{1e3 * (b - a):.1f}" for a, b in zip([t0] + stamps[:-1], stamps)))
        nb = (nfr - 1) / B
        print(f"{MODE} depth {DEPTH} ctx {CTXDEPTH} fs {FS} {W}x{H} B={B} lanes={lanes}: {(nfr - 1) / dt:.0f} pairs/s, {1e3 * dt / nb:.3f} ms per batch; inside calls: "
              + ", ".join(f"{k} {1e3 * v / nb:.3f}" for k, v in acc.items())
              + f"; python between calls {1e3 * (dt - sum(acc.values())) / nb:.3f}", flush=True)
