"""Where the host time of one small-image batch goes: wall time of the asynchronous ffl_flow_pairs call
(26 kernel launches), of the result read and of the synchronous pass-2 call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from funscript_flow_amd import _capi
from funscript_flow_amd.synth import sine_translate_frames

if os.environ.get("TORCH_INIT"):
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_capi.set_option("lanes", int(sys.argv[3]) if len(sys.argv) > 3 else 1)
fr = sine_translate_frames(B + 1, W, H, seed=1)
with _capi.Context(W, H, frame_slots=B + 2, flow_slots=3 * B, max_batch=B) as ctx:
    ctx.upload_frames(0, list(fr))
    f0, f1 = list(range(B)), list(range(1, B + 1))
    cs = [(W / 2, H / 2)] * B
    for rep in range(3):
        t = {"enqueue": 0.0, "sync": 0.0, "results": 0.0, "radial": 0.0}
        n = 30
        T0 = time.perf_counter()
        for s in range(n):
            slots = [(s % 3) * B + i for i in range(B)]
            a = time.perf_counter(); ctx.flow_pairs(f0, f1, slots)
            b = time.perf_counter(); t["enqueue"] += b - a
            ctx.sync(); c = time.perf_counter(); t["sync"] += c - b
            recs = ctx.pass1_results(slots); d = time.perf_counter(); t["results"] += d - c
            ctx.radial(slots, cs, [False] * B); e = time.perf_counter(); t["radial"] += e - d
        tot = time.perf_counter() - T0
        print(f"{W}x{H} B={B}: per batch ms: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in t.items()) + f", total {1e3 * tot / n:.3f}", flush=True)

# pipelined (what bench.py and PairEngine do): DEPTH batches queued ahead of the one being finalised
DEPTH = int(os.environ.get("DEPTH", "2"))
with _capi.Context(W, H, frame_slots=B + 2, flow_slots=(DEPTH + 1) * B, max_batch=B) as ctx:
    ctx.upload_frames(0, list(fr))
    for rep in range(3):
        t = {"enqueue": 0.0, "results": 0.0, "radial": 0.0}
        n = 30
        pending = []
        T0 = time.perf_counter()
        for s in range(n):
            slots = [(s % (DEPTH + 1)) * B + i for i in range(B)]
            a = time.perf_counter(); ctx.flow_pairs(f0, f1, slots); t["enqueue"] += time.perf_counter() - a
            pending.append(slots)
            if len(pending) > DEPTH:
                sl = pending.pop(0)
                a = time.perf_counter(); ctx.pass1_results(sl); b = time.perf_counter(); t["results"] += b - a
                ctx.radial(sl, cs, [False] * B); t["radial"] += time.perf_counter() - b
        ctx.sync()
        tot = time.perf_counter() - T0
        print(f"pipelined depth {DEPTH}: per batch ms: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in t.items()) + f", total {1e3 * tot / n:.3f}", flush=True)
