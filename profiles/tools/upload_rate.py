"""What bounds the PCIe-inclusive BGR rate?  (1) upload only (no flow kernels): pageable vs page-locked frames, by
copy_threads; (2) the full path by copy_threads.  1080p BGR frames, batches of 33 frames as PairEngine sends them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (HIP runtime first)
import numpy as np
import bench
from funscript_flow_amd import _capi
from funscript_flow_amd.synth import gray_to_bgr, sine_translate_frames

W, H, B = 1920, 1080, 32
base = sine_translate_frames(17, W, H, seed=1)
bgr = gray_to_bgr(base)
_capi.set_option("lanes", 2)
for n in (1, 2, 4, 8):
    _capi.set_option("copy_threads", n)
    with _capi.Context(W, H, max_batch=B, frame_slots=2 * B + 2, flow_slots=2 * B + 13) as ctx:
        store = ctx.pinned_frames(33, 3)
        for i in range(33):
            store[i] = bgr[i % 17]
        pin = [store[i] for i in range(33)]
        page = [bgr[i % 17] for i in range(33)]
        gpage = [base[i % 17] for i in range(33)]
        for name, fr, ch in (("bgr pageable", page, 3), ("bgr pinned", pin, 3), ("gray pageable", gpage, 1)):
            ctx.upload_frames(0, fr); ctx.sync()
            t0 = time.perf_counter()
            for r in range(8):
                ctx.upload_frames((r % 2) * 33, fr)
            ctx.sync()
            dt = time.perf_counter() - t0
            print(f"copy_threads {n}: upload only, {name}: {8 * 33 / dt:.0f} frames/s = {8 * 33 * W * H * ch / dt / 1e9:.1f} GB/s", flush=True)
        store = pin = None
    r = [bench.pcie_inclusive(W, H, B, 0, 1, 8 * B + 1, True, base)["value"] for _ in range(2)]
    print("copy_threads", n, "full path BGR pageable pairs/s", [round(v) for v in r], flush=True)
r = [bench.pcie_inclusive(W, H, B, 0, 1, 8 * B + 1, True, base, pinned=True)["value"] for _ in range(2)]
print("full path BGR pinned pairs/s", [round(v) for v in r], flush=True)
