#!/usr/bin/env python3
"""Read the in-kernel stamps of a -DFFL_STAMP build (diagnostic only): share of the level-0 folded k_blur_solve launch's
workgroup time per phase, as seen by thread 0 of 64 sampled workgroups.  Usage: built + run by profiles/tools/k5_stamps.sh"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from funscript_flow_amd import _capi  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

W, H, B = 1920, 1080, 32
fr = sine_translate_frames(B + 1, W, H, seed=1)
_capi.set_option("lanes", 1)
_capi.set_option("graph", 0)
with _capi.Context(W, H, frame_slots=B + 2, flow_slots=B, max_batch=B) as ctx:
    ctx.upload_frames(0, list(fr))
    for _ in range(3):
        ctx.flow_pairs(list(range(B)), list(range(1, B + 1)), list(range(B)))
    ctx.sync()
    out = (C.c_ulonglong * (64 * 16))()
    L = _capi.load()
    L.ffl_debug_read_stamps.argtypes = [C.c_void_p]
    assert L.ffl_debug_read_stamps(out) == 0
a = np.array(out, np.float64).reshape(64, 16)[:, :12]
names = ["loop", "U work", "wait U", "wait H(g0)", "V work", "wait V", "H work", "wait solve", "solve work", "wait solve2",
         "store+UM", "wait end"]
tot = a.sum(axis=1, keepdims=True)
share = (a / tot).mean(axis=0)
print("mean cycles per workgroup (strip of tiles):", int(tot.mean()))
for n, s in zip(names, share):
    print(f"  {n:12s} {100 * s:5.1f} %")
print("  work total  %5.1f %%   barrier waits %5.1f %%" % (100 * share[[1, 4, 6, 8, 10]].sum(), 100 * share[[2, 3, 5, 7, 9, 11]].sum()))
