#!/usr/bin/env python3
"""Audit of the compiler's s_waitcnt placement: for every kernel in a gfx950 .s file, how many global loads were
issued between consecutive vmcnt waits (the batch a wave has in flight when it stops).  Many batches of 1 inside a
loop = a serialised chain of memory round trips (what k_pass1 and phase U of k_blur_solve suffered from).
Usage: hipcc -S --cuda-device-only ... -o k.s ; python profiles/tools/wait_audit.py k.s"""
import re, sys
from collections import Counter
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    fn, hist, batch = None, Counter(), 0
    out = {}
    for l in lines:
        m = re.match(r"(_Z\w+):", l)
        if m:
            fn, hist, batch = m.group(1), Counter(), 0
            out[fn] = hist
            continue
        if fn is None:
            continue
        t = l.strip()
        if t.startswith(("global_load", "flat_load", "buffer_load")):
            batch += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if batch:
                hist[(batch, "all" if n == 0 else "partial")] += 1
            if n == 0:
                batch = 0
        elif t.startswith(".size"):
            fn = None
    for k, h in out.items():
        if not h:
            continue
        ones = sum(v for (b, kind), v in h.items() if b == 1 and kind == "all")
        tot = sum(h.values())
        print(f"{k[:70]:70s} waits {tot:4d}  of which 1-load-then-wait-all {ones:3d}   batches {dict(sorted(h.items()))}")
