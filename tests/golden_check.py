"""Compare one finished batch of the HIP path with committed golden scalars (tests/golden/bench_*.json, made
in the build container by oracle/gen_bench_golden.py).  Used by bench.py after its timed region and by the
`-m gpu` tests of the shipped configuration; this module only reads numbers -- it never runs a CPU version
of the path.  Test infrastructure: it lives beside the goldens it reads, outside the product package
(bench.py puts tests/ on its import path for it).

Bar (BASELINE north_star): argmax pixel index and value bit-exact, float reductions within 1e-4 relative,
flow fields bit-exact (crc32 of the raw float32 bytes of the first / middle / last pair)."""
import json
import os
import zlib

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_path(width, height, batch, seed):
    return os.path.join(GOLDEN_DIR, f"bench_{width}x{height}_b{batch}_s{seed}.json")


def load_golden(width, height, batch, seed):
    p = golden_path(width, height, batch, seed)
    return json.load(open(p)) if os.path.exists(p) else None


def check_batch(golden, frames, recs, dots, download_flow=None):
    """recs: the batch's (x, y, val, mean_mag, cut) tuples; dots: its pass-2 scalars; download_flow(j) returns the
    (H, W, 2) float32 flow of pair j (optional).  Returns (status, detail): status True (all equal), False (a
    mismatch: detail says which), or None (not comparable: the synthetic frames differ from the ones the golden was
    made from)."""
    if golden is None:
        return None, "no golden file for this workload"
    if zlib.crc32(np.ascontiguousarray(frames).tobytes()) != golden["frames_crc32"]:
        return None, "synthetic frames differ from the golden's (numpy/libm rounding): not comparable"
    B = golden["pairs_per_step"]
    if len(recs) != B or len(dots) != B:
        return False, f"batch of {len(recs)} pairs, golden has {B}"
    for j, r in enumerate(recs):
        if (int(r[0]), int(r[1])) != (golden["x"][j], golden["y"][j]):
            return False, f"pair {j}: argmax ({r[0]}, {r[1]}) != golden ({golden['x'][j]}, {golden['y'][j]})"
        if int(np.float32(r[2]).view(np.uint32)) != golden["val_bits"][j]:
            return False, f"pair {j}: argmax value bits differ"
        if abs(float(r[3]) - golden["mean_mag"][j]) > 1e-4 * golden["mean_mag"][j]:
            return False, f"pair {j}: mean_mag {r[3]} vs {golden['mean_mag'][j]}"
        if bool(r[4]) != golden["cut"][j]:
            return False, f"pair {j}: cut flag"
    g = np.asarray(golden["dots"], np.float64)
    scale = float(np.mean(np.abs(g)))
    err = np.abs(np.asarray(dots, np.float64) - g)
    if np.any(err > 1e-4 * np.maximum(np.abs(g), scale)):
        j = int(np.argmax(err))
        return False, f"pair {j}: pass-2 scalar {dots[j]} vs {g[j]}"
    n_flow = 0
    if download_flow is not None:
        for key, crc in golden["flow_crc32"].items():
            if zlib.crc32(np.ascontiguousarray(download_flow(int(key))).tobytes()) != crc:
                return False, f"pair {key}: flow field differs from the oracle's (crc32)"
            n_flow += 1
    return True, f"{B} records + {B} scalars + {n_flow} flow fields equal the oracle goldens"
