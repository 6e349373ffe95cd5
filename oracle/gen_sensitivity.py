"""oracle/gen_sensitivity.py -- TEST INFRASTRUCTURE (CPU only).

How far can the UNPINNED Farneback oracle be from what cv2 computes?  cv2 (opencv-python 4.11.0.86, uv.lock:238-239) is
not installable here and the reference holds no fixture, so the oracle cannot be pinned against it.  But every place
where the oracle KNOWINGLY orders its arithmetic differently from the published optflowgf.cpp / imgproc code is
available as a switchable variant of the same C file (farneback_oracle.c header), and this script measures what each of
them -- and all of them together, and an FMA-contracted build of the file -- does to

    the flow field                  max / mean |delta| in pixels
    the F3 "divergence" field       max |delta|, to be read against the top-1 - top-2 |div| margin
    the argmax pixel (FF:757)       moved or not
    mean_mag (FF:889-890)           relative change
    the per-pair scalar (FF:785)    change relative to max(|ref|, mean |weighted dot|)   (SURVEY F5's tolerance form)

on the workloads the repo benches: 256x256 (the reference's operating point) pure translation and a zooming clip,
640x360 (configs[0]) and a few 1920x1080 pairs (configs[1]).  The whole post path is re-run per variant: argmax per pair,
the +-6 centre smoothing over the workload's pairs (FF:1203-1214), pass 2 at the variant's own centres.

    python oracle/gen_sensitivity.py            ->  tests/golden/sensitivity.json

Data only.  tests/test_oracle_sensitivity.py re-measures the small workloads and asserts the bounds.
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import oracle as orc  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sensitivity.json")

# name -> (flags, fma build)
VARIANTS = {
    "box_sliding": (orc.V_BOX_SLIDING, False),
    "area2x_seq": (orc.V_AREA2X_SEQ, False),
    "gauss_row_ltr": (orc.V_GAUSS_ROW_LTR, False),
    "all_orderings": (orc.V_ALL, False),
    "fma": (0, True),
    "all_orderings_fma": (orc.V_ALL, True),
}

# name -> (W, H, seed, zoom, frame indices of the pairs' first frames)
WORKLOADS = {
    "256x256_translate": (256, 256, 1, 0.0, list(range(16))),
    "256x256_zoom": (256, 256, 3, 0.05, list(range(16))),
    "640x360": (640, 360, 0, 0.0, [0, 3, 6, 9]),
    "1920x1080": (1920, 1080, 1, 0.0, [0, 5, 11]),
}


def top2_margin(div):
    a = np.abs(div).ravel()
    i = np.argpartition(a, -2)[-2:]
    hi, lo = max(a[i]), min(a[i])
    return float(hi - lo)


def weighted_dot_scale(flow, c):
    """mean |dot * wx * wy|: the magnitude the signed mean of FF:785 is built from"""
    h, w, _ = flow.shape
    y, x = np.mgrid[0:h, 0:w]
    dot = flow[..., 0] * (x - c[0]) + flow[..., 1] * (y - c[1])
    wd = np.where(x > c[0], dot * (w - x) / w, dot * x / w)
    wd = np.where(y > c[1], wd * (h - y) / h, wd * y / h)
    return float(np.mean(np.abs(wd)))


def runner_up(div):
    """pixel of the second-largest |div| (what the argmax becomes if a last-bit difference flips a near-tie)"""
    a = np.abs(div).ravel()
    i = np.argpartition(a, -2)[-2:]
    j = int(i[np.argmin(a[i])])
    return j % div.shape[1], j // div.shape[1]


def tie_flip_effect(flows, recs, dots, scales, divs):
    """Worst case of a flipped near-tie: pair j's argmax replaced by its runner-up pixel, centres re-smoothed, pass 2
    re-run for the <= 13 pairs whose window contains j.  Returns per j the largest scalar change (same normalisation)
    and how far the runner-up pixel is from the winner."""
    n = len(flows)
    pos = [(r[0], r[1]) for r in recs]
    worst, dist = [], []
    for j in range(n):
        alt = list(pos)
        alt[j] = runner_up(divs[j])
        dist.append(float(np.hypot(alt[j][0] - pos[j][0], alt[j][1] - pos[j][1])))
        cs = orc.smooth_centers(alt)
        w = 0.0
        for i in range(max(0, j - 6), min(n, j + 7)):
            d = float(orc.radial_np(flows[i], cs[i], recs[i][3] > 7, False))
            w = max(w, abs(d - dots[i]) / scales[i])
        worst.append(w)
    return worst, dist


def post(flows):
    """pass 1 -> centre smoothing over these pairs -> pass 2, as process_video does on a chunk"""
    recs = []
    for f in flows:
        x, y, v = orc.max_divergence_np(f)
        recs.append((int(x), int(y), float(v), float(orc.mean_mag_np(f))))
    cs = orc.smooth_centers([(r[0], r[1]) for r in recs])
    dots = [float(orc.radial_np(f, c, r[3] > 7, False)) for f, c, r in zip(flows, cs, recs)]
    return recs, cs, dots


def measure(W, H, seed, zoom, starts, variants=None, threads=8):
    variants = variants or VARIANTS
    frames = sine_translate_frames(max(starts) + 2, W, H, seed=seed, zoom=zoom)
    have_fma = orc.lib_fma() is not None
    jobs = [("default", 0, False)] + [(n, fl, fma) for n, (fl, fma) in variants.items() if have_fma or not fma]
    orc.lib()

    def run(job):
        name, fl, fma = job
        return name, [orc.farneback_var(frames[j], frames[j + 1], fl, fma) for j in starts]

    with ThreadPoolExecutor(threads) as ex:
        flows = dict(ex.map(run, jobs))
    base = flows.pop("default")
    brecs, bcs, bdots = post(base)
    bdiv = [orc.divergence_c(f) for f in base]
    margins = [top2_margin(d) for d in bdiv]
    scales = [max(abs(d), weighted_dot_scale(f, c)) for f, c, d in zip(base, bcs, bdots)]
    out = {"size": [W, H], "seed": seed, "zoom": zoom, "pairs": starts,
           "flow_absmax": float(max(np.abs(f).max() for f in base)),
           "margin_top1_top2": margins,
           "default": {"x": [r[0] for r in brecs], "y": [r[1] for r in brecs], "dots": bdots,
                       "mean_mag": [r[3] for r in brecs], "scalar_scale": scales},
           "variants": {}}
    flip, dist = tie_flip_effect(base, brecs, bdots, scales, bdiv)
    out["tie_flip"] = {"what": "pair j's argmax replaced by its runner-up pixel: largest scalar change among the pairs whose "
                               "+-6 window contains j (same normalisation as scalar_rel_delta), runner-up distance in px",
                       "scalar_rel_delta": flip, "runner_up_distance_px": dist}
    for name, fl in flows.items():
        recs, cs, dots = post(fl)
        ddiv = [float(np.abs(orc.divergence_c(f) - d).max()) for f, d in zip(fl, bdiv)]
        out["variants"][name] = {
            "flow_max_abs_delta": [float(np.abs(f - b).max()) for f, b in zip(fl, base)],
            "flow_mean_abs_delta": [float(np.abs(f.astype(np.float64) - b).mean()) for f, b in zip(fl, base)],
            "div_max_abs_delta": ddiv,
            "argmax_moved": [bool((r[0], r[1]) != (b[0], b[1])) for r, b in zip(recs, brecs)],
            "argmax_decided_by_margin": [bool(m > 2 * d) for m, d in zip(margins, ddiv)],
            "mean_mag_rel_delta": [abs(r[3] - b[3]) / b[3] for r, b in zip(recs, brecs)],
            "scalar_rel_delta": [abs(d - b) / s for d, b, s in zip(dots, bdots, scales)],
        }
    return out


def frontend_luma(pairs=16, src=(1280, 720), seed=5):
    """The front-end's own unknown: 15-bit vs 14-bit luma coefficients.  Decoded BGR frames with
    unequal channel gains -> resize to 256x256 + RGB2GRAY with either set -> the whole pair path; same metrics."""
    W, H = src
    # three different textures (one per colour channel) under the same motion: channel values are independent, so the two
    # coefficient sets do round differently at some pixels (proportional channels never reach a rounding boundary)
    bgr = np.stack([sine_translate_frames(pairs + 1, W, H, seed=seed + c, zoom=0.03) for c in range(3)], axis=-1)
    f15 = [orc.frontend(f, False, (256, 256)) for f in bgr]
    f14 = [orc.frontend(f, False, (256, 256), luma14=True) for f in bgr]
    differing = float(np.mean([np.mean(a != b) for a, b in zip(f15, f14)]))
    maxdiff = int(max(np.abs(a.astype(int) - b).max() for a, b in zip(f15, f14)))
    base = [orc.farneback(f15[j], f15[j + 1]) for j in range(pairs)]
    var = [orc.farneback(f14[j], f14[j + 1]) for j in range(pairs)]
    brecs, bcs, bdots = post(base)
    recs, cs, dots = post(var)
    bdiv = [orc.divergence_c(f) for f in base]
    scales = [max(abs(d), weighted_dot_scale(f, c)) for f, c, d in zip(base, bcs, bdots)]
    return {"source": list(src), "pairs": pairs, "gray_pixels_differing": differing, "gray_max_abs_diff": maxdiff,
            "margin_top1_top2": [top2_margin(d) for d in bdiv],
            "flow_max_abs_delta": [float(np.abs(a - b).max()) for a, b in zip(var, base)],
            "flow_mean_abs_delta": [float(np.abs(a.astype(np.float64) - b).mean()) for a, b in zip(var, base)],
            "div_max_abs_delta": [float(np.abs(orc.divergence_c(a) - d).max()) for a, d in zip(var, bdiv)],
            "argmax_moved": [bool((r[0], r[1]) != (b[0], b[1])) for r, b in zip(recs, brecs)],
            "mean_mag_rel_delta": [abs(r[3] - b[3]) / b[3] for r, b in zip(recs, brecs)],
            "scalar_rel_delta": [abs(d - b) / s for d, b, s in zip(dots, bdots, scales)]}


def margin_survey(seeds=range(20, 28), W=256, H=256, pairs=16, threads=8):
    """top-1 - top-2 |div| margins of many more pairs of the default oracle (no variants): how often is a pair a near-tie?"""
    def one(seed):
        fr = sine_translate_frames(pairs + 1, W, H, seed=seed, zoom=0.03 * (seed % 3))
        return [top2_margin(orc.divergence_c(orc.farneback(fr[j], fr[j + 1]))) for j in range(pairs)]
    with ThreadPoolExecutor(threads) as ex:
        m = np.sort(np.concatenate(list(ex.map(one, seeds))))
    return {"size": [W, H], "seeds": list(seeds), "pairs": int(m.size), "min": float(m[0]),
            "percentiles": {str(p): float(np.percentile(m, p)) for p in (1, 5, 25, 50, 75)},
            "sorted_smallest_8": [float(v) for v in m[:8]]}


def summarize(res):
    s = {}
    for wname, w in res.items():
        for vname, v in w["variants"].items():
            e = s.setdefault(vname, {"flow_max_abs_delta": 0.0, "div_max_abs_delta": 0.0, "argmax_moved": 0, "pairs": 0,
                                     "mean_mag_rel_delta": 0.0, "scalar_rel_delta": 0.0})
            e["flow_max_abs_delta"] = max(e["flow_max_abs_delta"], *v["flow_max_abs_delta"])
            e["div_max_abs_delta"] = max(e["div_max_abs_delta"], *v["div_max_abs_delta"])
            e["argmax_moved"] += sum(v["argmax_moved"])
            e["pairs"] += len(v["argmax_moved"])
            e["mean_mag_rel_delta"] = max(e["mean_mag_rel_delta"], *v["mean_mag_rel_delta"])
            e["scalar_rel_delta"] = max(e["scalar_rel_delta"], *v["scalar_rel_delta"])
    return s


def main():
    res = {}
    for name, (W, H, seed, zoom, starts) in WORKLOADS.items():
        res[name] = measure(W, H, seed, zoom, starts)
        print(name, "done", file=sys.stderr)
    doc = {"what": "effect of the known OpenCV orderings the default oracle does not reproduce (oracle/gen_sensitivity.py)",
           "fma_variants_measured": orc.lib_fma() is not None,
           "min_margin_top1_top2": min(min(w["margin_top1_top2"]) for w in res.values()),
           "max_tie_flip_scalar_rel_delta": max(max(w["tie_flip"]["scalar_rel_delta"]) for w in res.values()),
           "margin_survey": margin_survey(), "frontend_luma14_vs_15": frontend_luma(), "summary": summarize(res),
           "workloads": res}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["summary"], indent=1))
    fl = doc["frontend_luma14_vs_15"]
    print("frontend luma 14 vs 15 bit: gray pixels differing", fl["gray_pixels_differing"], "argmax moved", sum(fl["argmax_moved"]), "of",
          fl["pairs"], "max scalar delta", max(fl["scalar_rel_delta"]), "max mean_mag delta", max(fl["mean_mag_rel_delta"]),
          "flow max", max(fl["flow_max_abs_delta"]))
    print("margin survey", doc["margin_survey"])
    print("min margin", doc["min_margin_top1_top2"], "worst tie flip", doc["max_tie_flip_scalar_rel_delta"])


if __name__ == "__main__":
    main()
