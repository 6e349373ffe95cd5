"""Bounds on how far the UNPINNED Farneback oracle can be from cv2 through the orderings it knowingly does not reproduce.

cv2.calcOpticalFlowFarneback (FF:878-879; opencv-python 4.11.0.86, uv.lock:238-239) cannot be run here, so the oracle
stays "parity unpinned".  What CAN be measured is the effect of every known difference between the oracle's arithmetic
order and the published optflowgf.cpp / imgproc code (farneback_oracle.c header): the sliding float-
difference box sums of FarnebackUpdateFlow_Blur, the INTER_AREA route of the x1/2 level, the left-to-right generic row
filter of the wide Gaussians, and FMA contraction.  oracle/gen_sensitivity.py wrote tests/golden/sensitivity.json; this
test re-measures the 256x256 and 640x360 workloads and asserts:

  * the variant code is a different ORDER of the same sums (unit checks against the exact window sum / 2x2 mean);
  * the argmax pixel (bit-exact bar) does not move wherever the top-1 - top-2 |div| margin exceeds twice the observed
    change of the divergence field -- and did not move on any measured pair;
  * mean_mag and the per-pair scalar change by far less than north_star's 1e-4 (tolerances written below);
  * the committed 1080p numbers obey the same bounds.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_sensitivity as gs  # noqa: E402
import oracle as orc  # noqa: E402

TOL_SCALAR = 1e-4      # north_star: float reductions within 1e-4 relative (SURVEY F5 normalisation)
OBSERVED_SCALAR = 1e-5  # what we actually hold the variants to (measured: <= 2.1e-7)
FLOW_ABS = {"256x256_translate": 1e-3, "256x256_zoom": 1e-3, "640x360": 1e-3, "1920x1080": 0.1}  # px, max over the field


@pytest.fixture(scope="module")
def committed():
    with open(os.path.join(ROOT, "tests", "golden", "sensitivity.json")) as f:
        return json.load(f)


def _check_workload(name, w):
    margins = w["margin_top1_top2"]
    assert w["variants"], "no variant was measured"
    for vname, v in w["variants"].items():
        for j, (m, d, moved) in enumerate(zip(margins, v["div_max_abs_delta"], v["argmax_moved"])):
            if m > 2 * d:
                assert not moved, f"{name}/{vname} pair {j}: argmax moved although margin {m:.2e} > 2 x {d:.2e}"
        assert not any(v["argmax_moved"]), f"{name}/{vname}: an argmax pixel moved"
        assert max(v["scalar_rel_delta"]) <= OBSERVED_SCALAR < TOL_SCALAR, (name, vname, max(v["scalar_rel_delta"]))
        assert max(v["mean_mag_rel_delta"]) <= OBSERVED_SCALAR, (name, vname, max(v["mean_mag_rel_delta"]))
        assert max(v["flow_max_abs_delta"]) <= FLOW_ABS[name], (name, vname, max(v["flow_max_abs_delta"]))
        assert max(v["flow_mean_abs_delta"]) <= 1e-5, (name, vname)


@pytest.mark.parametrize("name", ["256x256_translate", "256x256_zoom", "640x360"])
def test_known_opencv_orderings_do_not_move_argmax_or_scalars(name, committed):
    W, H, seed, zoom, starts = gs.WORKLOADS[name]
    w = gs.measure(W, H, seed, zoom, starts, threads=4)
    _check_workload(name, w)
    # the committed file describes the same experiment (skip the number-by-number part if this machine's libm rounds the
    # synthetic texture differently -- the bounds above were still asserted on what was measured here)
    c = committed["workloads"][name]
    if (w["default"]["x"], w["default"]["y"]) == (c["default"]["x"], c["default"]["y"]):
        np.testing.assert_allclose(w["margin_top1_top2"], c["margin_top1_top2"], rtol=0, atol=1e-6)
        for vname in ("box_sliding", "area2x_seq", "gauss_row_ltr", "all_orderings"):
            np.testing.assert_allclose(w["variants"][vname]["flow_max_abs_delta"], c["variants"][vname]["flow_max_abs_delta"],
                                       rtol=0, atol=1e-6)


def test_committed_sensitivity_file_obeys_the_bounds(committed):
    assert set(committed["workloads"]) == set(gs.WORKLOADS)
    for name, w in committed["workloads"].items():
        _check_workload(name, w)
        for vname in ("box_sliding", "area2x_seq", "gauss_row_ltr", "all_orderings"):
            assert vname in w["variants"]
    assert committed["fma_variants_measured"], "the committed file must include the FMA-contracted build"
    assert "fma" in committed["workloads"]["1920x1080"]["variants"]
    s = committed["summary"]
    assert all(e["argmax_moved"] == 0 and e["pairs"] == 39 for e in s.values())
    # near-ties are the residual risk: recorded, not hidden
    ms = committed["margin_survey"]
    assert ms["pairs"] >= 128 and ms["min"] > 0
    assert committed["max_tie_flip_scalar_rel_delta"] > TOL_SCALAR   # a flipped tie WOULD exceed 1e-4: DESIGN section 3 says so


def test_frontend_luma_coefficient_choice_is_an_input_level_difference(committed):
    """whether the pinned wheel's RGB2GRAY uses the 15-bit or the 14-bit coefficient set could not be
    verified.  The two differ by 1 LSB at ~0.1 % of the gray pixels -- an INPUT difference, which (unlike the orderings
    above) reaches the scalars at the 1e-4 level: recorded, bounded at 1e-3, argmax unmoved on the measured pairs."""
    fl = committed["frontend_luma14_vs_15"]
    assert fl["gray_max_abs_diff"] == 1 and 1e-4 < fl["gray_pixels_differing"] < 1e-2
    assert not any(fl["argmax_moved"])
    assert max(fl["scalar_rel_delta"]) < 1e-3 and max(fl["mean_mag_rel_delta"]) < 1e-3
    assert max(fl["flow_max_abs_delta"]) < 0.05
    small = gs.frontend_luma(pairs=5, src=(512, 288), seed=9)     # re-measured here on a smaller source
    assert small["gray_max_abs_diff"] <= 1
    assert max(small["scalar_rel_delta"]) < 2e-3 and max(small["mean_mag_rel_delta"]) < 2e-3
    for m, d, moved in zip(small["margin_top1_top2"], small["div_max_abs_delta"], small["argmax_moved"]):
        assert not (moved and m > 2 * d)


def test_sliding_box_is_the_same_sum_in_another_order():
    rng = np.random.default_rng(5)
    r2, r3, r4, r5, r6 = rng.normal(0, 3, (5, 70, 90)).astype(np.float32)     # M as UpdateMatrices forms it (A.4):
    M = np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3])  # G is PSD
    a, b = orc.blur_solve(M), orc.blur_solve_sliding(M)
    assert not np.array_equal(a, b)                      # it IS another rounding ...
    assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()  # ... of the same quantity
    # integer-valued M: every float difference and every running sum is exact -> the two orders agree bit for bit
    Mi = rng.integers(-50, 50, (5, 40, 50)).astype(np.float32)
    assert np.array_equal(orc.blur_solve(Mi), orc.blur_solve_sliding(Mi))


def test_area2x_and_row_ltr_are_reorderings_of_the_default_level_image():
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    for k in (0, 1, 2):
        base = orc.pyr_level(img, k)
        for fl in (orc.V_AREA2X_SEQ, orc.V_GAUSS_ROW_LTR):
            v = orc.pyr_level_var(img, k, fl)
            assert v.shape == base.shape
            assert np.abs(v - base).max() <= 4e-5          # a few ulp of 255
    # level 0 is untouched by both (same size, 3 taps); level 1 only by AREA2X; level 2 (9 taps) only by ROW_LTR
    assert np.array_equal(orc.pyr_level_var(img, 0, orc.V_ALL), orc.pyr_level(img, 0))
    assert np.array_equal(orc.pyr_level_var(img, 1, orc.V_GAUSS_ROW_LTR), orc.pyr_level(img, 1))
    assert np.array_equal(orc.pyr_level_var(img, 2, orc.V_AREA2X_SEQ), orc.pyr_level(img, 2))
    # a constant image is reproduced exactly by every ordering except for the kernel's own rounding
    flat = np.full((64, 96), 200, np.uint8)
    assert np.abs(orc.pyr_level_var(flat, 2, orc.V_ALL) - 200).max() <= 1e-4


def test_variant_driver_with_no_flag_is_the_oracle():
    from funscript_flow_amd.synth import sine_translate_frames
    fr = sine_translate_frames(2, 96, 80, seed=4)
    assert np.array_equal(orc.farneback_var(fr[0], fr[1], 0), orc.farneback(fr[0], fr[1]))


def test_workspace_pair_equals_the_malloc_pair():
    """bench.py's cpu_baseline workers use orc_pair_ws (pre-faulted workspace): same numbers as orc_pair."""
    from funscript_flow_amd.synth import sine_translate_frames
    fr = sine_translate_frames(3, 200, 136, seed=2)
    ws = orc.PairWorkspace(200, 136)
    for j in (0, 1, 0):
        f, x, y, v, mm = ws.pair(fr[j], fr[j + 1])
        g, gx, gy, gv, gmm = orc.pair_c(fr[j], fr[j + 1])
        assert np.array_equal(f, g) and (x, y, v, mm) == (gx, gy, gv, gmm)
