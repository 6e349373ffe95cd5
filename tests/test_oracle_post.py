"""Oracle post path (numpy + C restatements) against golden vectors produced by the REAL reference
functions max_divergence (FunscriptFlow.pyw:748-758) and radial_motion_weighted (FF:761-785);
generator: oracle/gen_golden.py."""
import json
import os

import numpy as np
import pytest

import oracle as orc


@pytest.fixture(scope="module")
def post(golden_dir):
    return np.load(os.path.join(golden_dir, "post_goldens.npz"))


def names(golden_dir):
    return list(np.load(os.path.join(golden_dir, "post_goldens.npz"))["names"])


@pytest.mark.parametrize("name", ["noise_36x64", "smooth_90x160", "noise_256x256", "ties_40x72", "negfirst_24x40",
                                  "farneback_180x320", "edge_32x48"])
def test_max_divergence_bit_exact(post, name):
    flow = post[f"{name}.flow"]
    gx, gy = post[f"{name}.maxdiv"]
    gv = post[f"{name}.maxdiv_val"]
    for fn in (orc.max_divergence_np, orc.max_divergence_c):
        x, y, v = fn(flow)
        assert (x, y) == (gx, gy), fn.__name__
        assert np.float32(v).tobytes() == np.float32(gv).tobytes(), fn.__name__


@pytest.mark.parametrize("name", ["noise_36x64", "smooth_90x160", "noise_256x256", "ties_40x72", "negfirst_24x40",
                                  "farneback_180x320", "edge_32x48"])
def test_radial_matches_reference(post, name):
    flow = post[f"{name}.flow"]
    for c, (gw, gp, gc) in zip(post[f"{name}.centers"], post[f"{name}.radial"]):
        # numpy restatement: same operations in the same order -> equal to the last bit
        assert orc.radial_np(flow, c, False, False) == gw
        assert orc.radial_np(flow, c, False, True) == gp
        assert orc.radial_np(flow, c, True, False) == gc == 0.0
        # C restatement: sequential float64 sum instead of numpy's pairwise sum
        scale = np.mean(np.abs(flow)) * max(flow.shape[:2])
        assert abs(orc.radial_c(flow, c, False, False) - gw) <= 1e-12 * scale
        assert abs(orc.radial_c(flow, c, False, True) - gp) <= 1e-12 * scale
        assert orc.radial_c(flow, c, True, False) == 0.0


def test_survey_recorded_values():
    """Values observed at survey time by calling the reference directly."""
    f = np.random.default_rng(0).standard_normal((36, 64, 2)).astype(np.float32)
    x, y, v = orc.max_divergence_np(f)
    assert (x, y) == (24, 0) and abs(float(v) - (-4.8598113)) < 1e-6
    assert orc.radial_np(f, [20.5, 11.25], False) == -0.015042943069554846
    assert orc.radial_np(f, [20.5, 11.25], False, True) == -0.35534192941860787


def test_divergence_is_cross_derivative():
    """F3: the reference's 'divergence' is du/dy + dv/dx: a pure expansion scores ~0, a shear scores."""
    h, w = 32, 48
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    expansion = np.stack([0.1 * (x - w / 2), 0.1 * (y - h / 2)], -1).astype(np.float32)
    shear = np.stack([0.1 * (y - h / 2), 0.1 * (x - w / 2)], -1).astype(np.float32)
    assert np.abs(orc.divergence_c(expansion)).max() == 0.0
    assert np.allclose(orc.divergence_c(shear), 0.2, atol=1e-6)


def test_mean_mag(post):
    flow = post["farneback_180x320.flow"]
    ref = orc.mean_mag_np(flow)
    assert ref.dtype == np.float32
    assert abs(orc.mean_mag_c(flow) - float(ref)) <= 1e-5 * float(ref)


def test_centre_smoothing_matches_process_video(golden_dir):
    """FF:1203-1214 pinned by driving the real process_video (chain_golden.*)."""
    d = np.load(os.path.join(golden_dir, "chain_golden.npz"))
    meta = json.load(open(os.path.join(golden_dir, "chain_golden.json")))
    bs = meta["settings"]["batch_size"]
    pos = d["pos_center"]
    # pairs never span chunks (F10): chunk c holds bs frames -> bs-1 pairs
    out, start, n_frames = [], 0, meta["n_frames"]
    for cs in range(0, n_frames, bs):
        n_pairs = min(bs, n_frames - cs) - 1
        if n_pairs < 1:
            continue
        out += orc.smooth_centers([tuple(p) for p in pos[start:start + n_pairs]])
        start += n_pairs
    assert start == len(pos)
    assert np.array_equal(np.array(out), d["centers"])
