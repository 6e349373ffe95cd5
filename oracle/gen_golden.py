"""oracle/gen_golden.py -- TEST INFRASTRUCTURE (build container only).

Generates tests/golden/*.npz|json by RUNNING THE REAL REFERENCE FUNCTIONS imported from
/root/reference/FunscriptFlow.pyw through oracle/ref_loader.py:

  post_goldens.npz   max_divergence (FF:748-758) and radial_motion_weighted (FF:761-785) on seeded
                     flow fields, incl. ties, integer centres, centres outside the image, POV, cut.
  chain_golden.npz   process_video (FF:1094-1404) driven end to end with a fake cv2 whose
  chain_golden.json  calcOpticalFlowFarneback is the C oracle: captures pos_center per pair, the
                     +-6 smoothed centres (FF:1203-1214) handed to radial_motion_weighted, the
                     per-pair scalars and the written .funscript actions.

Run:  python oracle/gen_golden.py      (needs /root/reference; never runs on the GPU box)
Fixtures are data only (inputs + expected outputs); no reference source is copied.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import oracle as orc  # noqa: E402
from ref_loader import load_reference  # noqa: E402
from funscript_flow_amd.synth import sine_translate_frames  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def smooth_flow(rng, h, w, amp=3.0):
    """Low-frequency random flow field (a few sinusoids), float32 (h,w,2)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    f = np.zeros((h, w, 2))
    for c in range(2):
        for _ in range(4):
            fx, fy = rng.uniform(0.01, 0.2, 2)
            ph = rng.uniform(0, 2 * np.pi)
            f[..., c] += rng.uniform(0.2, 1.0) * np.sin(fx * x + fy * y + ph)
    return (amp * f).astype(np.float32)


def gen_post(ref):
    rng = np.random.default_rng(1234)
    cases = {}
    meta = []

    def add(name, flow, centers):
        flow = np.ascontiguousarray(flow, np.float32)
        h, w, _ = flow.shape
        x, y, v = ref.max_divergence(flow)
        cases[f"{name}.flow"] = flow
        cases[f"{name}.maxdiv"] = np.array([int(x), int(y)], np.int64)
        cases[f"{name}.maxdiv_val"] = np.float32(v)
        cs, outs = [], []
        for c in centers:
            c = np.asarray(c, np.float64)
            r = [ref.radial_motion_weighted(flow, c, False, False), ref.radial_motion_weighted(flow, c, False, True),
                 ref.radial_motion_weighted(flow, c, True, False)]
            cs.append(c)
            outs.append(r)
        cases[f"{name}.centers"] = np.array(cs, np.float64)
        cases[f"{name}.radial"] = np.array(outs, np.float64)  # columns: weighted, pov, cut
        meta.append(name)

    def centers_for(h, w):
        return [(w / 2.0 + 0.37, h / 2.0 - 0.21), (float(w // 3), float(h // 4)), (0.0, 0.0), (w - 1.0, h - 1.0),
                (-12.5, h + 7.25), (w + 30.0, -3.0), (w / 2.0, h - 1.0)]

    add("noise_36x64", rng.standard_normal((36, 64, 2)), centers_for(36, 64))
    add("smooth_90x160", smooth_flow(rng, 90, 160), centers_for(90, 160))
    add("noise_256x256", rng.standard_normal((256, 256, 2)) * 2.5, centers_for(256, 256))
    # ties: piecewise-linear field whose |div| is maximal and equal on many pixels -> first in C order wins
    h, w = 40, 72
    y, x = np.mgrid[0:h, 0:w]
    tie = np.zeros((h, w, 2), np.float32)
    tie[..., 0] = np.where((y >= 10) & (y < 20), (y - 10) * 2.0, 0.0)  # du/dy = 2 on rows 11..18
    tie[..., 1] = np.where((x >= 30) & (x < 40), (x - 30) * -1.0, 0.0)
    add("ties_40x72", tie, centers_for(h, w))
    # negative winner with a positive tie later in C order
    neg = np.zeros((24, 40, 2), np.float32)
    neg[5:8, 7, 0] = [0.0, -3.0, 0.0]   # du/dy = -1.5 at (y=5), +1.5 at (y=7): |.| equal, first is negative
    add("negfirst_24x40", neg, centers_for(24, 40))
    # realistic flow: oracle Farneback on a synthetic pair (inputs to the post path are what matters)
    fr = sine_translate_frames(2, 320, 180, seed=7, amp=(4.0, 3.0), period=5)
    add("farneback_180x320", orc.farneback(fr[0], fr[1]), centers_for(180, 320))
    # edge winners: extremes on the border rows/cols use one-sided differences
    edge = smooth_flow(rng, 32, 48, amp=1.0)
    edge[0, 5, 0] += 40.0
    add("edge_32x48", edge, centers_for(32, 48))
    cases["names"] = np.array(meta)
    np.savez_compressed(os.path.join(GOLD, "post_goldens.npz"), **cases)
    print("post_goldens:", meta)


# ------------------------------------------------------------------ fake cv2 for process_video
def make_fake_cv2(frames_bgr, fps):
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FRAME_COUNT, cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH = 7, 5, 3
    cv2.CAP_PROP_FRAME_HEIGHT, cv2.CAP_PROP_POS_FRAMES, cv2.CAP_PROP_BUFFERSIZE = 4, 1, 38
    cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2GRAY, cv2.COLOR_BGR2GRAY = 4, 7, 6

    class VideoCapture:
        def __init__(self, path):
            self.pos = 0

        def isOpened(self):
            return True

        def get(self, prop):
            n, h, w, _ = frames_bgr.shape
            return {7: float(n), 5: float(fps), 3: float(w), 4: float(h)}[prop]

        def set(self, prop, val):
            if prop == 1:
                self.pos = int(val)
            return True

        def read(self):
            if 0 <= self.pos < len(frames_bgr):
                f = frames_bgr[self.pos].copy()
                self.pos += 1
                return True, f
            return False, None

        def release(self):
            pass

    def cvtColor(src, code, dst=None):
        if code == 4:
            out = src[..., ::-1].copy()
            if dst is not None:
                dst[...] = out
                return dst
            return out
        if code == 7:  # RGB -> gray, 8-bit fixed point
            return orc.bgr2gray(np.ascontiguousarray(src[..., ::-1]))
        if code == 6:
            return orc.bgr2gray(np.ascontiguousarray(src))
        raise NotImplementedError(code)

    def resize(src, size):
        if (src.shape[1], src.shape[0]) == tuple(size):
            return src
        raise NotImplementedError("fake cv2.resize: feed frames at the target size")

    def calcOpticalFlowFarneback(p0, p1, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        assert (pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags) == (0.5, 3, 15, 3, 5, 1.2, 0)
        return orc.farneback(p0, p1)

    def cartToPolar(x, y):
        return np.sqrt(x * x + y * y), np.arctan2(y, x)

    cv2.VideoCapture, cv2.cvtColor, cv2.resize = VideoCapture, cvtColor, resize
    cv2.calcOpticalFlowFarneback, cv2.cartToPolar = calcOpticalFlowFarneback, cartToPolar
    return cv2


class SerialPool:
    def __init__(self, processes=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def starmap(self, fn, args):
        return [fn(*a) for a in args]


class _Fut:
    def __init__(self, v):
        self.v = v

    def result(self):
        return self.v


class SerialExecutor:
    def __init__(self, max_workers=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, *a, **k):
        return _Fut(fn(*a, **k))


def gen_chain():
    n_frames, size, fps = 41, 256, 30.0
    gray = sine_translate_frames(n_frames, size, size, seed=3, amp=(3.0, 2.0), period=16, zoom=0.02)
    frames_bgr = np.repeat(gray[..., None], 3, axis=-1)
    ref = load_reference(make_fake_cv2(frames_bgr, fps))
    sys.modules["funscriptflow_reference"] = ref
    ref.Pool = SerialPool
    real_tpe = ref.concurrent.futures.ThreadPoolExecutor
    fake_cf = types.SimpleNamespace(ProcessPoolExecutor=SerialExecutor, ThreadPoolExecutor=real_tpe,
                                    as_completed=ref.concurrent.futures.as_completed)
    ref.concurrent = types.SimpleNamespace(futures=fake_cf)

    log = dict(pos=[], val=[], mean_mag=[], cut=[], centers=[], dots=[])
    real_pre, real_rad = ref.precompute_flow_info, ref.radial_motion_weighted

    def pre(p0, p1, config):
        r = real_pre(p0, p1, config)
        log["pos"].append([int(r["pos_center"][0]), int(r["pos_center"][1])])
        log["val"].append(float(r["val_pos"]))
        log["mean_mag"].append(float(r["mean_mag"]))
        log["cut"].append(bool(r["cut"]))
        return r

    def rad(flow, center, is_cut, pov_mode=False):
        v = real_rad(flow, center, is_cut, pov_mode)
        log["centers"].append([float(center[0]), float(center[1])])
        log["dots"].append(float(v))
        return v

    ref.precompute_flow_info, ref.radial_motion_weighted = pre, rad
    settings = {"threads": 2, "detrend_window": 2.0, "norm_window": 3.0, "batch_size": 24, "overwrite": True,
                "vr_mode": False, "pov_mode": False, "keyframe_reduction": True, "backend": "CPU"}
    with tempfile.TemporaryDirectory() as td:
        video = os.path.join(td, "clip.mp4")
        msgs = []
        err = ref.process_video(video, settings, msgs.append)
        assert not err, msgs
        with open(os.path.join(td, "clip.funscript")) as f:
            funscript = json.load(f)
    np.savez_compressed(os.path.join(GOLD, "chain_golden.npz"), pos_center=np.array(log["pos"], np.int64),
                        val_pos=np.array(log["val"], np.float32), mean_mag=np.array(log["mean_mag"], np.float32),
                        cut=np.array(log["cut"]), centers=np.array(log["centers"], np.float64),
                        dots=np.array(log["dots"], np.float64))
    import zlib
    with open(os.path.join(GOLD, "chain_golden.json"), "w") as f:
        json.dump({"settings": settings, "fps": fps, "n_frames": n_frames, "size": size,
                   "frames_crc32": zlib.crc32(gray.tobytes()),
                   "synth": {"seed": 3, "amp": [3.0, 2.0], "period": 16, "zoom": 0.02},
                   "funscript": funscript, "log": msgs}, f, indent=1)
    print("chain_golden: pairs", len(log["pos"]), "actions", len(funscript["actions"]))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    gen_post(load_reference())
    gen_chain()
