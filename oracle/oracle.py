"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end to oracle/liboracle.so (the C restatement, farneback_oracle.c) plus a numpy
restatement of the reference's post path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; funscript_flow_amd/ never does.

Reference lines followed (FF = /root/reference/FunscriptFlow.pyw):
  farneback()            cv2.calcOpticalFlowFarneback(p0,p1,None,0.5,3,15,3,5,1.2,0)   FF:878-879
                         (third-party opencv-python 4.11.0.86 -- parity unpinned, see the C header)
  max_divergence_np()    FF:748-758
  mean_mag_np()          FF:889-890
  radial_np()            FF:761-785
  smooth_centers()       FF:1203-1214
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_FMA = None
_LIB_FAST = None

# variant flags of orc_farneback_var (farneback_oracle.c header): OpenCV orderings the default oracle does not reproduce
V_BOX_SLIDING, V_AREA2X_SEQ, V_GAUSS_ROW_LTR = 1, 2, 4
V_ALL = V_BOX_SLIDING | V_AREA2X_SEQ | V_GAUSS_ROW_LTR

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False, name="liboracle.so"):
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("farneback_oracle.c", "frontend_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


def cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def lib_fma():
    """The same C file built with FMA contraction (sensitivity study only); None on a host without FMA3."""
    global _LIB_FMA
    if _LIB_FMA is None:
        if not cpu_has_fma():
            return None
        L = C.CDLL(build(name="liboracle_fma.so"))
        L.orc_farneback_var.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_uint]
        assert L.orc_fma_build() == 1
        _LIB_FMA = L
    return _LIB_FMA


def _host_id():
    """CPU model + instruction-set flags of this host: what a -march=native build is only valid for."""
    import hashlib
    model, flags = "?", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "?":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return f"{model} {hashlib.sha256(flags.encode()).hexdigest()[:12]}"


def lib_fast():
    """liboracle_fast.so: the SAME C source at -O3 -march=native with contraction / vectorisation allowed.  TIMING ONLY
    (bench.py's cpu_baseline `value`): never used to check anything.  Built for the host it runs on -- a copy that
    travelled here from another machine (the build container's) is rebuilt first, since -march=native code of another
    CPU may not even execute.  Returns None when it cannot be built here."""
    global _LIB_FAST
    if _LIB_FAST is None:
        so, tag = os.path.join(_HERE, "liboracle_fast.so"), os.path.join(_HERE, "liboracle_fast.so.host")
        srcs = [os.path.join(_HERE, f) for f in ("farneback_oracle.c", "frontend_oracle.c")]
        host = _host_id()
        try:
            stale = (not os.path.exists(so) or not os.path.exists(tag) or open(tag).read() != host
                     or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs))
            if stale:
                subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_fast.so"], stdout=subprocess.DEVNULL)
                with open(tag, "w") as f:
                    f.write(host)
            L = C.CDLL(so)
        except (OSError, subprocess.CalledProcessError):
            return None
        L.orc_ws_bytes.argtypes = [C.c_int, C.c_int]
        L.orc_ws_bytes.restype = C.c_size_t
        L.orc_pair_ws.argtypes = [C.c_void_p, C.c_size_t, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.orc_radial.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_radial.restype = C.c_double
        _LIB_FAST = L
    return _LIB_FAST


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_farneback_var.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_uint]
        L.orc_pyr_level_var.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_uint]
        L.orc_blur_solve_sliding.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.orc_ws_bytes.argtypes = [C.c_int, C.c_int]
        L.orc_ws_bytes.restype = C.c_size_t
        L.orc_pair_ws.argtypes = [C.c_void_p, C.c_size_t, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.orc_num_levels.argtypes = [C.c_int, C.c_int]
        L.orc_level_params.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_gaussian_kernel.argtypes = [C.c_int, C.c_double, _f32p]
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_pyr_level.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_resize_linear_u8c3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_swap_rb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_rgb2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_rgb2gray14.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_polyexp_prepare.argtypes = [_f32p, _f32p, _f32p, np.ctypeslib.ndpointer(np.float64)]
        L.orc_polyexp.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.orc_flow_upsample.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
        L.orc_update_matrices.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.orc_blur_solve.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.orc_farneback.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_farneback_dbg.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.orc_divergence.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.orc_max_divergence.argtypes = [_f32p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_float)]
        L.orc_mean_mag.argtypes = [_f32p, C.c_int, C.c_int]
        L.orc_mean_mag.restype = C.c_double
        L.orc_radial.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_radial.restype = C.c_double
        L.orc_pair.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_float), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


# ---------------------------------------------------------------- Farneback (C restatement)
def num_levels(w, h):
    return lib().orc_num_levels(w, h)


def level_params(w, h, k):
    lw, lh, ks, sg = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    lib().orc_level_params(w, h, k, C.byref(lw), C.byref(lh), C.byref(sg), C.byref(ks))
    return lw.value, lh.value, sg.value, ks.value


def gaussian_kernel(n, sigma):
    out = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, float(sigma), out)
    return out


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(bgr, w, h, w * 3, out)
    return out


def _rows_view(a):
    """(pointer, row stride) of an (h, w, 3) uint8 array whose pixels are contiguous inside a row"""
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
    if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < 0:
        a = np.ascontiguousarray(a)
    return a, a.strides[0]


def swap_rb(img):
    """cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)   FF:182"""
    h, w, _ = img.shape
    out = np.empty((h, w, 3), np.uint8)
    a, st = _rows_view(img)
    lib().orc_swap_rb(a.ctypes.data, w, h, st, out)
    return out


def resize_linear_u8c3(img, dw, dh):
    """cv2.resize(img, (dw, dh)) for an (h, w, 3) uint8 image   FF:186, FF:1076 (parity unpinned)"""
    h, w, _ = img.shape
    out = np.empty((dh, dw, 3), np.uint8)
    a, st = _rows_view(img)
    lib().orc_resize_linear_u8c3(a.ctypes.data, w, h, st, out, dw, dh)
    return out


def rgb2gray(img, luma14=False):
    """cv2.cvtColor(img, cv2.COLOR_RGB2GRAY) on an (h, w, 3) view (rows may be strided)   FF:1079, FF:1082
    luma14: the 14-bit coefficient set of older OpenCV releases (sensitivity study only)."""
    h, w, _ = img.shape
    out = np.empty((h, w), np.uint8)
    a, st = _rows_view(img)
    (lib().orc_rgb2gray14 if luma14 else lib().orc_rgb2gray)(a.ctypes.data, w, h, st, out)
    return out


def frontend(frame_bgr, vr_mode=False, size=(256, 256), luma14=False):
    """Decoded BGR frame -> the gray operand of the pair kernel, step by step as the reference does it:
    BGR2RGB (FF:182); non-VR: resize to `size` unless already that size (FF:185-186, FF:1057) then RGB2GRAY
    (FF:1082); VR: resize to twice `size`, keep rows [h:], columns [:w] (FF:1076-1079), RGB2GRAY."""
    w, h = size
    rgb = swap_rb(frame_bgr)
    if vr_mode:
        r = resize_linear_u8c3(rgb, 2 * w, 2 * h)
        return rgb2gray(r[h:, :w], luma14)
    if rgb.shape[1] != w or rgb.shape[0] != h:
        rgb = resize_linear_u8c3(rgb, w, h)
    return rgb2gray(rgb, luma14)


def pyr_level(img, k):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lw, lh, _, _ = level_params(w, h, k)
    out = np.empty((lh, lw), np.float32)
    lib().orc_pyr_level(img, w, h, w, k, out)
    return out


def polyexp_constants():
    g, xg, xxg = (np.empty(6, np.float32) for _ in range(3))
    ig = np.empty(4, np.float64)
    lib().orc_polyexp_prepare(g, xg, xxg, ig)
    return g, xg, xxg, ig


def polyexp(I):
    I = np.ascontiguousarray(I, np.float32)
    h, w = I.shape
    R = np.empty((5, h, w), np.float32)
    lib().orc_polyexp(I, w, h, R)
    return R


def flow_upsample(prev, w, h):
    prev = np.ascontiguousarray(prev, np.float32)
    ph, pw, _ = prev.shape
    out = np.empty((h, w, 2), np.float32)
    lib().orc_flow_upsample(prev, pw, ph, out, w, h)
    return out


def update_matrices(R0, R1, flow):
    h, w, _ = flow.shape
    M = np.empty((5, h, w), np.float32)
    lib().orc_update_matrices(np.ascontiguousarray(R0, np.float32), np.ascontiguousarray(R1, np.float32),
                              np.ascontiguousarray(flow, np.float32), w, h, M)
    return M


def blur_solve(M):
    _, h, w = M.shape
    flow = np.empty((h, w, 2), np.float32)
    lib().orc_blur_solve(np.ascontiguousarray(M, np.float32), w, h, flow)
    return flow


def farneback(p0, p1):
    """Restatement of cv2.calcOpticalFlowFarneback(p0,p1,None,0.5,3,15,3,5,1.2,0) (FF:878-879)."""
    p0 = np.ascontiguousarray(p0, np.uint8)
    p1 = np.ascontiguousarray(p1, np.uint8)
    h, w = p0.shape
    flow = np.empty((h, w, 2), np.float32)
    rc = lib().orc_farneback(p0, p1, w, h, w, flow)
    if rc:
        raise MemoryError("oracle farneback failed")
    return flow


def farneback_var(p0, p1, flags=0, fma=False):
    """The restatement with OpenCV-ordering variants switched on (flags = OR of V_*), optionally from the FMA-contracted
    build.  Sensitivity study only: the HIP kernels are compared with farneback() = flags 0, no FMA."""
    L = lib_fma() if fma else lib()
    if L is None:
        raise RuntimeError("this host CPU has no FMA3")
    p0 = np.ascontiguousarray(p0, np.uint8)
    p1 = np.ascontiguousarray(p1, np.uint8)
    h, w = p0.shape
    flow = np.empty((h, w, 2), np.float32)
    if L.orc_farneback_var(p0, p1, w, h, w, flow, int(flags)):
        raise MemoryError("oracle farneback failed")
    return flow


def pyr_level_var(img, k, flags):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lw, lh, _, _ = level_params(w, h, k)
    out = np.empty((lh, lw), np.float32)
    lib().orc_pyr_level_var(img, w, h, w, k, out, int(flags))
    return out


def blur_solve_sliding(M):
    _, h, w = M.shape
    flow = np.empty((h, w, 2), np.float32)
    lib().orc_blur_solve_sliding(np.ascontiguousarray(M, np.float32), w, h, flow)
    return flow


class PairWorkspace:
    """Pre-faulted scratch block + output buffers for orc_pair_ws (cpu_baseline workers: nothing is allocated or
    first-touched inside the timed loop)."""

    def __init__(self, w, h, fast=False):
        """fast: run liboracle_fast.so (timing only) instead of the parity build."""
        self.w, self.h = w, h
        self.L = lib_fast() if fast else lib()
        if self.L is None:
            raise RuntimeError("liboracle_fast.so could not be built on this host")
        self.nbytes = self.L.orc_ws_bytes(w, h)
        self.mem = np.zeros(self.nbytes, np.uint8)          # zeros() maps lazily: touch every page now
        self.mem[::4096] = 1
        self.flow = np.zeros((h, w, 2), np.float32)
        self.flow.reshape(-1)[::1024] = 1                   # one float per 4 KiB page: every page of the output too

    def radial(self, center, is_cut, pov_mode=False):
        """pass 2 of the workspace's latest flow, by the same build"""
        return self.L.orc_radial(self.flow, self.w, self.h, float(center[0]), float(center[1]), int(bool(is_cut)), int(bool(pov_mode)))

    def pair(self, p0, p1):
        x, y, v, mm = C.c_int(), C.c_int(), C.c_float(), C.c_double()
        rc = self.L.orc_pair_ws(self.mem.ctypes.data, self.nbytes, p0, p1, self.w, self.h, self.w, self.flow,
                                C.byref(x), C.byref(y), C.byref(v), C.byref(mm))
        if rc:
            raise MemoryError(f"orc_pair_ws failed ({rc})")
        return self.flow, x.value, y.value, np.float32(v.value), mm.value


def farneback_dbg(p0, p1, level, it):
    """Full run + the level-`level` intermediates captured before blur iteration `it`
    (it=0: after the initial UpdateMatrices; it=3: end of level)."""
    p0 = np.ascontiguousarray(p0, np.uint8)
    p1 = np.ascontiguousarray(p1, np.uint8)
    h, w = p0.shape
    lw, lh, _, _ = level_params(w, h, level)
    flow = np.empty((h, w, 2), np.float32)
    d = dict(I0=np.empty((lh, lw), np.float32), I1=np.empty((lh, lw), np.float32),
             R0=np.empty((5, lh, lw), np.float32), R1=np.empty((5, lh, lw), np.float32),
             M=np.empty((5, lh, lw), np.float32), flow=np.empty((lh, lw, 2), np.float32))
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().orc_farneback_dbg(p0, p1, w, h, w, flow, level, it, ptr(d["I0"]), ptr(d["I1"]), ptr(d["R0"]),
                                 ptr(d["R1"]), ptr(d["M"]), ptr(d["flow"]))
    if rc:
        raise MemoryError("oracle farneback failed")
    d["out"] = flow
    return d


def max_divergence_c(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    x, y, v = C.c_int(), C.c_int(), C.c_float()
    lib().orc_max_divergence(flow, w, h, C.byref(x), C.byref(y), C.byref(v))
    return x.value, y.value, np.float32(v.value)


def divergence_c(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    out = np.empty((h, w), np.float32)
    lib().orc_divergence(flow, w, h, out)
    return out


def mean_mag_c(flow):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    return lib().orc_mean_mag(flow, w, h)


def radial_c(flow, center, is_cut, pov_mode=False):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    return lib().orc_radial(flow, w, h, float(center[0]), float(center[1]), int(bool(is_cut)), int(bool(pov_mode)))


def pair_c(p0, p1):
    """Farneback + max_divergence + mean magnitude for one pair (what the cpu_baseline leg times)."""
    p0 = np.ascontiguousarray(p0, np.uint8)
    p1 = np.ascontiguousarray(p1, np.uint8)
    h, w = p0.shape
    flow = np.empty((h, w, 2), np.float32)
    x, y, v, mm = C.c_int(), C.c_int(), C.c_float(), C.c_double()
    rc = lib().orc_pair(p0, p1, w, h, w, flow, C.byref(x), C.byref(y), C.byref(v), C.byref(mm))
    if rc:
        raise MemoryError("oracle pair failed")
    return flow, x.value, y.value, np.float32(v.value), mm.value


# ---------------------------------------------------------------- post path (numpy restatement)
def max_divergence_np(flow):
    """FF:748-758.  div = d(u)/dy + d(v)/dx (sic), first argmax of |div| in C order."""
    u = flow[..., 0]
    v = flow[..., 1]
    du = np.empty_like(u)
    du[1:-1] = (u[2:] - u[:-2]) / np.float32(2.0)
    du[0] = u[1] - u[0]
    du[-1] = u[-1] - u[-2]
    dv = np.empty_like(v)
    dv[:, 1:-1] = (v[:, 2:] - v[:, :-2]) / np.float32(2.0)
    dv[:, 0] = v[:, 1] - v[:, 0]
    dv[:, -1] = v[:, -1] - v[:, -2]
    div = du + dv
    idx = int(np.argmax(np.abs(div)))
    y, x = divmod(idx, div.shape[1])
    return x, y, div[y, x]


def mean_mag_np(flow):
    """FF:889-890: cv2.cartToPolar magnitude (float32) then np.mean (float32)."""
    u = flow[..., 0]
    v = flow[..., 1]
    return np.mean(np.sqrt(u * u + v * v))


def radial_np(flow, center, is_cut, pov_mode=False):
    """FF:761-785, float64 throughout, same operation order."""
    if is_cut:
        return 0.0
    h, w, _ = flow.shape
    y, x = np.mgrid[0:h, 0:w]
    dx = x - center[0]
    dy = y - center[1]
    dot = flow[..., 0] * dx + flow[..., 1] * dy
    if pov_mode:
        return np.mean(dot)
    wd = np.where(x > center[0], dot * (w - x) / w, dot * x / w)
    wd = np.where(y > center[1], wd * (h - y) / h, wd * y / h)
    return np.mean(wd)


def smooth_centers(pos_centers):
    """FF:1203-1214: mean of the argmax positions of pairs j-6..j+6 inside the chunk."""
    n = len(pos_centers)
    out = []
    for j in range(n):
        lst = [pos_centers[j]]
        for i in range(1, 7):
            if j - i >= 0:
                lst.append(pos_centers[j - i])
            if j + i < n:
                lst.append(pos_centers[j + i])
        out.append(np.mean(np.array(lst), axis=0))
    return out
