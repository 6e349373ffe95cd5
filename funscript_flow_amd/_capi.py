"""ctypes binding of libffl_hip.so (include/ffl.h).  No fallback: if the HIP library is missing or no
device is usable, loading / context creation raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libffl_hip.so")

FFL_OK = 0
FFL_MAX_BATCH = 256
# kernel classes of ffl_profile_read
KERNEL_CLASSES = ["k_gray", "k_pyr_level", "k_polyexp", "k_frontend", "k_update_matrices", "k_blur_solve",
                  "k_pass1", "k_radial"]

# every symbol include/ffl.h declares (tests check that the library exports all of them)
EXPORTS = ["ffl_device_count", "ffl_create", "ffl_destroy", "ffl_last_error", "ffl_upload_frame", "ffl_upload_frames", "ffl_upload_frames_raw", "ffl_host_alloc", "ffl_host_free",
           "ffl_flow_pairs",
           "ffl_pass1_result", "ffl_pass1_results", "ffl_radial", "ffl_download_flow", "ffl_upload_flow", "ffl_submit_pair", "ffl_sync",
           "ffl_num_levels", "ffl_level_size", "ffl_download_frame", "ffl_debug_pair", "ffl_set_option", "ffl_ctx_set_option", "ffl_ctx_get_option", "ffl_graph_stats", "ffl_profile_enable",
           "ffl_profile_read", "ffl_kernel_name", "ffl_device_mem_info", "ffl_estimate_bytes"]


class FFLError(RuntimeError):
    pass


_lib = None


def load():
    """Load libffl_hip.so; raises FFLError if it has not been built (see __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFLError(f"{LIB_PATH} is missing: build it with `make -C funscript_flow_amd/csrc` "
                       "(there is no CPU fallback for the HIP backend)")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.ffl_device_count.restype = C.c_int
    L.ffl_create.argtypes = [C.c_int] * 6 + [C.POINTER(vp)]
    L.ffl_destroy.argtypes = [vp]
    L.ffl_destroy.restype = None
    L.ffl_last_error.argtypes = [vp]
    L.ffl_last_error.restype = C.c_char_p
    L.ffl_upload_frame.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t]
    L.ffl_upload_frames.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_ssize_t]
    L.ffl_upload_frames_raw.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_ssize_t, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int]
    L.ffl_flow_pairs.argtypes = [vp, C.c_int, ip, ip, ip, C.c_int]
    L.ffl_pass1_result.argtypes = [vp, C.c_int, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), ip]
    L.ffl_pass1_results.argtypes = [vp, C.c_int, ip, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), ip]
    L.ffl_radial.argtypes = [vp, C.c_int, ip, dp, dp, ip, C.c_int, dp]
    L.ffl_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.ffl_host_free.argtypes = [vp, vp]
    L.ffl_download_flow.argtypes = [vp, C.c_int, vp]
    L.ffl_download_frame.argtypes = [vp, C.c_int, vp]
    L.ffl_upload_flow.argtypes = [vp, C.c_int, vp, C.c_int]
    L.ffl_submit_pair.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int]
    L.ffl_sync.argtypes = [vp]
    L.ffl_num_levels.argtypes = [vp]
    L.ffl_level_size.argtypes = [vp, C.c_int, ip]
    L.ffl_debug_pair.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 6
    L.ffl_profile_enable.argtypes = [vp, C.c_uint]
    L.ffl_profile_read.argtypes = [vp, C.c_int, ip, dp]
    L.ffl_kernel_name.argtypes = [C.c_int]
    L.ffl_kernel_name.restype = C.c_char_p
    L.ffl_set_option.argtypes = [C.c_char_p, C.c_int]
    L.ffl_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.ffl_ctx_get_option.argtypes = [vp, C.c_char_p, ip]
    L.ffl_graph_stats.argtypes = [vp, ip, ip, ip]
    L.ffl_device_mem_info.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ffl_estimate_bytes.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    _lib = L
    return L


def set_option(name, value):
    """Process-wide DEFAULT of a tuning knob (ffl_set_option): contexts created afterwards start from it; live contexts
    keep their own copy (Context.set_option).  Results do not depend on options."""
    if load().ffl_set_option(name.encode(), int(value)) != FFL_OK:
        raise FFLError(f"ffl_set_option({name!r}, {value}) rejected")


def get_option(name):
    """The process-wide default of a knob (ffl_ctx_get_option with no context)."""
    v = C.c_int()
    if load().ffl_ctx_get_option(None, name.encode(), C.byref(v)) != FFL_OK:
        raise FFLError(f"unknown option {name!r}")
    return v.value


def device_count():
    return load().ffl_device_count()


def device_mem_info(device=0):
    """(free, total) bytes of device memory (hipMemGetInfo)."""
    f, t = C.c_size_t(), C.c_size_t()
    L = load()
    if L.ffl_device_mem_info(int(device), C.byref(f), C.byref(t)) != FFL_OK:
        raise FFLError(f"ffl_device_mem_info failed: {L.ffl_last_error(None).decode()}")
    return f.value, t.value


def estimate_bytes(width, height, frame_slots, flow_slots, max_batch):
    """(device, pinned host) bytes a Context with these arguments allocates under the current "lanes" option."""
    d, p = C.c_size_t(), C.c_size_t()
    L = load()
    if L.ffl_estimate_bytes(int(width), int(height), int(frame_slots), int(flow_slots), int(max_batch), C.byref(d), C.byref(p)) != FFL_OK:
        raise FFLError(f"ffl_estimate_bytes failed: {L.ffl_last_error(None).decode()}")
    return d.value, p.value


def _iarr(v):
    """int array argument: a C int pointer plus the object that keeps the memory alive (numpy does the conversion of a list
    or an array in C; element-by-element ctypes construction cost 30 us per 256 entries, three times per batch)"""
    a = np.ascontiguousarray(v, dtype=np.intc)
    return a.ctypes.data_as(C.POINTER(C.c_int)), a


def _darr(v):
    a = np.ascontiguousarray(v, dtype=np.float64)
    return a.ctypes.data_as(C.POINTER(C.c_double)), a


class Context:
    """One device context for frames of a fixed size (ffl_create / ffl_destroy)."""

    def __init__(self, width, height, device=0, frame_slots=None, flow_slots=None, max_batch=8):
        self.L = load()
        self.width, self.height, self.device, self.max_batch = int(width), int(height), int(device), int(max_batch)
        self.frame_slots = int(frame_slots if frame_slots is not None else 2 * max_batch + 2)
        self.flow_slots = int(flow_slots if flow_slots is not None else 13 + 2 * max_batch)
        h = C.c_void_p()
        rc = self.L.ffl_create(self.device, self.width, self.height, self.frame_slots, self.flow_slots, self.max_batch,
                               C.byref(h))
        if rc != FFL_OK:
            raise FFLError(f"ffl_create failed ({rc}): {self.L.ffl_last_error(None).decode()}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.ffl_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != FFL_OK:
            raise FFLError(f"ffl error {rc}: {self.L.ffl_last_error(self._h).decode()}")

    # ---- frames -------------------------------------------------------------------------------
    def upload_frame(self, fslot, frame):
        """frame: C-contiguous-rows uint8 (H,W) gray or (H,W,3) BGR, exactly what cv2 hands Python."""
        if frame.dtype != np.uint8 or frame.ndim not in (2, 3):
            raise FFLError("frame must be uint8 (H,W) or (H,W,3)")
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        if frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != ch):
            frame = np.ascontiguousarray(frame)
        self._chk(self.L.ffl_upload_frame(self._h, fslot, frame.ctypes.data, frame.shape[1], frame.shape[0], ch,
                                          frame.strides[0]))

    def upload_frames(self, first_slot, frames):
        """frames: sequence of equally shaped uint8 frames -> consecutive slots, one H2D transfer."""
        f0 = frames[0]
        shape, strides = f0.shape, f0.strides
        if f0.dtype == np.uint8 and f0.ndim in (2, 3) and strides[-1] == 1 and (f0.ndim == 2 or strides[1] == shape[2]):
            # the common case -- every frame an ndarray of frame 0's shape and strides -- in one pass: at 256x256 a batch is
            # 257 frames, and what is done per frame in Python here is what the upload call costs the host
            n = len(frames)
            ptrs = np.empty(n, np.uintp)
            uniform = True
            for i, f in enumerate(frames):
                if f.shape != shape or f.strides != strides or f.dtype != np.uint8:
                    uniform = False          # the general path below converts or refuses it
                    break
                ptrs[i] = f.__array_interface__["data"][0]
            if uniform:
                ch = 1 if f0.ndim == 2 else shape[2]
                self._chk(self.L.ffl_upload_frames(self._h, first_slot, n, ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), shape[1], shape[0],
                                                   ch, strides[0]))
                return
        fr = [f if (f.strides[-1] == 1 and (f.ndim == 2 or f.strides[1] == f.shape[2])) else np.ascontiguousarray(f)
              for f in frames]
        f0 = fr[0]
        ch = 1 if f0.ndim == 2 else f0.shape[2]
        if any(f.dtype != np.uint8 or f.shape != f0.shape or f.strides[0] != f0.strides[0] for f in fr):
            raise FFLError("upload_frames needs uint8 frames of one shape and row stride")
        ptrs = (C.c_void_p * len(fr))(*[f.ctypes.data for f in fr])
        self._chk(self.L.ffl_upload_frames(self._h, first_slot, len(fr), ptrs, f0.shape[1], f0.shape[0], ch,
                                           f0.strides[0]))

    def pinned_frames(self, n, channels=1, size=None):
        """(n, height, width[, 3]) uint8 array in page-locked memory of this context (ffl_host_alloc): frames a
        decoder writes into consecutive entries go to the device without the staging copy.  Do not overwrite an
        entry before the batch that uses it has returned results (or ctx.sync()).  `size=(w, h)`: decoded source
        frames of another size, for upload_frames_raw.  The memory belongs to the context: the array (and every
        view of it) must not be touched after ctx.close()."""
        w, h = size if size is not None else (self.width, self.height)
        shape = (n, h, w) + ((channels,) if channels != 1 else ())
        nbytes = int(np.prod(shape))
        p = C.c_void_p()
        self._chk(self.L.ffl_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(shape)

    def upload_frames_raw(self, first_slot, frames, resize, crop=(0, 0), rgb_order=False):
        """Decoded (h, w, 3) uint8 frames -> gray(resize(frame, resize)[crop window]) in consecutive slots
        (cv2.resize / cv2.cvtColor 8-bit rules on the device; FF:182-186, FF:1076-1082)."""
        fr = [f if (f.ndim == 3 and f.strides[2] == 1 and f.strides[1] == 3) else np.ascontiguousarray(f) for f in frames]
        f0 = fr[0]
        if any(f.dtype != np.uint8 or f.ndim != 3 or f.shape != f0.shape or f.shape[2] != 3 or
               f.strides[0] != f0.strides[0] for f in fr):
            raise FFLError("upload_frames_raw needs (h, w, 3) uint8 frames of one shape and row stride")
        ptrs = (C.c_void_p * len(fr))(*[f.ctypes.data for f in fr])
        self._chk(self.L.ffl_upload_frames_raw(self._h, first_slot, len(fr), ptrs, f0.shape[1], f0.shape[0],
                                               f0.strides[0], int(bool(rgb_order)), int(resize[0]), int(resize[1]),
                                               int(crop[0]), int(crop[1])))

    def flow_pairs(self, fslot0, fslot1, flow_slots, pov_mode=False):
        n = len(flow_slots)
        (p0, k0), (p1, k1), (ps, ks) = _iarr(fslot0), _iarr(fslot1), _iarr(flow_slots)
        if len(k0) != n or len(k1) != n:
            raise FFLError("flow_pairs: the three slot lists must have one entry per pair")
        self._chk(self.L.ffl_flow_pairs(self._h, n, p0, p1, ps, int(bool(pov_mode))))

    def submit_pair(self, slot, prev, nxt, pov_mode=False):
        prev, nxt = np.ascontiguousarray(prev), np.ascontiguousarray(nxt)
        ch = 1 if prev.ndim == 2 else prev.shape[2]
        self._chk(self.L.ffl_submit_pair(self._h, slot, prev.ctypes.data, nxt.ctypes.data, prev.shape[1], prev.shape[0],
                                         ch, prev.strides[0], int(bool(pov_mode))))

    def pass1_result(self, flow_slot, cut_threshold=7.0):
        x, y, c = C.c_int32(), C.c_int32(), C.c_int()
        v, mm = C.c_float(), C.c_float()
        self._chk(self.L.ffl_pass1_result(self._h, flow_slot, float(cut_threshold), C.byref(x), C.byref(y), C.byref(v),
                                          C.byref(mm), C.byref(c)))
        return x.value, y.value, np.float32(v.value), np.float32(mm.value), bool(c.value)

    def pass1_results(self, flow_slots, cut_threshold=7.0):
        """ffl_pass1_results: the records of many slots with one call (list of pass1_result tuples)."""
        n = len(flow_slots)
        x, y, c = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.intc)
        v, mm = np.empty(n, np.float32), np.empty(n, np.float32)
        i32, f32 = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        ps, ks = _iarr(flow_slots)
        self._chk(self.L.ffl_pass1_results(self._h, n, ps, float(cut_threshold), x.ctypes.data_as(i32), y.ctypes.data_as(i32),
                                           v.ctypes.data_as(f32), mm.ctypes.data_as(f32), c.ctypes.data_as(C.POINTER(C.c_int))))
        return list(zip(x.tolist(), y.tolist(), list(v), list(mm), (c != 0).tolist()))

    def radial(self, flow_slots, centers, is_cut, pov_mode=False):
        n = len(flow_slots)
        out = np.empty(n, np.float64)
        cen = np.asarray(centers, np.float64).reshape(n, 2)
        (ps, ks), (px, kx), (py, ky) = _iarr(flow_slots), _darr(cen[:, 0]), _darr(cen[:, 1])
        pc, kc = _iarr(np.asarray(is_cut, bool))
        self._chk(self.L.ffl_radial(self._h, n, ps, px, py, pc, int(bool(pov_mode)), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.tolist()

    def download_frame(self, fslot):
        out = np.empty((self.height, self.width), np.uint8)
        self._chk(self.L.ffl_download_frame(self._h, fslot, out.ctypes.data))
        return out

    def download_flow(self, flow_slot):
        out = np.empty((self.height, self.width, 2), np.float32)
        self._chk(self.L.ffl_download_flow(self._h, flow_slot, out.ctypes.data))
        return out

    def upload_flow(self, flow_slot, flow, pov_mode=False):
        flow = np.ascontiguousarray(flow, np.float32)
        if flow.shape != (self.height, self.width, 2):
            raise FFLError(f"flow shape {flow.shape} does not match context {(self.height, self.width, 2)}")
        self._chk(self.L.ffl_upload_flow(self._h, flow_slot, flow.ctypes.data, int(bool(pov_mode))))

    def sync(self):
        self._chk(self.L.ffl_sync(self._h))

    def set_option(self, name, value):
        """A knob of THIS context only (ffl_ctx_set_option); "lanes" is fixed at creation."""
        self._chk(self.L.ffl_ctx_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int()
        self._chk(self.L.ffl_ctx_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def graph_stats(self):
        """{"captured", "replayed", "capture_failures"} of this context's hipGraph path (ffl_graph_stats)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.ffl_graph_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"captured": a.value, "replayed": b.value, "capture_failures": c.value}

    # ---- test / measurement hooks --------------------------------------------------------------
    def num_levels(self):
        return self.L.ffl_num_levels(self._h)

    def level_size(self, level):
        wh = (C.c_int * 2)()
        self._chk(self.L.ffl_level_size(self._h, level, wh))
        return wh[0], wh[1]

    def debug_pair(self, f0, f1, level, it):
        lw, lh = self.level_size(level)
        d = dict(I0=np.empty((lh, lw), np.float32), I1=np.empty((lh, lw), np.float32),
                 R0=np.empty((5, lh, lw), np.float32), R1=np.empty((5, lh, lw), np.float32),
                 M=np.empty((5, lh, lw), np.float32), flow=np.empty((lh, lw, 2), np.float32))
        self._chk(self.L.ffl_debug_pair(self._h, f0, f1, level, it, d["I0"].ctypes.data, d["I1"].ctypes.data,
                                        d["R0"].ctypes.data, d["R1"].ctypes.data, d["M"].ctypes.data,
                                        d["flow"].ctypes.data))
        d["out"] = self.download_flow(0)
        return d

    def profile_enable(self, classes=True):
        """classes: True (all), False/None (off) or an iterable of kernel-class names."""
        if classes is True:
            mask = 0xFF
        elif not classes:
            mask = 0
        else:
            mask = sum(1 << KERNEL_CLASSES.index(c) for c in classes)
        self._chk(self.L.ffl_profile_enable(self._h, mask))

    def profile_read(self):
        out = {}
        for k, name in enumerate(KERNEL_CLASSES):
            n, ms = C.c_int(), C.c_double()
            self._chk(self.L.ffl_profile_read(self._h, k, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out
