"""ctypes binding of libwol.so (include/wol.h).  There is NO fallback: if the HIP library is
missing or no GPU is usable, every entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WOL_LIB_PATH") or os.path.join(_HERE, "libwol.so")  # env override: tuning builds

MEM_HOST, MEM_DEVICE = 0, 1

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p

SYMBOLS = ["wol_version", "wol_last_error", "wol_ctx_create", "wol_ctx_destroy", "wol_ctx_set_stream",
           "wol_ctx_synchronize", "wol_ctx_set_option", "wol_ctx_kernel_time", "wol_ctx_get_counter", "wol_reimage",
           "wol_near_neighbors", "wol_all_near_neighbors", "wol_tetra_cos_ang", "wol_lsi_dists",
           "wol_q_frame", "wol_cos_angs", "wol_lsi_frame", "wol_q_traj", "wol_q_traj_f32",
           "wol_willard_density_field", "wol_willard_density_points",
           "wol_radial_dist_same", "wol_radial_dist", "wol_pair_distance_histogram",
           "wol_wat_orient", "wol_wat_oh_vec", "wol_general_hbonds", "wol_general_hbonds_list"]

_lib = None


class WolError(RuntimeError):
    pass


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same soname,
    libamdhip64.so.7, as /opt/rocm's).  libwol.so must bind to the copy torch uses, otherwise two
    runtimes initialise the GPU and the second finds none; device pointers and streams handed over
    from torch also only make sense inside torch's runtime.  So: if torch is installed, map its
    libamdhip64 first (without importing torch); libwol's NEEDED entry then resolves to it by
    soname.  Without torch, libwol's RUNPATH (/opt/rocm/lib) applies."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Loads libwol.so; raises if it has not been built (python -m waterorderlib_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WolError("libwol.so not found at %s -- build it with `python -m waterorderlib_amd.build` "
                       "(there is no CPU fallback)" % LIB_PATH)
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.wol_last_error.restype = C.c_char_p
    L.wol_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
    L.wol_ctx_destroy.argtypes = [_vp]
    L.wol_ctx_set_stream.argtypes = [_vp, _vp]
    L.wol_ctx_synchronize.argtypes = [_vp]
    L.wol_ctx_set_option.argtypes = [_vp, C.c_char_p, C.c_double]
    L.wol_ctx_kernel_time.argtypes = [_vp, _dp, _lp, C.c_int]
    L.wol_ctx_get_counter.argtypes = [_vp, C.c_char_p, _lp]
    L.wol_reimage.argtypes = [_vp, _vp, C.c_int, _vp, _vp, _vp]
    L.wol_near_neighbors.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp]
    L.wol_all_near_neighbors.argtypes = [_vp, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp]
    L.wol_tetra_cos_ang.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp]
    L.wol_lsi_dists.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp]
    L.wol_q_frame.argtypes = [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64, C.c_int, _vp, C.c_double,
                              C.c_double, _vp, _vp, _vp, _vp, C.c_int]
    L.wol_cos_angs.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, _lp]
    L.wol_lsi_frame.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, _lp]
    L.wol_willard_density_field.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp,
                                            C.c_double, _vp, _vp]
    L.wol_willard_density_points.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_double, _vp, _vp]
    L.wol_radial_dist_same.argtypes = [_vp, _vp, C.c_int, C.c_double, C.c_int, C.c_double, _vp, _vp]
    L.wol_radial_dist.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_int, C.c_double, _vp, _vp]
    L.wol_pair_distance_histogram.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp]
    L.wol_wat_orient.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.wol_wat_oh_vec.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp]
    L.wol_general_hbonds.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp]
    L.wol_general_hbonds_list.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_double, C.c_double, C.c_int64,
                                          _vp, _vp, _vp]
    traj = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, _vp,
            C.c_int, _vp, C.c_int, C.c_double, _vp, C.c_int]
    L.wol_q_traj.argtypes = traj
    L.wol_q_traj_f32.argtypes = traj
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise WolError("libwol error %d: %s" % (rc, load().wol_last_error().decode()))


def ptr(a):
    """host numpy array -> void* (None passes NULL)"""
    if a is None:
        return None
    return a.ctypes.data_as(_vp)


class Context:
    """One per host thread / GPU (include/wol.h: wol_ctx)."""

    def __init__(self, device=0):
        L = load()
        h = _vp()
        check(L.wol_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            load().wol_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise WolError("context is closed")
        return self._h

    def set_option(self, key, value):
        check(load().wol_ctx_set_option(self.handle, key.encode(), float(value)))

    def set_stream(self, stream_ptr):
        """stream_ptr: hipStream_t as an int; 0/None = HIP's default stream (torch's default stream)"""
        check(load().wol_ctx_set_stream(self.handle, _vp(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        check(load().wol_ctx_synchronize(self.handle))

    def counter(self, key):
        v = C.c_int64(0)
        check(load().wol_ctx_get_counter(self.handle, key.encode(), C.byref(v)))
        return v.value

    def kernel_time(self, reset=False):
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        check(load().wol_ctx_kernel_time(self.handle, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value


_default_ctx = {}


def default_context(device=None):
    """Lazily created per-device context used by the reference-style module functions."""
    if device is None:
        device = int(os.environ.get("WOL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        try:
            import ctypes.util  # noqa: F401
        except Exception:
            pass
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx


def as_f64(a, order="C"):
    return np.require(a, dtype=np.float64, requirements=["C_CONTIGUOUS" if order == "C" else "F_CONTIGUOUS", "ALIGNED"])


def box3(b):
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1))
    if b.size < 3:
        raise ValueError("BoxDims must hold three box lengths")
    return np.ascontiguousarray(b[:3])
