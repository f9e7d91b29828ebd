"""ctypes binding of libwol.so (include/wol.h).  There is NO fallback: if the HIP library is
missing or no GPU is usable, every entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WOL_LIB_PATH") or os.path.join(_HERE, "libwol.so")  # env override: tuning builds

MEM_HOST, MEM_DEVICE = 0, 1
F64, F32 = 0, 1

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
           "wol_wat_orient", "wol_wat_oh_vec", "wol_general_hbonds", "wol_general_hbonds_list",
           "wol_q_frame_t", "wol_cos_angs_t", "wol_lsi_frame_t",
           "wol_shard_bounds", "wol_comm_unique_id", "wol_comm_init", "wol_comm_adopt", "wol_comm_destroy",
           "wol_q_traj_sharded", "wol_q_traj_sharded_f32",
           "wol_q_traj_pops", "wol_cos_angs_traj", "wol_lsi_traj", "wol_radial_dist_same_traj"]

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


def _preload_torch_rccl():
    """libwol resolves RCCL at run time and prefers the copy the process already maps; next to PyTorch-ROCm that must
    be torch's bundled librccl (same reason as the HIP runtime above).  Importing torch maps it; without torch in the
    process this maps it from the wheel if there is one, else libwol falls back to the system's librccl.so.1."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


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
    L.wol_q_frame_t.argtypes = L.wol_q_frame.argtypes[:-1] + [C.c_int, C.c_int]
    L.wol_cos_angs_t.argtypes = L.wol_cos_angs.argtypes + [C.c_int, C.c_int]
    L.wol_lsi_frame_t.argtypes = L.wol_lsi_frame.argtypes + [C.c_int, C.c_int]
    L.wol_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.wol_comm_unique_id.argtypes = [_vp]
    L.wol_comm_init.argtypes = [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]
    L.wol_comm_adopt.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(_vp)]
    L.wol_comm_destroy.argtypes = [_vp]
    sharded = [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, _vp,
               C.c_int, _vp, C.c_int, C.c_double, _vp, C.c_int]
    L.wol_q_traj_sharded.argtypes = sharded
    L.wol_q_traj_sharded_f32.argtypes = sharded
    L.wol_q_traj_pops.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, _vp,
                                  C.c_int, _vp, C.c_int, C.c_double, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int]
    L.wol_cos_angs_traj.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, C.c_int,
                                    C.c_int, _vp, _vp, _vp, C.c_int, C.c_int]
    L.wol_lsi_traj.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp, C.c_int,
                               C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]
    L.wol_radial_dist_same_traj.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp, C.c_double, C.c_int, C.c_double,
                                            _vp, _vp, C.c_int, C.c_int]
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


class Comm:
    """RCCL communicator of a frame-sharded job (include/wol.h: wol_comm_*), one rank per GPU.  Rank 0 creates the
    128-byte unique id (Comm.unique_id()), the caller distributes it, every rank constructs Comm(ctx, rank, world, id)."""

    def __init__(self, ctx, rank, world, unique_id):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        _preload_torch_rccl()
        h = _vp()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(load().wol_comm_init(ctx.handle, buf, int(rank), int(world), C.byref(h)))
        self._h, self.rank, self.world, self.ctx = h, int(rank), int(world), ctx

    @staticmethod
    def unique_id():
        _preload_torch_rccl()
        buf = C.create_string_buffer(128)
        check(load().wol_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, ctx, group=None):
        """the unique id travels through the (already initialised) torch.distributed process group"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(ctx, rank, world, box[0])

    @property
    def handle(self):
        if not self._h:
            raise WolError("communicator is closed")
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            load().wol_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(T, rank, world):
    a, b = C.c_int(0), C.c_int(0)
    check(load().wol_shard_bounds(int(T), int(rank), int(world), C.byref(a), C.byref(b)))
    return a.value, b.value


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
