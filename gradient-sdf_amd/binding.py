"""ctypes binding to the C-ABI of libgsdf.so (include/gsdf.h).

This is the thin host-side mirror used by tests/ and bench.py.  It contains no
arithmetic of the hot path: every call goes through the C-ABI into the gfx950
kernels.  If the shared library or a GPU is missing, loading / gsdf_create fail
loudly -- there is NO CPU fallback.

The library is linked without a DT_NEEDED on libamdhip64 (csrc/Makefile) so that
a process holds exactly one HIP runtime: when PyTorch (which bundles its own
copy) is already imported we bind to that one, otherwise to /opt/rocm's.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GSDF_LIB", os.path.join(CSRC, "libgsdf.so"))   # GSDF_LIB: kernel-variant experiments (tools/)
# the same sources with -DGSDF_EXPERIMENTS: path-forcing hooks for the tests, measurement switches for tools/
TEST_LIB_PATH = os.environ.get("GSDF_TEST_LIB", os.path.join(CSRC, "libgsdf_test.so"))   # GSDF_TEST_LIB: variants of the test build (tools/)

GSDF_OK, ERR_TABLE_FULL, ERR_KEY_RANGE, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE = 0, 1, 2, 3, 4, 5


class GsdfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gsdf error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("n_upd", C.c_int64), ("n_valid", C.c_int64), ("n_hit", C.c_int64),
                ("track_passes", C.c_int32), ("converged", C.c_int32), ("frames", C.c_int64),
                ("n_deferred", C.c_int64), ("fuse_timeouts", C.c_int64)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libgsdf.so, in-tree (csrc/Makefile)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "gsdf.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def _hip_runtime_path():
    if "torch" in sys.modules:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(p):
            return p
    for p in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"),
              "/opt/rocm/lib/libamdhip64.so"):
        if os.path.exists(p):
            return p
    raise OSError("libamdhip64.so not found (need ROCm): libgsdf has no CPU fallback")


_libs = {}
_hip = None


def load_test_lib():
    """libgsdf_test.so: the -DGSDF_EXPERIMENTS build with per-context gsdf_debug_flags(ctx, flags) (tests/, tools/ only)."""
    L = load(TEST_LIB_PATH)
    L.gsdf_debug_flags.restype = C.c_int
    L.gsdf_debug_flags.argtypes = [C.c_void_p, C.c_int]
    L.gsdf_debug_read.restype = C.c_int
    L.gsdf_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.gsdf_debug_trace.restype = C.c_int
    L.gsdf_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    for name, args in (("gsdf_debug_tile_counters", [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]),
                       ("gsdf_debug_set_tile_order", [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
                       ("gsdf_debug_get_tile_order", [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)])):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = args
    return L


def load(path=None):
    """Load libgsdf.so (must already be built: the .so travels in-tree to the GPU box)."""
    global _hip
    path = LIB_PATH if path is None else path
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OSError("%s missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % path)
    if _hip is None:
        _hip = C.CDLL(_hip_runtime_path(), mode=C.RTLD_GLOBAL)
    L = C.CDLL(path)
    fp, i32p, i64p, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_void_p
    sig = {
        "gsdf_last_error": (C.c_char_p, []),
        "gsdf_version": (C.c_char_p, []),
        "gsdf_create": (C.c_int, [C.POINTER(vp), C.c_float, C.c_float, C.c_int, C.c_int]),
        "gsdf_destroy": (None, [vp]),
        "gsdf_reset": (C.c_int, [vp]),
        "gsdf_set_zrange": (C.c_int, [vp, C.c_float, C.c_float]),
        "gsdf_normals_init": (C.c_int, [vp, C.c_int, C.c_int, fp, C.c_int]),
        "gsdf_normals_cache": (C.c_int, [vp, fp]),
        "gsdf_normals_compute": (C.c_int, [vp, fp, fp, fp, fp]),
        "gsdf_update": (C.c_int, [vp, fp, fp, fp]),
        "gsdf_update_dev": (C.c_int, [vp, vp, fp, fp]),
        "gsdf_track": (C.c_int, [vp, fp, fp, fp, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "gsdf_track_sampled": (C.c_int, [vp, fp, fp, fp, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "gsdf_set_pose": (C.c_int, [vp, fp]),
        "gsdf_get_pose": (C.c_int, [vp, fp]),
        "gsdf_track_and_fuse_dev": (C.c_int, [vp, vp, fp, C.c_int, C.c_float, C.c_float]),
        "gsdf_hint_next_depth_dev": (C.c_int, [vp, vp]),
        "gsdf_track_and_fuse_ahead_dev": (C.c_int, [vp, vp, vp, fp, C.c_int, C.c_float, C.c_float]),
        "gsdf_read_frame_log": (C.c_int, [vp, fp, C.c_int64, i64p]),
        "gsdf_sync": (C.c_int, [vp]),
        "gsdf_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "gsdf_count": (C.c_int, [vp, i64p]),
        "gsdf_export": (C.c_int, [vp, i32p, fp, C.c_int64, i64p, C.c_int, C.c_int]),
        "gsdf_enable_vis": (C.c_int, [vp, C.c_int]),
        "gsdf_export_vis": (C.c_int, [vp, i32p, C.POINTER(C.c_uint32), C.c_int, C.c_int64, i64p]),
        "gsdf_ba_setup": (C.c_int, [vp, C.c_int, fp, fp, C.POINTER(C.c_int), C.c_float]),
        "gsdf_ba_set_loss": (C.c_int, [vp, C.c_int, C.c_float]),
        "gsdf_ba_energy": (C.c_int, [vp, fp]),
        "gsdf_ba_solve_pose": (C.c_int, [vp, C.c_float]),
        "gsdf_ba_solve_dist": (C.c_int, [vp, C.c_float]),
        "gsdf_ba_optimize": (C.c_int, [vp, C.c_int, fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "gsdf_ba_get_poses": (C.c_int, [vp, fp]),
        "gsdf_ba_counters": (C.c_int, [vp, i64p, i64p]),
        "gsdf_merge_prepare": (C.c_int, [vp, C.c_int]),
        "gsdf_grow": (C.c_int, [vp, C.c_int]),
        "gsdf_set_auto_grow": (C.c_int, [vp, C.c_int]),
        "gsdf_capacity": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "gsdf_merge_from": (C.c_int, [vp, vp]),
        "gsdf_create_shards": (C.c_int, [C.POINTER(vp), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]),
        "gsdf_merge_raw": (C.c_int, [vp, i32p, fp, C.c_int64]),
        "gsdf_export_raw_dev": (C.c_int, [vp, vp, vp, C.c_int64, i64p]),
        "gsdf_merge_raw_dev": (C.c_int, [vp, vp, vp, C.c_int64]),
        "gsdf_query": (C.c_int, [vp, fp, C.c_int64, fp, fp, fp]),
        "gsdf_get_voxels": (C.c_int, [vp, i32p, C.c_int64, fp, i32p]),
        "gsdf_block_keys_dev": (C.c_int, [vp, vp, C.c_int64, C.POINTER(C.c_int64)]),
        "gsdf_pack_blocks_dev": (C.c_int, [vp, vp, C.c_int64, vp]),
        "gsdf_unpack_blocks_dev": (C.c_int, [vp, vp, C.c_int64, vp]),
        "gsdf_merge_allreduce": (C.c_int, [vp, vp, i64p, i64p]),
        "gsdf_merge_allreduce_with": (C.c_int, [vp, vp, i64p, i64p]),
        "gsdf_rccl_unique_id": (C.c_int, [C.c_char_p]),
        "gsdf_rccl_comm_init": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_int, C.c_int]),
        "gsdf_rccl_comm_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "gsdf_rccl_comm_destroy": (C.c_int, [vp]),
        "gsdf_raycast_dev": (C.c_int, [vp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]),
        "gsdf_raycast_counters": (C.c_int, [vp, i64p, i64p, C.c_int]),
        "gsdf_profile_read_n": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), i64p]),
        "gsdf_profile_read_launches": (C.c_int, [vp, C.c_int, fp, C.c_int64, i64p]),
        "gsdf_raycast": (C.c_int, [vp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]),
        "gsdf_extract_mesh": (C.c_int, [vp, C.c_float, C.POINTER(C.c_int8), fp, C.c_int64, C.POINTER(C.c_int64)]),
        "gsdf_dev_alloc": (C.c_int, [vp, C.POINTER(vp), C.c_int64]),
        "gsdf_dev_free": (C.c_int, [vp, vp]),
        "gsdf_dev_upload": (C.c_int, [vp, vp, vp, C.c_int64]),
        "gsdf_dev_download": (C.c_int, [vp, vp, vp, C.c_int64]),
        "gsdf_host_alloc": (C.c_int, [vp, C.POINTER(vp), C.c_int64]),
        "gsdf_host_free": (C.c_int, [vp, vp]),
        "gsdf_dev_upload_async": (C.c_int, [vp, vp, vp, C.c_int64]),
        "gsdf_mark": (C.c_int, [vp, i64p]),
        "gsdf_mark_wait": (C.c_int, [vp, C.c_int64]),
        "gsdf_mark_reached": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_int)]),
        "gsdf_dev_upload_ahead": (C.c_int, [vp, vp, vp, C.c_int64, i64p]),
        "gsdf_upload_wait": (C.c_int, [vp, C.c_int64]),
        "gsdf_timer_start": (C.c_int, [vp]),
        "gsdf_timer_stop_ms": (C.c_int, [vp, fp]),
        "gsdf_profile": (C.c_int, [vp, C.c_int]),
        "gsdf_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), i64p]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("GSDF_LIB_OLD") and not hasattr(L, name):
            continue                   # tools only: an older build of the library under A/B (GSDF_LIB=..., GSDF_LIB_OLD=1)
        f = getattr(L, name)       # AttributeError if the .so does not export a declared symbol
        f.restype = res
        f.argtypes = args
    _libs[path] = L
    return L


ABI_SYMBOLS = [
    "gsdf_last_error", "gsdf_version", "gsdf_create", "gsdf_destroy", "gsdf_reset", "gsdf_set_zrange",
    "gsdf_normals_init", "gsdf_normals_cache", "gsdf_normals_compute", "gsdf_update", "gsdf_update_dev",
    "gsdf_track", "gsdf_track_sampled", "gsdf_hint_next_depth_dev", "gsdf_track_and_fuse_ahead_dev", "gsdf_set_pose", "gsdf_get_pose", "gsdf_track_and_fuse_dev", "gsdf_read_frame_log",
    "gsdf_sync", "gsdf_get_stats", "gsdf_count", "gsdf_export", "gsdf_enable_vis", "gsdf_export_vis",
    "gsdf_ba_setup", "gsdf_ba_set_loss", "gsdf_ba_energy", "gsdf_ba_solve_pose", "gsdf_ba_solve_dist", "gsdf_ba_optimize", "gsdf_ba_get_poses", "gsdf_ba_counters", "gsdf_grow", "gsdf_set_auto_grow", "gsdf_capacity", "gsdf_merge_from", "gsdf_create_shards", "gsdf_merge_prepare",
    "gsdf_merge_raw", "gsdf_export_raw_dev",
    "gsdf_merge_raw_dev", "gsdf_block_keys_dev", "gsdf_pack_blocks_dev", "gsdf_unpack_blocks_dev",
    "gsdf_merge_allreduce", "gsdf_merge_allreduce_with", "gsdf_rccl_unique_id", "gsdf_rccl_comm_init", "gsdf_rccl_comm_count",
    "gsdf_rccl_comm_destroy",
    "gsdf_query", "gsdf_get_voxels", "gsdf_raycast", "gsdf_raycast_dev", "gsdf_raycast_counters", "gsdf_extract_mesh",
    "gsdf_dev_alloc", "gsdf_dev_free", "gsdf_dev_upload", "gsdf_dev_download", "gsdf_timer_start", "gsdf_timer_stop_ms",
    "gsdf_host_alloc", "gsdf_host_free", "gsdf_dev_upload_async", "gsdf_mark", "gsdf_mark_wait", "gsdf_mark_reached",
    "gsdf_dev_upload_ahead", "gsdf_upload_wait",
    "gsdf_profile", "gsdf_profile_read", "gsdf_profile_read_n", "gsdf_profile_read_launches",
]


def rccl_unique_id():
    """128-byte RCCL unique id (rank 0 creates it and hands it to the other ranks)."""
    L = load()
    buf = C.create_string_buffer(128)
    rc = L.gsdf_rccl_unique_id(buf)
    if rc != GSDF_OK:
        raise GsdfError(rc, L.gsdf_last_error().decode())
    return buf.raw


def rccl_comm_init(nranks, unique_id, rank, device):
    L = load()
    comm = C.c_void_p()
    rc = L.gsdf_rccl_comm_init(C.byref(comm), int(nranks), unique_id, int(rank), int(device))
    if rc != GSDF_OK:
        raise GsdfError(rc, L.gsdf_last_error().decode())
    return comm


def rccl_comm_count(comm):
    """ncclCommCount of the communicator."""
    L = load()
    n = C.c_int(0)
    rc = L.gsdf_rccl_comm_count(comm, C.byref(n))
    if rc != GSDF_OK:
        raise GsdfError(rc, L.gsdf_last_error().decode())
    return n.value


def rccl_comm_destroy(comm):
    load().gsdf_rccl_comm_destroy(comm)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class GradSdf:
    """MapGradPixelSdf + RigidPointOptimizer + NormalEstimator behind the C-ABI (one GPU)."""

    def __init__(self, voxel_size, trunc_dist, W, H, K, win=11, capacity_log2=22, device=0,
                 zmin=0.5, zmax=3.5, lib=None, _handle=None):
        self.L = load() if lib is None else lib         # lib: load_test_lib() for the path-forcing tests
        self.h = C.c_void_p()
        self._dev = []                                  # before anything can raise: close() / __del__ walk it
        if _handle is not None:                         # (GradSdf.shards: the context exists already)
            self.h = _handle
        else:
            self._chk(self.L.gsdf_create(C.byref(self.h), np.float32(voxel_size), np.float32(trunc_dist),
                                         int(capacity_log2), int(device)))
        self.W, self.H = int(W), int(H)
        self.K = _f32(K).reshape(9).copy()
        self._chk(self.L.gsdf_set_zrange(self.h, np.float32(zmin), np.float32(zmax)))
        self._chk(self.L.gsdf_normals_init(self.h, self.W, self.H, _fp(self.K), int(win)))

    @classmethod
    def shards(cls, n, voxel_size, trunc_dist, W, H, K, capacity_log2=22, device=0, **kw):
        """n contexts for n frame shards on one device, each stream in its own hardware queue (gsdf_create_shards); add them up
        with first.merge_from(other)."""
        L = kw.get("lib") or load()
        hs = (C.c_void_p * n)()
        rc = L.gsdf_create_shards(hs, int(n), np.float32(voxel_size), np.float32(trunc_dist), int(capacity_log2), int(device))
        if rc != GSDF_OK:
            raise GsdfError(rc, L.gsdf_last_error().decode())
        out = []
        try:
            for i in range(n):
                out.append(cls(voxel_size, trunc_dist, W, H, K, capacity_log2=capacity_log2, device=device, _handle=C.c_void_p(hs[i]), **kw))
        except Exception:
            # a constructor failed (set_zrange / normals_init): the handles not wrapped yet belong to nobody -- destroy them
            # (handle len(out) belongs to the half-built wrapper, whose __del__ closes it)
            for j in range(len(out) + 1, n):
                L.gsdf_destroy(C.c_void_p(hs[j]))
            for g in out:
                g.close()
            raise
        return out

    def _chk(self, rc):
        if rc != GSDF_OK:
            raise GsdfError(rc, self.L.gsdf_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            for p in self._dev:
                self.L.gsdf_dev_free(self.h, p)
            self._dev = []
            self.L.gsdf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._chk(self.L.gsdf_reset(self.h))

    def debug_flags(self, flags):
        """Path-forcing / measurement switches; exists only in the test build (lib=load_test_lib())."""
        self._chk(self.L.gsdf_debug_flags(self.h, int(flags)))

    # -- normals ------------------------------------------------------------------------------
    def normals_cache(self):
        out = np.empty((11, self.H, self.W), np.float32)
        self._chk(self.L.gsdf_normals_cache(self.h, _fp(out)))
        return out

    def normals(self, depth):
        d = _f32(depth).reshape(self.H, self.W)
        n = np.empty((3, self.H, self.W), np.float32)
        self._chk(self.L.gsdf_normals_compute(self.h, _fp(d), _fp(n[0]), _fp(n[1]), _fp(n[2])))
        return n

    # -- fusion -------------------------------------------------------------------------------
    def update(self, depth, R, t):
        d = _f32(depth).reshape(self.H, self.W)
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        self._chk(self.L.gsdf_update(self.h, _fp(d), _fp(R), _fp(t)))

    def download(self, dev_ptr, shape, dtype):
        """Copy a device buffer (pointer from upload / dev_alloc) into a new numpy array."""
        out = np.empty(shape, dtype)
        p = dev_ptr if isinstance(dev_ptr, C.c_void_p) else C.c_void_p(int(dev_ptr))
        self._chk(self.L.gsdf_dev_download(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    def upload(self, array):
        """Stage a host array in HBM; returns the device pointer (freed on close)."""
        a = np.ascontiguousarray(array)
        p = C.c_void_p()
        self._chk(self.L.gsdf_dev_alloc(self.h, C.byref(p), a.nbytes))
        self._dev.append(p)
        self._chk(self.L.gsdf_dev_upload(self.h, p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return p

    def update_dev(self, depth_dev, R, t):
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        self._chk(self.L.gsdf_update_dev(self.h, depth_dev, _fp(R), _fp(t)))

    # -- tracking -----------------------------------------------------------------------------
    def track(self, depth, pose7, iters=25, conv=1e-3, damping=1.0, sampling=None):
        """RigidPointOptimizer::optimize (sampling None) / ::optimize_sampled(depth, K, sampling)"""
        d = _f32(depth).reshape(self.H, self.W)
        p = _f32(pose7).reshape(7).copy()
        conv_flag, passes = C.c_int(0), C.c_int(0)
        if sampling is None:
            self._chk(self.L.gsdf_track(self.h, _fp(d), _fp(self.K), _fp(p), int(iters), np.float32(conv),
                                        np.float32(damping), C.byref(conv_flag), C.byref(passes)))
        else:
            self._chk(self.L.gsdf_track_sampled(self.h, _fp(d), _fp(self.K), _fp(p), int(iters), np.float32(conv),
                                                np.float32(damping), int(sampling), C.byref(conv_flag), C.byref(passes)))
        return bool(conv_flag.value), p, passes.value

    def set_pose(self, pose7):
        p = _f32(pose7).reshape(7)
        self._chk(self.L.gsdf_set_pose(self.h, _fp(p)))

    def get_pose(self):
        p = np.empty(7, np.float32)
        self._chk(self.L.gsdf_get_pose(self.h, _fp(p)))
        return p

    def track_and_fuse_dev(self, depth_dev, iters=25, conv=1e-3, damping=1.0):
        self._chk(self.L.gsdf_track_and_fuse_dev(self.h, depth_dev, _fp(self.K), int(iters), np.float32(conv),
                                                 np.float32(damping)))

    def track_and_fuse_ahead_dev(self, depth_dev, next_depth_dev, iters=25, conv=1e-3, damping=1.0):
        """hint_next_depth(next_depth_dev) + track_and_fuse_dev(depth_dev) as ONE call (next_depth_dev may be None)"""
        self._chk(self.L.gsdf_track_and_fuse_ahead_dev(self.h, depth_dev, next_depth_dev, _fp(self.K), int(iters), np.float32(conv),
                                                       np.float32(damping)))

    def hint_next_depth(self, depth_dev):
        """the frame the NEXT track_and_fuse_dev will get (call before the current frame's): its normals are then computed in the
        tail of the current frame's fusion launch -- time only, results unchanged"""
        self._chk(self.L.gsdf_hint_next_depth_dev(self.h, depth_dev))

    def frame_log(self):
        n = C.c_int64(0)
        self._chk(self.L.gsdf_read_frame_log(self.h, None, 0, C.byref(n)))
        rows = np.zeros((n.value, 10), np.float32)
        if n.value:
            self._chk(self.L.gsdf_read_frame_log(self.h, _fp(rows), n.value, C.byref(n)))
        return rows

    # -- state --------------------------------------------------------------------------------
    def sync(self):
        self._chk(self.L.gsdf_sync(self.h))

    def stats(self):
        s = Stats()
        self._chk(self.L.gsdf_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def count(self):
        n = C.c_int64(0)
        self._chk(self.L.gsdf_count(self.h, C.byref(n)))
        return n.value

    def export(self, sorted=True, raw=False):
        n = self.count()
        keys = np.empty((n, 3), np.int32)
        pay = np.empty((n, 5), np.float32)
        got = C.c_int64(0)
        if n:
            self._chk(self.L.gsdf_export(self.h, keys.ctypes.data_as(C.POINTER(C.c_int32)), _fp(pay), n,
                                         C.byref(got), int(sorted), int(raw)))
        return keys, pay

    def enable_vis(self, max_frames):
        self._vis_words = (int(max_frames) + 31) // 32
        self._chk(self.L.gsdf_enable_vis(self.h, int(max_frames)))

    def export_vis(self):
        n = self.count()
        keys = np.empty((n, 3), np.int32)
        words = np.zeros((n, self._vis_words), np.uint32)
        got = C.c_int64(0)
        if n:
            self._chk(self.L.gsdf_export_vis(self.h, keys.ctypes.data_as(C.POINTER(C.c_int32)),
                                             words.ctypes.data_as(C.POINTER(C.c_uint32)), self._vis_words, n, C.byref(got)))
        return keys, words

    # -- PhotoBA (PhotometricOptimizer) ----------------------------------------------------------
    def ba_setup(self, images_bgr, poses16, frame_idx, reg_weight=10.0):
        img = _f32(images_bgr)
        self._ba_n = img.shape[0]
        P = _f32(poses16).reshape(self._ba_n, 16)
        idx = np.ascontiguousarray(frame_idx, dtype=np.int32)
        self._chk(self.L.gsdf_ba_setup(self.h, self._ba_n, _fp(img), _fp(P), idx.ctypes.data_as(C.POINTER(C.c_int)),
                                       np.float32(reg_weight)))

    def ba_set_loss(self, loss, lam=0.5):
        self._chk(self.L.gsdf_ba_set_loss(self.h, int(loss), np.float32(lam)))

    def ba_energy(self):
        e = C.c_float(0)
        self._chk(self.L.gsdf_ba_energy(self.h, C.byref(e)))
        return e.value

    def ba_solve_pose(self, damping=1.0):
        self._chk(self.L.gsdf_ba_solve_pose(self.h, np.float32(damping)))

    def ba_solve_dist(self, damping=1.0):
        self._chk(self.L.gsdf_ba_solve_dist(self.h, np.float32(damping)))

    def ba_optimize(self, max_it=25):
        e = np.zeros(2 * max_it + 1, np.float32)
        ne, conv = C.c_int(0), C.c_int(0)
        self._chk(self.L.gsdf_ba_optimize(self.h, int(max_it), _fp(e), C.byref(ne), C.byref(conv)))
        return bool(conv.value), e[:ne.value]

    def merge_prepare(self, nranks):
        self._chk(self.L.gsdf_merge_prepare(self.h, int(nranks)))

    def grow(self, new_capacity_log2):
        self._chk(self.L.gsdf_grow(self.h, int(new_capacity_log2)))

    def capacity_log2(self):
        v = C.c_int(0)
        self._chk(self.L.gsdf_capacity(self.h, C.byref(v)))
        return v.value

    def set_auto_grow(self, max_capacity_log2):
        self._chk(self.L.gsdf_set_auto_grow(self.h, int(max_capacity_log2)))

    def ba_counters(self):
        """(voxels that took part, voxel x keyframe observations) of the last energy sweep read back"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.gsdf_ba_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def ba_poses(self):
        P = np.zeros((self._ba_n, 16), np.float32)
        self._chk(self.L.gsdf_ba_get_poses(self.h, _fp(P)))
        return P.reshape(self._ba_n, 4, 4)

    def merge_raw(self, keys, payload_raw):
        k = np.ascontiguousarray(keys, np.int32).reshape(-1, 3)
        p = _f32(payload_raw).reshape(-1, 5)
        self._chk(self.L.gsdf_merge_raw(self.h, k.ctypes.data_as(C.POINTER(C.c_int32)), _fp(p), k.shape[0]))

    def merge_from(self, other):
        """self += other (another context on the same device): gsdf_merge_from."""
        self._chk(self.L.gsdf_merge_from(self.h, other.h))

    def export_raw_dev(self, keys_ptr, payload_ptr, max_n):
        """Unsorted (key, raw sums) compaction into device buffers (e.g. torch tensors' data_ptr())."""
        n = C.c_int64(0)
        self._chk(self.L.gsdf_export_raw_dev(self.h, C.c_void_p(keys_ptr), C.c_void_p(payload_ptr), int(max_n), C.byref(n)))
        return n.value

    def merge_raw_dev(self, keys_ptr, payload_ptr, n):
        self._chk(self.L.gsdf_merge_raw_dev(self.h, C.c_void_p(keys_ptr), C.c_void_p(payload_ptr), int(n)))

    # -- dense block exchange (device pointers as ints, e.g. torch tensor.data_ptr()) --------------
    def block_keys_dev(self, keys_ptr, max_n):
        n = C.c_int64(0)
        self._chk(self.L.gsdf_block_keys_dev(self.h, C.c_void_p(keys_ptr), int(max_n), C.byref(n)))
        return n.value

    def pack_blocks_dev(self, keys_ptr, n, dense_ptr):
        self._chk(self.L.gsdf_pack_blocks_dev(self.h, C.c_void_p(keys_ptr), int(n), C.c_void_p(dense_ptr)))

    def unpack_blocks_dev(self, keys_ptr, n, dense_ptr):
        self._chk(self.L.gsdf_unpack_blocks_dev(self.h, C.c_void_p(keys_ptr), int(n), C.c_void_p(dense_ptr)))

    # -- the exchange as one C call ---------------------------------------------------------------------
    def merge_allreduce_rccl(self, comm):
        """gsdf_merge_allreduce over an ncclComm_t (c_void_p from rccl_comm_init): (blocks in the union, bytes all-reduced)."""
        nb, by = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.gsdf_merge_allreduce(self.h, comm, C.byref(nb), C.byref(by)))
        return nb.value, by.value

    def merge_allreduce_with(self, allgather, allreduce_sum, nranks):
        """gsdf_merge_allreduce_with: the exchange over caller-provided collectives on host buffers.
        allgather(send: np.uint8[bytes]) -> np.uint8[nranks * bytes];  allreduce_sum(buf: np.float32[n]) -> reduced copy."""
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

        def ag(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                out = np.ascontiguousarray(allgather(src.copy()), np.uint8).reshape(-1)
                assert out.size == nbytes * nranks
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except Exception:            # noqa: BLE001 -- an exception must not cross the C boundary
                return 1

        def ar(_user, buf, n):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float)), shape=(n,))
                out = np.ascontiguousarray(allreduce_sum(a.copy()), np.float32).reshape(-1)
                assert out.size == n
                C.memmove(buf, out.ctypes.data, out.size * 4)
                return 0
            except Exception:            # noqa: BLE001
                return 1

        class Ops(C.Structure):
            _fields_ = [("allgather", AG), ("allreduce_sum_f32", AR), ("user", C.c_void_p), ("nranks", C.c_int)]
        ops = Ops(AG(ag), AR(ar), None, int(nranks))
        nb, by = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.gsdf_merge_allreduce_with(self.h, C.byref(ops), C.byref(nb), C.byref(by)))
        return nb.value, by.value

    def query(self, pts):
        p = _f32(pts).reshape(-1, 3)
        n = p.shape[0]
        dist = np.empty(n, np.float32)
        grad = np.empty((n, 3), np.float32)
        w = np.empty(n, np.float32)
        self._chk(self.L.gsdf_query(self.h, _fp(p), n, _fp(dist), _fp(grad), _fp(w)))
        return dist, grad, w

    def get_voxels(self, keys):
        """tsdf_.at(idx) for int32 keys [n,3]: (payload [n,5] = dist, raw gx gy gz, weight; found [n] bool)."""
        k = np.ascontiguousarray(keys, np.int32).reshape(-1, 3)
        n = k.shape[0]
        pay = np.zeros((n, 5), np.float32)
        found = np.zeros(n, np.int32)
        self._chk(self.L.gsdf_get_voxels(self.h, k.ctypes.data_as(C.POINTER(C.c_int32)), n, _fp(pay),
                                         found.ctypes.data_as(C.POINTER(C.c_int32))))
        return pay, found.astype(bool)

    def raycast(self, R, t, zmin=0.5, zmax=3.5, W=None, H=None, K=None, normals=True):
        """Voxel-hash raycaster: (depth[H,W] camera z, 0 = no hit; normals[3,H,W] camera frame or None)."""
        W = self.W if W is None else int(W)
        H = self.H if H is None else int(H)
        K = self.K if K is None else _f32(K).reshape(9)
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        d = np.empty((H, W), np.float32)
        n = np.empty((3, H, W), np.float32) if normals else None
        self._chk(self.L.gsdf_raycast(self.h, _fp(K), _fp(R), _fp(t), W, H, C.c_float(zmin), C.c_float(zmax), _fp(d),
                                      _fp(n) if normals else None))
        return d, n

    def raycast_dev(self, R, t, depth_dev, normals_dev=None, zmin=0.5, zmax=3.5):
        """gsdf_raycast_dev: enqueue only, output into device buffers (W*H floats, 3*W*H floats or None)."""
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        self._chk(self.L.gsdf_raycast_dev(self.h, _fp(self.K), _fp(R), _fp(t), self.W, self.H, C.c_float(zmin), C.c_float(zmax),
                                          depth_dev, normals_dev))

    def raycast_counters(self, reset=False):
        """(samples the raycasts' definition evaluated, voxel records read) since the last reset."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.gsdf_raycast_counters(self.h, C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def extract_mesh(self, tri_table=None, iso=0.0):
        """GPU marching cubes: float32 [n, 3, 3] triangles in the reference's sweep order
        (tri_table: None = the reference's classic triTable, or int8 [256,16])."""
        tp = None
        if tri_table is not None:
            tt = np.ascontiguousarray(tri_table, dtype=np.int8).reshape(256 * 16)
            tp = tt.ctypes.data_as(C.POINTER(C.c_int8))
        n = C.c_int64(0)
        self._chk(self.L.gsdf_extract_mesh(self.h, C.c_float(iso), tp, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3, 3), np.float32)
        if n.value:
            self._chk(self.L.gsdf_extract_mesh(self.h, C.c_float(iso), tp, _fp(out), n.value, C.byref(n)))
        return out[:n.value]


    # -- timing -------------------------------------------------------------------------------
    def timer_start(self):
        self._chk(self.L.gsdf_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float(0)
        self._chk(self.L.gsdf_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def profile(self, enable):
        self._chk(self.L.gsdf_profile(self.h, int(enable)))

    def profile_read(self):
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        self._chk(self.L.gsdf_profile_read(self.h, ms, n))
        names = ("normals", "fusion", "track_pass")
        return {names[i]: {"ms": ms[i], "launches": n[i]} for i in range(3)}

    def profile_launches(self, slot):
        """durations (ms) of every launch of one slot since profile(1): 0 normals, 1 fusion, 2 tracker launches, 3 raycast"""
        n = C.c_int64(0)
        self._chk(self.L.gsdf_profile_read_launches(self.h, int(slot), None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            self._chk(self.L.gsdf_profile_read_launches(self.h, int(slot), _fp(out), n.value, C.byref(n)))
        return out

    def profile_read_all(self):
        """All slots (gsdf_profile_read_n): normals, fusion, track_pass, raycast, track_opt."""
        names = ("normals", "fusion", "track_pass", "raycast", "track_opt")
        ms = (C.c_double * len(names))()
        n = (C.c_int64 * len(names))()
        self._chk(self.L.gsdf_profile_read_n(self.h, len(names), ms, n))
        return {names[i]: {"ms": ms[i], "launches": n[i]} for i in range(len(names))}
