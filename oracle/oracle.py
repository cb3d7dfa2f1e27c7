"""ctypes wrapper around the CPU ORACLE (oracle/libgsdf_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/gsdf_oracle.h.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / the timed CPU baseline.  PARITY UNPINNED by the
reference (it ships no tests or fixtures); pinned by analytic known-answer tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsdf_oracle.so")


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "gsdf_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgsdf_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.gsdfo_create.restype = C.c_void_p
        L.gsdfo_create.argtypes = [C.c_float, C.c_float]
        L.gsdfo_destroy.argtypes = [C.c_void_p]
        L.gsdfo_set_zrange.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.gsdfo_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.gsdfo_set_box_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.gsdfo_set_nsq_order.argtypes = [C.c_void_p, C.c_int]
        L.gsdfo_llt_solve6.argtypes = [fp, fp, fp]
        L.gsdfo_normals_init.restype = C.c_int
        L.gsdfo_normals_init.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, C.c_int]
        L.gsdfo_normals_cache.argtypes = [C.c_void_p, fp]
        L.gsdfo_normals_compute.argtypes = [C.c_void_p, fp, fp, fp, fp]
        L.gsdfo_update.restype = C.c_int64
        L.gsdfo_update.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.POINTER(C.c_int64)]
        L.gsdfo_count.restype = C.c_int64
        L.gsdfo_count.argtypes = [C.c_void_p]
        L.gsdfo_frame_counter.restype = C.c_int64
        L.gsdfo_frame_counter.argtypes = [C.c_void_p]
        L.gsdfo_export.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp]
        L.gsdfo_export_vis.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        L.gsdfo_query.argtypes = [C.c_void_p, fp, C.c_int64, fp, fp, fp]
        L.gsdfo_set_map.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp, C.c_int64]
        L.gsdfo_set_payload.restype = C.c_int64
        L.gsdfo_set_payload.argtypes = [C.c_void_p, C.POINTER(C.c_int32), fp, C.c_int64]
        L.gsdfo_extract_pc.restype = C.c_int64
        L.gsdfo_extract_pc.argtypes = [C.c_void_p, fp]
        L.gsdfo_extract_mesh.restype = C.c_int64
        L.gsdfo_extract_mesh.argtypes = [C.c_void_p, C.c_float, fp, C.c_int64]
        L.gsdfo_raycast.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]
        L.gsdfo_track.restype = C.c_int
        L.gsdfo_track.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), fp, C.POINTER(C.c_int64)]
        L.gsdfo_quat_to_R.argtypes = [fp, fp]
        L.gsdfo_R_to_quat.argtypes = [fp, fp]
        L.gsdfo_se3_exp_mul.argtypes = [fp, fp]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    """CPU restatement of MapGradPixelSdf + RigidPointOptimizer + NormalEstimator."""

    def __init__(self, voxel_size, trunc_dist, W, H, K, win=11, zmin=0.5, zmax=3.5, threads=4, box_mode=(1, 0), nsq_order=2):
        """box_mode = (cached planes, per-frame filters): 1 = OpenCV's running box sums, 0 = a fresh sum per output.
        (1, 0) is the definition; the other settings exist to measure what the summation order changes."""
        self.L = lib()
        self.h = self.L.gsdfo_create(np.float32(voxel_size), np.float32(trunc_dist))
        self.L.gsdfo_set_box_mode(self.h, int(box_mode[0]), int(box_mode[1]))
        self.L.gsdfo_set_nsq_order(self.h, int(nsq_order))      # 2 = the definition (OpenCV 4 addWeighted, SIMD loop); 0 / 1: measurement
        self.W, self.H = int(W), int(H)
        self.K = _f32(K).reshape(9)
        self.L.gsdfo_set_zrange(self.h, np.float32(zmin), np.float32(zmax))
        self.L.gsdfo_set_threads(self.h, int(threads))
        rc = self.L.gsdfo_normals_init(self.h, self.W, self.H, _fp(self.K), int(win))
        if rc != 0:
            raise ValueError("gsdfo_normals_init failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gsdfo_destroy(self.h)
            self.h = None

    def set_threads(self, threads):
        """Threads of the OMP-structured variants (update(omp=True), track(omp=True))."""
        self.L.gsdfo_set_threads(self.h, int(threads))

    def normals_cache(self):
        out = np.empty((11, self.H, self.W), np.float32)
        self.L.gsdfo_normals_cache(self.h, _fp(out))
        return out

    def normals(self, depth):
        d = _f32(depth).reshape(self.H, self.W)
        n = np.empty((3, self.H, self.W), np.float32)
        self.L.gsdfo_normals_compute(self.h, _fp(d), _fp(n[0]), _fp(n[1]), _fp(n[2]))
        return n

    def update(self, depth, R, t, omp=False):
        d = _f32(depth).reshape(self.H, self.W)
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        nv = C.c_int64(0)
        n_upd = self.L.gsdfo_update(self.h, _fp(d), _fp(R), _fp(t), int(bool(omp)), C.byref(nv))
        return int(n_upd), int(nv.value)

    def count(self):
        return int(self.L.gsdfo_count(self.h))

    def frame_counter(self):
        return int(self.L.gsdfo_frame_counter(self.h))

    def export(self):
        n = self.count()
        keys = np.empty((n, 3), np.int32)
        pay = np.empty((n, 5), np.float32)
        if n:
            self.L.gsdfo_export(self.h, keys.ctypes.data_as(C.POINTER(C.c_int32)), _fp(pay))
        return keys, pay

    def export_vis(self, words_per_voxel):
        n = self.count()
        out = np.zeros((n, words_per_voxel), np.uint32)
        if n:
            self.L.gsdfo_export_vis(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), int(words_per_voxel))
        return out

    def query(self, pts):
        p = _f32(pts).reshape(-1, 3)
        n = p.shape[0]
        dist = np.empty(n, np.float32)
        grad = np.empty((n, 3), np.float32)
        w = np.empty(n, np.float32)
        self.L.gsdfo_query(self.h, _fp(p), n, _fp(dist), _fp(grad), _fp(w))
        return dist, grad, w

    def raycast(self, R, t, zmin=0.5, zmax=3.5, W=None, H=None, K=None):
        """Self-defined voxel-hash raycaster (absent from the reference): (depth[H,W], normals[3,H,W])."""
        W = self.W if W is None else int(W)
        H = self.H if H is None else int(H)
        K = self.K if K is None else _f32(K).reshape(9)
        R = _f32(R).reshape(9)
        t = _f32(t).reshape(3)
        d = np.zeros((H, W), np.float32)
        n = np.zeros((3, H, W), np.float32)
        self.L.gsdfo_raycast(self.h, _fp(K), _fp(R), _fp(t), W, H, np.float32(zmin), np.float32(zmax), _fp(d), _fp(n))
        return d, n

    def set_map(self, keys, payload):
        """Replace the map by (keys int32[n,3], payload float32[n,5] = dist,gx,gy,gz,weight) -- test plumbing."""
        k = np.ascontiguousarray(keys, np.int32).reshape(-1, 3)
        p = _f32(payload).reshape(-1, 5)
        self.L.gsdfo_set_map(self.h, k.ctypes.data_as(C.POINTER(C.c_int32)), _fp(p), k.shape[0])

    def set_payload(self, keys, payload):
        """Overwrite the payload of existing voxels (vis_ and key set stay); returns the number of unknown keys."""
        k = np.ascontiguousarray(keys, np.int32).reshape(-1, 3)
        p = _f32(payload).reshape(-1, 5)
        return int(self.L.gsdfo_set_payload(self.h, k.ctypes.data_as(C.POINTER(C.c_int32)), _fp(p), k.shape[0]))

    def extract_pc(self):
        """MapGradPixelSdf::extract_pc rows (x y z nx ny nz), voxels in (z,y,x) order."""
        n = int(self.L.gsdfo_extract_pc(self.h, None))
        rows = np.zeros((n, 6), np.float32)
        if n:
            self.L.gsdfo_extract_pc(self.h, _fp(rows))
        return rows

    def extract_mesh(self, iso=0.0):
        """LayeredMarchingCubesNoColor::computeIsoSurface: faces [n,3,3] in the reference's sweep order."""
        n = int(self.L.gsdfo_extract_mesh(self.h, np.float32(iso), None, 0))
        tris = np.zeros((n, 3, 3), np.float32)
        if n:
            self.L.gsdfo_extract_mesh(self.h, np.float32(iso), _fp(tris), n)
        return tris

    def track(self, depth, pose7, iters=25, conv=1e-3, damping=1.0, omp=False, sampling=1):
        """optimize_sampled(depth, K, sampling).  Returns (converged, pose7, iters_used, trace[iters_used,36], hits[iters_used])."""
        d = _f32(depth).reshape(self.H, self.W)
        p = _f32(pose7).reshape(7).copy()
        used = C.c_int(0)
        trace = np.zeros((iters, 36), np.float32)
        hits = np.zeros(iters, np.int64)
        conv_flag = self.L.gsdfo_track(self.h, _fp(d), _fp(self.K), _fp(p), int(iters), np.float32(conv),
                                       np.float32(damping), int(sampling), int(bool(omp)), C.byref(used), _fp(trace),
                                       hits.ctypes.data_as(C.POINTER(C.c_int64)))
        u = used.value
        return bool(conv_flag), p, u, trace[:u], hits[:u]


def quat_to_R(q_xyzw):
    q = _f32(q_xyzw).reshape(4)
    R = np.empty(9, np.float32)
    lib().gsdfo_quat_to_R(_fp(q), _fp(R))
    return R.reshape(3, 3)


def R_to_quat(R):
    R = _f32(R).reshape(9)
    q = np.empty(4, np.float32)
    lib().gsdfo_R_to_quat(_fp(R), _fp(q))
    return q


def llt_solve6(H, g):
    """H.llt().solve(g) in the oracle's statement of Eigen's operation order (float32)."""
    H = _f32(H).reshape(36)
    g = _f32(g).reshape(6)
    x = np.empty(6, np.float32)
    lib().gsdfo_llt_solve6(_fp(H), _fp(g), _fp(x))
    return x


def se3_exp_mul(xi, pose7):
    xi = _f32(xi).reshape(6)
    p = _f32(pose7).reshape(7).copy()
    lib().gsdfo_se3_exp_mul(_fp(xi), _fp(p))
    return p


class PhotoBA:
    """PhotometricOptimizer on an Oracle's map (ps_optimizer/PhotometricOptimizer.cpp restated)."""

    def __init__(self, oracle, images_bgr, poses16, frame_idx, reg_weight=10.0):
        self.L = lib()
        fp = C.POINTER(C.c_float)
        self.L.gsdfo_ba_create.restype = C.c_void_p
        self.L.gsdfo_ba_create.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_int, fp, fp, C.POINTER(C.c_int), C.c_float]
        self.L.gsdfo_ba_destroy.argtypes = [C.c_void_p]
        self.L.gsdfo_ba_energy.restype = C.c_float
        self.L.gsdfo_ba_energy.argtypes = [C.c_void_p]
        self.L.gsdfo_ba_energy_f64.restype = C.c_double
        self.L.gsdfo_ba_energy_f64.argtypes = [C.c_void_p]
        self.L.gsdfo_ba_solve_pose.argtypes = [C.c_void_p, C.c_float]
        self.L.gsdfo_ba_solve_dist.argtypes = [C.c_void_p, C.c_float]
        self.L.gsdfo_ba_optimize.restype = C.c_int
        self.L.gsdfo_ba_optimize.argtypes = [C.c_void_p, C.c_int, fp, C.POINTER(C.c_int)]
        self.L.gsdfo_ba_get_poses.argtypes = [C.c_void_p, fp]
        self.oracle = oracle
        img = _f32(images_bgr)
        self.n = img.shape[0]
        P = _f32(poses16).reshape(self.n, 16)
        idx = np.ascontiguousarray(frame_idx, dtype=np.int32)
        self.h = self.L.gsdfo_ba_create(oracle.h, _fp(oracle.K), self.n, oracle.W, oracle.H, _fp(img), _fp(P),
                                        idx.ctypes.data_as(C.POINTER(C.c_int)), np.float32(reg_weight))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gsdfo_ba_destroy(self.h)
            self.h = None

    def set_loss(self, loss, lam=0.5):
        """OptSettings::loss / lambda; 4 = TRUNC_L2 (the only value the reference's code distinguishes)."""
        self.L.gsdfo_ba_set_loss.restype = None
        self.L.gsdfo_ba_set_loss.argtypes = [C.c_void_p, C.c_int, C.c_float]
        self.L.gsdfo_ba_set_loss(self.h, int(loss), np.float32(lam))

    def energy(self):
        return float(self.L.gsdfo_ba_energy(self.h))

    def energy_f64(self):
        """The same float terms added in double: free of the reference's summation-order uncertainty."""
        return float(self.L.gsdfo_ba_energy_f64(self.h))

    def solve_pose(self, damping=1.0):
        self.L.gsdfo_ba_solve_pose(self.h, np.float32(damping))

    def solve_dist(self, damping=1.0):
        self.L.gsdfo_ba_solve_dist(self.h, np.float32(damping))

    def optimize(self, max_it=25):
        e = np.zeros(2 * max_it + 1, np.float32)
        ne = C.c_int(0)
        conv = self.L.gsdfo_ba_optimize(self.h, int(max_it), _fp(e), C.byref(ne))
        return bool(conv), e[:ne.value]

    def poses(self):
        P = np.zeros((self.n, 16), np.float32)
        self.L.gsdfo_ba_get_poses(self.h, _fp(P))
        return P.reshape(self.n, 4, 4)
