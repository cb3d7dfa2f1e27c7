"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
 - not gpu: the oracle still reproduces them bit for bit (guards the checker against drift);
 - gpu:     the HIP path, through the C-ABI, matches them (keys bit-exact, 1e-4 on floats)."""
import glob
import os

import numpy as np
import pytest

import hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
# full-size fixtures keep digests + samples instead of whole arrays (tests/golden/make_golden.py: LARGE)
LARGE_FILES = [f for f in ALL if "large" in np.load(f).files]
FILES = [f for f in ALL if f not in LARGE_FILES]
TOL = 1e-4


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load(path):
    z = np.load(path)
    depth = z["depth_u16"].astype(np.float32) * np.float32(z["unit"])
    return z, depth


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(O, path):
    z, depth = _load(path)
    n = depth.shape[0]
    o = O.Oracle(z["voxel_size"], z["trunc_dist"], int(z["W"]), int(z["H"]), z["K"])
    pr = z["probes"]
    nrm = o.normals(depth[0])[:, pr[:, 0], pr[:, 1]].T
    assert np.array_equal(nrm, z["normals_at_probes"], equal_nan=True)
    for i in range(n - 1):
        nu, nv = o.update(depth[i], z["R"][i], z["t"][i])
        assert (nu, nv) == tuple(z["counts"][i])
    keys, pay = o.export()
    assert np.array_equal(keys, z["keys"]) and np.array_equal(pay, z["payload"])
    rz, rn = o.raycast(z["R"][n - 2], z["t"][n - 2])
    assert np.array_equal(rz, z["raycast_depth"]) and np.array_equal(rn, z["raycast_normals"])
    conv, pose, used, trace, hits = o.track(depth[n - 1], z["track_start"])
    assert conv == bool(z["track_converged"]) and used == int(z["track_passes"])
    assert np.array_equal(pose, z["track_pose"]) and np.array_equal(trace, z["track_trace"], equal_nan=True)
    _, pose1, _, _, _ = o.track(depth[n - 1], z["track_start"], iters=1)
    assert np.array_equal(pose1, z["track_pose_1pass"])


def test_golden_files_exist():
    assert len(FILES) >= 3 and len(LARGE_FILES) >= 1


@pytest.mark.parametrize("path", LARGE_FILES, ids=[os.path.basename(f) for f in LARGE_FILES])
def test_oracle_reproduces_full_size_golden(O, path):
    """The 640x480 fixture at the bench configuration (S-tum, 1 cm, trunc 10): the oracle reproduces it bit for bit."""
    z, depth = _load(path)
    n = depth.shape[0]
    W, H = int(z["W"]), int(z["H"])
    o = O.Oracle(z["voxel_size"], z["trunc_dist"], W, H, z["K"])
    pr = z["probes"]
    assert np.array_equal(o.normals(depth[0])[:, pr[:, 0], pr[:, 1]].T, z["normals_at_probes"], equal_nan=True)
    for i in range(n - 1):
        assert o.update(depth[i], z["R"][i], z["t"][i]) == tuple(z["counts"][i])
    keys, pay = o.export()
    assert len(keys) == int(z["n_voxels"]) and digest(keys) == str(z["keys_sha256"]) and digest(pay) == str(z["payload_sha256"])
    sel = z["voxel_sample_index"]
    assert np.array_equal(keys[sel], z["voxel_sample_keys"]) and np.array_equal(pay[sel], z["voxel_sample_payload"])
    rz, rn = o.raycast(z["R"][n - 2], z["t"][n - 2])
    assert digest(rz) == str(z["raycast_depth_sha256"]) and int((rz > 0).sum()) == int(z["raycast_hits"])
    ps = z["pixel_sample_index"]
    assert np.array_equal(rn.reshape(3, -1)[:, ps], z["raycast_normals_sample"])
    conv, pose, used, trace, hits = o.track(depth[n - 1], z["track_start"])
    assert conv == bool(z["track_converged"]) and used == int(z["track_passes"]) and np.array_equal(pose, z["track_pose"])
    assert np.array_equal(trace, z["track_trace"], equal_nan=True)
    for s_, row in zip(z["track_samplings"], z["track_sampled"]):          # optimize_sampled's stride argument
        c_, p_, u_, _, h_ = o.track(depth[n - 1], z["track_start"], sampling=int(s_))
        assert np.array_equal(p_, row[:7]) and c_ == bool(row[7]) and u_ == int(row[8]) and int(h_[0]) == int(row[9])


@pytest.mark.gpu
@pytest.mark.parametrize("path", LARGE_FILES, ids=[os.path.basename(f) for f in LARGE_FILES])
def test_hip_path_matches_full_size_golden(pkg, path):
    """640x480, 1 cm voxels, trunc 10 (BASELINE configs[1]) through the C-ABI against the committed vectors: counters and
    the key set bit-exact (digest of ~9 x 10^5 sorted keys), payload / tracked pose / raycast within 1e-4."""
    z, depth = _load(path)
    n = depth.shape[0]
    W, H = int(z["W"]), int(z["H"])
    g = pkg.GradSdf(z["voxel_size"], z["trunc_dist"], W, H, z["K"], capacity_log2=22)
    pr = z["probes"]
    assert np.array_equal(g.normals(depth[0])[:, pr[:, 0], pr[:, 1]].T, z["normals_at_probes"], equal_nan=True)
    for i in range(n - 1):
        g.update(depth[i], z["R"][i], z["t"][i])
    st = g.stats()
    assert st["n_upd"] == int(z["counts"][:, 0].sum()) and st["n_valid"] == int(z["counts"][:, 1].sum())
    keys, pay = g.export(sorted=True)
    assert len(keys) == int(z["n_voxels"]) and digest(keys) == str(z["keys_sha256"])            # bit-exact occupancy
    colsum = pay.astype(np.float64).sum(axis=0)
    assert np.abs(colsum - z["payload_colsum"]).max() <= 1e-6 * np.abs(z["payload_colsum"]).max() + 1e-3
    sel = z["voxel_sample_index"]
    gp = z["voxel_sample_payload"]
    assert np.array_equal(keys[sel], z["voxel_sample_keys"])
    scale = np.maximum(1.0, gp[:, 4])
    assert np.abs(pay[sel, 0] - gp[:, 0]).max() <= TOL
    assert (np.abs(pay[sel, 1:] - gp[:, 1:]).max(axis=1) / scale).max() <= TOL
    rz, rn = g.raycast(z["R"][n - 2], z["t"][n - 2])
    ps = z["pixel_sample_index"]
    gz = z["raycast_depth_sample"]
    rzs = rz.reshape(-1)[ps]
    assert ((rzs > 0) == (gz > 0)).mean() > 0.995 and abs(int((rz > 0).sum()) - int(z["raycast_hits"])) <= 0.002 * W * H
    both = (rzs > 0) & (gz > 0)
    assert np.percentile(np.abs(rzs - gz)[both], 99.5) <= TOL
    assert np.percentile(np.abs(rn.reshape(3, -1)[:, ps] - z["raycast_normals_sample"])[:, both], 99.5) <= 1e-3
    conv, pose, passes = g.track(depth[n - 1], z["track_start"], iters=1)
    assert passes == 1 and np.abs(pose - z["track_pose_1pass"]).max() <= TOL
    conv, pose, passes = g.track(depth[n - 1], z["track_start"], iters=3)
    assert np.abs(pose - z["track_pose_3pass"]).max() <= 5 * TOL
    conv, pose, passes = g.track(depth[n - 1], z["track_start"])
    assert conv == bool(z["track_converged"]) and passes == int(z["track_passes"])
    assert np.abs(pose - z["track_pose"]).max() <= TOL
    # optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.h:65 / gsdf_track_sampled: pass counts equal, pose <= 1e-4
    for s_, row in zip(z["track_samplings"], z["track_sampled"]):
        conv, pose, passes = g.track(depth[n - 1], z["track_start"], sampling=int(s_))
        assert conv == bool(row[7]) and passes == int(row[8]), (int(s_), conv, passes, row[7:9])
        assert np.abs(pose - row[:7]).max() <= TOL, (int(s_), np.abs(pose - row[:7]).max())
    conv, pose, passes = g.track(depth[n - 1], z["track_start"], sampling=1)        # the same entry at stride 1 == gsdf_track
    assert conv == bool(z["track_converged"]) and passes == int(z["track_passes"]) and np.abs(pose - z["track_pose"]).max() <= TOL
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_matches_golden(pkg, path):
    z, depth = _load(path)
    n = depth.shape[0]
    g = pkg.GradSdf(z["voxel_size"], z["trunc_dist"], int(z["W"]), int(z["H"]), z["K"], capacity_log2=20)
    pr = z["probes"]
    nrm = g.normals(depth[0])[:, pr[:, 0], pr[:, 1]].T
    assert np.array_equal(nrm, z["normals_at_probes"], equal_nan=True)
    for i in range(n - 1):
        g.update(depth[i], z["R"][i], z["t"][i])
    st = g.stats()
    assert st["n_upd"] == int(z["counts"][:, 0].sum()) and st["n_valid"] == int(z["counts"][:, 1].sum())
    keys, pay = g.export(sorted=True)
    assert np.array_equal(keys, z["keys"])                                   # bit-exact occupancy
    gp = z["payload"]
    scale = np.maximum(1.0, gp[:, 4])
    assert np.abs(pay[:, 0] - gp[:, 0]).max() <= TOL
    assert (np.abs(pay[:, 1:] - gp[:, 1:]).max(axis=1) / scale).max() <= TOL
    # the self-defined raycaster: same hits, same depth (sign decisions on sums that differ in the last bits may flip)
    rz, rn = g.raycast(z["R"][n - 2], z["t"][n - 2])
    gz = z["raycast_depth"]
    assert ((rz > 0) == (gz > 0)).mean() > 0.995
    both = (rz > 0) & (gz > 0)
    if both.any():
        assert np.percentile(np.abs(rz - gz)[both], 99.5) <= TOL
        assert np.percentile(np.abs(rn - z["raycast_normals"])[:, both], 99.5) <= 1e-3
    # one and three Gauss-Newton passes: robust to summation order (a full 25-pass run that never
    # converges at this resolution amplifies last-bit differences and is not a parity quantity)
    conv, pose, passes = g.track(depth[n - 1], z["track_start"], iters=1)
    assert passes == 1 and not conv
    assert np.abs(pose - z["track_pose_1pass"]).max() <= TOL
    conv, pose, passes = g.track(depth[n - 1], z["track_start"], iters=3)
    assert np.abs(pose - z["track_pose_3pass"]).max() <= 5 * TOL
    if bool(z["track_converged"]) and int(z["track_passes"]) <= 8:   # slow convergers are borderline by nature
        conv, pose, passes = g.track(depth[n - 1], z["track_start"])
        assert conv and passes == int(z["track_passes"])
        assert np.abs(pose - z["track_pose"]).max() <= TOL
    g.close()
