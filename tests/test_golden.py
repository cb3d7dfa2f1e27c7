"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
 - not gpu: the oracle still reproduces them bit for bit (guards the checker against drift);
 - gpu:     the HIP path, through the C-ABI, matches them (keys bit-exact, 1e-4 on floats)."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
TOL = 1e-4


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
    assert len(FILES) >= 3


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
