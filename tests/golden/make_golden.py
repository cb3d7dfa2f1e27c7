#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot be built or run
here and ships no fixtures -- SURVEY.md F2/F3 -- so these vectors pin the oracle against ITSELF
across rounds; the analytic anchors live in tests/test_oracle_known_answers.py).

    python tests/golden/make_golden.py        # rewrites the fixtures

Fixture = data only: inputs (depth u16, unit, K, poses) and the oracle's outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

import hashlib


def digest(a):
    """SHA-256 of an array's bytes (C order)."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# Full-size case ("one 640x480 frame"): the bench configuration itself -- S-tum 640x480, 1 cm voxels, trunc 10.
# Its map has ~10^6 voxels, so the fixture keeps the inputs, the counters, DIGESTS of the key set / hit mask, column sums and
# a fixed sample of voxels and pixels instead of the whole arrays.
LARGE = {
    "tum_640x480": ("tum", 640, 480, 0.01, 10, 3, 0, 0.5),
}
SAMPLE = 4096
SAMPLINGS = (2, 3, 4)

CASES = {
    # name: (kind, W, H, voxel size, trunc voxels, frames, seed, step_deg)
    "spheres_64x48": ("spheres", 64, 48, 0.04, 5, 3, 11, 0.5),
    "spheres_160x120": ("spheres", 160, 120, 0.02, 5, 3, 7, 0.5),
    "tum_128x96": ("tum", 128, 96, 0.04, 5, 3, 0, 0.5),
}


def main():
    pkg = graft.package()
    O = graft.oracle_module()
    for name, (kind, W, H, vs, trunc, n, seed, step) in CASES.items():
        seq = pkg.synth.Sequence(kind, W, H, n_frames=n, seed=seed, step_deg=step)
        vs = np.float32(vs)
        T = np.float32(trunc) * vs
        o = O.Oracle(vs, T, W, H, seq.K)
        d16 = np.stack([seq.depth_u16(i) for i in range(n)])
        Rs = np.stack([seq.pose(i)[0] for i in range(n)])
        ts = np.stack([seq.pose(i)[1] for i in range(n)])
        depth = d16.astype(np.float32) * np.float32(seq.unit)
        rng = np.random.default_rng(seed)
        probes = np.stack([rng.integers(0, H, 32), rng.integers(0, W, 32)], 1)
        nrm = o.normals(depth[0])[:, probes[:, 0], probes[:, 1]].T.copy()
        counts = []
        for i in range(n - 1):                      # fuse all but the last frame at the GT poses
            counts.append(o.update(depth[i], Rs[i], ts[i]))
        keys, pay = o.export()
        ray_z, ray_n = o.raycast(Rs[n - 2], ts[n - 2])        # self-defined raycaster (absent from the reference)
        p0 = np.concatenate([ts[n - 2], O.R_to_quat(Rs[n - 2])]).astype(np.float32)
        conv, pose, used, trace, hits = o.track(depth[n - 1], p0)
        _, pose1, _, _, _ = o.track(depth[n - 1], p0, iters=1)       # one Gauss-Newton pass: no chaos amplification
        _, pose3, _, _, _ = o.track(depth[n - 1], p0, iters=3)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), kind=kind, W=W, H=H, voxel_size=vs, trunc_dist=T, unit=np.float32(seq.unit),
            K=seq.K, depth_u16=d16, R=Rs, t=ts, probes=probes, normals_at_probes=nrm, counts=np.array(counts, np.int64),
            keys=keys, payload=pay, track_start=p0, track_converged=conv, track_pose=pose, track_passes=used,
            track_trace=trace, track_hits=hits, track_pose_1pass=pose1, track_pose_3pass=pose3,
            raycast_depth=ray_z, raycast_normals=ray_n)
        print(name, "voxels", len(keys), "passes", used, "converged", conv)
    for name, (kind, W, H, vs, trunc, n, seed, step) in LARGE.items():
        seq = pkg.synth.Sequence(kind, W, H, n_frames=n, seed=seed, step_deg=step)
        vs = np.float32(vs)
        T = np.float32(trunc) * vs
        o = O.Oracle(vs, T, W, H, seq.K)
        d16 = np.stack([seq.depth_u16(i) for i in range(n)])
        Rs = np.stack([seq.pose(i)[0] for i in range(n)])
        ts = np.stack([seq.pose(i)[1] for i in range(n)])
        depth = d16.astype(np.float32) * np.float32(seq.unit)
        rng = np.random.default_rng(seed)
        probes = np.stack([rng.integers(0, H, 32), rng.integers(0, W, 32)], 1)
        nrm = o.normals(depth[0])[:, probes[:, 0], probes[:, 1]].T.copy()
        counts = [o.update(depth[i], Rs[i], ts[i]) for i in range(n - 1)]
        keys, pay = o.export()                                  # (z, y, x) order
        vsel = np.sort(rng.choice(len(keys), SAMPLE, replace=False))
        ray_z, ray_n = o.raycast(Rs[n - 2], ts[n - 2])
        psel = np.sort(rng.choice(W * H, SAMPLE, replace=False))
        p0 = np.concatenate([ts[n - 2], O.R_to_quat(Rs[n - 2])]).astype(np.float32)
        conv, pose, used, trace, hits = o.track(depth[n - 1], p0)
        _, pose1, _, _, _ = o.track(depth[n - 1], p0, iters=1)
        _, pose3, _, _, _ = o.track(depth[n - 1], p0, iters=3)
        # optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.h:65: strides 2, 3, 4 (rows: pose7, converged, passes)
        sampled = []
        for s_ in SAMPLINGS:
            c_, p_, u_, _, h_ = o.track(depth[n - 1], p0, sampling=s_)
            sampled.append(np.concatenate([p_, [float(c_), float(u_), float(h_[0])]]))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), large=True, track_samplings=np.array(SAMPLINGS), track_sampled=np.array(sampled, np.float32), kind=kind, W=W, H=H, voxel_size=vs, trunc_dist=T, unit=np.float32(seq.unit),
            K=seq.K, depth_u16=d16, R=Rs, t=ts, probes=probes, normals_at_probes=nrm, counts=np.array(counts, np.int64),
            n_voxels=len(keys), keys_sha256=digest(keys), payload_sha256=digest(pay), payload_colsum=pay.astype(np.float64).sum(axis=0),
            voxel_sample_index=vsel, voxel_sample_keys=keys[vsel], voxel_sample_payload=pay[vsel],
            raycast_hit_sha256=digest(ray_z > 0), raycast_hits=int((ray_z > 0).sum()), raycast_depth_sha256=digest(ray_z),
            pixel_sample_index=psel, raycast_depth_sample=ray_z.reshape(-1)[psel], raycast_normals_sample=ray_n.reshape(3, -1)[:, psel],
            track_start=p0, track_converged=conv, track_pose=pose, track_passes=used, track_trace=trace, track_hits=hits,
            track_pose_1pass=pose1, track_pose_3pass=pose3)
        print(name, "voxels", len(keys), "passes", used, "converged", conv)


if __name__ == "__main__":
    main()
