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


if __name__ == "__main__":
    main()
