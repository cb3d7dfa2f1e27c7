#!/usr/bin/env python3
"""BASELINE config C5: photometric bundle adjustment over 50 keyframes on a fused Gradient-SDF, 1 GPU -- measured.

Fuses a 640x480 S-tum stream at its ground-truth poses (1 cm voxels, trunc 10, visibility tracking on), picks 50 keyframes,
perturbs their poses, and times
  * gsdf_ba_optimize (PhotometricOptimizer::optimize, ps_optimizer/PhotometricOptimizer.cpp:611-662) end to end,
  * its three sweeps one by one (wall clock around the synchronous C-ABI entries, median of `--reps`),
  * the CPU oracle's optimize() FROM THE SAME STATE (same voxel values, same poses, same keyframes): `--oracle-it` iterations,
    1 core (the reference's PhotometricOptimizer is serial),
and states the sweeps' algorithmic bytes: per sweep every existing voxel record is read once (32 B), every voxel that takes part
reads its vis_ words, and every observation (voxel x keyframe that projects into the image) samples the keyframe bilinearly --
4 taps x 3 float channels = 48 B (energy / mean), + 2 image gradients of 4 pixel differences each = 96 B more in the pose and
distance sweeps; the distance sweep writes 4 B per voxel it moves.  Voxels and observations are counted on the device by the
energy sweep (gsdf_ba_counters).  One JSON line; per-kernel medians come from the rocprofv3 kernel trace of this command
(profiles/rNN_c5_kernel_summary.txt)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--keyframes", type=int, default=50)
    ap.add_argument("--max-it", type=int, default=10)
    ap.add_argument("--oracle-it", type=int, default=1, help="iterations of the CPU oracle's optimize() timed from the same state (0 = skip)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import __graft_entry__ as graft
    pkg = graft.package()
    W, H, n = 640, 480, args.frames
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01); T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    g.enable_vis(n)
    frames = [seq.frame(i) for i in range(n)]
    for f in frames:
        g.update(*f)
    kf = np.linspace(0, n - 1, args.keyframes).astype(np.int32)
    imgs = np.stack([pkg.synth.render_color_bgr(seq, int(i)) for i in kf])
    P = np.stack([pkg.synth.pose16(*seq.pose(int(i))) for i in kf])
    rng = np.random.default_rng(0)
    Pp = P.copy()
    Pp[1:, :3, 3] += (0.004 * rng.standard_normal((len(kf) - 1, 3))).astype(np.float32)
    n_vox = g.count()
    vw = (n + 31) // 32
    keys0, pay0 = g.export(sorted=True)          # the state both sides start from
    kv, vis0 = g.export_vis()

    out = {"config": "C5 PhotoBA (BASELINE.json configs[4])", "frame": "%dx%d" % (W, H), "voxel_size_m": float(vs), "keyframes": int(len(kf)),
           "fused_frames": n, "voxels": n_vox}
    # ---- the sweeps one by one (each entry synchronises) ----
    g.ba_setup(imgs, Pp, kf)
    g.ba_energy()                                 # warm
    t_energy = med(g.ba_energy, args.reps)
    act, obs = g.ba_counters()
    t_pose = med(g.ba_solve_pose, 1)              # (moves the poses: once, then restored by the next setup)
    g.ba_setup(imgs, Pp, kf)
    t_dist = med(g.ba_solve_dist, 1)
    act_d, obs_d = g.ba_counters()                # the distance sweep has no |dist| gate: every voxel with an observation
    b_energy = 32.0 * n_vox + 4.0 * vw * act + 48.0 * obs
    b_pose = b_energy + 96.0 * obs
    b_dist = 32.0 * n_vox + 4.0 * vw * n_vox + 144.0 * obs_d + 4.0 * act_d
    out["sweeps"] = {
        "voxels_taking_part": act, "observations": obs, "observations_per_voxel": round(obs / max(act, 1), 2),
        "energy": {"ms": round(t_energy * 1e3, 3), "algorithmic_bytes": round(b_energy), "achieved_GBs": round(b_energy / t_energy / 1e9, 1),
                   "frac_of_hbm_peak": round(b_energy / t_energy / 1e9 / HBM_PEAK_GBS, 4)},
        "pose": {"ms": round(t_pose * 1e3, 3), "algorithmic_bytes": round(b_pose), "achieved_GBs": round(b_pose / t_pose / 1e9, 1),
                 "frac_of_hbm_peak": round(b_pose / t_pose / 1e9 / HBM_PEAK_GBS, 4), "note": "incl. the 50 6x6 host solves and the pose upload"},
        "dist": {"ms": round(t_dist * 1e3, 3), "algorithmic_bytes": round(b_dist), "achieved_GBs": round(b_dist / t_dist / 1e9, 1),
                 "frac_of_hbm_peak": round(b_dist / t_dist / 1e9 / HBM_PEAK_GBS, 4), "voxels_taking_part": act_d, "observations": obs_d,
                 "note": "all voxels (no |dist| gate, :326-388); its 144 B per observation are served by the caches for the most part (neighbouring voxels "
                         "sample neighbouring pixels of the same 50 images, 184 MB): the figure is an L2-side rate, not HBM traffic"},
    }
    # ---- optimize() end to end, from the perturbed poses on the fused map ----
    g.reset()
    for f in frames:
        g.update(*f)
    g.ba_setup(imgs, Pp, kf)
    g.sync()
    t0 = time.perf_counter(); conv, en = g.ba_optimize(args.max_it); t_opt = time.perf_counter() - t0
    Pn = g.ba_poses()
    steps = int(len(en) - 1)
    out["optimize"] = {"wall_s": round(t_opt, 4), "energy_evaluations": int(len(en)), "iterations": steps // 2, "converged": bool(conv),
                       "ms_per_iteration": round(t_opt * 1e3 / max(steps // 2, 1), 3),
                       "E0": float(en[0]), "E_final": float(en[-1]),
                       "translation_err_before_m": float(np.abs(Pp[:, :3, 3] - P[:, :3, 3]).max()),
                       "translation_err_after_m": float(np.abs(Pn[:, :3, 3] - P[:, :3, 3]).max())}
    # ---- the CPU oracle from the same state ----
    if args.oracle_it > 0:
        O = graft.oracle_module()
        o = O.Oracle(vs, T, W, H, seq.K)
        for f in frames:
            o.update(*f)
        assert o.set_payload(keys0, pay0) == 0 and o.count() == len(keys0)
        ba = O.PhotoBA(o, imgs, Pp, kf)
        t0 = time.perf_counter(); conv_o, en_o = ba.optimize(args.oracle_it); t_o = time.perf_counter() - t0
        it_o = max((len(en_o) - 1) // 2, 1)
        out["cpu_oracle"] = {"kind": "port", "cores": 1, "iterations": int(it_o), "wall_s": round(t_o, 2), "s_per_iteration": round(t_o / it_o, 2),
                             "energies": [float(e) for e in en_o],
                             "gpu_energies_same_steps": [float(e) for e in en[:len(en_o)]],
                             "speedup_per_iteration": round((t_o / it_o) / (t_opt / max(steps // 2, 1)), 1)}
    print(json.dumps(out))
    g.close()


if __name__ == "__main__":
    main()
