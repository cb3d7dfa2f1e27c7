#!/usr/bin/env python3
"""BASELINE config C5: photometric bundle adjustment over 50 keyframes on a fused Gradient-SDF, 1 GPU.
Fuses a 640x480 S-tum stream at its ground-truth poses (visibility tracking on), picks 50 keyframes, perturbs
their poses and runs gsdf_ba_optimize; prints one JSON line with the energy trajectory and timings."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--keyframes", type=int, default=50)
    ap.add_argument("--max-it", type=int, default=10)
    args = ap.parse_args()
    import __graft_entry__ as graft
    pkg = graft.package()
    W, H, n = 640, 480, args.frames
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01); T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    g.enable_vis(n)
    for i in range(n):
        g.update(*seq.frame(i))
    kf = np.linspace(0, n - 1, args.keyframes).astype(np.int32)
    imgs = np.stack([pkg.synth.render_color_bgr(seq, int(i)) for i in kf])
    P = np.stack([pkg.synth.pose16(*seq.pose(int(i))) for i in kf])
    rng = np.random.default_rng(0)
    Pp = P.copy()
    Pp[1:, :3, 3] += (0.004 * rng.standard_normal((len(kf) - 1, 3))).astype(np.float32)
    g.ba_setup(imgs, Pp, kf)
    t0 = time.perf_counter(); E0 = g.ba_energy(); t_energy = time.perf_counter() - t0
    t0 = time.perf_counter(); conv, en = g.ba_optimize(args.max_it); t_opt = time.perf_counter() - t0
    Pn = g.ba_poses()
    print(json.dumps({"config": "C5 PhotoBA", "keyframes": int(len(kf)), "voxels": g.count(), "energy_sweep_ms": round(t_energy * 1e3, 2),
                      "optimize_s": round(t_opt, 3), "steps": int(len(en) - 1), "converged": conv, "E0": float(E0),
                      "E_final": float(en[-1]), "translation_err_before": float(np.abs(Pp[:, :3, 3] - P[:, :3, 3]).max()),
                      "translation_err_after": float(np.abs(Pn[:, :3, 3] - P[:, :3, 3]).max())}))
    g.close()


if __name__ == "__main__":
    main()
