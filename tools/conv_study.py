#!/usr/bin/env python3
"""Which of the two bench windows is the representative one?  On the S-tum stream a quarter of the frames of the default window
do not converge (25 passes, not fused); DESIGN.md explains it by the reference's nearest-voxel lookup (phi jumps at voxel
borders).  This tool stresses that explanation: the same stream
    base      seed 0, Kinect noise on, the fr1/xyz-like motion (<= 1.4 cm / frame)
    nonoise   the same without the Kinect disparity noise
    seed1..3  other noise seeds
    halfmotion  half the motion between two frames
tracked + fused (optimize(); if (converged) update()) by the GPU engine (--gpu: all `--frames` frames) and by the CPU oracle
(--oracle: the first `--oracle-frames`, one process per variant), and prints per variant the converged fraction and the mean
pass count, for the driver's window (frames 6..25), the default window (21..220) and the oracle's stretch.
Writes gpurun_out/conv_study_<side>.json."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {"base": dict(seed=0), "nonoise": dict(seed=0, noise=False), "seed1": dict(seed=1), "seed2": dict(seed=2), "seed3": dict(seed=3),
            "halfmotion": dict(seed=0, motion=0.5)}
W, H = 640, 480


def run_gpu(name, kw, n):
    import __graft_entry__ as G
    pkg = G.package()
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, **kw)
    vs = np.float32(0.01); T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    fr = [seq.frame(i) for i in range(n)]
    dev = [g.upload(f[0]) for f in fr]
    d0, R0, t0 = fr[0]
    p0 = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
    from bench import quat_to_R
    g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
    g.set_pose(p0)
    for i in range(1, n):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    log = g.frame_log()
    gt = np.array([f[2] for f in fr[1:]])
    g.close()
    return {"converged": [int(v) for v in log[:, 7]], "passes": [int(v) for v in log[:, 8]],
            "max_abs_translation_error_m": float(np.abs(log[:, :3] - gt).max())}


def run_oracle(name, kw, n):
    import __graft_entry__ as G
    pkg = G.package(); O = G.oracle_module()
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, **kw)
    vs = np.float32(0.01); T = np.float32(10) * vs
    o = O.Oracle(vs, T, W, H, seq.K, threads=1)
    d0, R0, t0 = seq.frame(0)
    pose = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
    o.update(d0, O.quat_to_R(pose[3:]), t0)
    conv, passes = [], []
    for i in range(1, n):
        d, R, t = seq.frame(i)
        c, pose, used, _, _ = o.track(d, pose)
        if c:
            o.update(d, O.quat_to_R(pose[3:]), pose[:3])
        conv.append(int(c)); passes.append(int(used))
    return {"converged": conv, "passes": passes}


def summary(r, lo, hi):
    c = np.array(r["converged"][lo - 1:hi]); p = np.array(r["passes"][lo - 1:hi])     # row i-1 = frame i
    return {"frames": "%d..%d" % (lo, lo + len(c) - 1), "converged": int(c.sum()), "of": int(len(c)), "mean_passes": round(float(p.mean()), 2) if len(p) else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--frames", type=int, default=221)
    ap.add_argument("--oracle-frames", type=int, default=65)
    ap.add_argument("--variant", default=None, help="one variant only (the oracle side starts one process per variant)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    names = [args.variant] if args.variant else list(VARIANTS)
    if args.oracle and not args.variant:
        import subprocess
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle", "--variant", v, "--oracle-frames", str(args.oracle_frames),
                                   "--out", args.out]) for v in names]
        rc = [p.wait() for p in procs]
        res = {v: json.load(open(os.path.join(args.out, "conv_study_oracle_%s.json" % v))) for v in names}
        json.dump(res, open(os.path.join(args.out, "conv_study_oracle.json"), "w"))
        for v in names:
            print("%-11s oracle  %s" % (v, summary(res[v], 1, args.oracle_frames - 1)))
        return max(rc)
    res = {}
    for v in names:
        t0 = time.time()
        if args.gpu:
            res[v] = run_gpu(v, VARIANTS[v], args.frames)
            print("%-11s gpu  driver window %s | default window %s | oracle's stretch %s  (%.0f s)" % (
                v, summary(res[v], 6, 25), summary(res[v], 21, 220), summary(res[v], 1, args.oracle_frames - 1), time.time() - t0), flush=True)
        else:
            res[v] = run_oracle(v, VARIANTS[v], args.oracle_frames)
            json.dump(res[v], open(os.path.join(args.out, "conv_study_oracle_%s.json" % v), "w"))
    if args.gpu:
        json.dump(res, open(os.path.join(args.out, "conv_study_gpu.json"), "w"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
