"""A/B of library variants on the tracked S-tum stream (the bench loop): frames/s, tracker passes, per-launch HIP-event
times of the three kernel groups, and the final pose of every variant against the first one's.
usage: python tools/track_variants.py [--frames N] lib1.so lib2.so ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--size", default="640x480")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
W, H = [int(v) for v in a.size.split("x")]
n = a.frames
seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
d0, R0, t0 = frames[0]
p0 = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
ref = None
for path in a.libs:
    L = pkg.binding.load(os.path.abspath(path))
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22, lib=L)
    dev = [g.upload(f[0]) for f in frames]
    best = 1e9
    for rep in range(3):
        g.reset()
        g.update_dev(dev[0], R0, t0)
        g.set_pose(p0)
        g.sync()
        t_0 = time.perf_counter()
        for i in range(1, n):
            g.track_and_fuse_dev(dev[i])
        g.sync()
        best = min(best, (time.perf_counter() - t_0) / (n - 1) * 1e6)
        log = g.frame_log()
    # event-timed replay (slower dispatch: only the per-kernel numbers are read from it)
    g.reset(); g.update_dev(dev[0], R0, t0); g.set_pose(p0); g.sync()
    g.profile(1)
    for i in range(1, n):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    pr = g.profile_read(); g.profile(0)
    msg = ""
    if ref is None:
        ref = log
    else:
        m = min(len(ref), len(log))
        same = int((ref[:m, 7] == log[:m, 7]).sum())
        msg = "converged flags equal %d/%d, passes equal %d/%d, max |d pose| %.1e" % (
            same, m, int((ref[:m, 8] == log[:m, 8]).sum()), m, float(np.abs(ref[:m, :7] - log[:m, :7]).max()))
    tl = max(pr["track_pass"]["launches"], 1)
    print("%-26s %7.1f us/frame (%.0f fps) conv %d/%d passes %.2f | events: track %.2f us x %d launches, fusion %.1f us, normals %.1f us | %s" % (
        os.path.basename(path), best, 1e6 / best, int(log[:, 7].sum()), len(log), float(log[:, 8].mean()),
        pr["track_pass"]["ms"] / tl * 1e3, tl, pr["fusion"]["ms"] / max(pr["fusion"]["launches"], 1) * 1e3,
        pr["normals"]["ms"] / max(pr["normals"]["launches"], 1) * 1e3, msg), flush=True)
    g.close()
