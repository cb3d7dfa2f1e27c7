"""Convergence behaviour of the tracked S-tum stream: GPU engine vs CPU oracle, frame by frame."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as G
pkg = G.package(); O = G.oracle_module()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = 640, 480
seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
o = O.Oracle(vs, T, W, H, seq.K, threads=1)
d0, R0, t0 = frames[0]
p0 = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
R0q = O.quat_to_R(p0[3:])
g.update(d0, R0q, t0); o.update(d0, R0q, t0)
g.set_pose(p0)
dev = [g.upload(f[0]) for f in frames]
for i in range(1, n): g.track_and_fuse_dev(dev[i])
g.sync()
log = g.frame_log()
po = p0.copy()
t0w = time.time()
same = 0
for i in range(1, n):
    co, po, used, trace, hits = o.track(frames[i][0], po)
    if co: o.update(frames[i][0], O.quat_to_R(po[3:]), po[:3])
    gt = frames[i][2]
    print("%3d oracle conv %d passes %2d |xi|^2 last %.2e err %.4f | gpu conv %d passes %2d err %.4f | dpose %.1e" % (
        i, co, used, trace[-1, 35], np.abs(po[:3] - gt).max(), int(log[i - 1, 7]), int(log[i - 1, 8]), np.abs(log[i - 1, :3] - gt).max(),
        np.abs(log[i - 1, :7] - po).max()))
    same += int(bool(log[i - 1, 7]) == co)
print("agree on converged: %d / %d; oracle time %.1f s" % (same, n - 1, time.time() - t0w))
