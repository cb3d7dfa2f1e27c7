"""Quick GPU probe: per-kernel HIP-event times of the hot path on a 640x480 stream."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
O = G.oracle_module()

kind = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H = 640, 480
seq = pkg.synth.Sequence(kind, W, H, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
t0 = time.time()
frames = [seq.frame(i) for i in range(n)]
print("render %.1fs" % (time.time() - t0))
g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
dev = [g.upload(f[0]) for f in frames]

# fused (GT poses)
g.profile(1)
g.timer_start()
for i in range(n):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
ms = g.timer_stop_ms()
g.sync()
st = g.stats()
print("FUSED: %d frames %.3f ms -> %.1f fps; n_upd/frame %.0f valid/frame %.0f voxels %d" % (n, ms, n / ms * 1e3, st["n_upd"] / n, st["n_valid"] / n, g.count()))
print(g.profile_read())
g.profile(0)
g.timer_start()
g.reset()
print("reset ms", g.timer_stop_ms())
g.timer_start()
for i in range(n):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
ms = g.timer_stop_ms()
print("FUSED (no events): %.3f ms -> %.1f fps" % (ms, n / ms * 1e3))

# fused + tracked
g.reset()
d0, R0, t0_ = frames[0]
p = np.concatenate([t0_, O.R_to_quat(R0)]).astype(np.float32)
g.update_dev(dev[0], O.quat_to_R(p[3:]), t0_)
g.set_pose(p)
g.profile(1)
g.timer_start()
for i in range(1, n):
    g.track_and_fuse_dev(dev[i])
ms = g.timer_stop_ms()
g.sync()
log = g.frame_log()
print("TRACKED: %d frames %.3f ms -> %.1f fps; converged %d, mean passes %.2f" % (n - 1, ms, (n - 1) / ms * 1e3, int(log[:, 7].sum()), log[:, 8].mean()))
print(g.profile_read())
err = [np.abs(log[i - 1, :3] - frames[i][2]).max() for i in range(1, n)]
print("max |t - t_gt| = %.4f" % max(err))
g.profile(0)
g.reset()
g.update_dev(dev[0], O.quat_to_R(p[3:]), t0_)
g.set_pose(p)
g.timer_start()
for i in range(1, n):
    g.track_and_fuse_dev(dev[i])
ms = g.timer_stop_ms()
print("TRACKED (no events): %.3f ms -> %.1f fps" % (ms, (n - 1) / ms * 1e3))
g.close()
