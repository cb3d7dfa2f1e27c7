"""Where does a tracker pass spend its time?  Event-timed k_track_pass launches of the tracked S-tum stream with the
measurement switches of the test build (tracker debug bits: 1 = no 6x6 solve / pose update, 2 = head only, no gather).
With a switch on the poses are wrong, so every frame runs all 25 passes: compare the per-launch times, not frames/s."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = 16
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
d0, R0, t0 = frames[0]
p0 = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
for flags, name in ((0, "full pass"), (1, "no solve / pose update"), (2, "head only (no gather, no sums)"), (3, "head without solve")):
    g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22, lib=pkg.binding.load_test_lib())   # -DGSDF_EXPERIMENTS build
    dev = [g.upload(f[0]) for f in frames]
    for i in range(6):
        g.update_dev(dev[i], frames[i][1], frames[i][2])
    g.set_pose(np.concatenate([frames[6][2], pkg.synth.R_to_quat_np(frames[6][1])]).astype(np.float32))
    g.debug_flags(flags << 16)
    g.sync()
    g.profile(1)
    for i in range(6, n):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    pr = g.profile_read()
    g.profile(0)
    print("%-34s k_track_pass %.2f us per launch (%d launches)" % (name, pr["track_pass"]["ms"] / max(pr["track_pass"]["launches"], 1) * 1e3,
                                                                     pr["track_pass"]["launches"]), flush=True)
    g.close()
