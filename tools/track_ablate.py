"""Where does a tracker pass spend its time?  (debug switches of k_track_pass; run under rocprofv3 --kernel-trace)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 12
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22, lib=pkg.binding.load_test_lib())   # -DGSDF_EXPERIMENTS build
for i in range(6):
    g.update(frames[i][0], frames[i][1], frames[i][2])
q = pkg.synth.R_to_quat_np(frames[6][1]).astype(np.float32)
pose = np.concatenate([frames[6][2], q]).astype(np.float32)
g.debug_flags(flags << 16)
import time
g.sync()
t0 = time.perf_counter()
for r in range(20):
    conv, p, passes = g.track(frames[6][0], pose, iters=25)
t1 = time.perf_counter()
print("flags", flags, "conv", conv, "passes", passes, "ms per optimize()", (t1 - t0) / 20 * 1e3, "us per pass", (t1 - t0) / 20 / max(passes, 1) * 1e6)
g.close()
