"""Fusion time with 5 mm voxels (4x the voxels per tile): row bands vs LDS overflow into the deferred list."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = 8
W, H = 640, 480
seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
vs = np.float32(float(sys.argv[1]) if len(sys.argv) > 1 else 0.005); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=25, lib=pkg.binding.load_test_lib())   # -DGSDF_EXPERIMENTS build
dev = [g.upload(f[0]) for f in frames]
for flags in (0, 256, 512, 0):
    g.debug_flags(flags)
    g.reset()
    for rep in range(2):
        g.profile(1)
        for i in range(n):
            g.update_dev(dev[i], frames[i][1], frames[i][2])
        g.sync()
        pr = g.profile_read()
        g.profile(0)
        print("vs=%.4f flags=%d rep%d fusion %.1f us/frame voxels %d" % (vs, flags, rep, pr["fusion"]["ms"] / n * 1e3, g.count()))
g.close()
