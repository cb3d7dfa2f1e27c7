"""Runs N GT-pose fusions of the S-tum stream (for rocprofv3 --pmc passes)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22)
dev = [g.upload(f[0]) for f in frames]
for i in range(n):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
g.sync()
print("n_upd/frame", g.stats()["n_upd"] / n, "voxels", g.count())
g.close()
