import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as G
pkg = G.package()
n = 221
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01)
g = pkg.GradSdf(vs, np.float32(10) * vs, 640, 480, seq.K, capacity_log2=22)
for i in range(n):
    d, R, t = seq.frame(i)
    g.update_dev(g.upload(d), R, t)
g.sync()
cap = 1 << 16
buf = g.upload(np.zeros(cap, np.int64))
nb = g.block_keys_dev(buf.value, cap)
print("voxels", g.count(), "blocks used", nb, "of", cap, "load %.2f" % (nb / cap), "fill %.2f" % (g.count() / (nb * 64.0)))
