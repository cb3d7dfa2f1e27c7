import sys, time, os, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as G
pkg = G.package()
n = 221
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01)
g = pkg.GradSdf(vs, np.float32(10) * vs, 640, 480, seq.K, capacity_log2=22)
frames = [seq.frame(i) for i in range(n)]
dev = [g.upload(f[0]) for f in frames]
for rep in range(2):
    g.reset(); g.sync()
    for c0 in range(21, 221, 40):
        t0 = time.perf_counter()
        for i in range(c0, min(c0 + 40, 221)):
            g.update_dev(dev[i], frames[i][1], frames[i][2])
        g.sync()
        print("rep", rep, "frames", c0, "..", min(c0 + 40, 221) - 1, "us/frame %.1f" % ((time.perf_counter() - t0) / (min(c0 + 40, 221) - c0) * 1e6), "voxels", g.count(), "deferred-ish stats", g.stats()["n_upd"])
