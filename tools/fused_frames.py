#!/usr/bin/env python3
"""GT-pose fusion of the bench stream frame by frame, each with a sync: device time, updates, deferred entries, timeouts per frame
(which frames of the stream are expensive, and why)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 221
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01)
g = pkg.GradSdf(vs, np.float32(10) * vs, 640, 480, seq.K, capacity_log2=22)
frames = [seq.frame(i) for i in range(n)]
dev = [g.upload(f[0]) for f in frames]
prev = g.stats()
rows = []
for i in range(n):
    g.sync(); t0 = time.perf_counter()
    g.update_dev(dev[i], frames[i][1], frames[i][2])
    g.sync(); dt = (time.perf_counter() - t0) * 1e6
    s = g.stats()
    rows.append((i, dt, s["n_upd"] - prev["n_upd"], s["n_deferred"] - prev["n_deferred"], s["fuse_timeouts"] - prev["fuse_timeouts"], g.count()))
    prev = s
for r in rows:
    if r[0] % 10 == 0 or r[1] > 150 or r[3] > 1000:
        print("frame %3d  %7.1f us  n_upd %8d  deferred %8d  timeouts %3d  voxels %8d" % r)
a = np.array(rows, dtype=np.float64)
print("median us %.1f  mean %.1f  frames > 150 us: %d" % (np.median(a[:, 1]), a[:, 1].mean(), (a[:, 1] > 150).sum()))
