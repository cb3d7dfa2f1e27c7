#!/usr/bin/env python3
"""Per-workgroup timeline of the one-launch optimize() (k_track_all, GSDF_PERSIST=1; test build, tracker debug bit 64):
per pass the time of the gather, of the row store, of the wait for everybody's rows (and the poll rounds it took) and of
the reduce + solve.  usage: GSDF_PERSIST=1 track_all_trace.py [frame (default 20)]"""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("GSDF_PERSIST", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
last = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = last + 1
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
L = pkg.binding.load_test_lib()
g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22, lib=L)
dev = [g.upload(f[0]) for f in frames]
d0, R0, t0 = frames[0]
g.update_dev(dev[0], R0, t0)
g.set_pose(np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32))
for i in range(1, n - 1):
    g.track_and_fuse_dev(dev[i])
g.sync()
g.debug_flags(64 | (64 << 16))
g.track_and_fuse_dev(dev[n - 1])
g.sync()
NW = 8192
buf = (ctypes.c_ulonglong * (NW * 16))()
assert L.gsdf_debug_trace(g.h, buf, NW) == 0
t = np.array(list(buf), dtype=np.int64).reshape(NW, 16)
log = g.frame_log()
print("frame %d: converged %d after %d passes" % (last, int(log[-1][7]), int(log[-1][8])))
t0 = None
for p in range(12):
    rows = t[2048 + p * 512:2048 + p * 512 + 256]
    if rows[:, 0].max() == 0:
        continue
    if t0 is None:
        t0 = rows[:, 0].min()
    us = lambda a: a / 100.0
    print("pass %d: starts %.2f..%.2f us | gather %.2f med %.2f max | reduce+store %.2f | wait for rows %.2f med %.2f max (%.1f rounds med, %d max) | "
          "sum+solve %.2f | pass ends %.2f..%.2f" % (
              p, us(rows[:, 0].min() - t0), us(rows[:, 0].max() - t0), np.median(us(rows[:, 1] - rows[:, 0])), us(rows[:, 1] - rows[:, 0]).max(),
              np.median(us(rows[:, 2] - rows[:, 1])), np.median(us(rows[:, 3] - rows[:, 2])), us(rows[:, 3] - rows[:, 2]).max(),
              np.median(rows[:, 6]), rows[:, 6].max(), np.median(us(rows[:, 4] - rows[:, 3])), us(rows[:, 4].min() - t0), us(rows[:, 4].max() - t0)))
g.close()
