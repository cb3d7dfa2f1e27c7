#!/usr/bin/env python3
"""Per-workgroup timeline of the tracker passes of ONE frame (test build, tracker debug bit 64): when the workgroups of a
launch start, how long the head (reduce + solve of the previous pass) and the gather take, how the launches follow each other.
usage: track_trace.py [frame (default 20)]"""
import ctypes, os, sys
import numpy as np
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
g.debug_flags(64 | (64 << 16))            # k_fuse trace allocates the buffer; tracker bit 64 stamps into it
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
    rows = t[2048 + p * 512:2048 + p * 512 + 512]
    rows = rows[rows[:, 0] > 0]                 # the workgroups of the pass (640 x 480: 240)
    if len(rows) == 0:
        continue
    if t0 is None:
        t0 = rows[:, 0].min()
    st = (rows[:, 0] - t0) / 100.0
    hd = (rows[:, 1] - rows[:, 0]) / 100.0
    has_g = rows[:, 3] > 0
    line = "launch %d: first workgroup starts %.2f us, last starts +%.2f; head %.2f (median)" % (p, st.min(), st.max() - st.min(), np.median(hd[rows[:, 1] > 0]) if (rows[:, 1] > 0).any() else -1)
    if has_g.any():
        ga = (rows[has_g, 2] - rows[has_g, 1]) / 100.0
        en = (rows[has_g, 3] - t0) / 100.0
        line += "; gather %.2f median / %.2f max; reduce+atomics %.2f; last workgroup ends %.2f (launch span %.2f)" % (
            np.median(ga), ga.max(), np.median((rows[has_g, 3] - rows[has_g, 2]) / 100.0), en.max(), en.max() - st.min())
    print(line)
    if (rows[:, 4] > 0).any():                 # head in parts: sums + state arrived | solved | pose handed to the other waves
        m = rows[:, 4] > 0
        ho = rows[m, 1] > 0                      # a head-only launch (optimize() ended in this head) has no hand-over to a gather
        line2 = "          head: sums and state arrive after %.2f, solve %.2f" % (
            np.median((rows[m, 4] - rows[m, 0]) / 100.0), np.median((rows[m, 5] - rows[m, 4]) / 100.0) if (rows[m, 5] > 0).all() else -1)
        line2 += ", hand-over %.2f" % np.median((rows[m, 1][ho] - np.maximum(rows[m, 5], rows[m, 4])[ho]) / 100.0) if ho.any() else " (head-only launch: optimize() ended here)"
        if has_g.any():
            line2 += " | tail: wave 0's sums %.2f, waits %.2f for the slowest wave, workgroup sum + atomics issued %.2f; ends %.2f after the first workgroup's end" % (
                np.median((rows[has_g, 6] - rows[has_g, 2]) / 100.0), np.median((rows[has_g, 7] - rows[has_g, 6]) / 100.0),
                np.median((rows[has_g, 3] - rows[has_g, 7]) / 100.0), (rows[has_g, 3].max() - rows[has_g, 3].min()) / 100.0)
        print(line2)

if os.environ.get("RAW"):
    for p in range(6):
        rows = t[2048 + p * 512:2048 + p * 512 + 256]
        print("pass", p, "wg0", ((rows[0, :4] - t0) / 100.0).round(2).tolist(), "wg1", ((rows[1, :4] - t0) / 100.0).round(2).tolist(), "wg128", ((rows[128, :4] - t0) / 100.0).round(2).tolist(), "wg255", ((rows[255, :4] - t0) / 100.0).round(2).tolist())
g.close()
