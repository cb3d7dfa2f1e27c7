#!/usr/bin/env python3
"""Per-workgroup timeline of one k_fuse launch (test build, debug bit 64: thread 0 of every workgroup stores a time stamp at
each phase boundary; 100 MHz clock).  Prints the phase durations by tile colour, when workgroups start, and how many
workgroups are in which phase over the launch.   usage: fuse_trace.py [frame_index (default 20)] [extra debug flags]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
last = int(sys.argv[1]) if len(sys.argv) > 1 else 20
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = last + 1
scene = sys.argv[3] if len(sys.argv) > 3 else "tum"          # "spheres": the object-scan scene of C1 / C4 (most tiles background)
seq = pkg.synth.Sequence(scene, 640, 480, n_frames=n, seed=0) if scene == "tum" else pkg.synth.Sequence(scene, 640, 480, n_frames=n, seed=0, step_deg=360.0 * 4 / 2000)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
L = pkg.binding.load_test_lib()
g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22, lib=L)
dev = [g.upload(f[0]) for f in frames]
for i in range(n - 1):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
g.sync()
g.debug_flags(64 | extra)
g.update_dev(dev[n - 1], frames[n - 1][1], frames[n - 1][2])
g.sync()
NW = 1200
buf = (ctypes.c_ulonglong * (NW * 16))()
assert L.gsdf_debug_trace(g.h, buf, NW) == 0
t = np.array(list(buf), dtype=np.int64).reshape(NW, 16)
t0 = t[:, 0].min()
ts = (t[:, :11] - t0) / 100.0          # us
tile = t[:, 15]; tx = tile & 0x7FFF; ty = (tile >> 16) & 0x7FFF
colour = (tx & 1) + 2 * (ty & 1)
xcc = t[:, 14] >> 32
hw = t[:, 14] & 0xFFFFFFFF
npass = t[:, 13]
single = npass == 1
print("launch span %.1f us; workgroups %d, single band %d; per XCC: %s" % (ts[:, 10].max(), NW, single.sum(), np.bincount(xcc.astype(int), minlength=8).tolist()))
empty = npass == 0
if empty.any():
    life = ts[empty, 10] - ts[empty, 0]
    print("tiles without a band: %d; their lives %.2f us median / %.2f p90 / %.2f max; slot time %.0f us of %.0f us (all workgroups); the last of them starts at %.1f us" % (
        empty.sum(), np.median(life), np.percentile(life, 90), life.max(), life.sum(), (ts[:, 10] - ts[:, 0]).sum(), ts[empty, 0].max()))
    band = ~empty
    print("tiles with a band: %d; lives %.1f us median / %.1f p90 / %.1f max; first starts %.1f, last starts %.1f, last ends %.1f us" % (
        band.sum(), np.median(ts[band, 10] - ts[band, 0]), np.percentile(ts[band, 10] - ts[band, 0], 90), (ts[band, 10] - ts[band, 0]).max(),
        ts[band, 0].min(), ts[band, 0].max(), ts[band, 10].max()))
names = ["prologue", "walk", "keys", "wait", "lookup(w0)", "lookup(all)", "rec loads", "stores", "drain+flag", "tail"]
d = np.diff(ts[:, :11], axis=1)
print("phase durations, us (mean | median | p90) over single-band workgroups")
for c in list(range(4)) + [None]:
    sel = single & ((colour == c) if c is not None else True)
    print(" colour %s (%d tiles): total %.1f" % (c if c is not None else "all", sel.sum(), (ts[sel, 10] - ts[sel, 0]).mean()))
    for k, nm in enumerate(names):
        x = d[sel, k]
        print("    %-12s %6.2f | %6.2f | %6.2f" % (nm, x.mean(), np.median(x), np.percentile(x, 90)))
pl = (t[:, 11] - t[:, 0]) / 100.0; ps = (t[:, 12] - t[:, 0]) / 100.0
print("prologue detail (single band, mean us): pixel loads arrived %.2f, statistics reduced %.2f, decisions made %.2f" % (pl[single].mean(), ps[single].mean(), d[single, 0].mean()))
# start times: the dispatch rounds
st = np.sort(ts[:, 0])
print("start times us: wg 0 %.1f, 256 %.1f, 511 %.1f, 512 %.1f, 600 %.1f, 800 %.1f, 1000 %.1f, 1199 %.1f" % tuple(st[[0, 256, 511, 512, 600, 800, 1000, 1199]]))
# occupancy of phases over time
edges = np.arange(0, ts[:, 10].max() + 2, 2.0)
print("time   running  prologue walk  flush(wait) flush(other)")
for a in edges[:-1]:
    m = a + 1.0
    run = (ts[:, 0] <= m) & (ts[:, 10] > m)
    pro = (ts[:, 0] <= m) & (ts[:, 1] > m)
    wk = (ts[:, 1] <= m) & (ts[:, 2] > m)
    wt = (ts[:, 3] <= m) & (ts[:, 4] > m)
    fl = (ts[:, 2] <= m) & (ts[:, 10] > m) & ~wt
    print("%5.0f  %6d  %6d  %5d  %6d  %6d" % (a, run.sum(), pro.sum(), wk.sum(), wt.sum(), fl.sum()))
# per CU (XCC, SE, SH, CU of HW_ID): how many workgroups it ran, when its last one ended; slot time used
cu = (xcc << 16) | (hw & 0xFF00)
ids = np.unique(cu)
cnt = np.array([(cu == i).sum() for i in ids]); end = np.array([ts[cu == i, 10].max() for i in ids]); busy = np.array([(ts[cu == i, 10] - ts[cu == i, 0]).sum() for i in ids])
span = ts[:, 10].max()
print("CUs seen: %d; workgroups per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
print("end of a CU's last workgroup, us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
print("slot time used: %.0f of %.0f wg-us (2 slots x %d CUs x %.1f us) = %.2f; mean workgroup life %.1f us" % (busy.sum(), 2 * len(ids) * span, len(ids), span, busy.sum() / (2 * len(ids) * span), (ts[:, 10] - ts[:, 0]).mean()))
for k in sorted(set(cnt)):
    print("   CUs with %d workgroups: %d, their last end %.1f us (mean)" % (k, (cnt == k).sum(), end[cnt == k].mean()))
life = ts[:, 10] - ts[:, 0]
order = np.argsort(ts[:, 0])
for a, b in ((0, 512), (512, 1024), (1024, NW)):
    sel = order[a:b]
    print("workgroups %4d..%4d by start: start %.1f..%.1f, mean life %.1f, walk %.1f, end %.1f..%.1f" % (a, b - 1, ts[sel, 0].min(), ts[sel, 0].max(), life[sel].mean(), d[sel, 1].mean(), ts[sel, 10].min(), ts[sel, 10].max()))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "fuse_trace.npy"), t)
g.close()
