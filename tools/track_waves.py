#!/usr/bin/env python3
"""Per-WAVE gather times of the tracker passes of one frame (test build, tracker debug bit 64): which waves are slow, and
whether that goes with their pixels (valid / hit counts), their place in the image or the block lookups.
usage: track_waves.py [frame (default 20)] [warm]      warm: tracker debug bit 8 -- waves 1-7 run their gather with the old pose under
the head's solve (an L2 warm-up experiment of round 6); the per-wave-index medians show whether the real gather gets shorter"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
last = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
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
g.debug_flags(64 | ((64 | (8 if warm else 0)) << 16))
g.track_and_fuse_dev(dev[n - 1])
g.sync()
NW = 8192
buf = (ctypes.c_ulonglong * (NW * 16))()
assert L.gsdf_debug_trace(g.h, buf, NW) == 0
t = np.array(list(buf), dtype=np.uint64).reshape(NW, 16)
for p in range(1, 4):
    rows = t[2048 + p * 512:2048 + p * 512 + 512]
    rows = rows[rows[:, 0] > 0]                           # the workgroups of the pass (640 x 480: 240)
    NWG = len(rows)
    if NWG == 0 or rows[:, 8:16].max() == 0:
        continue
    w = rows[:, 8:16].reshape(-1)                         # waves: workgroup-major
    tot = (w & np.uint64(0xFFFF)).astype(np.float64) / 100.0
    look = ((w >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0
    ok = ((w >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64)
    hit = ((w >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64)
    slow = tot > np.percentile(tot, 95)
    wg = np.arange(NWG * 8) // 8
    print("pass %d: gather per wave %.2f med / %.2f p95 / %.2f max us | until the lookups are done %.2f med / %.2f p95 | "
          "corr(time, hits) %.2f, corr(time, lookup time) %.2f" % (p, np.median(tot), np.percentile(tot, 95), tot.max(), np.median(look),
          np.percentile(look, 95), np.corrcoef(tot, hit)[0, 1], np.corrcoef(tot, look)[0, 1]))
    print("        slowest 5 %% of the waves: hits %.0f (all waves %.0f of 192), valid %.0f (%.0f), lookup part %.2f us (%.2f), "
          "workgroups they sit in: %d distinct of %d, image thirds of their first pixels: WG index quartiles %s" % (
              hit[slow].mean(), hit.mean(), ok[slow].mean(), ok.mean(), look[slow].mean(), look.mean(), len(set(wg[slow])), NWG,
              np.percentile(wg[slow], [25, 50, 75]).round().tolist()))
    heavy = np.tile(np.arange(8) < 4, NWG)
    print("        waves 0-3 (3 pixels per lane) %.2f med / %.2f p95 / %.2f max, waves 4-7 (2 pixels per lane) %.2f med / %.2f p95 / %.2f max" % (
        np.median(tot[heavy]), np.percentile(tot[heavy], 95), tot[heavy].max(), np.median(tot[~heavy]), np.percentile(tot[~heavy], 95), tot[~heavy].max()))
    by_wave = tot.reshape(NWG, 8)
    print("        median gather by wave index 0..7: %s us; launch: head done -> last gather done (thread 0 stamps) %.2f us median over workgroups" % (
        np.median(by_wave, axis=0).round(2).tolist(), float(np.median((rows[:, 2].astype(np.float64) - rows[:, 1].astype(np.float64)) / 100.0))))
    per_wg_max = tot.reshape(NWG, 8).max(axis=1)
    print("        slowest wave per workgroup: %.2f med / %.2f max; workgroups whose slowest wave is > 1.3x their median wave: %d" % (
        np.median(per_wg_max), per_wg_max.max(), int((per_wg_max > 1.3 * np.median(tot.reshape(NWG, 8), axis=1)).sum())))
g.close()
