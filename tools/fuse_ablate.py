"""Ablation of the fusion kernel (debug switches in k_fuse): where does the time go?"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
kind = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 640, 480
seq = pkg.synth.Sequence(kind, W, H, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
L = pkg.binding.load_test_lib()          # the -DGSDF_EXPERIMENTS build (libgsdf_test.so)
g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22, lib=L)
import ctypes as _ct
L.gsdf_version.restype = _ct.c_char_p
dev = [g.upload(f[0]) for f in frames]
names = {0: "full", 1: "no flush", 2: "no LDS accumulate (flush empty)", 3: "compute only", 4: "flush: probe only",
         8: "flush: plain RMW", 17: "no flush, no overflow-to-HBM", 65: "no flush, count overflows", 16: "no overflow-to-HBM",
         49: "no flush, no overflow, LDS w-add skipped"}
names = {0: 'full', 1: 'no flush', 2: 'no LDS lookups/adds (flush empty)', 3: 'compute only', 17: 'no flush, lookups only (no adds)', 33: 'no flush, adds only (slot from hash)', 8: 'flush without waiting (WRONG results)', 64: 'full, no skew', 65: 'no flush, no skew', 81: 'no flush, lookups only, no skew'}
names[256] = 'full, single band forced'; names[512] = 'full, 4 bands forced'
names[1024] = 'band limit 0.9'; names[2048] = 'band limit 1.1'
names[4096] = 'full, 1 workgroup per CU'; names[4097] = 'no flush, 1 workgroup per CU'; names[4099] = 'compute only, 1 workgroup per CU'
for flags in (0, 1, 17, 33, 3, 4096, 0):
    g.debug_flags(flags)
    g.reset()
    for rep in range(2):
        g.profile(1)
        for i in range(n):
            g.update_dev(dev[i], frames[i][1], frames[i][2])
        try:
            g.sync()
        except Exception as e:
            print("  status:", e)
        pr = g.profile_read()
        g.profile(0)
        print("flags=%d %-34s rep%d fusion %.1f us/frame  normals %.1f us" % (flags, names[flags], rep,
              pr["fusion"]["ms"] / n * 1e3, pr["normals"]["ms"] / n * 1e3))
g.close()
