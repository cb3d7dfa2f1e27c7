import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = 12
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
L = pkg.binding.load_test_lib()          # -DGSDF_EXPERIMENTS build
g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=22, lib=L)
g.debug_flags(128)
dev = [g.upload(f[0]) for f in frames]
import ctypes
def rd():
    o = (ctypes.c_ulonglong * 24)()
    L.gsdf_debug_read(g.h, o)
    return np.array(list(o), dtype=np.float64)
for i in range(n // 2):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
g.sync()
mid = rd()
for i in range(n // 2, n):
    g.update_dev(dev[i], frames[i][1], frames[i][2])
g.sync()
late = rd() - mid
wgl = 1200.0 * (n - n // 2)
print("frames %d..%d, mean per workgroup (clock ticks): prologue %.0f  ray walk %.0f  flush %.0f" % (n // 2, n - 1, late[4] / wgl, late[2] / wgl, late[3] / wgl))
st = g.stats()
print("go per frame", st["n_hit"] / n, "wave-samples per frame", 1200 * 8 * 10.5, "ratio", st["n_hit"] / n / (1200 * 8 * 10.5))
import ctypes
out = (ctypes.c_ulonglong * 24)()
L.gsdf_debug_read(g.h, out)
print("wave-level events per frame: bucket full", out[0] / n, "CAS lost", out[1] / n)
wg = 1200.0 * n
print("mean per workgroup (us): prologue %.2f  ray walk %.2f  flush %.2f" % (out[4] / wg / 100.0, out[2] / wg / 100.0, out[3] / wg / 100.0))
print("pixel loads arrived after %.2f us (mean per workgroup)" % (out[5] / wg / 100.0))
for c in range(4):
    f = [out[8 + 4 * c + k] / (wg / 4) / 100.0 for k in range(4)]
    print("flush, colour %d tiles (us per workgroup): keys+probe+wait %.2f  record loads %.2f  adds+stores issued %.2f  drain+flag %.2f" % (c, *f))
print("n_upd/frame", st["n_upd"] / n, "voxels", g.count())
g.close()
