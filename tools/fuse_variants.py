"""A/B of fusion-kernel variants (tools/build_variant.sh): GT-pose fusion of the S-tum stream through each library,
time per fusion (HIP events around k_fuse + k_fuse_resolve, and wall clock per frame incl. k_normals) and the
fused map compared with the first library's (keys identical, sums within float noise).
usage: python tools/fuse_variants.py [--frames N] [--size WxH] [--vs 0.01] [--cap 22] lib1.so lib2.so ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--size", default="640x480")
ap.add_argument("--vs", type=float, default=0.01)
ap.add_argument("--cap", type=int, default=22)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
W, H = [int(v) for v in a.size.split("x")]
n = a.frames
seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
vs = np.float32(a.vs); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
ref = None
for path in a.libs:
    L = pkg.binding.load(os.path.abspath(path))
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=a.cap, lib=L)
    dev = [g.upload(f[0]) for f in frames]
    res = []
    for rep in range(3):
        g.reset()
        g.sync()
        t0 = time.perf_counter()
        for i in range(n):
            g.update_dev(dev[i], frames[i][1], frames[i][2])
        g.sync()
        res.append((time.perf_counter() - t0) / n * 1e6)
    g.reset()
    g.profile(1)
    for i in range(n):
        g.update_dev(dev[i], frames[i][1], frames[i][2])
    g.sync()
    pr = g.profile_read()
    g.profile(0)
    st = g.stats()
    k, p = g.export(sorted=True, raw=True)
    msg = ""
    if ref is None:
        ref = (k, p)
    else:
        same = k.shape == ref[0].shape and np.array_equal(k, ref[0])
        if same:
            scale = np.maximum(1.0, np.abs(ref[1][:, 4:5]))
            msg = "keys identical, max |d sums| / max(1, w) = %.2e" % float((np.abs(p - ref[1]) / scale).max())
        else:
            msg = "KEY SETS DIFFER (%d vs %d voxels)" % (k.shape[0], ref[0].shape[0])
    print("%-28s wall %6.1f us/frame (best of 3: %s)  events: fusion %6.1f us  normals %5.1f us | n_upd/frame %d deferred %d timeouts %d | %s" % (
        os.path.basename(path), min(res), " ".join("%.1f" % r for r in res), pr["fusion"]["ms"] / n * 1e3, pr["normals"]["ms"] / n * 1e3,
        st["n_upd"] // n, st["n_deferred"], st["fuse_timeouts"], msg), flush=True)
    g.close()
