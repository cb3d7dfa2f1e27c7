import sys, time, os, numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0, ".")
import __graft_entry__ as G
pkg = G.package()
n = int(os.environ.get("NFR", "120"))
seq = pkg.synth.Sequence("tum", 640, 480, n_frames=n, seed=0)
vs = np.float32(0.01)
g = pkg.GradSdf(vs, np.float32(10) * vs, 640, 480, seq.K, capacity_log2=22)
frames = [seq.frame(i) for i in range(n)]
dev = [g.upload(f[0]) for f in frames]
for rep in range(2):
    g.reset(); g.sync()
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        g.update_dev(dev[i], frames[i][1], frames[i][2])
        t1 = time.perf_counter()
        g.sync()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts = np.array(ts) * 1e6
    print("rep", rep, "enqueue us: first", ts[0, 0].round(1), "median", np.median(ts[:, 0]).round(1), "| enqueue+sync us: frame0", ts[0, 1].round(1), "frames1-5", ts[1:6, 1].round(1), "median", np.median(ts[:, 1]).round(1), "last", ts[-1, 1].round(1))
    g.reset(); g.sync()
    t0 = time.perf_counter()
    for i in range(n):
        g.update_dev(dev[i], frames[i][1], frames[i][2])
    t1 = time.perf_counter()
    g.sync()
    t2 = time.perf_counter()
    print("   async: enqueue all %.1f us/frame, total %.1f us/frame" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
