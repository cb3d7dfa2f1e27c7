#!/usr/bin/env python3
"""Where does the staging-inclusive frame loop lose time against the resident one?  (bench.py: config.staged_fps vs value)

  python tools/staged_trace.py run             the workload: 4 windows of 20 tracked + fused frames with the frames resident in HBM,
                                               then 4 windows with every frame handed over from a page-locked host buffer
                                               (gsdf_dev_upload_ahead / gsdf_upload_wait / gsdf_track_and_fuse_dev / gsdf_mark), the
                                               phases separated by a 50 ms pause; run it under rocprofv3 --kernel-trace
  python tools/staged_trace.py report <dir>    splits the kernel trace at the pauses and prints, per phase, the frame period and the
                                               medians of kernel durations and of the gaps in front of the kernels"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import ctypes as C
    import __graft_entry__ as G
    from bench import quat_to_R
    pkg = G.package()
    W, H, Wm, K = 640, 480, 5, 20
    n = 1 + Wm + K
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01); T = np.float32(10) * vs
    frames = [seq.frame(i) for i in range(n)]
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    dev = [g.upload(f[0]) for f in frames]
    d0, R0, t0 = frames[0]
    p0 = np.concatenate([t0, pkg.synth.R_to_quat_np(R0)]).astype(np.float32)
    L = g.L
    nbytes = W * H * 4
    host = []
    for i in range(1 + Wm, n):
        hp = C.c_void_p()
        g._chk(L.gsdf_host_alloc(g.h, C.byref(hp), nbytes))
        C.memmove(hp, np.ascontiguousarray(frames[i][0], np.float32).ctypes.data, nbytes)
        host.append(hp)
    S, AHEAD, EVERY = 12, 4, 4
    slots = []
    for _ in range(S):
        dp = C.c_void_p()
        g._chk(L.gsdf_dev_alloc(g.h, C.byref(dp), nbytes)); g._dev.append(dp); slots.append(dp)

    def start():
        g.reset()
        g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
        g.set_pose(p0)
        for i in range(1, 1 + Wm):
            g.track_and_fuse_dev(dev[i])
        g.sync()

    for phase in ("resident", "staged"):
        for rep in range(5):
            start()
            time.sleep(0.05)                   # the pause the report splits the trace at
            t_a = time.perf_counter()
            if phase == "resident":
                for i in range(1 + Wm, n):
                    g.track_and_fuse_dev(dev[i])
            else:
                mark = [None] * S; unmarked = []; ids = {}; nxt = 0
                for j in range(K):
                    while nxt < K and nxt <= j + AHEAD:
                        sl = nxt % S
                        if mark[sl] is not None:
                            g._chk(L.gsdf_mark_wait(g.h, mark[sl]))
                        uid = C.c_int64(0)
                        g._chk(L.gsdf_dev_upload_ahead(g.h, slots[sl], host[nxt], nbytes, C.byref(uid)))
                        ids[nxt] = uid.value; nxt += 1
                    g._chk(L.gsdf_upload_wait(g.h, ids.pop(j)))
                    g.track_and_fuse_dev(slots[j % S])
                    unmarked.append(j % S)
                    if len(unmarked) >= EVERY or j == K - 1:
                        mk = C.c_int64(0)
                        g._chk(L.gsdf_mark(g.h, C.byref(mk)))
                        for sl in unmarked:
                            mark[sl] = mk.value
                        unmarked = []
            g.sync()
            print("%s window %d: %.1f frames/s" % (phase, rep, K / (time.perf_counter() - t_a)), flush=True)
            time.sleep(0.05)
    g.close()


def report(root):
    import csv, glob
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    ev = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:20], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # windows = runs of kernels between pauses > 20 ms that hold exactly 20 executed k_fuse launches
    wins, cur = [], []
    for e in ev:
        if cur and e[1] - cur[-1][2] > 20e6:
            wins.append(cur); cur = []
        cur.append(e)
    wins.append(cur)
    timed = [w for w in wins if sum(1 for e in w if e[0].startswith("k_fuse") and e[2] - e[1] > 20e3) == 20]
    print("%d windows of 20 fused frames found (expected 10: 5 resident, then 5 staged)" % len(timed))
    for wi, w in enumerate(timed):
        fuse = [e for e in w if e[0].startswith("k_fuse") and e[2] - e[1] > 20e3]
        period = (fuse[-1][2] - fuse[0][2]) / 19e3
        stats = {}
        prev = None
        for e in w:
            if prev is not None:
                stats.setdefault(e[0], []).append(((e[2] - e[1]) / 1e3, (e[1] - prev[2]) / 1e3))
            prev = e
        line = "window %d (%s): frame period %.1f us |" % (wi, "resident" if wi < len(timed) // 2 else "staged", period)
        for k in ("k_fuse", "k_track_pass"):
            v = [x for kk, vv in stats.items() if kk.startswith(k) for x in vv]
            big = [x for x in v if x[0] > 6.0]                      # executed launches
            line += " %s: med %.1f us, gap in front med %.2f p90 %.2f, sum of gaps per frame %.1f |" % (
                k, np.median([x[0] for x in big]), np.median([x[1] for x in big]), np.percentile([x[1] for x in big], 90), sum(x[1] for x in v) / 20)
        busy = sum(e[2] - e[1] for e in w if e[1] >= fuse[0][2] and e[2] <= fuse[-1][2]) / 19e3
        line += " kernels busy %.1f us per frame" % busy
        print(line)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
