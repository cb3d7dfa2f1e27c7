#!/usr/bin/env python3
"""Does the launch ORDER of the fusion tiles matter?  (test build)  Fuses the bench stream with GT poses; before every
fusion the tiles of each colour are re-sorted by what the previous frame's fusion counted for them (n_upd, descending =
longest first, or ascending), and the launches are timed with HIP events.   usage: fuse_order.py [frames (default 26)]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
scene = sys.argv[2] if len(sys.argv) > 2 else "tum"          # "spheres": the object-scan scene of C1 / C4 (four fifths of the tiles background)
W, H = 640, 480
seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0) if scene == "tum" else pkg.synth.Sequence(scene, W, H, n_frames=n, seed=0, step_deg=360.0 * 4 / 2000)
vs = np.float32(0.01); T = np.float32(10) * vs
frames = [seq.frame(i) for i in range(n)]
L = pkg.binding.load_test_lib()
os.environ["GSDF_DEFER"] = "0"
ntx, nty = W // 16, H // 16
NT = ntx * nty

def run(mode):
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22, lib=L)
    dev = [g.upload(f[0]) for f in frames]
    base = (C.c_uint32 * (2 * NT))(); nn = C.c_int(0)
    assert L.gsdf_debug_get_tile_order(g.h, base, C.byref(nn)) == 0
    base = np.array(base[:NT], dtype=np.uint32)
    colour = ((base & 0x7FFF) & 1) + 2 * (((base >> 16) & 0x7FFF) & 1)
    times = []
    for i in range(n):
        if i > 0 and mode != "static":
            cnt = (C.c_ulonglong * (4 * NT))()
            assert L.gsdf_debug_tile_counters(g.h, cnt, NT) == 0
            nu = np.array(cnt, dtype=np.int64).reshape(NT, 4)[:, 0]
            tx = (base & 0x7FFF).astype(np.int64); ty = ((base >> 16) & 0x7FFF).astype(np.int64)
            cost = nu[ty * ntx + tx]
            if mode == "topo":
                # global longest-first list with dependencies: a tile is emitted behind its neighbours of lower colour
                cmap = np.zeros((nty, ntx), np.int64); cmap[ty, tx] = cost
                done = np.zeros((nty, ntx), bool)
                out = []
                def emit(x, y):
                    if done[y, x]:
                        return
                    col = (x & 1) + 2 * (y & 1)
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            xx, yy = x + dx, y + dy
                            if (dx or dy) and 0 <= xx < ntx and 0 <= yy < nty and (xx & 1) + 2 * (yy & 1) < col:
                                emit(xx, yy)
                    done[y, x] = True
                    out.append(x | (y << 16))
                for j in np.argsort(-cost, kind="stable"):
                    emit(int(tx[j]), int(ty[j]))
                order = np.array(out, dtype=np.uint32)
            elif mode == "band_first":
                # round 6: within every residue class of the launch position (b % 8 = the XCD, = the image stripe) the tiles that had
                # something to fuse in the previous frame first, the others behind them, both in the static (colour-major) order:
                # a launch of 1200 workgroups takes ~20 us to DISPATCH, and a band tile at the end of it starts that late
                order = base.copy()
                for r in range(8):
                    pos = np.arange(r, NT, 8)
                    ent = base[pos]
                    band = cost[pos] > 0
                    order[pos] = np.concatenate([ent[band], ent[~band]])
            elif mode == "desc_stripe":
                # longest first within each of the 8 image stripes of a colour, stripes interleaved like the static order
                order = []
                for c in range(4):
                    idx = np.where(colour == c)[0]
                    stripe = (tx[idx] * 8 // ntx)
                    lists = [list(idx[stripe == r][np.argsort(-cost[idx[stripe == r]], kind="stable")]) for r in range(8)]
                    b = sum(len(o) for o in order)
                    left = len(idx)
                    while left:
                        r = b % 8
                        if not lists[r]:
                            r = int(np.argmax([len(l) for l in lists]))
                        order.append(base[lists[r].pop(0):][:1])
                        b += 1; left -= 1
                order = np.concatenate(order).astype(np.uint32)
            else:
                order = []
                for c in range(4):
                    idx = np.where(colour == c)[0]
                    key = -cost[idx] if mode == "desc" else cost[idx]
                    order.append(base[idx[np.argsort(key, kind="stable")]])
                order = np.concatenate(order).astype(np.uint32)
            assert L.gsdf_debug_set_tile_order(g.h, order.ctypes.data_as(C.POINTER(C.c_uint32)), NT) == 0
        g.sync()
        g.profile(1)
        g.update_dev(dev[i], frames[i][1], frames[i][2])
        g.sync()
        pr = g.profile_read()
        g.profile(0)
        times.append(pr["fusion"]["ms"] * 1e3)
    k, p = g.export(sorted=True, raw=True)
    g.close()
    return np.array(times), k, p

ref = None
for mode in (("static", "band_first", "static", "band_first") if scene != "tum" or os.environ.get("BAND_FIRST") else ("static", "desc", "desc_stripe", "static", "desc", "desc_stripe")):
    t, k, p = run(mode)
    if ref is None:
        ref = (k, p)
    same = np.array_equal(k, ref[0]) and np.abs(p - ref[1]).max() < 1e-3
    print("%-10s: k_fuse mean of frames 6..%d %.2f us (median %.2f, min %.2f, max %.2f); map equal to the static order's: %s" % (
        mode, n - 1, t[6:].mean(), np.median(t[6:]), t[6:].min(), t[6:].max(), same))
