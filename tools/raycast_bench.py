#!/usr/bin/env python3
"""Raycaster timing on the bench map: the look-ahead kernel (k_raycast) against the sample-at-a-time kernel of rounds 1-2
(k_raycast_v1, test build, debug bit 16384) -- same definition, outputs compared bit for bit -- with the counters of the
roofline entry (samples, records).  `python tools/raycast_bench.py [frames] [reps]`"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pkg = graft.package()
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n_frames, seed=0)
    vs = np.float32(0.01)
    L = pkg.binding.load_test_lib()
    g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22, lib=L)
    fr = [seq.frame(i) for i in range(n_frames)]
    for d, R, t in fr:
        g.update(d, R, t)
    R, t = fr[-1][1], fr[-1][2]
    N = W * H
    out = {}
    bufs = {}
    for name, flag in (("k_raycast", 0), ("k_raycast_v1", 16384)):
        g.debug_flags(flag)
        p = C.c_void_p()
        g._chk(g.L.gsdf_dev_alloc(g.h, C.byref(p), 4 * N * 4))
        g._dev.append(p)
        nrm = C.c_void_p(p.value + 4 * N)
        g.raycast_dev(R, t, p, nrm)                      # warm-up
        g.sync()
        g.raycast_counters(reset=True)
        g.profile(1)
        for _ in range(reps):
            g.raycast_dev(R, t, p, nrm)
        g.sync()
        pr = g.profile_read_all()["raycast"]
        g.profile(0)
        n_wg = ((W + 15) // 16) * ((H + 15) // 16)
        rows = np.zeros((n_wg, 8), np.uint64)
        g.L.gsdf_debug_raycast_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        g.L.gsdf_debug_raycast_rows(g.h, rows.ctypes.data_as(C.c_void_p), n_wg)
        samples, records = g.raycast_counters(reset=True)
        if flag == 0:
            life = (rows[:, 5] - rows[:, 4]).astype(np.float64) / 100.0          # us
            start = (rows[:, 4] - rows[:, 4].min()).astype(np.float64) / 100.0
            itf, its = rows[:, 2].astype(np.float64) / (reps + 0), rows[:, 3].astype(np.float64) / (reps + 0)
            itmax = rows[:, 6].astype(np.float64)
            order = np.argsort(-life)[:8]
            fill = rows[:, 7]                   # test build: how full the slowest wave of each workgroup was, iteration by iteration
            le32, le16, le8 = (fill & 0xFFF).astype(np.float64), ((fill >> 12) & 0xFFF).astype(np.float64), ((fill >> 24) & 0xFFF).astype(np.float64)
            lsum = (fill >> 36).astype(np.float64)
            slow_wg = itmax >= np.percentile(itmax, 90)
            fill_stats = {"slowest_decile_iters_mean": round(float(itmax[slow_wg].mean()), 1),
                          "slowest_decile_iters_le32_mean": round(float(le32[slow_wg].mean()), 1),
                          "slowest_decile_iters_le16_mean": round(float(le16[slow_wg].mean()), 1),
                          "slowest_decile_iters_le8_mean": round(float(le8[slow_wg].mean()), 1),
                          "slowest_decile_lane_fill": round(float((lsum[slow_wg] / np.maximum(64.0 * itmax[slow_wg], 1)).mean()), 3),
                          "all_lane_fill": round(float((lsum / np.maximum(64.0 * itmax, 1)).mean()), 3),
                          "all_iters_le32_frac": round(float(le32.sum() / max(itmax.sum(), 1)), 3)}
            wg_stats = {"life_us_mean": round(float(life.mean()), 1), "life_us_p50": round(float(np.median(life)), 1), "life_us_max": round(float(life.max()), 1),
                        "last_start_us": round(float(start.max()), 1), "last_end_us": round(float((start + life).max()), 1),
                        "iters_fast_mean": round(float(itf.mean()), 2), "iters_slow_mean": round(float(its.mean()), 2), "iters_slow_max": float(its.max()),
                        "slowest_wave_iters_mean": round(float(itmax.mean()), 1), "slowest_wave_iters_max": float(itmax.max()),
                        "us_per_iteration_of_slowest_wave_p50": round(float(np.median(life / np.maximum(itmax, 1))), 3),
                        "corr_life_iters": round(float(np.corrcoef(life, itmax)[0, 1]), 3),
                        "slowest_wave_fill": fill_stats,
                        "slowest": [{"wg": int(i), "life_us": round(float(life[i]), 1), "start_us": round(float(start[i]), 1), "fast": float(itf[i]), "slow": float(its[i]),
                                     "samples": int(rows[i, 0] // reps), "slowest_wave_iters": float(itmax[i])} for i in order]}
        us = pr["ms"] * 1e3 / max(pr["launches"], 1)
        bufs[name] = g.download(p, (4, H, W), np.float32)
        hits = int((bufs[name][0] > 0).sum())
        out[name] = {"us": round(us, 2), "launches": pr["launches"], "hits": hits,
                     "samples_per_launch": samples // max(reps, 1), "records_per_launch": records // max(reps, 1)}
        if flag == 0:
            out[name]["workgroups"] = wg_stats
        if samples:
            b = 8.0 * samples / reps + 32.0 * records / reps + 16.0 * N
            out[name]["algorithmic_bytes"] = round(b)
            out[name]["GBps"] = round(b / (us * 1e-6) / 1e9, 1)
    out["identical"] = bool(np.array_equal(bufs["k_raycast"].view(np.uint32), bufs["k_raycast_v1"].view(np.uint32)))
    out["voxels"] = g.count()
    out["depth_vs_input_p50_mm"] = float(np.median(np.abs(bufs["k_raycast"][0] - fr[-1][0])[bufs["k_raycast"][0] > 0]) * 1e3)
    print(json.dumps(out))
    g.close()


if __name__ == "__main__":
    main()
