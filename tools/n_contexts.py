#!/usr/bin/env python3
"""N shard contexts on ONE GPU (gsdf_create_shards, N = 1..4) on the 250-frame shard of bench.py's sharded flavour: frames/s incl.
the local merges (gsdf_merge_from of contexts 1.. into context 0).  -> appended to profiles/r05_two_contexts.txt"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench

seq, frames = bench.render_frames("spheres", 640, 480, range(250), seed=0, n_frames=250, step_deg=360.0 * 4 / 2000)
import __graft_entry__ as graft
pkg = graft.package()
vs = np.float32(0.01)
T = np.float32(10) * vs
F = 250
for n in (1, 2, 3, 4):
    ctx = pkg.GradSdf.shards(n, vs, T, 640, 480, seq.K, capacity_log2=23)
    bounds = [(F * i) // n for i in range(n + 1)]
    devs = [[ctx[i].upload(f[0]) for f in frames[bounds[i]:bounds[i + 1]]] for i in range(n)]
    res = []
    for rep in range(4):
        for g in ctx:
            g.reset()
        t0 = time.perf_counter()
        m = max(len(d) for d in devs)
        for j in range(m):
            for i in range(n):
                if j < len(devs[i]):
                    f = frames[bounds[i] + j]
                    ctx[i].update_dev(devs[i][j], f[1], f[2])
            if j % 32 == 31:
                for g in ctx:
                    g.sync()
        for g in ctx:
            g.sync()
        for i in range(1, n):
            ctx[0].merge_from(ctx[i])
        res.append(F / (time.perf_counter() - t0))
    print("%d context(s): %s frames/s (frames after merge %d, voxels %d)" % (n, [round(r, 1) for r in res[1:]], ctx[0].stats()["frames"], ctx[0].count()))
    for g in ctx:
        g.close()
