#!/usr/bin/env python3
"""Two shard contexts on ONE GPU against one (SURVEY.md 8e "G logical shards on 1 GPU"; VERDICT r4 #7a): the 250-frame shard of
bench.py's sharded flavour fused by one context, and by two contexts on two streams (enqueued alternately, synchronised every
8 / 16 / 32 frame pairs) + gsdf_merge_from; also with an RCCL communicator alive and after an exchange.  -> profiles/r05_two_contexts.txt"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
if '--torch' in sys.argv:
    import torch   # bench.py's situation: libgsdf then binds to the HIP runtime PyTorch bundles (another ROCm release)
    print('torch', torch.__version__, 'imported first')
seq, frames = bench.render_frames("spheres", 640, 480, range(250), seed=0, n_frames=250, step_deg=360.0*4/2000)
import __graft_entry__ as graft
pkg = graft.package()
vs = np.float32(0.01); T = np.float32(10)*vs
F = 250
if '--default-streams' in sys.argv:     # two ordinary contexts: their streams may share a hardware queue
    g = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=23)
    g2 = pkg.GradSdf(vs, T, 640, 480, seq.K, capacity_log2=23)
else:                                   # gsdf_create_shards: a hardware queue each
    g, g2 = pkg.GradSdf.shards(2, vs, T, 640, 480, seq.K, capacity_log2=23)
dev = [g.upload(f[0]) for f in frames]
half = 125
dev2 = [g2.upload(f[0]) for f in frames[half:]]
def one():
    g.reset(); t0=time.perf_counter()
    for j,(d,f) in enumerate(zip(dev,frames)):
        g.update_dev(d,f[1],f[2])
        if j%32==31: g.sync()
    g.sync(); return F/(time.perf_counter()-t0)
def two(sync_every=16):
    g.reset(); g2.reset(); t0=time.perf_counter()
    for j in range(half):
        g.update_dev(dev[j],frames[j][1],frames[j][2])
        g2.update_dev(dev2[j],frames[half+j][1],frames[half+j][2])
        if j%sync_every==sync_every-1: g.sync(); g2.sync()
    g.sync(); g2.sync(); tm=time.perf_counter(); g.merge_from(g2); t1=time.perf_counter()
    return F/(t1-t0), (t1-tm)*1e3
if '--profile-first' in sys.argv:
    # bench.py's situation: an event-timed replay ran earlier in the process (gsdf profile mode records HIP events around launches)
    g.profile(1)
    for j in range(20): g.update_dev(dev[j], frames[j][1], frames[j][2])
    g.sync(); print('profile mode used once:', g.profile_read()['fusion']); g.profile(0)
import threading
def two_threads(sync_every=32):
    g.reset(); g2.reset()
    gate = threading.Barrier(3)
    def body(ctx, dv, fr):
        gate.wait()
        for j, (d, f) in enumerate(zip(dv, fr)):
            ctx.update_dev(d, f[1], f[2])
            if j % sync_every == sync_every - 1: ctx.sync()
        ctx.sync()
    th = [threading.Thread(target=body, args=(g, dev[:half], frames[:half])), threading.Thread(target=body, args=(g2, dev2, frames[half:]))]
    for t in th: t.start()
    gate.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    tm = time.perf_counter(); g.merge_from(g2); t1 = time.perf_counter()
    return F/(t1-t0), (t1-tm)*1e3
for r in range(3):
    print("two host threads:", two_threads(32))
for r in range(3):
    print("one", round(one(),1), "two16", two(16), "two32", two(32), "two8", two(8))
comm = pkg.binding.rccl_comm_init(1, pkg.binding.rccl_unique_id(), 0, 0)
for r in range(2):
    print("with comm alive: one", round(one(),1), "two16", two(16))
g.reset()
for j,(d,f) in enumerate(zip(dev,frames)): g.update_dev(d,f[1],f[2])
g.merge_allreduce_rccl(comm)
for r in range(2):
    print("after an exchange: one", round(one(),1), "two16", two(16))
