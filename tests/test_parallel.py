"""Multi-GPU path on CPU: world-size-2 gloo processes run the frame-sharded fusion protocol
(shard -> fuse -> all-gather of (key, raw sums) lists -> additive merge) with the CPU oracle standing in
for the per-rank fusion, and must reproduce the single-process map.  The GPU path runs the same
exchange with device tensors over RCCL (gradient-sdf_amd/parallel.py:exchange_and_merge)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.package()
    O = graft.oracle_module()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, n = 96, 72, 5
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=9, step_deg=3.0)
    vs = np.float32(0.04)
    o = O.Oracle(vs, np.float32(5) * vs, W, H, seq.K)
    lo, hi = pkg.parallel.shard_range(n, rank, world)
    for i in range(lo, hi):
        o.update(*seq.frame(i))
    keys, pay = o.export()
    raw = pay.copy()
    raw[:, 0] = pay[:, 0] * pay[:, 4]                    # dist -> sum w*d (the wire format carries sums)
    lists = pkg.parallel.allgather_lists(keys, raw, dist)
    k, p = pkg.parallel.merge_numpy([(a.numpy(), b.numpy()) for a, b in lists])
    # the dense form: one all-reduce of per-voxel sums over the union of the ranks' 4x4x4 blocks
    ops = pkg.parallel.NumpyBlockOps(keys, raw)
    n_blocks = pkg.parallel.allreduce_merge(ops, dist)
    # barrier + max-over-ranks reduction used by bench.py's timing
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), keys=k, pay=p, tmax=t.numpy(), lo=lo, hi=hi, keys_ar=ops.keys, pay_ar=ops.pay,
             n_blocks=n_blocks)
    dist.destroy_process_group()


def test_shard_ranges_tile_the_frames(pkg):
    for n in (1, 5, 8, 2000):
        for world in (1, 2, 3, 8):
            r = [pkg.parallel.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_merge_numpy_reduces_by_key(pkg):
    k1 = np.array([[1, 2, 3], [0, 0, 0]], np.int32)
    k2 = np.array([[0, 0, 0], [-5, 7, 1]], np.int32)
    p1 = np.ones((2, 5), np.float32)
    p2 = 2 * np.ones((2, 5), np.float32)
    k, p = pkg.parallel.merge_numpy([(k1, p1), (k2, p2)])
    assert k.tolist() == [[0, 0, 0], [-5, 7, 1], [1, 2, 3]]          # (z, y, x) order
    assert p[:, 0].tolist() == [3.0, 2.0, 1.0]


def test_frame_sharded_fusion_world2_gloo(pkg, O, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    assert (int(a["lo"]), int(a["hi"]), int(b["lo"]), int(b["hi"])) == (0, 3, 3, 5)
    assert a["tmax"][0] == 2.0 and b["tmax"][0] == 2.0
    assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["pay"], b["pay"])
    # the all-reduce form gives the same map on both ranks, and the same map as the all-gather form
    assert np.array_equal(a["keys_ar"], b["keys_ar"]) and np.array_equal(a["pay_ar"], b["pay_ar"]) and int(a["n_blocks"]) > 10
    assert np.array_equal(a["keys_ar"], a["keys"])
    assert np.abs(a["pay_ar"] - a["pay"]).max() <= 1e-5 * max(1.0, float(np.abs(a["pay"]).max()))
    # single-process reference over all frames
    W, H, n = 96, 72, 5
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=9, step_deg=3.0)
    vs = np.float32(0.04)
    o = O.Oracle(vs, np.float32(5) * vs, W, H, seq.K)
    for i in range(n):
        o.update(*seq.frame(i))
    keys, pay = o.export()
    assert np.array_equal(a["keys"], keys)                               # union of key sets, bit-exact
    w = a["pay"][:, 4]
    assert np.abs(w - pay[:, 4]).max() <= 1e-4 * max(1.0, pay[:, 4].max())
    assert np.abs(a["pay"][:, 0] / w - pay[:, 0]).max() <= 1e-4
    assert np.abs(a["pay"][:, 1:4] - pay[:, 1:4]).max() <= 1e-4 * max(1.0, pay[:, 4].max())


# ---- the exchange as ONE C-ABI call (gsdf_merge_allreduce / gsdf_merge_allreduce_with) on real hardware ---------------

def _gpu_worker(rank, world, port, out_dir):
    """One rank of the frame-sharded GT-pose fusion on the GPU: fuse the own frame range through the C-ABI, then the C++
    exchange entry with gloo as its transport (two ranks share the one GPU of the test box, which RCCL does not allow)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, n = 320, 240, 8
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=9, step_deg=3.0)
    vs = np.float32(0.02)
    # (rank 1's table is twice the size of rank 0's -- as if it had grown during its scan: the exchange brings rank 0's up first)
    g = pkg.GradSdf(vs, np.float32(5) * vs, W, H, seq.K, capacity_log2=20 + rank, device=0)
    g.enable_vis(40)                                     # vis_ takes part in the exchange (bits = frames of ALL ranks)
    lo, hi = pkg.parallel.shard_range(n, rank, world)
    for i in range(lo, hi):
        g.update(*seq.frame(i))

    def allgather(send):
        t = torch.from_numpy(send)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).numpy()

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        return t.numpy()

    own = g.count()
    n_blocks, nbytes = g.merge_allreduce_with(allgather, allreduce, world)
    keys, pay = g.export(sorted=True, raw=True)
    kv, vis = g.export_vis()
    frames = g.stats()["frames"]
    again = ""
    try:
        g.merge_allreduce_with(allgather, allreduce, world)          # one-shot: a second exchange would double every sum
    except pkg.binding.GsdfError as e:
        again = str(e)
    np.savez(os.path.join(out_dir, "gpu_rank%d.npz" % rank), keys=keys, pay=pay, own=own, n_blocks=n_blocks, nbytes=nbytes,
             vis_keys=kv, vis=vis, frames=frames, again=again)
    g.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_merge_allreduce_c_entry_world2_on_one_gpu(pkg, O, tmp_path):
    """gsdf_merge_allreduce_with from two processes (one context each): afterwards both hold the map of ALL frames --
    key set bit-exact against the oracle's single-process fusion, sums within float noise."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "gpu_rank0.npz")
    b = np.load(tmp_path / "gpu_rank1.npz")
    assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["pay"], b["pay"])   # every rank holds the same sums
    assert int(a["n_blocks"]) == int(b["n_blocks"]) > 100 and int(a["nbytes"]) == int(a["n_blocks"]) * 64 * (5 * 4 + 2 * 4)
    assert int(a["own"]) < len(a["keys"]) and int(b["own"]) < len(a["keys"])             # the shards really differed
    W, H, n = 320, 240, 8
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=9, step_deg=3.0)
    vs = np.float32(0.02)
    o = O.Oracle(vs, np.float32(5) * vs, W, H, seq.K)
    for i in range(n):
        o.update(*seq.frame(i))
    keys, pay = o.export()
    assert np.array_equal(a["keys"], keys)
    w = a["pay"][:, 4]
    scale = np.maximum(1.0, pay[:, 4])
    assert (np.abs(w - pay[:, 4]) / scale).max() <= 1e-4
    assert np.abs(a["pay"][:, 0] / w - pay[:, 0]).max() <= 1e-4
    assert (np.abs(a["pay"][:, 1:4] - pay[:, 1:4]).max(axis=1) / scale).max() <= 1e-4
    # vis_ and Sdf::counter_ are part of the exchange (MapGradPixelSdf.cpp:113-115, SURVEY.md 8e "vis = OR of frame bits"):
    # rank 1's frames 0..3 are integrated frames 4..7 of the merged map; both ranks end with the oracle's bit-vectors
    assert int(a["frames"]) == int(b["frames"]) == n == o.frame_counter()
    assert np.array_equal(a["vis_keys"], keys) and np.array_equal(b["vis_keys"], keys)
    vo = o.export_vis(2)
    assert np.array_equal(a["vis"], vo) and np.array_equal(b["vis"], vo)
    assert (vo[:, 0] >> 4).any() and (vo[:, 0] & 0xF).any()                              # bits of both shards are present
    assert "one-shot" in str(a["again"]) and "one-shot" in str(b["again"])               # re-merging is refused on every rank


def _gpu_worker8(rank, world, port, out_dir):
    """One of EIGHT ranks (the node size of BASELINE configs[3]) on the one GPU of the test box: uneven frame shards (19 frames
    over 8 ranks), vis_ enabled, the exchange over gloo callbacks."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, n = 160, 120, 19
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=3, step_deg=4.0)
    vs = np.float32(0.04)
    g = pkg.GradSdf(vs, np.float32(5) * vs, W, H, seq.K, capacity_log2=18, device=0)
    g.enable_vis(32)
    lo, hi = pkg.parallel.shard_range(n, rank, world)
    dev = [g.upload(seq.frame(i)[0]) for i in range(lo, hi)]
    for j, i in enumerate(range(lo, hi)):
        g.update_dev(dev[j], seq.frame(i)[1], seq.frame(i)[2])       # the pipelined entry: the last fusion waits until the exchange asks

    def allgather(send):
        t = torch.from_numpy(send)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).numpy()

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        return t.numpy()

    g.merge_allreduce_with(allgather, allreduce, world)
    keys, pay = g.export(sorted=True, raw=True)
    kv, vis = g.export_vis()
    np.savez(os.path.join(out_dir, "r8_%d.npz" % rank), keys=keys, pay=pay, vis=vis, frames=g.stats()["frames"], lo=lo, hi=hi)
    g.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_merge_allreduce_eight_ranks_uneven_shards(pkg, O, tmp_path):
    """Eight ranks with shards of 3 / 2 frames: every rank ends with the oracle's key set, its vis_ bit-vectors (frame f of
    rank r = integrated frame lo(r) + f) and Sdf::counter_ = 19."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    W, H, n = 160, 120, 19
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=3, step_deg=4.0)
    vs = np.float32(0.04)
    o = O.Oracle(vs, np.float32(5) * vs, W, H, seq.K)
    for i in range(n):
        o.update(*seq.frame(i))
    keys, pay = o.export()
    vo = o.export_vis(1)
    shards = []
    for r in range(8):
        z = np.load(tmp_path / ("r8_%d.npz" % r))
        shards.append((int(z["lo"]), int(z["hi"])))
        assert np.array_equal(z["keys"], keys) and int(z["frames"]) == n
        bad = np.nonzero((z["vis"] != vo).any(axis=1))[0]
        assert bad.size == 0, "rank %d: vis_ differs on %d voxels, e.g. %s: got %s, oracle %s (shard %s)" % (
            r, bad.size, keys[bad[:4]].tolist(), [hex(int(v)) for v in z["vis"][bad[:4], 0]], [hex(int(v)) for v in vo[bad[:4], 0]], shards[-1])
        w = z["pay"][:, 4]
        assert (np.abs(w - pay[:, 4]) / np.maximum(1.0, pay[:, 4])).max() <= 1e-4
        assert np.abs(z["pay"][:, 0] / w - pay[:, 0]).max() <= 1e-4
    assert shards[0] == (0, 3) and shards[-1] == (17, 19) and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))


@pytest.mark.gpu
def test_merge_allreduce_over_rccl_one_rank(pkg):
    """gsdf_merge_allreduce over a real RCCL communicator (gsdf_rccl_comm_init; one rank -- the box has one GPU): block ids,
    pack, ncclAllGather / ncclAllReduce on the context's stream, unpack.  With one rank the map must come back unchanged,
    bit for bit; a second call finds the same union (with ONE rank the exchange is the identity, so it may be repeated)."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("tum", W, H, n_frames=4, seed=1)
    vs = np.float32(0.01)
    g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=21)
    for i in range(seq.n):
        g.update(*seq.frame(i))
    k0, p0 = g.export(sorted=True, raw=True)
    comm = pkg.binding.rccl_comm_init(1, pkg.binding.rccl_unique_id(), 0, 0)
    try:
        nb, nbytes = g.merge_allreduce_rccl(comm)
        assert nb > 100 and nbytes == nb * 64 * 5 * 4
        k1, p1 = g.export(sorted=True, raw=True)
        assert np.array_equal(k0, k1) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
        nb2, _ = g.merge_allreduce_rccl(comm)
        assert nb2 == nb
        g.update(*seq.frame(0))                           # the map stays usable: fusing after the exchange works
        assert g.count() == len(k0)
    finally:
        pkg.binding.rccl_comm_destroy(comm)
    g.close()


@pytest.mark.gpu
def test_merge_from_two_contexts_on_one_gpu(pkg, O):
    """gsdf_merge_from: two shard contexts on ONE device, each fusing its half of the frames on its own stream (enqueued
    alternately, so the launches overlap), then dst += src: the oracle's key set bit for bit, sums within float noise, vis_
    bit-vectors with the source's frames shifted behind the destination's, Sdf::counter_ = all frames; the source stays as it was."""
    W, H, n = 320, 240, 10
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=5, step_deg=3.0)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    a, b = pkg.GradSdf.shards(2, vs, T, W, H, seq.K, capacity_log2=20)   # gsdf_create_shards: a hardware queue each
    a.enable_vis(32); b.enable_vis(32)
    fr = [seq.frame(i) for i in range(n)]
    da = [a.upload(f[0]) for f in fr[:6]]
    db = [b.upload(f[0]) for f in fr[6:]]
    for j in range(6):                                             # alternately: both streams hold work at the same time
        a.update_dev(da[j], fr[j][1], fr[j][2])
        if j < 4:
            b.update_dev(db[j], fr[6 + j][1], fr[6 + j][2])
    kb0, pb0 = None, None
    a.merge_from(b)
    kb0, pb0 = b.export(sorted=True, raw=True)
    o = O.Oracle(vs, T, W, H, seq.K)
    for d, R, t in fr:
        o.update(d, R, t)
    keys, pay = o.export()
    ka, pa = a.export(sorted=True, raw=True)
    assert np.array_equal(ka, keys) and a.stats()["frames"] == n
    w = pa[:, 4]
    scale = np.maximum(1.0, pay[:, 4])
    assert (np.abs(w - pay[:, 4]) / scale).max() <= 1e-4
    assert np.abs(pa[:, 0] / w - pay[:, 0]).max() <= 1e-4
    assert (np.abs(pa[:, 1:4] - pay[:, 1:4]).max(axis=1) / scale).max() <= 1e-4
    kv, vis = a.export_vis()
    assert np.array_equal(kv, keys) and np.array_equal(vis, o.export_vis(1))
    ob = O.Oracle(vs, T, W, H, seq.K)                              # the source is unchanged: still the map of its own four frames
    for d, R, t in fr[6:]:
        ob.update(d, R, t)
    assert np.array_equal(kb0, ob.export()[0]) and b.stats()["frames"] == 4
    with pytest.raises(pkg.GsdfError):
        a.merge_from(a)
    with pytest.raises(pkg.GsdfError):                             # at most four shard contexts (one hardware queue each)
        pkg.GradSdf.shards(5, vs, T, W, H, seq.K, capacity_log2=16)
    c = pkg.GradSdf(np.float32(0.04), np.float32(5) * np.float32(0.04), W, H, seq.K, capacity_log2=16)
    with pytest.raises(pkg.GsdfError) as e:                        # another voxel size: not the same map
        a.merge_from(c)
    assert "voxel size" in str(e.value)
    c.close()
    a.close(); b.close()


@pytest.mark.gpu
def test_merge_from_checks_the_room_in_dst_before_touching_it(pkg, O):
    """a dst too small for both maps used to come back half merged with the frame counter advanced.  Now the blocks
    of both maps are counted first: without leave to grow the call fails with GSDF_ERR_TABLE_FULL and dst is bit for bit what
    it was; with gsdf_set_auto_grow dst is doubled as far as needed and the merge equals the oracle's map of all frames."""
    W, H, n = 320, 240, 6
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=3)
    vs = np.float32(0.02); T = np.float32(5) * vs                     # one frame: ~3 000 blocks, the other five: ~3 300
    fr = [seq.frame(i) for i in range(n)]
    src = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
    for d, R, t in fr[1:]:
        src.update(d, R, t)
    dst = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=18)          # 4096 block entries
    dst.update(*fr[0])
    k0, p0 = dst.export(sorted=True, raw=True)
    blocks_dst = len(np.unique(k0 >> 2, axis=0))
    ks, _ = src.export(sorted=True, raw=True)
    blocks_src = len(np.unique(ks >> 2, axis=0))
    assert (blocks_dst + blocks_src) * 100 > 4096 * 90, (blocks_dst, blocks_src)      # the case under test: no room
    with pytest.raises(pkg.GsdfError) as e:
        dst.merge_from(src)
    assert e.value.code == pkg.binding.ERR_TABLE_FULL and "unchanged" in str(e.value)
    k1, p1 = dst.export(sorted=True, raw=True)
    assert np.array_equal(k0, k1) and np.array_equal(p0, p1) and dst.stats()["frames"] == 1 and dst.capacity_log2() == 18
    dst.set_auto_grow(22)
    dst.merge_from(src)
    assert dst.capacity_log2() >= 19 and dst.stats()["frames"] == n
    o = O.Oracle(vs, T, W, H, seq.K)
    for d, R, t in fr:
        o.update(d, R, t)
    keys, pay = o.export()
    ka, pa = dst.export(sorted=True, raw=True)
    assert np.array_equal(ka, keys)
    assert np.abs(pa[:, 0] / pa[:, 4] - pay[:, 0]).max() <= 1e-4
    src.close(); dst.close()


# ---- the RCCL-typed path with PEERS ------------------------------------------------------------------
# gsdf_merge_allreduce(ctx, ncclComm_t) had only ever run with nranks = 1: RCCL refuses two ranks on one device and the pool
# offers one GPU per call.  tests/fake_rccl.c is a test double for the seven nccl* entry points libgsdf resolves with
# dlsym(RTLD_DEFAULT, ...): N processes sharing device 0, shared-memory rendezvous, device buffers staged on the stream they
# were given.  With it the code that talks to RCCL -- rccl_transport (ncclAllGather of ncclInt8, ncclAllReduce of ncclFloat32
# and of ncclUint32 for vis_), the header / token scheme, the agree() points, growth to the largest rank's capacity -- runs
# with 2 and 8 ranks, uneven shards and vis_ on, against the oracle.  What it does NOT test is RCCL itself or xGMI.

def _build_fake_rccl():
    src = os.path.join(ROOT, "tests", "fake_rccl.c")
    out = os.path.join(ROOT, "tests", "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               src, "-o", out, "-lrt"])
    return out


def test_fake_rccl_builds_and_exports_what_libgsdf_resolves():
    """(CPU) the test double compiles against <rccl/rccl.h> and exports the seven entry points gsdf_merge.hip:50-73 looks up."""
    out = _build_fake_rccl()
    syms = subprocess.run(["nm", "-D", "--defined-only", out], capture_output=True, text=True, check=True).stdout
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclAllGather", "ncclAllReduce",
                 "ncclGetErrorString"):
        assert (" T " + name) in syms, name


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_merge_allreduce_rccl_path_with_peers(pkg, O, tmp_path, world):
    """gsdf_merge_allreduce(ctx, comm) called by `world` processes over the RCCL test double: every rank ends with the oracle's
    key set, sums within float noise, the oracle's vis_ bit-vectors (frame f of rank r = integrated frame lo(r) + f) and
    Sdf::counter_ = 19; all tables have the largest rank's capacity; a second exchange is refused on every rank."""
    import rccl_rank as RR
    fake = _build_fake_rccl()
    id_file = str(tmp_path / "unique_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank.py"), str(r), str(world), id_file, str(tmp_path), fake],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    seq = pkg.synth.Sequence("spheres", RR.W, RR.H, n_frames=RR.N_FRAMES, seed=RR.SEED, step_deg=RR.STEP_DEG)
    vs = np.float32(RR.VS)
    o = O.Oracle(vs, np.float32(RR.TRUNC) * vs, RR.W, RR.H, seq.K)
    for i in range(RR.N_FRAMES):
        o.update(*seq.frame(i))
    keys, pay = o.export()
    vo = o.export_vis(1)
    z0 = np.load(tmp_path / "rccl_0.npz")
    shards = []
    for r in range(world):
        z = np.load(tmp_path / ("rccl_%d.npz" % r))
        shards.append((int(z["lo"]), int(z["hi"])))
        assert np.array_equal(z["keys"], keys) and int(z["frames"]) == RR.N_FRAMES == o.frame_counter()
        assert np.array_equal(z["vis"], vo)                                  # the ncclUint32 all-reduce
        assert np.array_equal(z["pay"].view(np.uint32), z0["pay"].view(np.uint32))   # every rank holds the same sums, bit for bit
        w = z["pay"][:, 4]
        assert (np.abs(w - pay[:, 4]) / np.maximum(1.0, pay[:, 4])).max() <= 1e-4
        assert np.abs(z["pay"][:, 0] / w - pay[:, 0]).max() <= 1e-4
        assert (np.abs(z["pay"][:, 1:4] - pay[:, 1:4]).max(axis=1) / np.maximum(1.0, pay[:, 4])).max() <= 1e-4
        assert int(z["own"]) < len(keys)                                     # the shards really differed
        assert int(z["cap"]) == 19                                           # grown to the largest rank's capacity
        assert int(z["n_blocks"]) == int(z0["n_blocks"]) > 50
        assert "one-shot" in str(z["again"])
    assert shards[0][0] == 0 and shards[-1][1] == RR.N_FRAMES and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    # what went through the double: all-gathers (header, key arrays, agreements), ONE float all-reduce, ONE unsigned one (vis_)
    ag, ar_f32, ar_u32, total = [int(v) for v in z0["stats"]]
    assert ag >= 2 and ar_f32 == 1 and ar_u32 == 1 and total >= int(z0["nbytes"])


# ---- BASELINE configs[3] AS CONFIGURED (640x480, 1 cm voxels, trunc 10, sphere orbit): shards -> ONE exchange -> mesh ------

C4_W, C4_H, C4_RANKS, C4_PER_RANK, C4_CAP = 640, 480, 4, 32, 23
C4_STEP_DEG = 360.0 * 4 / 2000                      # the 2000-frame job's orbit (tools/run_c4.py, bench.py)


def _c4_worker(rank, world, port, out_dir):
    """One rank of C4 at full frame size: 32 consecutive frames of the sphere orbit fused with the ground-truth poses
    (main_scan_3d.cpp:250-254) through the pipelined entry, then the C-ABI exchange (gloo callbacks: the ranks share one GPU)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import hashlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = world * C4_PER_RANK
    seq = pkg.synth.Sequence("spheres", C4_W, C4_H, n_frames=n, seed=0, step_deg=C4_STEP_DEG)
    vs = np.float32(0.01)
    g = pkg.GradSdf(vs, np.float32(10) * vs, C4_W, C4_H, seq.K, capacity_log2=C4_CAP, device=0)
    g.enable_vis(n)
    lo, hi = pkg.parallel.shard_range(n, rank, world)
    fr = [seq.frame(i) for i in range(lo, hi)]
    dev = [g.upload(f[0]) for f in fr]
    for d, f in zip(dev, fr):
        g.update_dev(d, f[1], f[2])

    def allgather(send):
        t = torch.from_numpy(send)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).numpy()

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
        return t.numpy()

    own = g.count()
    nb, nbytes = g.merge_allreduce_with(allgather, allreduce, world)
    keys, pay = g.export(sorted=True, raw=True)
    kv, vis = g.export_vis()
    digest = hashlib.sha256(keys.tobytes() + pay.tobytes() + vis.tobytes()).hexdigest()
    out = dict(own=own, nb=nb, nbytes=nbytes, frames=g.stats()["frames"], digest=digest, n=len(keys))
    if rank == 0:
        out.update(keys=keys, pay=pay, vis=vis, tris=g.extract_mesh())
    np.savez(os.path.join(out_dir, "c4_rank%d.npz" % rank), **out)
    g.close()
    dist.destroy_process_group()


def _tri_cells(tris, vs):
    """cube cell of every triangle (a marching-cubes triangle lies inside the cube that emitted it)"""
    c = np.floor(tris.reshape(-1, 3, 3).mean(axis=1) / vs - 1e-3).astype(np.int64)
    return c


@pytest.mark.gpu
def test_c4_as_configured_shards_exchange_mesh(pkg, O, tmp_path):
    """BASELINE configs[3] at its frame size and voxel size: four ranks x 32 frames of the sphere orbit (640x480, 1 cm, trunc 10,
    ground-truth poses) -> gsdf_merge_allreduce_with -> every rank holds the map ONE context gets from all 128 frames: key set
    bit-exact, sums within float re-association (1e-5 relative; the shards add 32 frames each and then the four sums, the single
    context adds 128 frames in a row), vis_ bit-vectors and Sdf::counter_ equal; and the marching-cubes mesh of the merged map
    is the mesh of the single map: the same cubes emit triangles (the ones whose corner distance sits within rounding of the
    iso value may differ: < 0.1 %), vertices within 1e-5 m.  (Bit-for-bit equality of the meshes is not defined: the vertex
    positions are ratios of sums whose last bits depend on the order of addition.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_c4_worker, args=(C4_RANKS, port, str(tmp_path)), nprocs=C4_RANKS, join=True)
    z = [np.load(tmp_path / ("c4_rank%d.npz" % r)) for r in range(C4_RANKS)]
    n = C4_RANKS * C4_PER_RANK
    assert len({str(a["digest"]) for a in z}) == 1                                 # every rank ends with the same map, bit for bit
    assert all(int(a["frames"]) == n for a in z)
    assert all(int(a["own"]) < int(a["n"]) for a in z)                              # the shards really differed
    assert int(z[0]["nb"]) > 5000 and int(z[0]["nbytes"]) == int(z[0]["nb"]) * 64 * (5 * 4 + ((n + 31) // 32) * 4)
    # the single context
    seq = pkg.synth.Sequence("spheres", C4_W, C4_H, n_frames=n, seed=0, step_deg=C4_STEP_DEG)
    vs = np.float32(0.01)
    g = pkg.GradSdf(vs, np.float32(10) * vs, C4_W, C4_H, seq.K, capacity_log2=C4_CAP)
    g.enable_vis(n)
    for i in range(n):
        d, R, t = seq.frame(i)
        g.update_dev(g.upload(d), R, t)
    keys, pay = g.export(sorted=True, raw=True)
    kv, vis = g.export_vis()
    tris = g.extract_mesh()
    assert g.stats()["frames"] == n
    g.close()
    a = z[0]
    assert len(keys) > 200000
    assert np.array_equal(a["keys"], keys)                                          # occupancy: bit-exact
    scale = np.maximum(1.0, np.abs(pay[:, 4:5]))
    assert (np.abs(a["pay"] - pay) / scale).max() <= 1e-5
    assert np.array_equal(a["vis"], vis)
    # ... and the ORACLE on the same 128 full-size frames (the full-size exchange is held to the oracle, not only
    # to another GPU context): key set bit-exact, distance <= 1e-4, weight / gradient sums <= 1e-4 relative to the weight
    # (the rule of test_gpu_parity._cmp_tables), vis_ bit-vectors and the frame counter equal.  ~1 min on one core.
    o = O.Oracle(vs, np.float32(10) * vs, C4_W, C4_H, seq.K)
    for i in range(n):
        d, R, t = seq.frame(i)
        o.update(d, R, t)
    ko, po = o.export()
    assert np.array_equal(a["keys"], ko) and o.frame_counter() == n
    w = a["pay"][:, 4]
    osc = np.maximum(1.0, po[:, 4])
    assert np.abs(a["pay"][:, 0] / w - po[:, 0]).max() <= 1e-4
    assert (np.abs(w - po[:, 4]) / osc).max() <= 1e-4
    assert (np.abs(a["pay"][:, 1:4] - po[:, 1:4]).max(axis=1) / osc).max() <= 1e-4
    assert np.array_equal(a["vis"], o.export_vis((n + 31) // 32))
    # the meshes
    ta, tb = a["tris"], tris
    assert len(tb) > 20000 and abs(len(ta) - len(tb)) <= 1e-3 * len(tb)
    ca, cb = _tri_cells(ta, float(vs)), _tri_cells(tb, float(vs))
    pack = lambda c: ((c[:, 0] + (1 << 20)) << 42) | ((c[:, 1] + (1 << 20)) << 21) | (c[:, 2] + (1 << 20))
    ka, kb = pack(ca), pack(cb)
    ua, na = np.unique(ka, return_counts=True)
    ub, nb_ = np.unique(kb, return_counts=True)
    common, ia, ib = np.intersect1d(ua, ub, return_indices=True)
    same_count = common[na[ia] == nb_[ib]]
    assert len(same_count) >= (1 - 1e-3) * max(len(ua), len(ub))                   # the same cubes emit the same number of triangles
    # cubes with exactly one triangle in both meshes: the vertices agree (both lists are in sweep order)
    one = np.intersect1d(ua[na == 1], ub[nb_ == 1])
    sa = np.flatnonzero(np.isin(ka, one)); sb = np.flatnonzero(np.isin(kb, one))
    sa = sa[np.argsort(ka[sa], kind="stable")]; sb = sb[np.argsort(kb[sb], kind="stable")]
    assert len(sa) == len(sb) > 1000
    dv = np.abs(ta[sa].reshape(-1, 9) - tb[sb].reshape(-1, 9)).max(axis=1)
    assert np.quantile(dv, 0.999) <= 1e-5 and (dv > 1e-4).mean() <= 1e-3


# ---- bench.py --gpus N: the launch path ---------------------------------------------------------------------------------

def _run_bench(extra, timeout=900):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_n_starts_n_ranks_dry():
    """`python bench.py --gpus N` without a launcher starts N rank processes itself (the driver's command shape); each rank
    checks WORLD_SIZE == --gpus, they rendezvous, and only rank 0 prints."""
    out = _run_bench(["--gpus", "3", "--dry-run"], timeout=300)
    assert out == {"dry_run": True, "n_gpus": 3, "max_over_ranks": 3.0, "local_rank": 0}


def test_bench_names_resolve():
    """bench.py is only executed end to end on the GPU box; here: every plain name it CALLS is defined somewhere in the file
    (a function lost in an edit would otherwise only show up as an error string inside the JSON line)."""
    import ast
    import builtins
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    defined = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defined.add(node.name)
            if isinstance(node, ast.FunctionDef):
                a = node.args
                defined.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
            defined.add(node.id)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            defined.update((al.asname or al.name).split(".")[0] for al in node.names)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            defined.add(node.name)
    called = {n.func.id for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)}
    assert not (called - defined), sorted(called - defined)
    assert {"main", "sharded_flavour", "spawn_ranks", "render_frames", "parse_args"} <= defined


def test_bench_rejects_world_size_mismatch():
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu(pkg):
    """The N > 1 path of bench.py end to end on a one-GPU box: two self-started ranks share device 0, replicas of the tracked
    stream + the sharded GT-pose flavour with ONE exchange (gsdf_merge_allreduce_with over gloo: RCCL refuses two ranks on one
    device), one JSON line from rank 0."""
    out = _run_bench(["--gpus", "2", "--single-device", "--dist-backend", "gloo", "--steps", "6", "--warmup", "3", "--repeats", "2",
                      "--c4-frames", "12", "--raycast-reps", "2", "--cpu-frames", "0"])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["config"]["value_runs"]) == 2
    sh = out["config"]["sharded"]
    assert "error" not in sh, sh
    assert sh["ranks"] == 2 and sh["frames_total"] == 24 and sh["frames_counter_after_merge"] == 24
    assert sh["exchange_blocks"] > 100 and sh["exchange_bytes"] == sh["exchange_blocks"] * 64 * 5 * 4
    assert sh["voxels_merged"] > sh["voxels_own_shard"] and sh["mesh_faces"] > 1000
    assert out["roofline"]["raycast"]["avg_launch_us"] > 0 and out["roofline"]["raycast"]["hit_fraction"] > 0.5
    assert out["cpu_baseline"] is None                               # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_gpus_8_end_to_end_over_the_rccl_double():
    """`python bench.py --gpus 8` had never run end to end anywhere.  On ONE GPU over the RCCL test double
    (TEST INFRASTRUCTURE: LD_PRELOAD=tests/libfake_rccl.so and HIP_VISIBLE_DEVICES=0 in every rank, torch.distributed over gloo):
    bench.py starts its 8 ranks, they rendezvous, run the tracked replica windows with the max-over-ranks timing, exchange the
    unique id, gsdf_rccl_comm_init x 8, gsdf_merge_prepare, fuse their shards on two contexts each, gsdf_merge_allreduce with 8
    peers, mesh on rank 0, ONE JSON line from rank 0 -- labelled as the double so that it can never pass for a scaling number."""
    fake = _build_fake_rccl()
    out = _run_bench(["--gpus", "8", "--steps", "10", "--warmup", "3", "--repeats", "2", "--c4-frames", "16", "--rccl-double", fake,
                      "--raycast-reps", "0", "--no-staged", "--rank-timeout", "800"], timeout=900)
    assert out["n_gpus"] == 8 and out["steps"] == 10 and out["value"] > 0 and out["scaling"] == "weak"
    assert "ONE GPU" in out["config"]["parallelism"]
    sh = out["config"]["sharded"]
    assert sh is not None and sh["ranks"] == 8 and sh["rccl_ranks"] == 8 and sh["contexts_per_gpu"] == 2
    assert sh["transport"].startswith("rccl test double")
    assert sh["same_map_as_one_context"] is True
    assert sh["frames_total"] == 8 * 16 and sh["frames_counter_after_merge"] == 8 * 16
    assert sh["voxels_merged"] > sh["voxels_own_shard"] and sh["exchange_blocks"] > 1000 and sh["mesh_faces"] > 1000
    assert out["cpu_baseline"] is None or out["n_gpus"] == 8          # the CPU baseline is an N = 1 item
