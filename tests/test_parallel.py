"""Multi-GPU path on CPU: world-size-2 gloo processes run the frame-sharded fusion protocol
(shard -> fuse -> all-gather of (key, raw sums) lists -> additive merge) with the CPU oracle standing in
for the per-rank fusion, and must reproduce the single-process map.  The GPU path runs the same
exchange with device tensors over RCCL (gradient-sdf_amd/parallel.py:exchange_and_merge)."""
import os
import socket
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
