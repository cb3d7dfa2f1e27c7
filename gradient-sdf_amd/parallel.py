"""Frame-sharded Gradient-SDF fusion across the GPUs of one node (BASELINE config C4).

`MapGradPixelSdf::update` with a KNOWN pose depends only on (depth, pose) and mutates the map
additively (W = sum w, S = sum w d, G = sum w R n; key set = union), so frames are independent units:
each rank (one process per GPU, torch.distributed over RCCL/xGMI) fuses a contiguous frame range into
its private table and ONE exchange step follows -- an all-gather of the compacted (key, raw sums)
lists (32 B per occupied voxel) and a local additive merge, after which every rank holds the full map.
The tracked path cannot be sharded (frame i needs the map of all frames < i): replicas only.

All functions work on CPU tensors under gloo (used by the world-size-2 tests) and on device tensors
under nccl (= RCCL on ROCm).  This module contains no hot-path arithmetic: the GPU work goes through the
C-ABI (gsdf_export_raw_dev / gsdf_merge_raw_dev).
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of `rank`; the ranges tile [0, n_frames) exactly."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_lists(keys, payload, dist, device="cpu"):
    """All-gather variable-length (keys int32[n,3], payload float32[n,5]) lists.
    Returns a list over ranks of (keys_r, payload_r) torch tensors on `device`."""
    import torch
    world = dist.get_world_size()
    keys = torch.as_tensor(keys, dtype=torch.int32, device=device).reshape(-1, 3)
    payload = torch.as_tensor(payload, dtype=torch.float32, device=device).reshape(-1, 5)
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    kp = torch.zeros((m, 3), dtype=torch.int32, device=device)
    pp = torch.zeros((m, 5), dtype=torch.float32, device=device)
    kp[:keys.shape[0]] = keys
    pp[:payload.shape[0]] = payload
    kall = [torch.empty_like(kp) for _ in range(world)]
    pall = [torch.empty_like(pp) for _ in range(world)]
    dist.all_gather(kall, kp)          # two collectives of 12 B and 20 B per voxel
    dist.all_gather(pall, pp)
    return [(kall[r][:counts[r]], pall[r][:counts[r]]) for r in range(world)]


def merge_numpy(tables):
    """Host reference of the additive merge: reduce-by-key over (keys, raw payload) lists.
    Returns (keys sorted by (z,y,x), summed payload)."""
    keys = np.concatenate([np.asarray(k, np.int64).reshape(-1, 3) for k, _ in tables])
    pay = np.concatenate([np.asarray(p, np.float64).reshape(-1, 5) for _, p in tables])
    off = 1 << 20
    packed = (keys[:, 0] + off) | ((keys[:, 1] + off) << 21) | ((keys[:, 2] + off) << 42)
    uniq, inv = np.unique(packed, return_inverse=True)
    out = np.zeros((len(uniq), 5))
    np.add.at(out, inv, pay)
    k = np.stack([(uniq & 0x1FFFFF) - off, ((uniq >> 21) & 0x1FFFFF) - off, ((uniq >> 42) & 0x1FFFFF) - off], 1)
    return k.astype(np.int32), out.astype(np.float32)


def exchange_and_merge(g, dist):
    """GPU path: all-gather every rank's (key, raw sums) list over RCCL and merge the other ranks'
    lists into this rank's table `g` (a binding.GradSdf).  Afterwards all ranks hold the full map."""
    import torch
    rank = dist.get_rank()
    n = g.count()
    keys = torch.empty((max(n, 1), 3), dtype=torch.int32, device="cuda")
    pay = torch.empty((max(n, 1), 5), dtype=torch.float32, device="cuda")
    got = g.export_raw_dev(keys.data_ptr(), pay.data_ptr(), n) if n else 0
    lists = allgather_lists(keys[:got], pay[:got], dist, device="cuda")
    for r, (k, p) in enumerate(lists):
        if r == rank or k.shape[0] == 0:
            continue
        k = k.contiguous()
        p = p.contiguous()
        g.merge_raw_dev(k.data_ptr(), p.data_ptr(), k.shape[0])
    torch.cuda.synchronize()
    return sum(k.shape[0] for k, _ in lists)
