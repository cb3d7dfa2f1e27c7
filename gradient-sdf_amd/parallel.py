"""Frame-sharded Gradient-SDF fusion across the GPUs of one node (BASELINE config C4).

`MapGradPixelSdf::update` with a KNOWN pose depends only on (depth, pose) and mutates the map
additively (W = sum w, S = sum w d, G = sum w R n; key set = union), so frames are independent units:
each rank (one process per GPU, torch.distributed over RCCL/xGMI) fuses a contiguous frame range into
its private table and ONE exchange step follows, after which every rank holds the full map.  Two forms:
  * `allreduce_merge`  -- the all-reduce of per-voxel sums named in BASELINE.json: the map is a hash of 4x4x4 voxel
    blocks; the ranks all-gather + unique their block ids, pack 64 x 5 raw sums per block of the union into one dense
    device buffer (zeros where a rank has nothing), all-reduce it (sum) and store the result.  Volume per rank:
    1280 B per block of the union, independent of the number of ranks -- the form for overlapping shards (8 GPUs
    scanning one scene);
  * `exchange_and_merge` -- all-gather of the compacted (key, raw sums) lists (32 B per occupied voxel per rank) and a
    local additive merge: cheaper when the shards barely overlap.
The tracked path cannot be sharded (frame i needs the map of all frames < i): replicas only.

All functions work on CPU tensors under gloo (used by the world-size-2 tests) and on device tensors
under nccl (= RCCL on ROCm).  This module contains no hot-path arithmetic: the GPU work goes through the
C-ABI (gsdf_export_raw_dev / gsdf_merge_raw_dev).  The C++ form of the same exchange is gsdf_merge_allreduce
(include/gsdf.h: pack -> ncclAllReduce -> unpack on the context's own stream); this module is the torch.distributed
harness around the same device kernels.

Stream ordering: libgsdf works on its own non-blocking HIP stream, torch (and the nccl backend, whose collectives return
before they finish) on torch's current stream.  Every hand-over between the two is therefore fenced with
`_torch_done()` (torch -> libgsdf) -- the libgsdf entries used here synchronise their stream before they return.
"""
import numpy as np


def _torch_done(device):
    """Wait until everything torch has queued on its current stream (fills, copies, RCCL collectives) is complete, so that
    libgsdf -- which runs on its own stream -- may read or overwrite the buffers."""
    if str(device) != "cpu":
        import torch
        torch.cuda.current_stream().synchronize()


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
    lists = [(k.contiguous(), p.contiguous()) for k, p in lists]
    _torch_done("cuda")                        # the all-gathers (and the copies above) have landed
    for r, (k, p) in enumerate(lists):
        if r == rank or k.shape[0] == 0:
            continue
        g.merge_raw_dev(k.data_ptr(), p.data_ptr(), k.shape[0])
    torch.cuda.synchronize()
    return sum(k.shape[0] for k, _ in lists)


# ---- dense block all-reduce ------------------------------------------------------------------------------------
BLOCK_VOX = 64            # voxels per 4x4x4 block (csrc/gsdf_table.h)


def union_block_keys(local_keys, dist, device="cpu"):
    """All-gather the ranks' block-id lists (int64 tensors of different lengths) and return the sorted union --
    identical on every rank."""
    import torch
    world = dist.get_world_size()
    local_keys = torch.as_tensor(local_keys, dtype=torch.int64, device=device).reshape(-1)
    n = torch.tensor([local_keys.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros(m, dtype=torch.int64, device=device)
    pad[:local_keys.shape[0]] = local_keys
    allk = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allk, pad)
    return torch.unique(torch.cat([allk[r][:counts[r]] for r in range(world)]))


def allreduce_merge(ops, dist, device="cpu"):
    """One all-reduce of per-voxel (weight, weighted distance, weighted gradient) over the union of the ranks' blocks.
    `ops` provides block_keys() -> int64 tensor, pack(union) -> float32 tensor [n, 64, 5], unpack(union, dense)
    (GpuBlockOps below for a binding.GradSdf; NumpyBlockOps for the CPU tests).  Returns the number of blocks."""
    union = union_block_keys(ops.block_keys(), dist, device=device).contiguous()
    _torch_done(device)                          # the union is complete before libgsdf reads it
    dense = ops.pack(union)                      # libgsdf's stream is synchronised when pack returns
    dist.all_reduce(dense)                       # sum; RCCL over xGMI under the nccl backend
    _torch_done(device)                          # the reduction has finished before the sums are stored
    ops.unpack(union, dense)
    return int(union.shape[0])


class GpuBlockOps:
    """Block exchange primitives of a binding.GradSdf on device tensors (gsdf_block_keys_dev / _pack_ / _unpack_)."""

    def __init__(self, g, capacity_log2):
        self.g = g
        self.max_blocks = 1 << (int(capacity_log2) - 6)

    def block_keys(self):
        import torch
        buf = torch.empty(self.max_blocks, dtype=torch.int64, device="cuda")
        n = self.g.block_keys_dev(buf.data_ptr(), self.max_blocks)
        return buf[:n]

    def pack(self, union):
        import torch
        union = union.contiguous()
        # empty, not zeros: the pack kernel writes every element, and a fill queued on torch's stream could land after it
        dense = torch.empty((union.shape[0], BLOCK_VOX, 5), dtype=torch.float32, device="cuda")
        _torch_done("cuda")
        self.g.pack_blocks_dev(union.data_ptr(), union.shape[0], dense.data_ptr())
        return dense

    def unpack(self, union, dense):
        import torch
        union = union.contiguous()
        self.g.unpack_blocks_dev(union.data_ptr(), union.shape[0], dense.contiguous().data_ptr())
        torch.cuda.synchronize()


def block_id(keys):
    """Block id (csrc/gsdf_table.h gsdf_block_key) and in-block index of voxel keys int[n,3]."""
    k = np.asarray(keys, np.int64).reshape(-1, 3)
    b = (k >> 2) + (1 << 18)
    bid = b[:, 0] | (b[:, 1] << 19) | (b[:, 2] << 38)
    local = (k[:, 0] & 3) | ((k[:, 1] & 3) << 2) | ((k[:, 2] & 3) << 4)
    return bid, local


class NumpyBlockOps:
    """The same three primitives on a host (keys, raw payload) table: the CPU stand-in used by the gloo tests."""

    def __init__(self, keys, payload):
        self.keys = np.asarray(keys, np.int32).reshape(-1, 3)
        self.pay = np.asarray(payload, np.float32).reshape(-1, 5)      # raw sums: s, gx, gy, gz, w  (export order)

    def block_keys_numpy(self):
        bid, _ = block_id(self.keys)
        return np.unique(bid)

    def block_keys(self):
        import torch
        return torch.from_numpy(self.block_keys_numpy())

    def pack_numpy(self, u):
        """dense float32 [len(u), 64, 5] (w, s, gx, gy, gz) of the blocks listed in the sorted id array `u`."""
        dense = np.zeros((len(u), BLOCK_VOX, 5), np.float32)
        bid, local = block_id(self.keys)
        row = np.searchsorted(u, bid)
        dense[row, local] = self.pay[:, [4, 0, 1, 2, 3]]                 # dense order: w, s, gx, gy, gz
        return dense

    def pack(self, union):
        import torch
        return torch.from_numpy(self.pack_numpy(union.numpy()))

    def unpack(self, union, dense):
        u = union.numpy()
        d = dense.numpy()
        b, l = np.nonzero(d[:, :, 0] > 0)                                # a voxel exists iff w > 0
        bx = (u[b] & 0x7FFFF) - (1 << 18)
        by = ((u[b] >> 19) & 0x7FFFF) - (1 << 18)
        bz = ((u[b] >> 38) & 0x7FFFF) - (1 << 18)
        keys = np.stack([bx * 4 + (l & 3), by * 4 + ((l >> 2) & 3), bz * 4 + ((l >> 4) & 3)], 1).astype(np.int32)
        pay = d[b, l][:, [1, 2, 3, 4, 0]].astype(np.float32)
        order = np.lexsort((keys[:, 0], keys[:, 1], keys[:, 2]))
        self.keys, self.pay = keys[order], pay[order]
