#!/usr/bin/env python3
"""BASELINE config C4: frame-sharded GT-pose fusion of a synthetic sphere stream with one RCCL exchange -- by default the
all-reduce of per-voxel sums over the union of the ranks' 4x4x4 blocks (--exchange allgather: all-gather of the (key, raw
sums) lists + additive merge) -- then the marching-cubes export on the device.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_c4.py --frames 2000

One process per GPU; with N = 1 the exchange is skipped.  Prints one JSON line on rank 0."""
import argparse, ctypes, gc, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--out", default="/tmp/c4_mesh.ply")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--exchange", default="c-abi", choices=["c-abi", "allreduce", "allgather"],
                    help="c-abi: gsdf_merge_allreduce (pack -> ncclAllReduce -> unpack inside libgsdf, communicator created before "
                         "the timed exchange); allreduce / allgather: the torch.distributed harness of parallel.py")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange also with one rank (degenerate, for testing)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    import __graft_entry__ as graft
    pkg = graft.package()
    W, H = 640, 480
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=args.frames, seed=0, step_deg=360.0 * 4 / args.frames)
    vs = np.float32(0.01); T = np.float32(10) * vs
    lo, hi = pkg.parallel.shard_range(args.frames, rank, world)
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=23, device=local)
    chunk, t_fuse = 64, 0.0
    # 64 staging buffers, allocated ONCE and refilled chunk by chunk.  (Allocating and freeing them per chunk -- what this tool
    # did until round 4 -- makes every other chunk take 15-30 ms instead of 2.7 with the libraries built since the exchange
    # links rocPRIM's select / sort (bisected: the same sources without those two kernels' code object do not show it, nor does
    # the round-3 library; no rocPRIM kernel has run at that point): the stall is on the device side of the first
    # synchronisation after the re-allocation, with no fusion time-outs or deferred entries -- freeing and mapping 77 MB of
    # device memory per chunk is at the mercy of where the runtime's allocator places it.  Buffers that stay do not show it.)
    nbytes = W * H * 4
    slots = []
    for _ in range(chunk):
        p = ctypes.c_void_p()
        g._chk(g.L.gsdf_dev_alloc(g.h, ctypes.byref(p), nbytes)); g._dev.append(p); slots.append(p)
    for c0 in range(lo, hi, chunk):                      # stage frames in HBM chunk by chunk
        fr = [seq.frame(i) for i in range(c0, min(c0 + chunk, hi))]
        dev = slots[:len(fr)]
        for p, f in zip(dev, fr):
            a = np.ascontiguousarray(f[0], np.float32)
            g._chk(g.L.gsdf_dev_upload(g.h, p, a.ctypes.data_as(ctypes.c_void_p), nbytes))
        gc.collect(); gc.disable()     # (a generation-2 collection of this process takes 30-40 ms and used to land inside chunks)
        g.sync(); t0 = time.perf_counter()
        worst = (0.0, "")
        for j, (d, f) in enumerate(zip(dev, fr)):
            ta = time.perf_counter()
            g.update_dev(d, f[1], f[2])
            tb = time.perf_counter()
            if j % 32 == 31:
                g.sync()             # as bench.py does: long unsynchronised runs of launches make the HIP runtime stall the host (10-40 ms, sporadically)
            tc = time.perf_counter()
            worst = max(worst, (tb - ta, "update_dev %d" % j), (tc - tb, "sync %d" % j))
        if os.environ.get("GSDF_C4_DEBUG"):
            print("   slowest call: %s %.2f ms" % (worst[1], worst[0] * 1e3), file=sys.stderr)
        g.sync(); t_fuse += time.perf_counter() - t0
        gc.enable()
        if os.environ.get("GSDF_C4_DEBUG"):
            st = g.stats()
            print("chunk at frame %d: %.2f ms for %d frames | timeouts %d deferred %d" % (c0, (time.perf_counter() - t0) * 1e3, len(dev), st["fuse_timeouts"], st["n_deferred"]), file=sys.stderr, flush=True)
    exchanged = 0
    comm = None
    if (world > 1 or args.force_exchange) and args.exchange == "c-abi":
        # communicator set-up is NOT part of the exchange time: rank 0's RCCL id travels through torch.distributed
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            idt = torch.frombuffer(bytearray(pkg.binding.rccl_unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            idt = idt.cuda() if args.dist_backend == "nccl" else idt
            dist.broadcast(idt, src=0)
        comm = pkg.binding.rccl_comm_init(world, bytes(idt.cpu().numpy().tobytes()), rank, local)
        g.merge_prepare(world)               # like the communicator: set-up outside the timed exchange
        if world > 1:
            dist.barrier()
    t0 = time.perf_counter()
    if world > 1 or args.force_exchange:
        if args.exchange != "c-abi" and world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(args.dist_backend, rank=0, world_size=1)
        if args.exchange == "c-abi":
            nb, nbytes = g.merge_allreduce_rccl(comm)
            exchanged = nb * 64
        elif args.exchange == "allreduce":
            exchanged = pkg.parallel.allreduce_merge(pkg.parallel.GpuBlockOps(g, 23), dist, device="cuda") * 64
        else:
            exchanged = pkg.parallel.exchange_and_merge(g, dist)
    t_merge = time.perf_counter() - t0
    if comm is not None:
        pkg.binding.rccl_comm_destroy(comm)
    if world > 1:
        tt = torch.tensor([t_fuse, t_merge], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_fuse, t_merge = [float(v) for v in tt.tolist()]
    if rank == 0:
        hl = ctypes.CDLL(os.path.join(ROOT, "gradient-sdf_amd", "host", "libgsdf_host.so"))
        hl.gsdf_host_extract_mesh.restype = ctypes.c_long
        hl.gsdf_host_extract_mesh.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_char_p]
        t0 = time.perf_counter()
        faces = hl.gsdf_host_extract_mesh(g.h, vs, args.out.encode())
        print(json.dumps({"config": "C4 frame-sharded fusion", "n_gpus": world, "frames": args.frames,
                          "fused_fps_total": round(args.frames / t_fuse, 1), "fuse_s": round(t_fuse, 3),
                          "exchange": args.exchange, "merge_s": round(t_merge, 4), "exchanged_voxels": exchanged, "voxels": g.count(),
                          "mesh_faces": faces, "mesh_s": round(time.perf_counter() - t0, 2), "mesh": args.out}))
    g.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
