#!/usr/bin/env python3
"""bench.py -- depth frames/sec fused+tracked on MI355X (BASELINE.json metric).

A *step* = one pass of the hot path over one depth frame of the synthetic stream:
RigidPointOptimizer::optimize (<= 25 Gauss-Newton passes) + MapGradPixelSdf::update when it
converged (main_scan_3d.cpp:255-266), enqueued through the C-ABI entry gsdf_track_and_fuse_dev
with the frame already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
TUM fr1/xyz-format synthetic stream, 640x480, 1 cm voxels, trunc = 10 voxels, capacity 2^22.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment bench.py starts the N rank
processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set); under torch.distributed.run it is one of the
ranks.  Either way a rank asserts WORLD_SIZE == --gpus.

Multi-GPU, two flavours in one run:
  * fused+tracked (`value`): tracking makes frame i depend on the map of all frames < i, so the path does not shard --
    N independent replica streams (seed = rank), no data-path collective, `value` = frames of all ranks / max-over-ranks time;
  * `config.sharded` (BASELINE configs[3], the flavour that DOES shard): GT-pose fusion of contiguous frame shards of ONE
    sphere-orbit stream (main_scan_3d.cpp:250-254), `--c4-frames` per rank, then ONE gsdf_merge_allreduce (RCCL all-reduce of
    the per-voxel sums over the union of blocks) on a communicator created outside the timed region, then the mesh on rank 0.

Output: ONE JSON line on rank 0 (contract in the task statement) including
  value         median of `--repeats` timed windows (each: reset, frame 0, W warm-up steps, barrier, K timed steps, barrier);
                every run is listed in config.value_runs
  roofline      dominant kernel (k_fuse): algorithmic bytes per launch / mean HIP-event duration, measured live in a replay
                of the same frames; `traffic` (HBM bytes per launch from rocprofv3 PMC passes over THIS command, committed
                under profiles/ and named in `traffic_source`); `tracker` and `raycast`: the same for k_track_pass / k_raycast
  cpu_baseline  the CPU oracle ("port" of the reference's serial path) timed on the SAME frames as `value` (the warm-up frames
                are fused untimed first), plus its OMP-structured variants (4 threads as in the reference, and all host cores);
                rank 0 at N = 1 only

Two windows are worth knowing: the driver's `--steps 20 --warmup 5` covers frames 6..25 of the stream, which all
converge in 3-4 Gauss-Newton passes; the default 200 steps run into the stretch where the reference's tracker does not
converge (25 passes, frame not fused; tests/test_gpu_parity.py::test_tracked_bench_stream_matches_oracle_frame_by_frame).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md), ~6300 achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed windows; value = their median")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--voxel-size", type=float, default=0.01)
    ap.add_argument("--trunc", type=float, default=10.0)
    ap.add_argument("--hash-capacity-log2", type=int, default=22)
    ap.add_argument("--cpu-frames", type=int, default=8, help="timed frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--c4-frames", type=int, default=250,
                    help="frames PER RANK of the sharded GT-pose flavour (8 ranks x 250 = the 2000 frames of BASELINE configs[3]); 0 = skip")
    ap.add_argument("--raycast-reps", type=int, default=10, help="raycasts of the bench map timed for roofline.raycast (0 = skip)")
    ap.add_argument("--no-staged", action="store_true", help="skip the staging-inclusive flavour (config.staged_fps)")
    ap.add_argument("--contexts-per-gpu", type=int, default=2,
                    help="shard contexts per GPU in the sharded flavour (each on its own stream, merged locally before the exchange)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip config.default_window (frames 21..220 of the same stream) and config.c3 (BASELINE configs[2])")
    ap.add_argument("--extras-timeout", type=float, default=150.0, help="seconds the two extra configurations may take before the line is printed without them")
    ap.add_argument("--only-main", action="store_true", help="fused+tracked windows and the roofline replays only (profiling runs)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--single-device", action="store_true",
                    help="dry run of the N>1 code path on a 1-GPU box: every rank uses device 0, the exchange goes through "
                         "gsdf_merge_allreduce_with over torch.distributed (RCCL refuses two ranks on one device)")
    ap.add_argument("--rccl-double", default="", metavar="LIBFAKE_RCCL_SO",
                    help="TEST INFRASTRUCTURE: run the whole --gpus N line on ONE GPU with the RCCL test double (tests/libfake_rccl.so) "
                         "standing in for RCCL: every rank is started with LD_PRELOAD=<this> and HIP_VISIBLE_DEVICES=0, torch.distributed "
                         "uses gloo, and the sharded flavour goes through gsdf_rccl_comm_init / gsdf_merge_allreduce exactly as with the "
                         "real library.  The line is labelled `transport: rccl test double`; it is never a scaling number.")
    ap.add_argument("--no-next-hint", action="store_true", help="do not name the next frame ahead (gsdf_hint_next_depth_dev): A/B of that entry")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only (CPU test of --gpus N): the ranks rendezvous over gloo, barrier, max-reduce, rank 0 prints a stub line")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="seconds the sharded flavour may take before the line is printed without it")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="seconds after which the self-started ranks are stopped")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """--gpus N without a launcher: start N rank processes of this script (one per GPU) and wait for them.  The first rank that
    fails takes the others with it (exact pids), so a broken rank cannot leave its peers inside a collective forever."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
        if args.rccl_double:                                    # every rank on device 0, the double in front of any real RCCL
            env["LD_PRELOAD"] = os.path.abspath(args.rccl_double) + ((":" + env["LD_PRELOAD"]) if env.get("LD_PRELOAD") else "")
            env["HIP_VISIBLE_DEVICES"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.time() + args.rank_timeout
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:                                  # exact pids, never a pattern
                    q.terminate()
        if live and time.time() > deadline:
            print("bench.py: ranks still running after %.0f s, stopping them" % args.rank_timeout, file=sys.stderr)
            for q in live:
                q.kill()
            worst = worst or 124
        time.sleep(0.05)
    return worst


def render_frames(kind, W, H, indices, seed, **kw):
    """Synthetic frames (numpy renderer, ~50 ms each) on a few forked workers -- called before torch / HIP are touched."""
    import __graft_entry__ as graft
    pkg = graft.package()
    seq = pkg.synth.Sequence(kind, W, H, seed=seed, **kw)
    indices = list(indices)
    workers = max(1, min(8, (os.cpu_count() or 8) // max(1, int(os.environ.get("WORLD_SIZE", "1"))), len(indices)))
    if workers > 1 and "torch" not in sys.modules:
        try:
            import multiprocessing as mp
            pool = mp.get_context("fork").Pool(workers)
            try:
                frames = pool.map(seq.frame, indices, chunksize=max(1, len(indices) // (4 * workers)))
            finally:
                # close + join, not the context manager's terminate(): under rocprofv3 a worker that gets SIGTERM runs the
                # profiler's signal handler and may never exit -- the parent then waits for it for good (seen once in a PMC pass)
                pool.close()
                pool.join()
            return seq, frames
        except Exception:                                       # noqa: BLE001 -- any pool trouble: render serially
            pass
    return seq, [seq.frame(i) for i in indices]


NEXT_HINT = True            # --no-next-hint: A/B of gsdf_hint_next_depth_dev


def track_fuse(g, dev, i, last):
    """One step: gsdf_track_and_fuse_dev of frame i.  The frames are resident, so the next one is known: it is named first
    (gsdf_hint_next_depth_dev) and its normals are computed in the tail of this frame's fusion launch instead of beside the
    next frame's tracker passes -- time only, same poses and map (tests/test_gpu_parity.py::test_next_depth_hint_is_invisible_except_in_time).
    `last`: the last frame index of the loop (no hint beyond it)."""
    if NEXT_HINT and i < last:
        g.track_and_fuse_ahead_dev(dev[i], dev[i + 1])          # gsdf_hint_next_depth_dev + gsdf_track_and_fuse_dev in one call
    else:
        g.track_and_fuse_dev(dev[i])


def quat_to_R(q):
    x, y, z, w = [np.float32(v) for v in q]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    return np.array([[1 - (ty * y + tz * z), ty * x - tz * w, tz * x + ty * w],
                     [ty * x + tz * w, 1 - (tx * x + tz * z), tz * y - tx * w],
                     [tz * x - ty * w, tz * y + tx * w, 1 - (tx * x + ty * y)]], np.float32)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    if args.single_device:
        local_rank = 0
    if args.rccl_double:
        # the ranks share device 0; torch's own collectives (barrier, max over ranks, the unique-id broadcast) go over gloo -- the
        # preloaded double would otherwise stand in for torch's RCCL as well, and it only has the seven entries libgsdf resolves
        local_rank = 0
        args.dist_backend = "gloo"
        if world > 1 and os.path.abspath(args.rccl_double) not in os.environ.get("LD_PRELOAD", ""):
            print("bench.py: --rccl-double needs LD_PRELOAD=%s in every rank (bench.py sets it when it starts the ranks itself)" % args.rccl_double, file=sys.stderr)
            sys.exit(2)
    if args.dry_run:
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": float(t.item()), "local_rank": local_rank}))
        return
    global NEXT_HINT
    NEXT_HINT = not args.no_next_hint
    if args.only_main:
        args.c4_frames = 0
        args.raycast_reps = 0
        args.cpu_frames = 0
        args.no_staged = True
        args.no_extras = True

    W, H = args.width, args.height
    K, Wm = args.steps, args.warmup
    n_frames = 1 + Wm + K
    vs = np.float32(args.voxel_size)
    T = np.float32(args.trunc) * vs
    # ---- synthetic inputs first (forked render workers must not inherit a HIP runtime) ---------------------------
    seq, frames = render_frames("tum", W, H, range(n_frames), seed=rank, n_frames=n_frames)
    c4 = None
    if args.c4_frames > 0:
        F = args.c4_frames
        total = F * world
        c4_seq, c4_frames = render_frames("spheres", W, H, range(rank * F, (rank + 1) * F), seed=0, n_frames=total,
                                          step_deg=360.0 * 4 / 2000)       # 2000 frames = 4 orbits (tools/run_c4.py)
        c4 = {"seq": c4_seq, "frames": c4_frames, "F": F, "total": total}

    # the two other single-GPU configurations under the driver's clock: N = 1, BASELINE's headline arguments only
    extras_in = None
    is_headline = ((W, H) == (640, 480) and abs(float(vs) - 0.01) < 1e-6 and args.trunc == 10.0 and args.hash_capacity_log2 == 22)
    if world == 1 and not args.no_extras and is_headline:
        DW_WARM, DW_K = 20, 200                              # bench.py's own default window: frames 21..220
        dw_seq, dw_frames = render_frames("tum", W, H, range(1 + DW_WARM + DW_K), seed=rank, n_frames=1 + DW_WARM + DW_K)
        C3_W, C3_H, C3_WARM, C3_K = 1280, 960, 5, 20         # configs[2] on the driver's window
        c3_seq, c3_frames = render_frames("tum", C3_W, C3_H, range(1 + C3_WARM + C3_K), seed=rank, n_frames=1 + C3_WARM + C3_K)
        extras_in = {"dw": (dw_seq, dw_frames, DW_WARM, DW_K), "c3": (c3_seq, c3_frames, C3_WARM, C3_K, C3_W, C3_H)}

    # torch first: libgsdf binds to the HIP runtime already in the process (gradient-sdf_amd/binding.py)
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"

    import __graft_entry__ as graft
    pkg = graft.package()

    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=args.hash_capacity_log2, device=local_rank)
    dev = [g.upload(f[0]) for f in frames]          # inputs resident in HBM before the timed region

    def q_from_R(R):
        return pkg.synth.R_to_quat_np(R).astype(np.float32)

    def sync_all(ctx=None):
        (ctx or g).sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def max_over_ranks(values):
        if world == 1:
            return [float(v) for v in values]
        tt = torch.tensor(list(values), dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [float(v) for v in tt.tolist()]

    d0, R0, t0 = frames[0]
    p0 = np.concatenate([t0, q_from_R(R0)]).astype(np.float32)

    def start_stream():
        """frame 0: setup at the ground-truth pose (main_scan_3d.cpp:242), then W untimed warm-up steps"""
        g.reset()
        g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
        g.set_pose(p0)
        for i in range(1, 1 + Wm):
            track_fuse(g, dev, i, Wm + K)

    # ---- burn-in: whole windows, untimed ------------------------------------------------------------------------------
    # Python's cyclic garbage collector is switched off for the windows: a generation-2 collection -- ~40 ms with the frame
    # lists of this process alive -- had landed in the second of five timed windows (475 frames/s between four at 9 400) after
    # an unrelated edit moved the allocation count that triggers it.  At least two whole windows are run untimed first, and
    # further ones until a window takes no more than 1.5 x the fastest so far (at most eight), so that whatever else happens
    # once per process is over: `config.burn_in_windows` / `burn_in_runs` say what they saw.
    import gc
    gc.collect()
    gc.disable()                  # no collector pauses inside anything that is timed from here on
    burn = []
    while len(burn) < 8:
        start_stream()
        sync_all()
        t_b = time.perf_counter()
        for i in range(1 + Wm, 1 + Wm + K):
            track_fuse(g, dev, i, Wm + K)
        sync_all()
        burn.append(max_over_ranks([time.perf_counter() - t_b])[0])
        if len(burn) >= 2 and burn[-1] <= 1.5 * min(burn):
            break

    # ---- timed region: exactly K steps, `repeats` times ----------------------------------------------------------
    runs = []
    st_w = None
    for rep in range(max(1, args.repeats)):
        start_stream()
        sync_all()
        st_w = g.stats()
        sync_all()
        t_start = time.perf_counter()
        for i in range(1 + Wm, 1 + Wm + K):
            track_fuse(g, dev, i, Wm + K)
        sync_all()
        runs.append(max_over_ranks([time.perf_counter() - t_start])[0])
    elapsed = float(np.median(runs))          # (the collector stays off: the flavours measured below are timed too)

    st = g.stats()
    log = g.frame_log()
    timed = log[Wm:Wm + K]
    n_conv = int(timed[:, 7].sum())
    passes = float(timed[:, 8].mean()) if len(timed) else 0.0
    gt_t = np.array([frames[i][2] for i in range(1 + Wm, 1 + Wm + K)])
    trans_err = float(np.abs(timed[:, :3] - gt_t).max()) if len(timed) else 0.0
    n_upd_timed = st["n_upd"] - st_w["n_upd"]
    n_hit_timed = st["n_hit"] - st_w["n_hit"]
    voxels = g.count()

    # ---- staging-inclusive flavour: the same K frames, but every timed frame starts in a page-locked HOST buffer ----
    # main_scan_3d.cpp:213: the reference's loop loads the frame inside the loop.  Here: gsdf_dev_upload_ahead (the copy runs on
    # the library's copy stream, up to `ahead` frames in front of the kernels) into a ring of device slots, gsdf_upload_wait,
    # gsdf_track_and_fuse_dev, gsdf_mark (the slot is reused once the stream has passed its mark) -- what the Scan3D CLI's
    # FramePipeline does, without the PNG decode.  Timed like `value`; reported as config.staged_fps, never as `value`.
    staged_runs = []
    if not args.no_staged:
        import ctypes as C
        Lb = g.L
        nbytes = W * H * 4
        first, last = 1 + Wm, Wm + K
        host = []
        for i in range(first, last + 1):
            hp = C.c_void_p()
            g._chk(Lb.gsdf_host_alloc(g.h, C.byref(hp), nbytes))
            C.memmove(hp, np.ascontiguousarray(frames[i][0], np.float32).ctypes.data, nbytes)
            host.append(hp)
        S_SLOTS, AHEAD, MARK_EVERY = 12, 4, int(os.environ.get("GSDF_BENCH_MARK_EVERY", "4"))
        slots = []
        for _ in range(S_SLOTS):
            dp = C.c_void_p()
            g._chk(Lb.gsdf_dev_alloc(g.h, C.byref(dp), nbytes))
            g._dev.append(dp)
            slots.append(dp)
        def start_uploads(upto, state):
            """copies of the timed frames state['nxt'] .. upto into the ring, on the copy stream, ahead of the kernels"""
            while state["nxt"] < K and state["nxt"] <= upto:
                nxt = state["nxt"]
                sl = nxt % S_SLOTS
                assert sl not in state["unmarked"]  # S_SLOTS > AHEAD + MARK_EVERY: the slot's last reader is behind a recorded mark
                if state["mark"][sl] is not None:
                    g._chk(Lb.gsdf_mark_wait(g.h, state["mark"][sl]))
                uid = C.c_int64(0)
                g._chk(Lb.gsdf_dev_upload_ahead(g.h, slots[sl], host[nxt], nbytes, C.byref(uid)))
                state["ids"][nxt] = uid.value
                state["nxt"] = nxt + 1

        for rep in range(max(1, args.repeats)):
            start_stream()
            # The timed window is a slice of a continuous stream: while the warm-up frames are processed, the copies of the next
            # AHEAD frames are already on their way (that is what "ahead of the stream" means), so they are started here, in front
            # of the barrier that opens the timed region.  Without this a 20-frame window pays the latency of its first copies
            # (~100 us, 5 % of the window) that a stream pays once.  The device-side frame period with staging is +1.2 us (+1 %):
            # profiles/r04_staged_trace.txt (tools/staged_trace.py under rocprofv3 --kernel-trace).
            stt = {"nxt": 0, "mark": [None] * S_SLOTS, "unmarked": [], "ids": {}}
            start_uploads(AHEAD, stt)
            sync_all()
            slot_mark, unmarked, ids = stt["mark"], stt["unmarked"], stt["ids"]
            t_start = time.perf_counter()
            dbg = os.environ.get("GSDF_BENCH_DEBUG")
            tacc = [0.0] * 5
            for j in range(K):
                ta = time.perf_counter()
                start_uploads(j + AHEAD, stt)
                tb = time.perf_counter()
                g._chk(Lb.gsdf_upload_wait(g.h, ids.pop(j)))
                # the next frame's copy was started AHEAD - 1 frames ago: once it is known to be over (a host-side look, normally no
                # wait) the frame can be named ahead like a resident one (gsdf_hint_next_depth_dev's contract: the image is in place)
                nxt_dev = None
                if NEXT_HINT and j + 1 < K and (j + 1) in ids:
                    g._chk(Lb.gsdf_upload_wait(g.h, ids[j + 1]))
                    nxt_dev = slots[(j + 1) % S_SLOTS]
                tc = time.perf_counter()
                if nxt_dev is not None:
                    g.track_and_fuse_ahead_dev(slots[j % S_SLOTS], nxt_dev)
                else:
                    g.track_and_fuse_dev(slots[j % S_SLOTS])
                td = time.perf_counter()
                # A mark is an event on the kernels' stream -- a packet between this frame's fusion and the next frame's first
                # tracker pass -- so one is recorded every MARK_EVERY frames only and releases all the slots submitted since the
                # one before (the ring is MARK_EVERY slots longer for it).  Measured: a mark per frame 8 990, every fourth 9 100.
                unmarked.append(j % S_SLOTS)
                if len(unmarked) >= MARK_EVERY or j == K - 1:
                    mk = C.c_int64(0)
                    g._chk(Lb.gsdf_mark(g.h, C.byref(mk)))
                    for sl in unmarked:
                        slot_mark[sl] = mk.value
                    del unmarked[:]
                te = time.perf_counter()
                if dbg:
                    tacc[0] += tb - ta; tacc[1] += tc - tb; tacc[2] += td - tc; tacc[3] += te - td
            if dbg:
                print("staged window %d: host us per frame: start uploads %.1f, upload_wait %.1f, track_and_fuse %.1f, mark %.1f" % (
                    rep, tacc[0] / K * 1e6, tacc[1] / K * 1e6, tacc[2] / K * 1e6, tacc[3] / K * 1e6), file=sys.stderr)
            sync_all()
            staged_runs.append(max_over_ranks([time.perf_counter() - t_start])[0])
        staged_log = g.frame_log()[Wm:Wm + K]
        # the staged window must have done the same work as the resident one
        staged_same = bool(len(staged_log) == len(timed) and np.array_equal(staged_log[:, 7:9], timed[:, 7:9]))
        for hp in host:
            g._chk(Lb.gsdf_host_free(g.h, hp))

    # fused-only flavour (GT poses, update only) over the same K frames.  Measured BEFORE the event-timed replay:
    # recording timing events switches the HIP queue to a slower, profiled dispatch for the rest of the process.
    # Two rounds, the second one counts: the first long run of back-to-back launches in a process makes the HIP runtime grow
    # its launch resources once (one launch call of ~40 ms).
    fused_fps = 0.0
    for rnd in range(2):
        g.reset()
        sync_all()
        tf = time.perf_counter()
        for j, i in enumerate(range(1 + Wm, 1 + Wm + K)):
            g.update_dev(dev[i], frames[i][1], frames[i][2])
            if j % 32 == 31:
                g.sync()             # hundreds of launches queued without a sync make the HIP runtime throttle the host
        t_enq = time.perf_counter() - tf
        sync_all()
        fused_fps = K / max_over_ranks([time.perf_counter() - tf])[0]
    if os.environ.get("GSDF_BENCH_DEBUG"):
        print("fused-only: enqueue %.1f us/frame, total %.1f us/frame" % (t_enq / K * 1e6, 1e6 / fused_fps), file=sys.stderr)

    # ---- roofline of the dominant kernel: replay the same K frames with HIP events around k_fuse ---
    # (a separate pass so that event records do not perturb `value`; same frames, same poses)
    poses = log[:, :7].copy()
    g.reset()
    g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
    for i in range(1, 1 + Wm):
        if log[i - 1, 7] > 0:
            g.update_dev(dev[i], quat_to_R(poses[i - 1, 3:]), poses[i - 1, :3])
    g.sync()
    st_a = g.stats()
    g.profile(1)
    n_fuse = 0
    for i in range(1 + Wm, 1 + Wm + K):
        if log[i - 1, 7] > 0:
            g.update_dev(dev[i], quat_to_R(poses[i - 1, 3:]), poses[i - 1, :3])
            n_fuse += 1
    g.sync()
    prof = g.profile_read()
    g.profile(0)
    st_b = g.stats()
    fuse_ms = prof["fusion"]["ms"] / max(prof["fusion"]["launches"], 1)
    n_upd_launch = (st_b["n_upd"] - st_a["n_upd"]) / max(n_fuse, 1)
    alg_bytes = 16.0 * W * H + 52.0 * n_upd_launch            # fusion = 16 N_pix + 52 N_upd
    achieved = alg_bytes / (fuse_ms * 1e-3) / 1e9 if fuse_ms > 0 else 0.0

    # ---- the tracker's roofline entry: replay the tracked stream with HIP events around every k_track_pass launch ---
    # one pass moves 4 N_pix (depth) + 32 N_hit (one voxel record per hit) bytes
    start_stream()
    g.sync()
    st_c = g.stats()
    g.profile(1)
    for i in range(1 + Wm, 1 + Wm + K):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    prof_t = g.profile_read()
    trk_each = np.sort(g.profile_launches(2))[::-1]           # every tracker launch by itself, longest first
    g.profile(0)
    st_d = g.stats()
    log_t = g.frame_log()[Wm:Wm + K]
    trk_passes = float(log_t[:, 8].sum())
    trk_bytes = 4.0 * W * H * trk_passes + 32.0 * float(st_d["n_hit"] - st_c["n_hit"])
    trk_ms = prof_t["track_pass"]["ms"]
    trk_achieved = trk_bytes / (trk_ms * 1e-3) / 1e9 if trk_ms > 0 else 0.0
    # Per EXECUTED pass: of the launches the host issues (batches of 5, then 8) only `passes` gather; the launch
    # behind a frame's last pass is head-only (reduce, solve, publish) and those behind the end of optimize() return at once.
    # The launches that ran a pass are the `passes` longest ones; their median duration is the per-pass figure.
    n_exec = int(min(trk_passes, len(trk_each)))
    trk_pass_us = float(np.median(trk_each[:n_exec])) * 1e3 if n_exec else 0.0
    trk_per_pass = (trk_bytes / max(trk_passes, 1.0)) / (trk_pass_us * 1e-6) / 1e9 if trk_pass_us > 0 else 0.0

    # ---- the raycaster's roofline entry: render the bench map (all 1 + W + K frames tracked and fused) from the last pose ---
    # bytes per render = 8 per sample the definition evaluates (one block-key probe) + 32 per voxel record read + 16 per pixel
    # written (depth + normal); samples / records counted on the device (gsdf_raycast_counters)
    raycast = None
    if args.raycast_reps > 0:
        import ctypes
        N = W * H
        buf = ctypes.c_void_p()
        g._chk(g.L.gsdf_dev_alloc(g.h, ctypes.byref(buf), 4 * N * 4))
        g._dev.append(buf)
        nrm = ctypes.c_void_p(buf.value + 4 * N)
        pose_last = g.get_pose()
        Rl, tl = quat_to_R(pose_last[3:]), pose_last[:3]
        g.raycast_dev(Rl, tl, buf, nrm)
        g.sync()
        g.raycast_counters(reset=True)
        g.profile(1)
        for _ in range(args.raycast_reps):
            g.raycast_dev(Rl, tl, buf, nrm)
        g.sync()
        pr = g.profile_read_all()["raycast"]
        g.profile(0)
        samples, records = g.raycast_counters(reset=True)
        rc_us = pr["ms"] * 1e3 / max(pr["launches"], 1)
        rc_bytes = (8.0 * samples + 32.0 * records) / max(args.raycast_reps, 1) + 16.0 * N
        depth_r = g.download(buf, (H, W), np.float32)
        hit = depth_r > 0
        raycast = {"kernel": "k_raycast", "avg_launch_us": round(rc_us, 2), "launches": int(pr["launches"]),
                   "algorithmic_bytes_per_launch": round(rc_bytes), "achieved": round(rc_bytes / (rc_us * 1e-6) / 1e9, 1) if rc_us > 0 else 0.0,
                   "frac": round(rc_bytes / (rc_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rc_us > 0 else 0.0,
                   "samples_per_launch": int(samples // max(args.raycast_reps, 1)), "records_per_launch": int(records // max(args.raycast_reps, 1)),
                   "hit_fraction": round(float(hit.mean()), 4),
                   "median_abs_diff_to_input_depth_mm": round(float(np.median(np.abs(depth_r - frames[-1][0])[hit])) * 1e3, 3) if hit.any() else None}

    # device n_upd against the oracle's, on the CPU-sample frames at the oracle's poses (filled in below)
    def device_n_upd(i, R, t):
        g.reset()
        g.update_dev(dev[i], R, t)
        return g.stats()["n_upd"]

    # ---- CPU baseline: the oracle (port of the reference's serial path) on the frames `value` is timed on ----------
    cpu = None
    n_upd_checked = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        O = graft.oracle_module()
        nc = min(args.cpu_frames, K)
        first = 1 + Wm                                     # first timed frame of `value`
        o = O.Oracle(vs, T, W, H, seq.K)
        o.update(frames[0][0], quat_to_R(p0[3:]), t0)
        pose = p0.copy()
        for i in range(1, first):                          # the warm-up frames: tracked and fused, untimed
            conv, pose, _, _, _ = o.track(frames[i][0], pose)
            if conv:
                o.update(frames[i][0], O.quat_to_R(pose[3:]), pose[:3])
        warm_keys, warm_pay = o.export()
        pose_w = pose.copy()
        cpu_passes, fused = [], []
        tc = time.perf_counter()
        for i in range(first, first + nc):
            conv, pose, used, _, _ = o.track(frames[i][0], pose)
            cpu_passes.append(int(used))
            if conv:
                nu, _ = o.update(frames[i][0], O.quat_to_R(pose[3:]), pose[:3])
                fused.append((i, O.quat_to_R(pose[3:]).copy(), pose[:3].copy(), int(nu)))
        dt = time.perf_counter() - tc
        # N_upd of the roofline comes from the device counter; here it is checked against the oracle's on these frames
        n_upd_checked = bool(fused) and all(device_n_upd(i, R, t) == nu for i, R, t, nu in fused)
        # the reference's OMP-structured variant (critical-section fusion, 4-thread tracker reduction,
        # MapGradPixelSdfOmp.cpp:82,112 / RigidPointOptimizerOmp.cpp:68-69) on a shorter sample, from the same warm map
        no = min(4, nc)

        def omp_leg(track_threads):
            o2 = O.Oracle(vs, T, W, H, seq.K, threads=4)
            o2.set_map(warm_keys, warm_pay)
            pose = pose_w.copy()
            tc = time.perf_counter()
            for i in range(first, first + no):
                o2.set_threads(track_threads)
                conv, pose, _, _, _ = o2.track(frames[i][0], pose, omp=True)
                o2.set_threads(4)
                if conv:
                    o2.update(frames[i][0], O.quat_to_R(pose[3:]), pose[:3], omp=True)
            return time.perf_counter() - tc
        dto = omp_leg(4)
        # ... and with every host core in the tracker's parallel-for.  Only the tracker: the fusion keeps
        # the 4 threads the reference's tracker leaves set (omp_set_num_threads(4), RigidPointOptimizerOmp.cpp:68) -- its
        # `omp critical` (MapGradPixelSdfOmp.cpp:112) serialises the map update whatever the thread count, and with hundreds
        # of threads the contention makes a frame take minutes.
        ncores = os.cpu_count() or 1
        dta = omp_leg(ncores)
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        cpu = {"value": round(nc / dt, 3) if nc else 0.0, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "frames %d..%d of the same stream = the first %d frames `value` is timed on (frames 0..%d tracked + fused "
                         "untimed first), serial oracle, %s host cores present" % (first, first + nc - 1, nc, first - 1, os.cpu_count()),
               "passes_per_frame": cpu_passes, "gpu_passes_per_frame": [int(v) for v in timed[:nc, 8]],
               "cpu_model": model,
               "omp4_value": round(no / dto, 3) if no else 0.0,
               "omp4_note": "reference's OMP structure (fusion inside omp critical, 4-thread tracker), first %d timed frames" % no,
               "omp_all_value": round(no / dta, 3) if no else 0.0,
               "omp_all_note": "the same with %d threads (all host cores) in the tracker's parallel-for, 4 in the fusion, first %d timed frames" % (ncores, no)}

    g.close()

    # HBM traffic of one k_fuse launch: PMC counters cannot be read from inside the process, so they come from the committed
    # rocprofv3 --pmc passes over this very command (profiles/pmc_latest.json names file and command); reported only for
    # the workload they were collected on, null otherwise
    traffic = None
    l2_atomics = None
    traffic_source = None
    try:
        is_c3_ = (W, H) == (1280, 960) and abs(float(vs) - 0.005) < 1e-6 and args.trunc == 10.0 and args.hash_capacity_log2 == 25
        with open(os.path.join(ROOT, "profiles", "pmc_latest_c3.json" if is_c3_ else "pmc_latest.json")) as f:
            if is_c3_ or ((W, H) == (640, 480) and abs(float(vs) - 0.01) < 1e-6 and args.trunc == 10.0 and args.hash_capacity_log2 == 22):
                pmc = json.load(f)
                traffic = pmc.get("traffic_bytes_per_fusion")
                l2_atomics = round(pmc.get("k_fuse", {}).get("TCC_ATOMIC", 0))
                traffic_source = "%s (rocprofv3 --pmc, `%s`)" % (pmc.get("source"), pmc.get("command"))
    except (OSError, ValueError):
        pass

    # what the line is quoted on follows the arguments: only the two single-GPU BASELINE configurations carry their names
    cm = "%gcm" % (float(vs) * 100.0)
    is_c2 = (W, H) == (640, 480) and abs(float(vs) - 0.01) < 1e-6 and args.trunc == 10.0 and args.hash_capacity_log2 == 22
    is_c3 = (W, H) == (1280, 960) and abs(float(vs) - 0.005) < 1e-6 and args.trunc == 10.0 and args.hash_capacity_log2 == 25
    if is_c2:
        workload = "S-tum: TUM fr1/xyz-format synthetic stream, 640x480, 1 cm voxels, trunc 10, capacity 2^22 (BASELINE.json configs[1])"
    elif is_c3:
        workload = "S-stress: the S-tum stream at 1280x960, 5 mm voxels, trunc 10, capacity 2^25 (BASELINE.json configs[2])"
    else:
        workload = "S-tum stream at %dx%d, %s voxels, trunc %g, capacity 2^%d (not a BASELINE.json configuration)" % (
            W, H, cm, args.trunc, args.hash_capacity_log2)

    printed = threading.Lock()
    extras = {}

    def emit(sharded):
        """Rank 0 prints THE line (once); everything it needs is known before the sharded flavour starts."""
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            total_frames = K * world
            out = {
                "metric": "depth frames/sec fused+tracked, %dx%d @%s voxels" % (W, H, cm),
                "value": round(total_frames / elapsed, 2),
                "unit": "frames/s",
                "n_gpus": world,
                "steps": K,
                "warmup": Wm,
                "ms_per_step": round(elapsed / K * 1e3, 4),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "config": {
                    "workload": workload,
                    "width": W, "height": H, "voxel_size_m": float(vs), "trunc_voxels": args.trunc,
                    "hash_capacity_log2": args.hash_capacity_log2, "tracker": "25 iters, conv 1e-3, damping 1",
                    "next_depth_hint": bool(NEXT_HINT),   # gsdf_hint_next_depth_dev in the resident loops (see track_fuse)
                    "parallelism": "replicas x%d (tracked path does not shard)" % world + (
                        " -- ALL RANKS ON ONE GPU (--rccl-double: launch-path test, value is not an N-GPU number)" if args.rccl_double else ""),
                    "value_is": "median of %d timed windows" % len(runs),
                    "value_runs": [round(total_frames / r, 1) for r in runs],
                "burn_in_windows": len(burn), "burn_in_runs": [round(total_frames / r, 1) for r in burn],
                    "converged_frames": n_conv, "mean_tracker_passes": round(passes, 2),
                    "max_abs_translation_error_m": round(trans_err, 5), "voxels": voxels,
                    "n_upd_per_frame": round(n_upd_timed / max(n_conv, 1)), "n_hit_per_pass": round(n_hit_timed / max(passes * K, 1)),
                    "n_upd_oracle_checked": n_upd_checked,
                    "fused_only_fps": round(fused_fps * world, 1),
                    # the same K frames handed over as page-locked HOST buffers (upload ahead of the stream + wait + track + fuse)
                    "staged_fps": round(total_frames / float(np.median(staged_runs)), 1) if staged_runs else None,
                    "staged_runs": [round(total_frames / r, 1) for r in staged_runs] if staged_runs else None,
                    "staged_same_passes_as_resident": staged_same if staged_runs else None,
                    "raycast_us": raycast["avg_launch_us"] if raycast else None,
                    # the same engine on the other two single-GPU windows / configurations, timed like `value` (never `value`)
                    "streams_per_gpu": extras.get("streams_per_gpu"),
                    "default_window": extras.get("default_window"),
                    "c3": extras.get("c3"),
                    "sharded": sharded,
                },
                "roofline": {
                    "bound": "hbm", "kernel": "k_fuse", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": round(alg_bytes), "avg_launch_us": round(fuse_ms * 1e3, 2),
                    "launches": prof["fusion"]["launches"],
                    "l2_atomics_per_launch": l2_atomics,       # the C2 table is cache resident, so report atomics too
                    # `frac` / `achieved`: per EXECUTED pass = (bytes of all passes / passes) / median duration of the launches that
                    # ran one; `per_launch`: the same bytes over the time of ALL tracker launches, head-only and empty ones included
                    "tracker": {"kernel": "k_track_pass", "achieved": round(trk_per_pass, 1), "frac": round(trk_per_pass / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes": round(trk_bytes), "passes": int(trk_passes),
                                "pass_launch_us_median": round(trk_pass_us, 2),
                                "per_launch": {"achieved": round(trk_achieved, 1), "frac": round(trk_achieved / HBM_PEAK_GBS, 4),
                                               "launches": prof_t["track_pass"]["launches"],
                                               "avg_launch_us": round(trk_ms * 1e3 / max(prof_t["track_pass"]["launches"], 1), 2)}},
                    "raycast": raycast,
                },
                "cpu_baseline": cpu,
            }
        # the JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio, which would otherwise be
        # flushed behind it at exit
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                       # noqa: BLE001
            pass
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()

    # ---- the other two single-GPU numbers, under the same clock as `value` (and the same kind of watchdog as the sharded flavour) ----
    if extras_in is not None:
        def give_up_extras():
            extras.setdefault("streams_per_gpu", {"error": "did not finish within %.0f s" % args.extras_timeout})
            extras.setdefault("default_window", {"error": "did not finish within %.0f s" % args.extras_timeout})
            extras.setdefault("c3", {"error": "did not finish within %.0f s" % args.extras_timeout})
            emit({"error": "skipped: the extra configurations did not finish in time"} if c4 is not None else None)
            os._exit(0)
        dog_x = threading.Timer(args.extras_timeout, give_up_extras)
        dog_x.daemon = True
        dog_x.start()
        for name, fn in (("streams_per_gpu", lambda: two_streams_flavour(pkg, args, seq, frames, local_rank, vs, T, W, H, Wm, K)),
                         ("default_window", lambda: default_window_flavour(pkg, args, extras_in["dw"], local_rank, vs, T, W, H)),
                         ("c3", lambda: c3_flavour(pkg, args, extras_in["c3"], local_rank))):
            try:
                extras[name] = fn()
            except Exception as e:                              # noqa: BLE001 -- the headline line must still be printed
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        dog_x.cancel()

    # ---- the flavour that shards: GT-pose fusion of frame shards + ONE all-reduce of the per-voxel sums (configs[3]) ------
    # It is the only part of the run with a data-path collective on a communicator of its own.  The headline numbers are
    # complete before it starts, so a watchdog bounds it: if the exchange does not come back (a fabric or RCCL problem on
    # the node), every rank prints / leaves on its own after --sharded-timeout seconds instead of hanging the whole line.
    sharded = None
    if c4 is not None:
        def give_up():
            emit({"error": "sharded flavour did not finish within %.0f s" % args.sharded_timeout})
            os._exit(0)
        dog = threading.Timer(args.sharded_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            sharded = sharded_flavour(pkg, args, c4, rank, world, local_rank, torch, dist, vs, T, W, H, coll_dev)
        except Exception as e:                                  # noqa: BLE001 -- the headline line must still be printed
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
    if world > 1:
        dist.destroy_process_group()
    emit(sharded)


def _tracked_windows(g, dev, frames, Wm, K, repeats):
    """`repeats` timed windows of K tracked + fused frames behind frame 0 (GT pose) and Wm untimed frames, as in main():
    returns (seconds per window, the frame log rows of the timed frames of the last window, stats before / after it)."""
    d0, R0, t0 = frames[0]
    p0 = np.concatenate([t0, pkg_quat(R0)]).astype(np.float32)

    def start():
        g.reset()
        g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
        g.set_pose(p0)
        for i in range(1, 1 + Wm):
            track_fuse(g, dev, i, Wm + K)
    runs, st_w = [], None
    for rep in range(1 + repeats):                            # one whole window untimed first
        start()
        g.sync()
        st_w = g.stats()
        t_start = time.perf_counter()
        for i in range(1 + Wm, 1 + Wm + K):
            track_fuse(g, dev, i, Wm + K)
        g.sync()
        if rep:
            runs.append(time.perf_counter() - t_start)
    return runs, g.frame_log()[Wm:Wm + K], st_w, g.stats(), p0


def pkg_quat(R):
    import __graft_entry__ as graft
    return graft.package().synth.R_to_quat_np(R).astype(np.float32)


def two_streams_flavour(pkg, args, seq, frames, local_rank, vs, T, W, H, Wm, K):
    """config.streams_per_gpu: TWO independent replica streams of the headline workload in one process, one context (one HIP
    stream) and one host thread each (tracked mode does not shard -- "N independent streams"; here N = 2 on ONE
    GPU).  A single dependent stream leaves the chip idle between a third (fusion tail) and nine tenths (tracker passes) of the
    time; the second stream's launches fill some of it.  Aggregate frames/s of both streams, reported BESIDE `value`, never as it."""
    import threading
    ctxs = pkg.GradSdf.shards(2, vs, T, W, H, seq.K, capacity_log2=args.hash_capacity_log2, device=local_rank)   # a hardware queue each
    devs = [[g.upload(f[0]) for f in frames] for g in ctxs]
    d0, R0, t0 = frames[0]
    p0 = np.concatenate([t0, pkg_quat(R0)]).astype(np.float32)

    def start(g, dev):
        g.reset()
        g.update_dev(dev[0], quat_to_R(p0[3:]), t0)
        g.set_pose(p0)
        for i in range(1, 1 + Wm):
            track_fuse(g, dev, i, Wm + K)
        g.sync()
    runs, logs = [], None
    for rep in range(4):                                      # the first window is untimed
        for g, dev in zip(ctxs, devs):
            start(g, dev)
        gate = threading.Barrier(3)
        done = [0.0, 0.0]

        errs = []

        def body(k):
            g, dev = ctxs[k], devs[k]
            try:
                gate.wait()
                for i in range(1 + Wm, 1 + Wm + K):
                    track_fuse(g, dev, i, Wm + K)
                g.sync()
            except Exception as e:                              # noqa: BLE001 -- reported by the caller's thread below
                errs.append(e)
            done[k] = time.perf_counter()
        th = [threading.Thread(target=body, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        gate.wait()
        t_start = time.perf_counter()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        if rep:
            runs.append(max(done) - t_start)
    logs = [g.frame_log()[Wm:Wm + K] for g in ctxs]
    for g in ctxs:
        g.close()
    el = float(np.median(runs))
    return {"streams": 2, "aggregate_fps": round(2 * K / el, 1), "runs": [round(2 * K / r, 1) for r in runs],
            "frames_per_stream": K, "converged_frames": [int(l[:, 7].sum()) for l in logs],
            "note": "two independent replica streams (same frames) in one process, one context + one host thread each; aggregate of both"}


def default_window_flavour(pkg, args, dw, local_rank, vs, T, W, H):
    """config.default_window: frames 21..220 of the SAME stream as `value` (bench.py's own default --steps 200 --warmup 20), on
    which a quarter of the frames runs all 25 passes and is not fused ("Non-converging frames")."""
    seq, frames, Wm, K = dw
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=args.hash_capacity_log2, device=local_rank)
    dev = [g.upload(f[0]) for f in frames]
    runs, timed, _, _, _ = _tracked_windows(g, dev, frames, Wm, K, 3)
    g.close()
    el = float(np.median(runs))
    return {"frames": "%d..%d of the stream `value` is timed on (= python bench.py without arguments)" % (1 + Wm, Wm + K),
            "fps": round(K / el, 1), "runs": [round(K / r, 1) for r in runs], "ms_per_frame": round(el / K * 1e3, 4),
            "converged_frames": int(timed[:, 7].sum()), "frames_total": K, "mean_tracker_passes": round(float(timed[:, 8].mean()), 2)}


def c3_flavour(pkg, args, c3, local_rank):
    """config.c3: BASELINE configs[2] -- the S-tum stream at 1280x960, 5 mm voxels, trunc 10, capacity 2^25 (1 GiB of voxel
    records: outside the Infinity Cache) -- on the driver's window (frames 6..25), with k_fuse's roofline figures measured the
    same way as the headline's (HIP events around the executed launches of a replay of the same frames at the same poses)."""
    seq, frames, Wm, K, W, H = c3
    vs = np.float32(0.005)
    T = np.float32(10.0) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=25, device=local_rank)
    dev = [g.upload(f[0]) for f in frames]
    runs, timed, _, _, p0 = _tracked_windows(g, dev, frames, Wm, K, 3)
    log = g.frame_log()
    poses = log[:, :7].copy()
    g.reset()
    g.update_dev(dev[0], quat_to_R(p0[3:]), frames[0][2])
    for i in range(1, 1 + Wm):
        if log[i - 1, 7] > 0:
            g.update_dev(dev[i], quat_to_R(poses[i - 1, 3:]), poses[i - 1, :3])
    g.sync()
    st_a = g.stats()
    g.profile(1)
    n_fuse = 0
    for i in range(1 + Wm, 1 + Wm + K):
        if log[i - 1, 7] > 0:
            g.update_dev(dev[i], quat_to_R(poses[i - 1, 3:]), poses[i - 1, :3])
            n_fuse += 1
    g.sync()
    prof = g.profile_read()
    g.profile(0)
    st_b = g.stats()
    g.close()
    fuse_ms = prof["fusion"]["ms"] / max(prof["fusion"]["launches"], 1)
    n_upd = (st_b["n_upd"] - st_a["n_upd"]) / max(n_fuse, 1)
    alg = 16.0 * W * H + 52.0 * n_upd
    ach = alg / (fuse_ms * 1e-3) / 1e9 if fuse_ms > 0 else 0.0
    el = float(np.median(runs))
    return {"workload": "S-stress: the S-tum stream at 1280x960, 5 mm voxels, trunc 10, capacity 2^25 (BASELINE.json configs[2]), frames %d..%d" % (1 + Wm, Wm + K),
            "fps": round(K / el, 1), "runs": [round(K / r, 1) for r in runs], "ms_per_frame": round(el / K * 1e3, 4),
            "converged_frames": int(timed[:, 7].sum()), "frames_total": K, "mean_tracker_passes": round(float(timed[:, 8].mean()), 2),
            "k_fuse_us": round(fuse_ms * 1e3, 2), "k_fuse_launches": int(prof["fusion"]["launches"]),
            "algorithmic_bytes_per_launch": round(alg), "achieved_gbs": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}


def sharded_flavour(pkg, args, c4, rank, world, local_rank, torch, dist, vs, T, W, H, coll_dev):
    """BASELINE configs[3] on N ranks: every rank fuses its contiguous shard of ONE sphere-orbit stream with the ground-truth
    poses (main_scan_3d.cpp:250-254) into its own map, then ONE exchange -- gsdf_merge_allreduce over an RCCL communicator that
    exists before the timed region -- after which every rank holds the map of all frames; rank 0 extracts the mesh.
    Weak scaling: --c4-frames per rank.  Two rounds, the second one is reported (the first one warms RCCL's channels)."""
    seq, frames, F, total = c4["seq"], c4["frames"], c4["F"], c4["total"]
    ncx = max(1, min(2, args.contexts_per_gpu))
    g2 = None
    if ncx > 1:     # two shard contexts whose streams sit in two hardware queues of their own (gsdf_create_shards)
        g, g2 = pkg.GradSdf.shards(2, vs, T, W, H, seq.K, capacity_log2=23, device=local_rank)
    else:
        g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=23, device=local_rank)
    dev = [g.upload(f[0]) for f in frames]

    def barrier():
        g.sync()
        if world > 1:
            dist.barrier()

    def vmax(values):
        if world == 1:
            return [float(v) for v in values]
        tt = torch.tensor(list(values), dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [float(v) for v in tt.tolist()]

    # ---- transport, outside the timed region ----
    comm, transport, rccl_ranks = None, None, None
    use_rccl = not args.single_device and (world == 1 or args.dist_backend == "nccl" or bool(args.rccl_double))
    if use_rccl:
        ok = 1.0
        try:
            idt = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                idt = torch.frombuffer(bytearray(pkg.binding.rccl_unique_id()), dtype=torch.uint8).clone()
            if world > 1:
                idt = idt.to(coll_dev)
                dist.broadcast(idt, src=0)
            uid = bytes(idt.cpu().numpy().tobytes())
            if args.rccl_double and not uid.startswith(b"/gsdf_fake_rccl_"):
                raise RuntimeError("--rccl-double: libgsdf resolved another RCCL than the test double")
            comm = pkg.binding.rccl_comm_init(world, uid, rank, local_rank)
            rccl_ranks = pkg.binding.rccl_comm_count(comm)
        except Exception as e:                                   # noqa: BLE001
            print("bench.py rank %d: RCCL communicator: %s" % (rank, e), file=sys.stderr)
            ok = 0.0
        # every rank takes the same route
        ok = -vmax([-ok])[0] if world > 1 else ok
        if ok > 0 and args.rccl_double:
            transport = ("rccl test double (tests/fake_rccl.c preloaded: %d processes share ONE GPU, shared-memory rendezvous; the code path of "
                         "gsdf_rccl_comm_init / gsdf_merge_allreduce is the production one, the transport is not RCCL / xGMI -- NOT a scaling number)" % world)
        elif ok > 0:
            transport = "rccl (gsdf_merge_allreduce: pack -> ncclAllReduce -> unpack on the context's stream)"
        else:
            if comm is not None:
                pkg.binding.rccl_comm_destroy(comm)
                comm = None
            use_rccl = False
    if not use_rccl:
        transport = "torch.distributed %s through gsdf_merge_allreduce_with (host staging)" % args.dist_backend

    def ag(send):
        if world == 1:
            return send
        t = torch.from_numpy(send).to(coll_dev)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).cpu().numpy()

    def ar(buf):
        if world == 1:
            return buf
        t = torch.from_numpy(buf).to(coll_dev)
        dist.all_reduce(t)
        return t.cpu().numpy()

    # ---- one context per GPU first (the form of rounds 1-4): its rate, and its map as the yardstick for the two-context form ----
    one_fps, ref_map, dev2, half = None, None, None, F
    if ncx > 1:
        for rnd in range(2):
            g.reset()
            barrier()
            t0 = time.perf_counter()
            for j, (d, f) in enumerate(zip(dev, frames)):
                g.update_dev(d, f[1], f[2])
                if j % 32 == 31:
                    g.sync()
            g.sync()
            one_fps = total / vmax([time.perf_counter() - t0])[0]
        ref_map = g.export(sorted=True, raw=True)
        half = (F + 1) // 2                                       # g takes the first frames of the shard, g2 the rest (contiguous)
        dev2 = [g2.upload(f[0]) for f in frames[half:]]

    g.merge_prepare(world)            # like the communicator: the exchange's scratch and first-use code loads, outside the timed region
    res = {}
    for rnd in range(2):
        g.reset()
        if g2 is not None:
            g2.reset()
        barrier()
        t0 = time.perf_counter()
        if g2 is not None:
            # SURVEY.md 8e "G logical shards on 1 GPU": TWO shard contexts on this GPU, each on its own stream, enqueued alternately
            # -- a fusion launch leaves a third of the chip's workgroup slots idle in its tail, which the other context's launch
            # fills (measured 1.22-1.28x one context, tools/two_contexts.py) -- then the local sum (gsdf_merge_from, ~0.1 ms),
            # then the exchange between the GPUs
            for j in range(half):
                g.update_dev(dev[j], frames[j][1], frames[j][2])
                if half + j < F:
                    g2.update_dev(dev2[j], frames[half + j][1], frames[half + j][2])
                if j % 32 == 31:
                    g.sync(); g2.sync()
            g.sync(); g2.sync()
            t_m = time.perf_counter()
            g.merge_from(g2)
            t_merge = time.perf_counter() - t_m
        else:
            for j, (d, f) in enumerate(zip(dev, frames)):
                g.update_dev(d, f[1], f[2])
                if j % 32 == 31:
                    g.sync()
            g.sync()
            t_merge = 0.0
        t_fuse = time.perf_counter() - t0
        own = g.count()
        same = None
        if g2 is not None and rnd == 1:
            k2, p2 = g.export(sorted=True, raw=True)
            same = bool(k2.shape == ref_map[0].shape and np.array_equal(k2, ref_map[0]) and
                        float((np.abs(p2 - ref_map[1]).max(axis=1) / np.maximum(1.0, ref_map[1][:, 4])).max()) <= 1e-5)
        barrier()
        t1 = time.perf_counter()
        if use_rccl:
            nb, nbytes = g.merge_allreduce_rccl(comm)
        else:
            nb, nbytes = g.merge_allreduce_with(ag, ar, world)
        g.sync()
        t_exch = time.perf_counter() - t1
        t_fuse, t_exch, t_both, t_merge = vmax([t_fuse, t_exch, t_fuse + t_exch, t_merge])
        res = {"frames_per_rank": F, "frames_total": total, "ranks": world, "rccl_ranks": rccl_ranks, "transport": transport,
               "contexts_per_gpu": ncx,
               "sharded_fused_fps": round(total / t_fuse, 1), "sharded_fused_fps_incl_exchange": round(total / t_both, 1),
               "one_context_fused_fps": round(one_fps, 1) if one_fps else None,
               "same_map_as_one_context": same,
               "fuse_ms": round(t_fuse * 1e3, 3), "local_merge_ms": round(t_merge * 1e3, 3) if g2 is not None else None,
               "exchange_ms": round(t_exch * 1e3, 3), "exchange_bytes": int(nbytes),
               "exchange_blocks": int(nb), "voxels_own_shard": int(own)}
    res["voxels_merged"] = int(g.count())
    res["frames_counter_after_merge"] = int(g.stats()["frames"])
    if rank == 0:
        t2 = time.perf_counter()
        tris = g.extract_mesh()
        res["mesh_faces"] = int(len(tris))
        res["export_ms"] = round((time.perf_counter() - t2) * 1e3, 2)
    barrier()
    if comm is not None:
        pkg.binding.rccl_comm_destroy(comm)
    if g2 is not None:
        g2.close()
    g.close()
    return res


if __name__ == "__main__":
    main()
