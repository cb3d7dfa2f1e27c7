"""One rank of the RCCL-typed exchange test (tests/test_parallel.py::test_merge_allreduce_rccl_path_with_peers): a process of
its own that loads tests/libfake_rccl.so into the global symbol scope BEFORE libgsdf resolves the nccl* entry points
(gsdf_merge.hip:50-73, dlsym(RTLD_DEFAULT, ...)), fuses its shard of the frames on device 0 and calls
gsdf_merge_allreduce(ctx, comm) itself.

    python tests/rccl_rank.py <rank> <world> <id file> <out dir> <libfake_rccl.so>
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H, N_FRAMES, SEED, STEP_DEG, VS, TRUNC = 160, 120, 19, 3, 4.0, 0.04, 5


def main():
    rank, world, id_file, out_dir, fake = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import __graft_entry__ as graft
    pkg = graft.package()
    L = pkg.binding.load()                                   # HIP runtime (RTLD_GLOBAL) + libgsdf.so; no torch => no real RCCL in the process
    assert "torch" not in sys.modules
    fk = C.CDLL(fake, mode=C.RTLD_GLOBAL)                    # the nccl* symbols libgsdf will find (its hip* resolve from the global scope)
    fk.fake_rccl_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    if rank == 0:
        uid = pkg.binding.rccl_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        with open(id_file, "rb") as f:
            uid = f.read()
    assert uid.startswith(b"/gsdf_fake_rccl_"), "libgsdf resolved another RCCL than the test double"
    comm = pkg.binding.rccl_comm_init(world, uid, rank, 0)
    assert pkg.binding.rccl_comm_count(comm) == world

    seq = pkg.synth.Sequence("spheres", W, H, n_frames=N_FRAMES, seed=SEED, step_deg=STEP_DEG)
    vs = np.float32(VS)
    # the last rank's table is twice the others' -- as if auto-grow had doubled it during its scan: the exchange brings the
    # smaller ones up first (the ranks learn the largest capacity from the header all-gather and agree that growing worked)
    g = pkg.GradSdf(vs, np.float32(TRUNC) * vs, W, H, seq.K, capacity_log2=18 + (1 if rank == world - 1 else 0), device=0)
    g.enable_vis(32)
    lo, hi = pkg.parallel.shard_range(N_FRAMES, rank, world)  # uneven: 19 frames over 8 ranks = 3 / 2 per rank
    dev = [g.upload(seq.frame(i)[0]) for i in range(lo, hi)]
    for j, i in enumerate(range(lo, hi)):
        g.update_dev(dev[j], seq.frame(i)[1], seq.frame(i)[2])
    own = g.count()
    nb, nbytes = g.merge_allreduce_rccl(comm)                # <- the code path under test
    keys, pay = g.export(sorted=True, raw=True)
    kv, vis = g.export_vis()
    again = ""
    try:
        g.merge_allreduce_rccl(comm)                         # one-shot: refused on every rank without entering a collective
    except pkg.binding.GsdfError as e:
        again = str(e)
    stats = (C.c_uint64 * 4)()
    fk.fake_rccl_stats(comm, stats)
    np.savez(os.path.join(out_dir, "rccl_%d.npz" % rank), keys=keys, pay=pay, vis=vis, frames=g.stats()["frames"], lo=lo, hi=hi,
             own=own, n_blocks=nb, nbytes=nbytes, again=again, cap=g.capacity_log2(), stats=np.array(list(stats), np.uint64))
    g.close()
    pkg.binding.rccl_comm_destroy(comm)


if __name__ == "__main__":
    main()
