"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact voxel keys / occupancy; <= 1e-4 on SDF distance, gradient and
pose (BASELINE.json north_star); the cached planes and normals are bit-exact by construction."""
import ctypes

import numpy as np
import pytest

from conftest import pose7_from
from lockstep import lockstep, flip_classes
from test_oracle_serial_vs_omp import SERIAL_VS_OMP_FLIPS

pytestmark = pytest.mark.gpu

TOL = 1e-4
# frames of the free-running 64-frame bench stream that agree with the oracle EXACTLY (flag, pass count) before the first one that
# differs; measured on MI355X in round 6 (see the MEASURED line the test prints with -rP)
N_STRICT_FLOOR = 25         # measured: 33 (first sensitive frame: 2, i.e. the bound derived from the oracle's |xi|^2 series is 1)


def _mk(pkg, O, kind="spheres", W=160, H=120, vs=0.02, trunc=5, cap=18, seed=1, n=4, lib=None, **kw):
    seq = pkg.synth.Sequence(kind, W, H, n_frames=n, seed=seed, **kw)
    vs = np.float32(vs)
    T = np.float32(trunc) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=cap, lib=lib)
    o = O.Oracle(vs, T, W, H, seq.K)
    return seq, g, o


def _cmp_tables(g, o, weight_scale=1.0):
    kg, pg = g.export(sorted=True)
    ko, po = o.export()
    assert kg.shape == ko.shape, (kg.shape, ko.shape)
    assert np.array_equal(kg, ko), "voxel key sets differ"
    assert np.abs(pg[:, 0] - po[:, 0]).max() <= TOL                      # SDF distance
    assert np.array_equal(pg[:, 4] > 0, po[:, 4] > 0)
    # weight and the un-normalised gradient are sums of up to `weight` terms: relative bound
    scale = np.maximum(1.0, po[:, 4])
    assert (np.abs(pg[:, 4] - po[:, 4]) / scale).max() <= TOL
    assert (np.abs(pg[:, 1:4] - po[:, 1:4]).max(axis=1) / scale).max() <= TOL
    # The stored gradient is the un-normalised sum of w R n (|R n| = 1, so |grad| <= weight): the bound above is relative to the
    # number of terms.  What the reference USES is its direction, 1.2 grad/|grad| (MapGradPixelSdf.h:113-114): that is held to
    # the north_star's 1e-4 ABSOLUTE wherever the direction is defined at that precision: |grad| >= 1 % of the weight (a voxel
    # whose few samples' normals nearly cancel has no stable direction in the reference either) and >= 0.05 (the fusion
    # kernel's fixed-point terms are truncated to 2^-21 = 4.8e-7 per sample and component: a voxel whose whole gradient is a
    # few 1e-3 -- one sample at the far end of the weight ramp -- keeps its direction to ~1e-3 only; the tracker's pose,
    # which is what these directions feed, is held to 1e-4 by its own tests).
    ng, no = np.linalg.norm(pg[:, 1:4], axis=1), np.linalg.norm(po[:, 1:4], axis=1)
    ok = no >= np.maximum(1e-2 * po[:, 4], 0.05)
    assert ok.mean() > 0.95
    assert np.abs(pg[ok, 1:4] / ng[ok, None] - po[ok, 1:4] / no[ok, None]).max() <= TOL
    return kg.shape[0]


def test_normals_cache_bit_exact(pkg, O):
    seq, g, o = _mk(pkg, O)
    a, b = g.normals_cache(), o.normals_cache()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    g.close()


def test_normals_bit_exact_with_holes(pkg, O):
    seq, g, o = _mk(pkg, O)
    d, _, _ = seq.frame(0)
    a, b = g.normals(d), o.normals(d)
    m = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), m)
    assert np.array_equal(a[m].view(np.uint32), b[m].view(np.uint32))
    g.close()


@pytest.mark.parametrize("kind,W,H,vs,trunc", [("spheres", 160, 120, 0.02, 5), ("tum", 160, 120, 0.02, 10),
                                               ("spheres", 133, 77, 0.01, 10)])
def test_fusion_keys_bit_exact(pkg, O, kind, W, H, vs, trunc):
    seq, g, o = _mk(pkg, O, kind=kind, W=W, H=H, vs=vs, trunc=trunc, cap=20, n=4)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        nu, nv = o.update(d, R, t)
    n = _cmp_tables(g, o)
    st = g.stats()
    assert st["frames"] == seq.n
    assert n > 1000
    g.close()


@pytest.mark.parametrize("flags,name", [(0, "test build, nothing forced"),
                                        (4, "every tile defers (near-tile path, float atomics in k_fuse_resolve)"),
                                        (512, "every tile walked as 4 row bands (far-tile path)"),
                                        (256, "single band only (far tiles overflow the LDS table into the deferred list)"),
                                        (512 + 4, "4 bands, deferred"),
                                        (8192, "every hand-off wait expires at once (timed-out tiles defer)"),
                                        (1024, "the kernel with the larger LDS table (far scenes)"),
                                        (1024 + 512, "larger table, 4 bands"),
                                        (1024 + 4, "larger table, every tile defers")])
def test_fusion_forced_paths_match_oracle(pkg, O, flags, name):
    """The flush has a fast path (ordered tiles, plain read-modify-write handed from tile to tile) and
    fallbacks chosen per tile from its depth; force each of them on the same input.  The switches exist only
    in the test build of the library (libgsdf_test.so, -DGSDF_EXPERIMENTS), per context."""
    seq, g, o = _mk(pkg, O, kind="tum", W=320, H=240, vs=0.01, trunc=10, cap=21, n=3, lib=pkg.binding.load_test_lib())
    g.debug_flags(flags)
    nu = nv = 0
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        a, b = o.update(d, R, t)
        nu += a; nv += b
    g.sync()
    assert _cmp_tables(g, o) > 10000
    st = g.stats()
    assert st["n_upd"] == nu and st["n_valid"] == nv                          # every sample and pixel counted exactly once
    if flags == 8192:
        assert st["fuse_timeouts"] > 100 and st["n_deferred"] > 10000         # the timed-out tiles took the deferred route
    else:
        assert st["fuse_timeouts"] == 0
        assert (st["n_deferred"] > 10000) == bool(flags & 4) or flags == 256  # only the forced-deferred runs defer wholesale
    g.close()


def test_fusion_handoff_loses_nothing_at_full_size(pkg):
    """Size-independent property at the bench size (640x480, 1 cm): the tile-to-tile hand-off of the flush
    (plain read-modify-write, agent-scope accesses, flags) must give the same sums as the path that defers
    every contribution to float atomics.  A lost or stale hand-off would drop a whole tile's contribution to
    a voxel (>= 1e-2 relative); float reordering alone stays below 1e-5."""
    W, H, n = 640, 480, 24
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=3)
    vs = np.float32(0.01); T = np.float32(10) * vs
    out, stats = [], []
    for flags in (0, 4):
        # the production library for the fast path, the test build with every tile deferring for the other
        g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22, lib=pkg.binding.load_test_lib() if flags else None)
        if flags:
            g.debug_flags(flags)
        for i in range(n):
            d, R, t = seq.frame(i)
            g.update(d, R, t)
        g.sync()
        out.append(g.export(sorted=True, raw=True))
        stats.append(g.stats())
        g.close()
    (ka, pa), (kb, pb) = out
    assert stats[0]["fuse_timeouts"] == 0 and stats[0]["n_deferred"] < 10000   # the fast path really ran (a few LDS overflows defer)
    assert stats[1]["n_deferred"] > 1000000
    assert np.array_equal(ka, kb)
    assert ka.shape[0] > 1000000
    scale = np.maximum(1.0, np.abs(pb[:, 4:5]))
    assert (np.abs(pa - pb) / scale).max() < 1e-5


def test_fusion_counters_match_oracle(pkg, O):
    seq, g, o = _mk(pkg, O, n=2)
    tot_u = tot_v = 0
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        nu, nv = o.update(d, R, t)
        tot_u += nu
        tot_v += nv
    st = g.stats()
    assert st["n_upd"] == tot_u and st["n_valid"] == tot_v
    g.close()


def test_query_matches_oracle(pkg, O):
    seq, g, o = _mk(pkg, O, n=2)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        o.update(d, R, t)
    keys, _ = o.export()
    rng = np.random.default_rng(0)
    sel = keys[rng.integers(0, len(keys), 4000)]
    pts = (sel.astype(np.float32) + rng.uniform(-0.45, 0.45, sel.shape).astype(np.float32)) * np.float32(0.02)
    pts = np.concatenate([pts, rng.uniform(-5, 5, (500, 3)).astype(np.float32)])
    dg, gg, wg = g.query(pts)
    do, go, wo = o.query(pts)
    assert np.array_equal(wg > 0, wo > 0)
    assert np.abs(dg - do).max() <= TOL
    assert np.abs(gg - go).max() <= TOL
    g.close()


def test_tracker_matches_oracle(pkg, O):
    seq, g, o = _mk(pkg, O, W=640, H=480, vs=0.01, trunc=10, cap=21, n=3, seed=0)
    d0, R0, t0 = seq.frame(0)
    g.update(d0, R0, t0)
    o.update(d0, R0, t0)
    d1, R1, t1 = seq.frame(1)
    p0 = pose7_from(O, R0, t0)
    cg, pg, passes = g.track(d1, p0)
    co, po, used, trace, hits = o.track(d1, p0)
    assert cg == co and cg
    assert passes == used
    assert np.abs(pg - po).max() <= TOL
    # hit counts depend on the pose of each pass, which differs in the last bits after pass 1
    assert abs(g.stats()["n_hit"] - int(hits.sum())) <= 1e-3 * hits.sum()
    g.close()


@pytest.mark.parametrize("sampling", [2, 3, 4, 5, 7, 100000])
def test_tracker_sampled_matches_oracle(pkg, O, sampling):
    """gsdf_track_sampled == RigidPointOptimizer::optimize_sampled(depth, K, sampling) (RigidPointOptimizer.h:65, .cpp:62):
    640x480 (not a multiple of 3 or 7: ragged last column / row of samples), frame 2 against the map of frames 0 and 1: first-pass
    hit count exact and pose <= 1e-4 at every stride; the whole optimize() -- pass counts equal, pose <= 1e-4 -- at the strides
    where the oracle's stop test is not a coin toss (2, 3, 4, 5: converged in 3-4 passes with |xi|^2 at least 10 % off the
    threshold, the stream tests' rule for a deterministic frame; at stride 7 its 5th pass ends at |xi|^2 = 9.5e-7 and a 2e-7 m nudge of the start pose makes it 5, 6 or 8 passes;
    against a one-frame map every stride cycles at 6e-6 for all 25 passes and the oracle's serial and OMP builds end 8e-5 apart)"""
    seq, g, o = _mk(pkg, O, kind="tum", W=640, H=480, vs=0.01, trunc=10, cap=21, n=3, seed=0)
    for i in range(2):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        o.update(d, R, t)
    d2, _, _ = seq.frame(2)
    _, R1, t1 = seq.frame(1)
    p0 = pose7_from(O, R1, t1)
    n0 = g.stats()["n_hit"]
    cg, pg, passes = g.track(d2, p0, sampling=sampling, iters=1)
    co, po, used, trace, hits = o.track(d2, p0, sampling=sampling, iters=1)
    assert g.stats()["n_hit"] - n0 == int(hits[0])            # the same pixels: same voxels hit from the same pose
    assert passes == used == 1
    if sampling <= 7:
        assert np.abs(pg - po).max() <= TOL
    if sampling <= 5:
        cg, pg, passes = g.track(d2, p0, sampling=sampling)
        co, po, used, trace, hits = o.track(d2, p0, sampling=sampling)
        xi2 = trace[:used, 35]
        assert co and used <= 4 and bool((np.abs(xi2 / 1e-6 - 1.0) >= 0.1).all()), (used, xi2)     # the premise: a clean run
        assert cg == co and passes == used, (cg, co, passes, used)
        assert np.abs(pg - po).max() <= TOL
    elif sampling > 7:
        # a stride beyond the image leaves pixel (0, 0) alone (one hit, above): H has rank 1, llt() meets a pivot that is zero
        # up to rounding and what it "solves" is decided by that rounding -- not a parity quantity in the reference either
        assert int(hits[0]) == 1 and not cg and not co
    # and the unsampled entry afterwards is untouched by the sampled one's geometry
    cg, pg, passes = g.track(d2, p0)
    co, po, used, _, _ = o.track(d2, p0)
    assert cg == co and passes == used and np.abs(pg - po).max() <= TOL
    with pytest.raises(RuntimeError):
        g.track(d2, p0, sampling=0)
    g.close()


def test_tracker_no_overlap_returns_false(pkg, O):
    seq, g, o = _mk(pkg, O)
    d0, R0, t0 = seq.frame(0)
    p0 = pose7_from(O, R0, t0)
    cg, pg, passes = g.track(d0, p0, iters=5)       # empty map: H = 0 -> NaN -> idle passes
    co, po, used, _, _ = o.track(d0, p0, iters=5)
    assert cg is False and co is False and passes == used == 5
    assert np.array_equal(pg, po)
    g.close()


def test_track_and_fuse_stream_matches_oracle_loop(pkg, O):
    """The device-side Scan3D loop (main_scan_3d.cpp:255-266) against the oracle's host loop."""
    seq, g, o = _mk(pkg, O, W=640, H=480, vs=0.01, trunc=10, cap=21, n=6, seed=0)
    frames = [seq.frame(i) for i in range(seq.n)]
    d0, R0, t0 = frames[0]
    p = pose7_from(O, R0, t0)
    R0q = O.quat_to_R(p[3:])
    g.update(d0, R0q, t0)
    o.update(d0, R0q, t0)
    g.set_pose(p)
    dev = [g.upload(f[0]) for f in frames]
    for i in range(1, seq.n):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    log = g.frame_log()
    po = p.copy()
    for i in range(1, seq.n):
        co, po, used, _, _ = o.track(frames[i][0], po)
        if co:
            o.update(frames[i][0], O.quat_to_R(po[3:]), po[:3])
        assert bool(log[i - 1, 7]) == co
        # frame 1 starts from identical state: the 1e-4 bar.  Later frames start from maps and poses that already
        # differ in the last bits (the oracle sums ~230 k float terms sequentially, the GPU pairwise / in double), so
        # the trajectories drift apart slowly: allow one more TOL per frame.  Measured on THIS stream between the
        # reference's own serial and OMP builds, each free-running (tools/serial_vs_omp.py free,
        # profiles/r05_serial_vs_omp.txt): 1.5e-6, 2.8e-5, 7.9e-6, 8.5e-5, 7.5e-5 at frames 1..5 -- the same slope.
        assert np.abs(log[i - 1, :7] - po).max() <= TOL * i
    kg, _ = g.export()
    ko, _ = o.export()
    inter = len(set(map(tuple, kg)) & set(map(tuple, ko)))
    assert inter / max(len(kg), len(ko)) > 0.97      # poses differ in the last bits -> keys near-identical
    g.close()


def test_merge_raw_equals_single_table(pkg, O):
    """Frame-sharded fusion: two private tables merged additively == one table."""
    seq, g, o = _mk(pkg, O, n=4)
    ga = pkg.GradSdf(np.float32(0.02), np.float32(5) * np.float32(0.02), 160, 120, seq.K, capacity_log2=18)
    gb = pkg.GradSdf(np.float32(0.02), np.float32(5) * np.float32(0.02), 160, 120, seq.K, capacity_log2=18)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        (ga if i % 2 == 0 else gb).update(d, R, t)
        o.update(d, R, t)
    kb, pb = gb.export(raw=True)
    ga.merge_raw(kb, pb)
    _cmp_tables(ga, o)
    k1, p1 = g.export()
    k2, p2 = ga.export()
    assert np.array_equal(k1, k2)
    for h in (g, ga, gb):
        h.close()


def test_table_full_is_reported(pkg, O):
    seq = pkg.synth.Sequence("tum", 160, 120, n_frames=1, seed=0)
    g = pkg.GradSdf(np.float32(0.01), np.float32(0.1), 160, 120, seq.K, capacity_log2=10)
    d, R, t = seq.frame(0)
    with pytest.raises(pkg.GsdfError) as e:
        g.update(d, R, t)
    assert e.value.code == pkg.binding.ERR_TABLE_FULL
    g.close()


def test_device_export_merge_is_the_shard_exchange(pkg, O):
    """gsdf_export_raw_dev -> (all-gather) -> gsdf_merge_raw_dev: the pack/unpack either side of the RCCL
    exchange, here as two logical shards on one GPU."""
    seq, g, o = _mk(pkg, O, n=4, cap=19)
    gb = pkg.GradSdf(np.float32(0.02), np.float32(5) * np.float32(0.02), 160, 120, seq.K, capacity_log2=19)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        (g if i < 2 else gb).update(d, R, t)
        o.update(d, R, t)
    n = gb.count()
    kbuf = g.upload(np.zeros((n, 3), np.int32))
    pbuf = g.upload(np.zeros((n, 5), np.float32))
    got = gb.export_raw_dev(kbuf.value, pbuf.value, n)
    assert got == n
    g.merge_raw_dev(kbuf.value, pbuf.value, n)
    _cmp_tables(g, o)
    g.close()
    gb.close()


def test_dense_block_allreduce_is_the_shard_exchange(pkg, O):
    """gsdf_block_keys_dev -> (all-gather, unique) -> gsdf_pack_blocks_dev -> (all-reduce) -> gsdf_unpack_blocks_dev: the dense
    form of the exchange, here as two logical shards on one GPU with the sum done on the host."""
    seq, ga, o = _mk(pkg, O, n=4, cap=19)
    gb = pkg.GradSdf(np.float32(0.02), np.float32(5) * np.float32(0.02), 160, 120, seq.K, capacity_log2=19)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        (ga if i < 2 else gb).update(d, R, t)
        o.update(d, R, t)
    cap = 1 << (19 - 6)
    lists = []
    for g in (ga, gb):
        buf = g.upload(np.zeros(cap, np.int64))
        n = g.block_keys_dev(buf.value, cap)
        assert 0 < n <= cap
        lists.append(g.download(buf, (n,), np.int64))
    union = np.unique(np.concatenate(lists))
    dense = []
    for g in (ga, gb):
        kd = g.upload(union)
        dd = g.upload(np.zeros((len(union), 64, 5), np.float32))
        g.pack_blocks_dev(kd.value, len(union), dd.value)
        dense.append(g.download(dd, (len(union), 64, 5), np.float32))
    total = dense[0] + dense[1]                                        # what the all-reduce leaves on every rank
    for g in (ga, gb):
        kd = g.upload(union)
        dd = g.upload(total)
        g.unpack_blocks_dev(kd.value, len(union), dd.value)
        _cmp_tables(g, o)
    ga.close()
    gb.close()


def test_block_ids_and_dense_layout_match_the_host_statement(pkg, O):
    """parallel.NumpyBlockOps (used by the world-2 gloo test) and the device kernels agree on block ids, the in-block
    voxel order and the dense (w, s, gx, gy, gz) layout."""
    seq, g, o = _mk(pkg, O, n=3, cap=19)
    for i in range(seq.n):
        g.update(*seq.frame(i))
    cap = 1 << (19 - 6)
    buf = g.upload(np.zeros(cap, np.int64))
    n = g.block_keys_dev(buf.value, cap)
    dev_ids = np.sort(g.download(buf, (n,), np.int64))
    keys, raw = g.export(sorted=True, raw=True)
    ops = pkg.parallel.NumpyBlockOps(keys, raw)
    host_ids = ops.block_keys_numpy()
    assert np.array_equal(dev_ids, host_ids)
    kd = g.upload(host_ids)
    dd = g.upload(np.zeros((n, 64, 5), np.float32))
    g.pack_blocks_dev(kd.value, n, dd.value)
    dense_dev = g.download(dd, (n, 64, 5), np.float32)
    dense_host = ops.pack_numpy(host_ids)
    assert np.array_equal(dense_dev, dense_host)
    g.close()


def test_stress_config_c3_small_slice(pkg, O):
    """BASELINE config C3 geometry (5 mm voxels, trunc 10, K scaled) on a 1280x960 frame, capacity 2^23."""
    W, H = 1280, 960
    seq = pkg.synth.Sequence("tum", W, H, n_frames=1, seed=0)
    vs = np.float32(0.005)
    T = np.float32(10) * vs
    assert float(T) == pytest.approx(0.049999997, abs=1e-9)
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=23)
    o = O.Oracle(vs, T, W, H, seq.K)
    d, R, t = seq.frame(0)
    g.update(d, R, t)
    nu, nv = o.update(d, R, t)
    st = g.stats()
    assert st["n_upd"] == nu and st["n_valid"] == nv
    _cmp_tables(g, o)
    g.close()


def test_edge_frames_empty_invalid_and_ragged(pkg, O):
    """Edge cases the path must survive: all-invalid depth, depth outside [zmin,zmax], a frame whose
    size is not a multiple of the 16x16 / 32x8 tiles, NaN-free output."""
    W, H = 77, 45
    K = pkg.synth.intrinsics(W, H)
    vs = np.float32(0.02)
    g = pkg.GradSdf(vs, np.float32(0.1), W, H, K, capacity_log2=18)
    o = O.Oracle(vs, np.float32(0.1), W, H, K)
    zero = np.zeros((H, W), np.float32)
    g.update(zero, np.eye(3), np.zeros(3))
    o.update(zero, np.eye(3), np.zeros(3))
    assert g.count() == 0 and g.stats()["frames"] == 1
    far = np.full((H, W), 7.5, np.float32)                  # >= zmax: every pixel skipped
    g.update(far, np.eye(3), np.zeros(3))
    assert g.count() == 0
    conv, pose, passes = g.track(far, np.array([0, 0, 0, 0, 0, 0, 1], np.float32), iters=3)
    assert not conv and passes == 3
    rng = np.random.default_rng(1)
    d = (1.0 + 0.3 * rng.random((H, W))).astype(np.float32)
    d[rng.random((H, W)) < 0.2] = 0.0                       # ragged holes
    R = O.quat_to_R(np.array([0.05, -0.1, 0.02, 0.99], np.float32) / np.float32(np.linalg.norm([0.05, -0.1, 0.02, 0.99])))
    t = np.array([0.3, -0.2, 0.1], np.float32)
    g.update(d, R, t)
    o.update(far, np.eye(3), np.zeros(3))
    o.update(d, R, t)
    _cmp_tables(g, o)
    assert g.stats()["frames"] == o.frame_counter() == 3
    g.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fusion_keys_bit_exact_random_poses(pkg, O, seed):
    """Random rotations / translations (oblique rays stress the rounding of all three key components)."""
    rng = np.random.default_rng(seed)
    W, H = 128, 96
    seq = pkg.synth.Sequence("tum", W, H, n_frames=1, seed=seed)
    vs = np.float32(0.015)
    T = np.float32(7) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=21)
    o = O.Oracle(vs, T, W, H, seq.K)
    d, _, _ = seq.frame(0)
    for _ in range(3):
        q = rng.standard_normal(4).astype(np.float32)
        q /= np.float32(np.linalg.norm(q))
        R = O.quat_to_R(q)
        t = rng.uniform(-3, 3, 3).astype(np.float32)
        g.update(d, R, t)
        o.update(d, R, t)
    _cmp_tables(g, o)
    g.close()


def test_vis_bitvectors_match_oracle(pkg, O):
    """vis_ (MapGradPixelSdf.cpp:113-115): bit f of a voxel <=> it was updated by integrated frame f."""
    seq, g, o = _mk(pkg, O, n=5, cap=19, step_deg=4.0)
    g.enable_vis(40)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        o.update(d, R, t)
    kg, vg = g.export_vis()
    ko, _ = o.export()
    vo = o.export_vis(2)
    assert np.array_equal(kg, ko)
    assert np.array_equal(vg, vo)
    assert (vg[:, 0] > 0).all() and vg[:, 1].max() == 0 and int(vg[:, 0].max()) < (1 << seq.n)
    g.close()


def test_raycast_matches_definition_and_input_depth(pkg, O):
    """Voxel-hash raycaster (north_star; absent from the reference): the HIP kernel against its CPU definition
    (oracle.gsdfo_raycast, built on weights()/tsdf()), and self-consistency: rendering the fused map from
    a fused pose gives the input depth back (within a voxel) and normals close to the input normals."""
    seq, g, o = _mk(pkg, O, kind="tum", W=320, H=240, vs=0.01, trunc=10, cap=21, n=4)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        o.update(d, R, t)
    d, R, t = seq.frame(1)
    zg, ng = g.raycast(R, t)
    zo, no = o.raycast(R, t)
    hit_g, hit_o = zg > 0, zo > 0
    assert (hit_g == hit_o).mean() > 0.999                      # borderline sign decisions may flip (sums differ in the last bits)
    both = hit_g & hit_o
    assert both.mean() > 0.5
    dz = np.abs(zg - zo)[both]
    assert np.percentile(dz, 99.9) <= TOL and np.median(dz) <= 1e-6
    assert np.percentile(np.abs(ng - no)[:, both], 99.9) <= 1e-3
    # self-consistency against the frame that was fused from this pose
    m = both & (d > 0.5) & (d < 3.5)
    assert np.median(np.abs(zg - d)[m]) < 0.01                  # one voxel
    n_in = g.normals(d)
    cosang = (n_in * ng).sum(axis=0)[m & np.isfinite(n_in).all(axis=0)]
    assert np.median(cosang) > 0.95
    g.close()


def test_argument_errors_of_the_newer_entries(pkg):
    """Error behaviour of entries added after the first ABI: bad arguments are reported, nothing crashes."""
    import ctypes as C
    L = pkg.binding.load()
    with pytest.raises(pkg.binding.GsdfError):
        pkg.GradSdf(np.float32(0.01), np.float32(0.1), 64, 48, pkg.synth.intrinsics(64, 48), capacity_log2=31)
    g = pkg.GradSdf(np.float32(0.02), np.float32(0.1), 64, 48, pkg.synth.intrinsics(64, 48), capacity_log2=14)
    with pytest.raises(pkg.binding.GsdfError):
        g.raycast(np.eye(3), np.zeros(3), zmin=1.0, zmax=0.5)
    z, n = g.raycast(np.eye(3), np.zeros(3))                  # empty map: no hits, no NaN
    assert (z == 0).all() and (n == 0).all()
    st = g.stats()
    assert st["n_deferred"] == 0 and st["fuse_timeouts"] == 0
    g.close()


# ---- round-2 additions: the bench workload, the stress config and the exports under test ---------------------------

def test_tracked_bench_stream_matches_oracle_frame_by_frame(pkg, O):
    """The bench workload itself (BASELINE configs[1]: S-tum 640x480, 1 cm voxels, trunc 10, 2^22), tracked, 64 frames,
    through gsdf_track_and_fuse_dev against the oracle's main_scan_3d.cpp:255-266 loop: per frame the same `converged`
    flag and the same number of Gauss-Newton passes, frame-1 pose within 1e-4.

    Frames that do NOT converge are part of the reference's behaviour on this stream and are asserted here: the
    nearest-voxel lookup with the 1.2 gradient scale (MapGradPixelSdf.h:113) makes phi discontinuous at voxel
    borders, and on some frames Gauss-Newton settles into a cycle with |xi|^2 of 5e-6 .. 5e-5, above the 1e-6
    threshold: optimize() runs all 25 passes, returns false, the frame is not fused (main_scan_3d.cpp:261) and the
    last iterate stays the start pose of the next frame (RigidOptimizer.h:64).

    What can and cannot be equal.  GPU and oracle differ in the last bits of their sums.  A frame whose iteration is
    DETERMINISTIC -- it converges in a few passes with every |xi|^2 at least 10 % away from the threshold, or it settles
    into a stable cycle (the last six |xi|^2 within 5 % of each other, all above the threshold) -- must agree exactly.
    On the other frames Gauss-Newton wanders chaotically with |xi|^2 between 1e-6 and 1e-4: whether it dips below the
    threshold within 25 passes is decided by last-bit noise (MEASURED between the reference's own serial and OMP builds,
    whose reductions differ: tests/test_oracle_serial_vs_omp.py -- 2 of 47 frames of this stream end differently from
    identical state), and after such a frame the two runs hold different maps.  So: strict equality on every frame up to
    the first one that differs -- which must be a non-deterministic one, i.e. it cannot come before the first frame the
    oracle itself classifies as sensitive -- and agreement in the aggregate over the whole stream."""
    W, H, n = 640, 480, 64
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01); T = np.float32(10) * vs
    frames = [seq.frame(i) for i in range(n)]
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    o = O.Oracle(vs, T, W, H, seq.K, threads=1)
    d0, R0, t0 = frames[0]
    p0 = pose7_from(O, R0, t0)
    R0q = O.quat_to_R(p0[3:])
    g.update(d0, R0q, t0); o.update(d0, R0q, t0)
    g.set_pose(p0)
    dev = [g.upload(f[0]) for f in frames]
    for i in range(1, n):
        g.track_and_fuse_dev(dev[i])
    g.sync()
    log = g.frame_log()
    assert log.shape == (n - 1, 10)
    po = p0.copy()
    strict = True
    n_strict = strict_not_conv = 0
    first_sensitive = None                                # first frame whose iteration the oracle's own |xi|^2 series marks as non-deterministic
    conv_o, err_o, err_g = [], [], []
    for i in range(1, n):
        co, po, used, trace, _ = o.track(frames[i][0], po)
        if co:
            o.update(frames[i][0], O.quat_to_R(po[3:]), po[:3])
        xi2 = trace[:used, 35]
        clean = co and used <= 6 and bool((np.abs(xi2 / 1e-6 - 1.0) >= 0.1).all())
        cycle = (not co) and used == 25 and xi2[-6:].min() > 2e-6 and xi2[-6:].max() <= 1.05 * xi2[-6:].min()
        if first_sensitive is None and not (clean or cycle):
            first_sensitive = i
        if strict:
            same = bool(log[i - 1, 7]) == co and int(log[i - 1, 8]) == used
            if same:
                assert np.abs(log[i - 1, :7] - po).max() <= TOL * i, "frame %d pose" % i
                n_strict += 1
                strict_not_conv += int(not co)
            else:
                # the first frame that differs must be a sensitive one; from here on only the aggregate is compared
                assert not (clean or cycle), "frame %d: converged %d / %d passes vs oracle %d / %d, |xi|^2 %s" % (
                    i, log[i - 1, 7], log[i - 1, 8], co, used, xi2)
                strict = False
        conv_o.append(co)
        gt = frames[i][2]
        if co:
            err_o.append(np.abs(po[:3] - gt).max())
        if log[i - 1, 7]:
            err_g.append(np.abs(log[i - 1, :3] - gt).max())
    conv_o = np.array(conv_o)
    conv_g = log[:, 7] > 0
    # every deterministic frame in front of the first sensitive one agreed exactly (the assertion inside the loop), so:
    print("MEASURED n_strict %d first_sensitive %s strict_not_conv %d flags_equal %.3f conv_g %d conv_o %d" % (
        n_strict, first_sensitive, strict_not_conv, (conv_g == conv_o).mean(), conv_g.sum(), conv_o.sum()))
    assert first_sensitive is not None and n_strict >= first_sensitive - 1, (n_strict, first_sensitive)
    # ... and an absolute floor near the measured value (the relative bound alone would follow a regression down)
    assert n_strict >= N_STRICT_FLOOR and strict_not_conv >= 1         # ... among them frames that run all 25 passes and are not fused
    assert not bool(log[0, 7]) and int(log[0, 8]) == 25   # frame 1: one fused frame in the map, 25 passes, not fused
    # the whole stream in the aggregate
    # the yardstick: the reference's own serial and OMP builds, each free-running on this stream, give the same flag on 61 of 63
    # frames (0.968) and converge on 47 / 49 of them (profiles/r05_serial_vs_omp.txt); the engine against the serial oracle:
    # 61 of 63, 49 / 47 (round 5; the engine's float atomics make chaotic frames vary a little from run to run).  Allowed: 57 of 63.
    assert (conv_g == conv_o).mean() >= 0.9, (conv_g == conv_o).mean()
    assert abs(int(conv_g.sum()) - int(conv_o.sum())) <= 5
    assert (~conv_g).sum() >= 5 and (~conv_o).sum() >= 5  # both runs hit the non-converging stretch of the stream
    assert max(err_g) < 0.012 and max(err_o) < 0.012      # every fused frame within ~1 voxel of the ground truth
    st = g.stats()
    assert st["frames"] == 1 + int(log[:, 7].sum())       # Sdf::counter_ counts the setup frame + the converged ones
    g.close()


def test_stress_config_c3_full_capacity(pkg, O):
    """BASELINE configs[2] as configured: 1280x960, 5 mm voxels, trunc 10, hash capacity 2^25 (1 GiB of voxel records,
    beyond the Infinity Cache): 3 frames fused (counters exact, map against the oracle) and one optimize() at 1280x960."""
    W, H, n = 1280, 960, 4
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.005); T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=25)
    o = O.Oracle(vs, T, W, H, seq.K, threads=1)
    nu = nv = 0
    for i in range(3):
        d, R, t = seq.frame(i)
        g.update(d, R, t)
        a, b = o.update(d, R, t)
        nu += a; nv += b
    st = g.stats()
    assert st["n_upd"] == nu and st["n_valid"] == nv and st["frames"] == 3
    assert st["fuse_timeouts"] == 0
    assert _cmp_tables(g, o) > 3000000
    d3, R3, t3 = seq.frame(3)
    p = pose7_from(O, seq.frame(2)[1], seq.frame(2)[2])
    cg, pg, passes = g.track(d3, p, iters=6)
    co, po, used, _, _ = o.track(d3, p, iters=6)
    assert cg == co and passes == used
    assert np.abs(pg - po).max() <= TOL
    g.close()


def _mc_case_tables():
    """include/gsdf_mc_tables.h parsed as data."""
    import os, re
    from conftest import ROOT
    txt = open(os.path.join(ROOT, "include", "gsdf_mc_tables.h")).read()
    e = re.search(r"GSDF_MC_EDGE_TABLE\[256\] = \{(.*?)\};", txt, re.S).group(1)
    t = re.search(r"GSDF_MC_TRI_TABLE\[256 \* 16\] = \{(.*?)\};", txt, re.S).group(1)
    edge = np.array([int(v, 16) for v in re.findall(r"0x[0-9a-f]+", e)])
    tri = np.array([int(v) for v in re.findall(r"-?\d+", t)]).reshape(256, 16)
    return edge, tri


def test_device_marching_cubes_is_the_reference_sweep(pkg, O):
    """gsdf_extract_mesh (one lane per voxel through the block map) against the ORACLE's restatement of
    LayeredMarchingCubesNoColor::computeIsoSurface (two-layer sweep, classic edgeTable / triTable) on exactly the same
    voxel values (the GPU's exported map loaded into the oracle): bit-identical triangle list, same order."""
    for kind, W, H, vs in (("spheres", 160, 120, 0.02), ("tum", 320, 240, 0.01)):
        seq = pkg.synth.Sequence(kind, W, H, n_frames=4, seed=2)
        vsf = np.float32(vs)
        g = pkg.GradSdf(vsf, np.float32(5) * vsf, W, H, seq.K, capacity_log2=21)
        for i in range(seq.n):
            g.update(*seq.frame(i))
        keys, pay = g.export(sorted=True)
        o = O.Oracle(vsf, np.float32(5) * vsf, W, H, seq.K)
        o.set_map(keys, pay)
        tg = g.extract_mesh()
        to = o.extract_mesh()
        assert tg.shape == to.shape and tg.shape[0] > 1000
        assert np.array_equal(tg.view(np.uint32), to.view(np.uint32))
        # the oracle-fused map gives (nearly) the same surface: same face count within 1 %
        o2 = O.Oracle(vsf, np.float32(5) * vsf, W, H, seq.K)
        for i in range(seq.n):
            o2.update(*seq.frame(i))
        assert abs(o2.extract_mesh().shape[0] - tg.shape[0]) <= 0.01 * tg.shape[0]
        g.close()


def test_extract_pc_rows_match_oracle(pkg, O, tmp_path):
    """MapGradPixelSdf::extract_pc (MapGradPixelSdf.cpp:177-220) of the facade -- through the C wrapper the C4 runner
    uses -- against the oracle's restatement on the same voxel values: same rows, same order."""
    import ctypes, os, subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "gradient-sdf_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    seq = pkg.synth.Sequence("spheres", 160, 120, n_frames=16, seed=5, step_deg=0.5)
    vs = np.float32(0.02)
    g = pkg.GradSdf(vs, np.float32(5) * vs, 160, 120, seq.K, capacity_log2=19)
    for i in range(seq.n):
        g.update(*seq.frame(i))
    hl = ctypes.CDLL(os.path.join(host, "libgsdf_host.so"))
    hl.gsdf_host_extract_pc.restype = ctypes.c_long
    hl.gsdf_host_extract_pc.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_char_p]
    path = str(tmp_path / "cloud.ply")
    n = hl.gsdf_host_extract_pc(g.h, ctypes.c_float(vs), path.encode())
    keys, pay = g.export(sorted=True)
    o = O.Oracle(vs, np.float32(5) * vs, 160, 120, seq.K)
    o.set_map(keys, pay)
    rows = o.extract_pc()
    assert n == rows.shape[0] and n > 50
    lines = open(path).read().split("\n")
    assert lines[0] == "ply" and lines[2] == "element vertex %d" % n
    got = np.array([[float(v) for v in ln.split()] for ln in lines[10:10 + n]], np.float64)
    # the file holds 6 significant digits (operator<< of float, as in the reference)
    assert np.allclose(got, rows, rtol=6e-6, atol=1e-9)
    g.close()


def test_get_voxels_returns_the_stored_record(pkg, O):
    """gsdf_get_voxels = tsdf_.at(idx) (getSdf, MapGradPixelSdf.h:127-129): dist, RAW gradient sum, weight of a voxel,
    exactly the row of the export; missing voxels are flagged."""
    seq, g, o = _mk(pkg, O, n=3)
    for i in range(seq.n):
        g.update(*seq.frame(i))
    keys, pay = g.export(sorted=True)
    rng = np.random.default_rng(3)
    sel = rng.integers(0, len(keys), 500)
    probe = np.concatenate([keys[sel], np.array([[10000, 10000, 10000], [-(1 << 21), 0, 0]], np.int32)])
    got, found = g.get_voxels(probe)
    assert found[:500].all() and not found[500:].any()
    assert np.array_equal(got[:500].view(np.uint32), pay[sel].view(np.uint32))
    assert (got[500:] == 0).all()
    g.close()


def _contention_worker(rank, out_dir, n_rounds):
    """One of two processes that fuse the same stream on the same GPU at the same time (own context each)."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    pkg = graft.package()
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=4, seed=1)
    vs = np.float32(0.01)
    g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22)
    frames = [seq.frame(i) for i in range(seq.n)]
    dev = [g.upload(f[0]) for f in frames]
    open(os.path.join(out_dir, "ready%d" % rank), "w").close()
    import time
    t0 = time.time()
    while not all(os.path.exists(os.path.join(out_dir, "ready%d" % r)) for r in (0, 1)) and time.time() - t0 < 120:
        time.sleep(0.001)
    keys = pay = None
    timeouts = deferred = 0
    for rnd in range(n_rounds):                       # many rounds: the two processes' kernels interleave on the device
        g.reset()
        for d, f in zip(dev, frames):
            g.update_dev(d, f[1], f[2])
        g.sync()
        st = g.stats()
        timeouts += st["fuse_timeouts"]; deferred += st["n_deferred"]
        k, p = g.export(sorted=True)
        if keys is None:
            keys, pay = k, p
        else:                                          # every round must give the same map (sums up to float-atomic order)
            assert np.array_equal(k, keys)
            assert np.abs(p - pay).max() <= 1e-4 * max(1.0, float(np.abs(pay).max()))
    np.savez(os.path.join(out_dir, "proc%d.npz" % rank), keys=keys, pay=pay, timeouts=timeouts, deferred=deferred)
    g.close()


def test_two_processes_fuse_on_one_gpu_at_the_same_time(pkg, O, tmp_path):
    """Two processes share the GPU: each tile hand-off of the ordered flush now competes with a foreign kernel for the CUs
    (waits get longer, and a wait that expires sends its tile through the deferred list -- the forced form of that path is
    test_fusion_forced_paths_match_oracle[8192]).  Whatever the interleaving, both maps must equal the oracle's."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_contention_worker, args=(r, str(tmp_path), 12)) for r in (0, 1)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=4, seed=1)
    vs = np.float32(0.01)
    o = O.Oracle(vs, np.float32(10) * vs, W, H, seq.K)
    for i in range(seq.n):
        o.update(*seq.frame(i))
    ko, po = o.export()
    for r in (0, 1):
        z = np.load(tmp_path / ("proc%d.npz" % r))
        assert np.array_equal(z["keys"], ko), "voxel key sets differ"
        assert np.abs(z["pay"][:, 0] - po[:, 0]).max() <= TOL
        scale = np.maximum(1.0, po[:, 4])
        assert (np.abs(z["pay"][:, 4] - po[:, 4]) / scale).max() <= TOL
        assert (np.abs(z["pay"][:, 1:4] - po[:, 1:4]).max(axis=1) / scale).max() <= TOL
        print("process %d: %d hand-off waits expired, %d deferred contributions over 12 rounds" % (r, int(z["timeouts"]), int(z["deferred"])))


def test_full_size_properties_order_and_sharding(pkg):
    """Size-independent properties at the bench configuration (S-tum 640x480, 1 cm voxels, capacity 2^22), where the CPU oracle
    would take minutes: 40 frames fused forwards, backwards, and as two frame shards merged additively (gsdf_export raw sums ->
    gsdf_merge_raw) give ONE map -- key sets bit-identical, sums within float
    rounding of their different orders -- and exactly the same counters (every sample counted once wherever it is fused)."""
    W, H, n = 640, 480, 40
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01)
    T = np.float32(10) * vs
    frames = [seq.frame(i) for i in range(n)]

    def fuse(order):
        g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
        for i in order:
            g.update(*frames[i])
        return g

    fwd, bwd = fuse(range(n)), fuse(range(n - 1, -1, -1))
    kf, pf = fwd.export(sorted=True, raw=True)
    kb, pb = bwd.export(sorted=True, raw=True)
    assert len(kf) > 1_000_000 and np.array_equal(kf, kb)
    scale = np.maximum(1.0, pf[:, 4:5])
    assert (np.abs(pf - pb) / scale).max() <= 1e-5
    sf, sb = fwd.stats(), bwd.stats()
    assert sf["n_upd"] == sb["n_upd"] and sf["n_valid"] == sb["n_valid"] and sf["frames"] == n
    # two shards (even / odd frames), merged additively through the raw export
    a, b = fuse(range(0, n, 2)), fuse(range(1, n, 2))
    sa, sb2 = a.stats(), b.stats()
    assert sa["n_upd"] + sb2["n_upd"] == sf["n_upd"] and sa["n_valid"] + sb2["n_valid"] == sf["n_valid"]
    a.merge_raw(*b.export(raw=True))
    ka, pa = a.export(sorted=True, raw=True)
    assert np.array_equal(ka, kf)
    assert (np.abs(pa - pf) / scale).max() <= 1e-5
    for g in (fwd, bwd, a, b):
        g.close()


@pytest.mark.gpu
def test_pipelined_gt_pose_fusion_is_invisible_except_in_time(pkg, O):
    """gsdf_update_dev defers the launch of a frame's fusion until the next frame arrives (whose normals that launch computes in
    its tail).  Whatever the caller does in between must see the map WITH the waiting frame: counts, exports, stats, tracking,
    raycasts; a reset drops it together with the map; the frame counter and vis_ bits follow the call order."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("tum", W, H, n_frames=6, seed=0)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    fr = [seq.frame(i) for i in range(6)]
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
    g.enable_vis(8)
    o = O.Oracle(vs, T, W, H, seq.K)
    dev = [g.upload(f[0]) for f in fr]
    counts = []
    for i in range(4):
        g.update_dev(dev[i], fr[i][1], fr[i][2])
        o.update(*fr[i])
        if i in (0, 2):                               # a query between two frames: the waiting fusion is launched first
            assert g.count() == o.count()
            assert g.stats()["frames"] == i + 1
        counts.append(o.count())
    kg, pg = g.export(sorted=True)
    ko, po = o.export()
    assert np.array_equal(kg, ko) and np.abs(pg[:, 0] - po[:, 0]).max() <= TOL
    kv, vis = g.export_vis()
    assert np.array_equal(kv, ko) and np.array_equal(vis, o.export_vis(1))
    # tracking right after a deferred fusion reads the map with it
    g.update_dev(dev[4], fr[4][1], fr[4][2])
    o.update(*fr[4])
    p0 = pose7_from(O, fr[4][1], fr[4][2])
    cg, pose_g, passes_g = g.track(fr[5][0], p0, iters=3)
    co, pose_o, passes_o, _, _ = o.track(fr[5][0], p0, iters=3)
    assert passes_g == passes_o and np.abs(pose_g - pose_o).max() <= 5 * TOL
    # ... and so does a raycast (whose block filters are rebuilt because the map changed)
    g.update_dev(dev[5], fr[5][1], fr[5][2])
    o.update(*fr[5])
    zg, _ = g.raycast(fr[5][1], fr[5][2])
    zo, _ = o.raycast(fr[5][1], fr[5][2])
    assert ((zg > 0) == (zo > 0)).mean() > 0.999
    # a reset drops the frame that was still waiting together with the map
    g.update_dev(dev[0], fr[0][1], fr[0][2])
    g.reset()
    assert g.count() == 0 and g.stats()["frames"] == 0
    g.update_dev(dev[0], fr[0][1], fr[0][2])
    assert g.count() == counts[0]
    g.close()


def _quat_to_R_f32(q):
    """Eigen's toRotationMatrix in float32, operation by operation as the library's gsdf_quat_to_R (csrc/gsdf_math.h) does it"""
    f = np.float32
    x, y, z, w = (f(v) for v in q)
    tx, ty, tz = f(2) * x, f(2) * y, f(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = f(1)
    return np.array([one - (tyy + tzz), txy - twz, txz + twy, txy + twz, one - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, one - (txx + tyy)], np.float32).reshape(3, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 2, 3, 25])
def test_frame_loop_call_equals_optimize_then_update(pkg, O, iters):
    """gsdf_track_and_fuse_dev (one call per frame: the frame's normals tiles ride in its first tracker launches -- two of them
    when optimize() is limited to one pass, three otherwise -- and the fusion is gated on the device) leaves the same pose and
    the same map, bit for bit, as the two blocking calls optimize() + update() it stands for (main_scan_3d.cpp:258-263), at any
    iteration limit; frames that do not converge within the limit are not fused by either."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("tum", W, H, n_frames=5, seed=0)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    fr = [seq.frame(i) for i in range(5)]
    res = []
    for one_call in (True, False):
        g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
        g.update(*fr[0])
        pose = pose7_from(O, fr[0][1], fr[0][2])
        g.set_pose(pose)
        flags = []
        for i in range(1, 5):
            if one_call:
                g.track_and_fuse_dev(g.upload(fr[i][0]), iters=iters)
                g.sync()
                row = g.frame_log()[-1]
                flags.append((int(row[7]), int(row[8])))
                pose = g.get_pose()
            else:
                conv, pose, passes = g.track(fr[i][0], pose, iters=iters)
                flags.append((int(conv), int(passes)))
                if conv:
                    g.update(fr[i][0], _quat_to_R_f32(pose[3:]), pose[:3])
        k, p = g.export(sorted=True)
        res.append((flags, np.array(pose, np.float32), k, p))
        g.close()
    (f1, p1, k1, v1), (f2, p2, k2, v2) = res
    assert f1 == f2, (f1, f2)
    assert np.array_equal(p1, p2)
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)


@pytest.mark.gpu
def test_staging_ahead_of_the_stream(pkg, O):
    """gsdf_dev_upload_ahead / gsdf_upload_wait (the CLI's frame staging): copies started on the library's copy stream while
    earlier frames are still being fused arrive intact, in order, and the frames fused from them give the map of the blocking
    path; waiting twice, or for an id of the past, returns at once."""
    import ctypes as C
    W, H = 320, 240
    seq = pkg.synth.Sequence("tum", W, H, n_frames=6, seed=1)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    fr = [seq.frame(i) for i in range(6)]
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
    ref = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
    L = g.L
    nbytes = W * H * 4
    host, dev, ids = [], [], []
    for i in range(6):
        hp, dp = C.c_void_p(), C.c_void_p()
        assert L.gsdf_host_alloc(g.h, C.byref(hp), nbytes) == 0 and L.gsdf_dev_alloc(g.h, C.byref(dp), nbytes) == 0
        C.memmove(hp, np.ascontiguousarray(fr[i][0], np.float32).ctypes.data, nbytes)
        host.append(hp); dev.append(dp)
    for i in range(6):                                  # all copies in flight before the first fusion is even enqueued
        uid = C.c_int64(0)
        assert L.gsdf_dev_upload_ahead(g.h, dev[i], host[i], nbytes, C.byref(uid)) == 0
        ids.append(uid.value)
    assert ids == sorted(ids) and len(set(ids)) == 6
    for i in range(6):
        assert L.gsdf_upload_wait(g.h, ids[i]) == 0
        g.update_dev(dev[i], fr[i][1], fr[i][2])
        ref.update(*fr[i])
    assert L.gsdf_upload_wait(g.h, ids[2]) == 0 and L.gsdf_upload_wait(g.h, ids[5]) == 0
    kg, pg = g.export(sorted=True)
    kr, pr = ref.export(sorted=True)
    assert np.array_equal(kg, kr) and np.array_equal(pg, pr)
    for hp, dp in zip(host, dev):
        assert L.gsdf_dev_free(g.h, dp) == 0 and L.gsdf_host_free(g.h, hp) == 0
    g.close(); ref.close()


@pytest.mark.gpu
def test_one_staging_buffer_reused_across_gt_pose_fusions(pkg, O, monkeypatch):
    """upload(buf) -> update_dev(buf) -> upload(buf) -> update_dev(buf) ... on ONE device buffer: gsdf_update_dev keeps the
    fusion of a frame back until its successor arrives, so the copy of the next frame into the same buffer must launch the
    waiting fusion first (gsdf_dev_upload / gsdf_dev_upload_async do, when their destination overlaps its depth image).
    The map must equal the one of a context that launches every fusion at once (GSDF_DEFER=0), bit for bit, and the oracle's;
    a copy ahead of the stream into that buffer (which no stream order protects) is refused."""
    import ctypes as C
    W, H, n = 320, 240, 5
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=2)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    fr = [seq.frame(i) for i in range(n)]
    nbytes = W * H * 4
    maps = []
    for defer, entry in (("1", "gsdf_dev_upload"), ("1", "gsdf_dev_upload_async"), ("0", "gsdf_dev_upload")):
        monkeypatch.setenv("GSDF_DEFER", defer)
        g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=20)
        buf = C.c_void_p()
        assert g.L.gsdf_dev_alloc(g.h, C.byref(buf), nbytes) == 0
        keep = []
        for d, R, t in fr:
            a = np.ascontiguousarray(d, np.float32)
            keep.append(a)                                  # the async copy reads the host array until the stream has passed it
            assert getattr(g.L, entry)(g.h, buf, a.ctypes.data_as(C.c_void_p), nbytes) == 0
            g.update_dev(buf, R, t)
        if defer == "1":                                    # a fusion waits: a copy AHEAD of the stream into its depth image is an error
            uid = C.c_int64(0)
            assert g.L.gsdf_dev_upload_ahead(g.h, buf, keep[0].ctypes.data_as(C.c_void_p), nbytes, C.byref(uid)) != 0
            assert "has not run yet" in g.L.gsdf_last_error().decode()
        g.sync()
        maps.append(g.export(sorted=True, raw=True))
        assert g.stats()["frames"] == n
        assert g.L.gsdf_dev_free(g.h, buf) == 0
        g.close()
    (k0, p0), (k1, p1), (k2, p2) = maps
    assert np.array_equal(k0, k2) and np.array_equal(p0.view(np.uint32), p2.view(np.uint32))
    assert np.array_equal(k1, k2) and np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
    o = O.Oracle(vs, T, W, H, seq.K)
    for d, R, t in fr:
        o.update(d, R, t)
    assert np.array_equal(k0, o.export()[0])


def _lockstep(pkg, O, g, o, frames, pose):
    """tests/lockstep.py: engine and oracle on identical state, frame by frame (the rules are stated there)."""
    return lockstep(O, g, o, frames, pose)


@pytest.mark.gpu
def test_c1_every_frame_from_identical_state(pkg, O):
    """BASELINE configs[0] (30-frame RenderSpheres-style sequence, 640x480, 1 cm voxels, trunc 10, tracked), engine and oracle in
    lockstep (_lockstep): all 29 optimize() calls from identical state at 1e-4.  The free-running comparison of the two
    trajectories (where a borderline frame leaves them a threshold-sized step apart) is tests/test_host.py's."""
    W, H, n = 640, 480, 30
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=n, seed=0, step_deg=0.5)
    vs = np.float32(0.01)
    T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    o = O.Oracle(vs, T, W, H, seq.K)
    pose = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    depth = lambda i: seq.depth_u16(i).astype(np.float32) * np.float32(0.001)
    g.update(depth(0), np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    o.update(depth(0), np.eye(3), np.zeros(3))
    n_conv, n_long, flips = _lockstep(pkg, O, g, o, ((i, depth(i)) for i in range(1, n)), pose)
    # frames whose stop test falls differently: no more than between the reference's own serial and OMP builds on this stream
    # (measured in the same harness: tests/test_oracle_serial_vs_omp.py, profiles/r05_serial_vs_omp.txt) plus one
    assert len(flips) <= SERIAL_VS_OMP_FLIPS["c1"] + 1, flips
    assert n_conv >= 20
    assert _cmp_tables(g, o) > 100000                             # and the maps stayed the same: keys bit-exact, values 1e-4
    g.close()


@pytest.mark.gpu
def test_bench_stream_every_frame_from_identical_state(pkg, O):
    """The bench stream (BASELINE configs[1]: S-tum 640x480, 1 cm voxels, trunc 10, 2^22), 48 frames, engine and oracle in
    lockstep (_lockstep): the 1e-4 bar on EVERY frame as it stands -- not TOL x frame number as in the free-running stream tests
    above -- including the stretch whose frames run all 25 passes and are not fused (there the comparison is after 4 passes and
    at the decision; the limit cycle itself is described in test_tracked_bench_stream_matches_oracle_frame_by_frame)."""
    W, H, n = 640, 480, 48
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01); T = np.float32(10) * vs
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=22)
    o = O.Oracle(vs, T, W, H, seq.K, threads=1)
    d0, R0, t0 = seq.frame(0)
    pose = pose7_from(O, R0, t0)
    R0q = O.quat_to_R(pose[3:])
    g.update(d0, R0q, t0); o.update(d0, R0q, t0)
    n_conv, n_long, flips = _lockstep(pkg, O, g, o, ((i, seq.frame(i)[0]) for i in range(1, n)), pose)
    assert n_conv >= 25 and n_long >= 3, (n_conv, n_long)
    # flips: no more than the reference's own serial and OMP builds show against each other on this stretch (2 of 47 frames, one
    # of them a 25-pass frame that the other build ends after 12: tests/test_oracle_serial_vs_omp.py) plus one
    assert len(flips) <= SERIAL_VS_OMP_FLIPS["bench"] + 1, flips
    assert _cmp_tables(g, o) > 500000
    g.close()


@pytest.mark.gpu
def test_grow_keeps_the_map_and_auto_grow_removes_table_full(pkg, O, monkeypatch):
    """The reference's map grows without bound (MapGradPixelSdf.h:65-68).  gsdf_grow: every block moves into a larger table --
    voxels, vis_ bit-vectors and the frame counter unchanged bit for bit, fusion and tracking go on as if nothing had happened.
    gsdf_set_auto_grow: a scan that overflows its table (GSDF_ERR_TABLE_FULL without it) completes and equals the oracle's.
    The scene: the S-tum room at 160x120 / 2 cm voxels with twelve times the motion -- 2688 blocks after the first frame, 4395
    after the fourteenth; a table of 2^18 records has 4096."""
    W, H, n = 160, 120, 14
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=4, motion=12)
    vs = np.float32(0.02); T = np.float32(5) * vs
    fr = [seq.frame(i) for i in range(n)]
    o = O.Oracle(vs, T, W, H, seq.K)
    # 1. explicit growth in the middle of a scan
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=18)
    g.enable_vis(n)
    for d, R, t in fr[:3]:
        g.update(d, R, t); o.update(d, R, t)
    k0, p0 = g.export(sorted=True, raw=True)
    v0 = g.export_vis()[1]
    g.grow(20)
    k1, p1 = g.export(sorted=True, raw=True)
    assert np.array_equal(k0, k1) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32)) and np.array_equal(v0, g.export_vis()[1])
    assert g.stats()["frames"] == 3
    with pytest.raises(pkg.GsdfError):
        g.grow(20)                                              # not larger
    pose = pose7_from(O, fr[2][1], fr[2][2])
    cg, pg, passes = g.track(fr[3][0], pose, iters=3)          # the tracker reads the moved blocks
    co, po, used, _, _ = o.track(fr[3][0], pose, iters=3)
    assert passes == used and np.abs(pg - po).max() < TOL
    for d, R, t in fr[3:]:
        g.update(d, R, t); o.update(d, R, t)
    _cmp_tables(g, o)
    assert np.array_equal(g.export_vis()[1], o.export_vis((n + 31) // 32))
    n_vox = g.count()
    g.close()
    # 2. the same scan into the table that is too small
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=18)
    with pytest.raises(pkg.GsdfError) as e:
        for d, R, t in fr:
            g.update(d, R, t)
    assert e.value.code == pkg.binding.ERR_TABLE_FULL
    g.close()
    # 3. auto-grow with its defaults and NO synchronisation between the frames: the host queues the whole scan
    # without waiting; the library counts the blocks behind every frame, knows how old the count it sees is, and waits by
    # itself where the estimate (count + lag x growth) comes near the limit
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=18)
    g.set_auto_grow(22)
    dev = [g.upload(f[0]) for f in fr]
    for dptr, (d, R, t) in zip(dev, fr):
        g.update_dev(dptr, R, t)
    g.sync()
    assert g.count() == n_vox
    _cmp_tables(g, o)
    assert g.capacity_log2() > 18
    g.close()
    # 4. a sticky TABLE_FULL does not stop an explicit gsdf_grow (the error itself stays: samples were dropped)
    g = pkg.GradSdf(vs, T, W, H, seq.K, capacity_log2=18)
    with pytest.raises(pkg.GsdfError):
        for d, R, t in fr:
            g.update(d, R, t)
    g.grow(20)
    assert g.capacity_log2() == 20
    with pytest.raises(pkg.GsdfError) as e:
        g.sync()
    assert e.value.code == pkg.binding.ERR_TABLE_FULL
    g.close()


@pytest.mark.gpu
def test_where_the_gated_fusion_is_queued_is_invisible(pkg, O, monkeypatch):
    """The frame loop queues the device-gated fusion behind the first and the last batch of tracker launches and, for a frame
    that needs more batches, once the progress word says that optimize() has ended (GSDF_LAZY_FUSE=1, the default) -- or behind
    every batch (0).  Pass counts, convergence flags and the key set are identical, poses and sums agree to the last bits, on a stretch of
    the bench stream with frames that converge late and frames that never do."""
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=56, seed=0)
    vs = np.float32(0.01)
    frames = [seq.frame(i) for i in range(seq.n)]
    out = []
    for lazy in ("1", "0"):
        monkeypatch.setenv("GSDF_LAZY_FUSE", lazy)
        g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22)
        d0, R0, t0 = frames[0]
        p = pose7_from(O, R0, t0)
        g.update(d0, O.quat_to_R(p[3:]), t0)
        g.set_pose(p)
        dev = [g.upload(f[0]) for f in frames]
        for i in range(1, seq.n):
            g.track_and_fuse_dev(dev[i])
        g.sync()
        log = g.frame_log().copy()
        keys, pay = g.export(sorted=True)
        out.append((log, keys, pay))
        g.close()
    (la, ka, pa), (lb, kb, pb) = out
    # (not bit for bit between two RUNS: the handful of deferred contributions of a launch are float atomics, whose order is free)
    assert np.array_equal(la[:, 7:], lb[:, 7:])                               # converged flags, pass counts
    assert np.abs(la[:, :7] - lb[:, :7]).max() <= 1e-6
    assert np.array_equal(ka, kb)
    assert np.abs(pa - pb).max() <= 1e-5 * max(1.0, float(np.abs(pa).max()))
    conv = la[:, 7] != 0
    assert 0 < conv.sum() < len(conv), "the stretch should hold converged and non-converged frames"


def test_tracker_head_takes_the_exact_forms_outside_the_fast_range(pkg, O):
    """The tracker's head computes llt().solve and SE3::exp in <= 1-ulp forms for the common case (rotation step below 0.1 rad,
    positive pivots) and in the exact forms otherwise.  A start pose 0.2 rad off makes the first Gauss-Newton step rotate by more
    than 0.1 rad: the exact SE3::exp path (full-range sinf / cosf); an empty map makes every pivot zero: the exact
    llt path (Eigen stops at the pivot and solves on what it has: NaN, SURVEY gotcha 9).  One pass each, against the oracle."""
    seq, g, o = _mk(pkg, O, kind="tum", W=320, H=240, vs=0.02, trunc=5, cap=20, n=3, seed=2)
    d0, R0, t0 = seq.frame(0)
    p0 = pose7_from(O, R0, t0)
    cg, pg, passes = g.track(d0, p0, iters=1)                   # empty map: zero pivots
    co, po, used, _, _ = o.track(d0, p0, iters=1)
    assert passes == used == 1 and np.array_equal(pg, po) and not cg and not co
    R0q = O.quat_to_R(p0[3:])
    g.update(d0, R0q, t0); o.update(d0, R0q, t0)
    start = O.se3_exp_mul(np.array([0.01, -0.01, 0.0, 0.0, 0.2, 0.0], np.float32), p0)
    cg, pg, passes = g.track(d0, start, iters=1)
    co, po, used, trace, _ = o.track(d0, start, iters=1)
    xi = trace[0, 29:35]
    assert float(np.linalg.norm(xi[3:])) > 0.1, xi                # the step really is outside the fast range
    assert passes == used == 1
    assert np.abs(pg[:3] - po[:3]).max() < TOL and np.abs(np.abs(pg[3:]) - np.abs(po[3:])).max() < TOL, (pg, po)
    g.close()


def test_tracker_fast_head_stays_within_ulps_of_the_exact_head(pkg, O):
    """the head's common case (rcp / rsq + one Newton step, FMA dots, polynomial sin / cos) is a <= 1-ulp form per
    operation, not the oracle's exact one.  Test build, tracker debug bit 4 = every head takes the exact forms: from the same
    start pose and map the two heads must give the same pass count and poses within a few ulps of the pose, pass by pass --
    so that later drift in the fast forms is caught here and not as a mysterious flip in a stream test."""
    L = pkg.binding.load_test_lib()
    seq, g, o = _mk(pkg, O, kind="tum", W=640, H=480, vs=0.01, trunc=10, cap=21, n=3, seed=0, lib=L)
    d0, R0, t0 = seq.frame(0)
    g.update(d0, R0, t0)
    d1, _, _ = seq.frame(1)
    p0 = pose7_from(O, R0, t0)
    worst = 0.0
    for iters in (1, 2, 3, 4, 25):
        g.debug_flags(0)
        cf, pf, nf = g.track(d1, p0, iters=iters)
        g.debug_flags(4 << 16)
        ce, pe, ne = g.track(d1, p0, iters=iters)
        assert (cf, nf) == (ce, ne), (iters, cf, nf, ce, ne)
        ulp = np.spacing(np.maximum(np.abs(pe), np.float32(1.0)).astype(np.float32))
        worst = max(worst, float((np.abs(pf - pe) / ulp).max()))
    print("MEASURED fast-vs-exact head: worst %.1f ulp(max(1, |pose|))" % worst)
    # each pass adds a few 1-ulp operations on top of sums that agree to ~1e-7 relative (the f64 group atomics' order is free, so
    # two runs of the SAME head differ by a few ulp already); measured on MI355X in round 6: 22 ulp over the five runs
    assert worst <= 64.0, worst
    g.debug_flags(0)
    g.close()


@pytest.mark.gpu
def test_next_depth_hint_is_invisible_except_in_time(pkg, O):
    """gsdf_hint_next_depth_dev (VERDICT r5 #2b): with the hint the NEXT frame's normals are computed in the tail of THIS frame's
    fusion launch (also when this frame is not fused: the gate does not apply to those workgroups) and the next frame's tracker
    launches carry no normals tiles.  The normals depend on the depth alone, so pass counts, convergence flags and the key set
    must be identical and poses / sums agree to the last bits -- on a stretch of the bench stream with frames that converge and
    frames that do not; a hint that names another buffer, a withdrawn one, and a hinted buffer overwritten through the staging
    entry are ignored without harm."""
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=40, seed=0)
    vs = np.float32(0.01)
    frames = [seq.frame(i) for i in range(seq.n)]
    out = []
    for mode in ("plain", "hint", "hint-abused"):
        g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22)
        d0, R0, t0 = frames[0]
        p = pose7_from(O, R0, t0)
        g.update(d0, O.quat_to_R(p[3:]), t0)
        g.set_pose(p)
        dev = [g.upload(f[0]) for f in frames]
        spare = g.upload(frames[5][0])
        for i in range(1, seq.n):
            if mode != "plain" and i + 1 < seq.n:
                nxt = dev[i + 1]
                if mode == "hint-abused":
                    if i % 5 == 1:
                        nxt = spare                                  # names a frame that will not come next: ignored at the next call
                    elif i % 5 == 2:
                        g.hint_next_depth(nxt); nxt = None           # given and withdrawn
                g.hint_next_depth(nxt)
            g.track_and_fuse_dev(dev[i])
            if mode == "hint-abused" and i % 5 == 3 and i + 1 < seq.n:
                # the hinted frame is overwritten (with its own contents) through the staging entry after its normals were queued:
                # the precomputed normals are forgotten, the frame's own tracker launches compute them again
                nxt_host = np.ascontiguousarray(frames[i + 1][0], np.float32)
                g._chk(g.L.gsdf_dev_upload(g.h, dev[i + 1], nxt_host.ctypes.data_as(ctypes.c_void_p), nxt_host.nbytes))
        g.sync()
        log = g.frame_log().copy()
        keys, pay = g.export(sorted=True)
        out.append((log, keys, pay))
        g.close()
    (la, ka, pa) = out[0]
    conv = la[:, 7] != 0
    assert 0 < conv.sum() < len(conv), "the stretch should hold converged and non-converged frames"
    for lb, kb, pb in out[1:]:
        # (not bit for bit between two RUNS: the handful of deferred contributions of a launch are float atomics, whose order is free)
        assert np.array_equal(la[:, 7:9], lb[:, 7:9])                             # converged flags, pass counts
        assert np.abs(la[:, :7] - lb[:, :7]).max() <= 1e-6
        assert np.array_equal(ka, kb)
        assert np.abs(pa - pb).max() <= 1e-5 * max(1.0, float(np.abs(pa).max()))


@pytest.mark.gpu
def test_closing_head_inside_the_fusion_launch_is_invisible(pkg, O, monkeypatch):
    """VERDICT r5 #2a: the frame's first gated fusion launch stands in for the last tracker launch of the first batch and performs
    its head -- for the usual frame the closing one of optimize() -- in workgroup 0 (k_fuse<.., HEAD>; GSDF_FUSE_HEAD=1, the
    default), or that launch is a tracker launch as before (0).  The arithmetic is the same function (trk_solve_update) on the
    same sums, so pass counts, convergence flags, poses and the map must be IDENTICAL up to the float atomics of the deferred
    lists -- with and without next-frame hints, with first batches of 2, 3 and 5 launches (the head then is the one of pass 0, 1
    or 3: frames that are still running when the fusion launch comes continue with a head-less tracker launch), on a stretch of
    the bench stream with frames that converge in 3-4 passes, late, and never."""
    W, H = 640, 480
    seq = pkg.synth.Sequence("tum", W, H, n_frames=48, seed=0)
    vs = np.float32(0.01)
    frames = [seq.frame(i) for i in range(seq.n)]

    def run(head, hint, first_batch):
        monkeypatch.setenv("GSDF_FUSE_HEAD", head)
        monkeypatch.setenv("GSDF_FIRST_BATCH", str(first_batch))
        g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22)
        d0, R0, t0 = frames[0]
        p = pose7_from(O, R0, t0)
        g.update(d0, O.quat_to_R(p[3:]), t0)
        g.set_pose(p)
        dev = [g.upload(f[0]) for f in frames]
        for i in range(1, seq.n):
            if hint and i + 1 < seq.n:
                g.hint_next_depth(dev[i + 1])
            g.track_and_fuse_dev(dev[i])
        g.sync()
        log = g.frame_log().copy()
        keys, pay = g.export(sorted=True)
        st = g.stats()
        g.close()
        return log, keys, pay, st

    # The reference run of every first-batch size is the one WITHOUT the stand-in and without hints: the batch size itself is not
    # invisible to the last bits (measured in round 6, also with the old code: the host waits at other moments, sees the fusion's
    # "tiles too big for the small LDS table" note earlier or later, and a tile flushed in two bands instead of one adds its sums in
    # two steps -- another last bit, which the frames that never converge amplify to ~5e-3 within their 25 passes; flags and pass
    # counts still agree on all 47 frames).  Within one batch size everything is compared to the last bits.
    bad = []
    for fb in (5, 3, 2):
        la, ka, pa, sa = run("0", False, fb)
        conv = la[:, 7] != 0
        assert 0 < conv.sum() < len(conv) and la[:, 8].max() == 25 and la[:, 8].min() <= 4
        for head, hint in (("1", False), ("1", True), ("0", True)):
            lb, kb, pb, sb = run(head, hint, fb)
            same_flags = bool(np.array_equal(la[:, 7:10], lb[:, 7:10]))                    # converged, passes, hits of the last pass
            dpose = float(np.abs(la[:, :7] - lb[:, :7]).max())
            same_keys = bool(ka.shape == kb.shape and np.array_equal(ka, kb))
            dpay = float(np.abs(pa - pb).max()) if same_keys else float("nan")
            print("MEASURED first_batch %d, head %s hint %d against head 0 hint 0: flags / passes / hits equal %s, pose diff max %.2e, keys equal %s, "
                  "sums diff %.2e, frames %d / %d, n_upd equal %s" % (fb, head, hint, same_flags, dpose, same_keys, dpay, sa["frames"], sb["frames"],
                                                                      sa["n_upd"] == sb["n_upd"]))
            # (not bit for bit between two RUNS: the handful of deferred contributions of a launch are float atomics, whose order is free)
            if not (same_flags and dpose <= 1e-6 and same_keys and dpay <= 1e-5 * max(1.0, float(np.abs(pa).max())) and
                    sa["frames"] == sb["frames"] and sa["n_upd"] == sb["n_upd"]):
                bad.append((fb, head, hint))
    assert not bad, bad


@pytest.mark.gpu
def test_no_frame_is_lost_when_contexts_share_the_gpu(pkg, O):
    """Round 6, found by test_bench_gpus_8_end_to_end_over_the_rccl_double (1 run in 3): in a k_fuse<.., HEAD> launch a workgroup of a
    later dispatch round could take the solver's sticky `done` word (fresh) for "optimize() ended before this launch" and then use a
    STALE st->done = 0 from its XCD's L2 -- it left without fusing its tile and without its ticket, so the launch had no last
    workgroup: no frame-log row, the deferred list never added.  The window opens when lines of the state block are evicted between
    two dispatch rounds, i.e. when other work shares the GPU.  Here: four contexts in four host threads run the hinted frame loop
    over the same 30 frames (all converge within the first batch: every fusion launch is a HEAD launch that runs), twenty times each;
    every context must log every frame, fuse every frame (Sdf::counter_) with every tile (n_upd: a lost tile is ~4 000 updates), and
    end with the poses and the map of a context that ran alone -- up to the last bits the float atomics of the deferred lists and
    the host's timing leave open (see test_closing_head_inside_the_fusion_launch_is_invisible)."""
    import threading
    W, H = 640, 480
    n = 30
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.01)
    frames = [seq.frame(i) for i in range(n)]

    def run(out, idx, reps):
        g = pkg.GradSdf(vs, np.float32(10) * vs, W, H, seq.K, capacity_log2=22)
        d0, R0, t0 = frames[0]
        p = pose7_from(O, R0, t0)
        dev = [g.upload(f[0]) for f in frames]
        res = []
        for _ in range(reps):
            g.reset()
            g.update(d0, O.quat_to_R(p[3:]), t0)
            g.set_pose(p)
            for i in range(1, n):
                if i + 1 < n:
                    g.hint_next_depth(dev[i + 1])
                g.track_and_fuse_dev(dev[i])
            g.sync()
            log = g.frame_log().copy()
            st = g.stats()
            keys, pay = g.export(sorted=True)
            res.append((log, st["frames"], st["n_upd"], keys, pay))
        g.close()
        out[idx] = res

    alone = [None]
    run(alone, 0, 1)
    log_a, frames_a, n_upd_a, keys_a, pay_a = alone[0][0]
    assert len(log_a) == n - 1 and frames_a == 1 + int(log_a[:, 7].sum())
    out = [None] * 4
    th = [threading.Thread(target=run, args=(out, i, 20)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for res in out:
        assert res is not None and len(res) == 20
        for log, n_frames, n_upd, keys, pay in res:
            assert len(log) == n - 1, "a frame has no log row: its fusion launch had no last workgroup"
            assert np.array_equal(log[:, 7:9], log_a[:, 7:9])                       # converged flags, pass counts
            assert n_frames == frames_a and abs(int(n_upd) - int(n_upd_a)) <= 500   # every converged frame fused, every tile of it
            assert np.abs(log[:, :7] - log_a[:, :7]).max() <= 1e-5
            assert abs(len(keys) - len(keys_a)) <= 1e-4 * len(keys_a)
