"""Analytic known-answer tests that pin the CPU oracle: the reference ships
no tests or golden vectors, so these are what the oracle's restatement is anchored on."""
import numpy as np
import pytest

from conftest import pose7_from

VS = np.float32(0.01)
T10 = np.float32(10) * VS


def round_half_away(x):
    return np.where(x >= 0, np.floor(x + np.float32(0.5)), -np.floor(-x + np.float32(0.5)))


def test_constants_match_survey(O, pkg):
    # T = 10 * 0.01 in float32 = 0.099999994, inv_vs = 100.0
    assert float(T10) == pytest.approx(0.099999994, abs=1e-9)
    assert np.float32(1.0 / float(VS)) == np.float32(100.0)
    assert int(np.floor(T10 / VS)) == 10
    assert np.float32(1.0 / float(T10)) == np.float32(10.000001)


def test_plane_keys_dist_and_normals(O, pkg):
    """Fronto-parallel plane at z0, identity pose: keys = round((z0 + k vs) (x0,y0,1) / vs), dist =
    clamp(vs*vz - z0) for EVERY voxel, FALS normal = (0,0,+1) (inward-pointing)."""
    W, H = 160, 120
    K = pkg.synth.intrinsics(W, H)
    z0 = np.float32(1.5)
    depth = np.full((H, W), z0, np.float32)
    o = O.Oracle(VS, T10, W, H, K)
    n = o.normals(depth)
    inner = (slice(8, H - 8), slice(8, W - 8))
    # Q*b in float32 cancels ~4 digits (|Q||b| >> |n|): the reference's own formulation, not the oracle's
    assert np.abs(n[0][inner]).max() < 5e-4 and np.abs(n[1][inner]).max() < 5e-4
    assert np.abs(n[2][inner] - 1).max() < 1e-6
    n_upd, n_valid = o.update(depth, np.eye(3), np.zeros(3))
    assert n_valid == W * H
    keys, pay = o.export()
    c = o.normals_cache()
    x0, y0 = c[0], c[1]
    exp = set()
    for k in range(-10, 11):
        s = z0 + np.float32(k) * VS
        vx = round_half_away(np.float32(100.0) * (s * x0)).astype(np.int64)
        vy = round_half_away(np.float32(100.0) * (s * y0)).astype(np.int64)
        vz = round_half_away(np.float32(100.0) * (s * np.ones_like(x0))).astype(np.int64)
        sdf = VS * vz.astype(np.float32) - z0
        keep = sdf <= T10          # weight > 0
        exp |= set(zip(vx[keep].tolist(), vy[keep].tolist(), vz[keep].tolist()))
    got = set(map(tuple, keys.tolist()))
    assert got == exp
    sdf = VS * keys[:, 2].astype(np.float32) - z0
    assert np.abs(pay[:, 0] - np.clip(sdf, -T10, T10)).max() < 1e-6
    # gradient = (0,0,1) * sum of weights (up to normal rounding)
    assert np.abs(pay[:, 3] - pay[:, 4]).max() <= 1e-4 * pay[:, 4].max()
    assert np.abs(pay[:, 1:3]).max() <= 1e-3 * pay[:, 4].max()


def test_sphere_gradient_direction(O, pkg):
    """Stored gradients point along the analytic sphere normal, inward (MapGradPixelSdf.cpp:112)."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=4, seed=3, noise=False, step_deg=2.0)
    seq.spheres = np.array([[0.05, 0.1, -0.05, 0.45]])
    o = O.Oracle(VS, T10, W, H, seq.K)
    for i in range(seq.n):
        d, R, t = seq.frame(i)
        o.update(d, R, t)
    keys, pay = o.export()
    m = (pay[:, 4] >= 5) & (np.abs(pay[:, 0]) < 0.02)
    assert m.sum() > 2000
    ctr = seq.spheres[0, :3]
    pos = keys[m].astype(np.float64) * float(VS)
    radial = ctr - pos
    radial /= np.linalg.norm(radial, axis=1, keepdims=True)          # inward = towards the centre
    gdir = pay[m, 1:4].astype(np.float64)
    gdir /= np.linalg.norm(gdir, axis=1, keepdims=True)
    ang = np.degrees(np.arccos(np.clip((radial * gdir).sum(1), -1, 1)))
    assert np.median(ang) < 3.0 and np.percentile(ang, 95) < 10.0


def test_tracker_recovers_known_twist(O, pkg):
    W, H = 640, 480
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=1, seed=0, noise=False)
    o = O.Oracle(VS, T10, W, H, seq.K)
    d, R, t = seq.frame(0)
    p_true = pose7_from(O, R, t)
    o.update(d, O.quat_to_R(p_true[3:]), t)
    xi = np.array([0.012, -0.008, 0.01, 0.004, -0.006, 0.005], np.float32)   # small twist
    p_start = O.se3_exp_mul(xi, p_true)
    conv, p, used, trace, hits = o.track(d, p_start)
    assert conv and used < 25
    assert np.abs(p[:3] - p_true[:3]).max() < 2e-3
    assert np.abs(p[3:] - p_true[3:]).max() < 1e-3
    assert trace[0, 35] > trace[used - 1, 35]              # |xi|^2 shrinks
    assert trace[used - 1, 35] < 1e-6


def test_tracker_no_overlap_is_not_converged(O, pkg):
    """All-zero H -> llt gives NaN -> 25 idle passes -> false (SURVEY.md gotcha 9)."""
    W, H = 64, 48
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=1, seed=0)
    o = O.Oracle(VS, T10, W, H, seq.K)
    d, R, t = seq.frame(0)
    p0 = pose7_from(O, R, t)
    conv, p, used, trace, hits = o.track(d, p0, iters=7)
    assert not conv and used == 7 and np.array_equal(p, p0) and hits.sum() == 0


def test_fusion_order_independent_and_omp_equals_serial(O, pkg):
    W, H = 160, 120
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=4, seed=2, step_deg=3.0)
    fr = [seq.frame(i) for i in range(seq.n)]
    outs = []
    for order, omp in (([0, 1, 2, 3], False), ([3, 1, 0, 2], False), ([0, 1, 2, 3], True)):
        o = O.Oracle(np.float32(0.02), np.float32(0.1), W, H, seq.K, threads=4)
        for i in order:
            o.update(*fr[i], omp=omp)
        outs.append(o.export())
    k0, p0 = outs[0]
    for k, p in outs[1:]:
        assert np.array_equal(k0, k)
        assert np.abs(p - p0).max() <= 1e-4 * max(1.0, np.abs(p0).max())
    vis = o.export_vis(1)
    assert vis.max() < 16 and vis.min() >= 1               # every voxel was seen by at least one frame


def test_holes_produce_finite_or_nan_normals_only_where_expected(O, pkg):
    W, H = 96, 72
    K = pkg.synth.intrinsics(W, H)
    depth = np.full((H, W), 1.2, np.float32)
    depth[20:50, 30:70] = 0.0                              # hole larger than the 11x11 window
    o = O.Oracle(VS, T10, W, H, K)
    n = o.normals(depth)
    nan = ~np.isfinite(n[2])
    assert nan.any() and not nan[depth > 0].any()          # 0/0 only deep inside the hole
    n_upd, n_valid = o.update(depth, np.eye(3), np.zeros(3))
    assert n_valid <= int((depth > 0).sum())
    keys, pay = o.export()
    assert np.isfinite(pay).all()


def test_se3_helpers(O):
    q = np.array([0.1, -0.2, 0.3, 0.9], np.float32)
    q /= np.linalg.norm(q)
    R = O.quat_to_R(q)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and abs(np.linalg.det(R) - 1) < 1e-6
    q2 = O.R_to_quat(R)
    assert np.abs(q2 - q).max() < 1e-6
    ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    assert np.array_equal(O.se3_exp_mul(np.zeros(6, np.float32), ident), ident)
    xi = np.array([0.1, 0.2, -0.1, 0.05, -0.02, 0.03], np.float32)
    p = O.se3_exp_mul(-xi, O.se3_exp_mul(xi, ident))
    assert np.abs(p - ident).max() < 1e-6
    # pure translation twist: exp moves t by exactly the twist
    p = O.se3_exp_mul(np.array([0.5, -1, 2, 0, 0, 0], np.float32), ident)
    assert np.allclose(p[:3], [0.5, -1, 2]) and np.array_equal(p[3:], ident[3:])
    # rotation about z by 90 degrees
    p = O.se3_exp_mul(np.array([0, 0, 0, 0, 0, np.pi / 2], np.float32), ident)
    assert np.abs(O.quat_to_R(p[3:]) - np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])).max() < 1e-6


def test_query_is_first_order_taylor(O, pkg):
    W, H = 160, 120
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=1, seed=1)
    vs = np.float32(0.02)
    o = O.Oracle(vs, np.float32(0.1), W, H, seq.K)
    o.update(*seq.frame(0))
    keys, pay = o.export()
    k = keys[:200]
    ctr = k.astype(np.float32) * vs
    d, g, w = o.query(ctr)
    assert np.array_equal(w, pay[:200, 4])
    assert np.abs(d - pay[:200, 0]).max() < 1e-7                 # at the voxel centre phi = dist
    gn = pay[:200, 1:4] / np.linalg.norm(pay[:200, 1:4], axis=1, keepdims=True)
    assert np.abs(g - 1.2 * gn).max() < 1e-5                    # 1.2 * normalized gradient (MapGradPixelSdf.h:113)
    off = np.float32(0.004) * np.ones(3, np.float32)
    d2, _, _ = o.query(ctr + off)
    assert np.abs(d2 - (d - g @ off)).max() < 1e-6
    dm, gm, wm = o.query(np.array([[50.0, 50.0, 50.0]], np.float32))
    assert wm[0] == 0 and dm[0] == 0


def test_se3_exp_matches_matrix_exponential(O):
    """The restated Sophus SE3::exp / group product against scipy's expm of the 4x4 twist matrix
    (independent float64 check of the third-party arithmetic the oracle has to restate)."""
    from scipy.linalg import expm
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    for scale in (1e-6, 1e-3, 0.05, 0.7, 2.5):
        xi = (scale * rng.standard_normal(6)).astype(np.float32)
        q0 = rng.standard_normal(4)
        q0 /= np.linalg.norm(q0)
        t0 = rng.uniform(-2, 2, 3)
        pose = np.concatenate([t0, q0]).astype(np.float32)
        got = O.se3_exp_mul(xi, pose)
        M = np.zeros((4, 4))
        w = xi[3:].astype(np.float64)
        M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        M[:3, 3] = xi[:3]
        T0 = np.eye(4)
        T0[:3, :3] = Rotation.from_quat(pose[3:].astype(np.float64)).as_matrix()
        T0[:3, 3] = pose[:3]
        T = expm(M) @ T0
        Rg = O.quat_to_R(got[3:]).astype(np.float64)
        assert np.abs(Rg - T[:3, :3]).max() < 5e-6
        assert np.abs(got[:3] - T[:3, 3]).max() < 5e-6 * max(1.0, np.abs(T[:3, 3]).max())
        assert abs(np.linalg.norm(got[3:]) - 1) < 1e-6


def test_quaternion_conversions_match_scipy(O):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(6)
    for _ in range(50):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        R = O.quat_to_R(q.astype(np.float32))
        assert np.abs(R - Rotation.from_quat(q).as_matrix()).max() < 1e-6
        q2 = O.R_to_quat(R)
        assert min(np.abs(q2 - q).max(), np.abs(q2 + q).max()) < 1e-6      # q and -q are the same rotation


def test_tracker_step_solves_the_normal_equations(O, pkg):
    """trace rows expose E, g, H, xi of every pass: xi must be damping * H^-1 g (Eigen LLT restated) and
    H symmetric positive definite."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("spheres", W, H, n_frames=2, seed=0, noise=False)
    o = O.Oracle(VS, T10, W, H, seq.K)
    d, R, t = seq.frame(0)
    o.update(d, R, t)
    d1, R1, t1 = seq.frame(1)
    conv, p, used, trace, hits = o.track(d1, pose7_from(O, R, t), iters=3)
    for row in trace:
        g = row[1:7].astype(np.float64)
        Hm = np.zeros((6, 6))
        q = 7
        for i in range(6):
            for j in range(i, 6):
                Hm[i, j] = Hm[j, i] = row[q]
                q += 1
        assert np.all(np.linalg.eigvalsh(Hm) > 0)
        xi = np.linalg.solve(Hm, g)
        assert np.abs(row[29:35] - xi).max() <= 2e-3 * max(1e-6, np.abs(xi).max())
        assert row[35] == pytest.approx(float((row[29:35].astype(np.float64) ** 2).sum()), rel=1e-5)
        assert row[28] == hits[0] or row[28] > 0


def test_raycast_recovers_the_plane(O, pkg):
    """The self-defined voxel-hash raycaster (absent from the reference): a fronto-parallel plane fused once is
    rendered back at its depth (within half a voxel) with normal +z; a map seen from behind gives no hit."""
    W, H = 160, 120
    K = pkg.synth.intrinsics(W, H)
    z0 = np.float32(1.5)
    depth = np.full((H, W), z0, np.float32)
    o = O.Oracle(VS, T10, W, H, K)
    o.update(depth, np.eye(3), np.zeros(3))
    z, n = o.raycast(np.eye(3), np.zeros(3))
    inner = (slice(8, H - 8), slice(8, W - 8))
    # rays near the image border run along the edge of the fused columns (1 cm voxels, 1.1 cm between pixel rays at this
    # depth), where two consecutive samples need not both exist
    assert (z[inner] > 0).all() and (z > 0).mean() > 0.97
    assert np.abs(z - z0)[z > 0].max() < 0.5 * float(VS)
    assert np.abs(n[2][inner] - 1).max() < 1e-3
    # camera 3 m further down the axis, looking back at the plane from its far side: the SDF goes + -> -, no hit
    Rb = np.diag([-1.0, 1.0, -1.0]).astype(np.float32)
    zb, _ = o.raycast(Rb, np.array([0, 0, 3.0], np.float32))
    assert (zb == 0).all()
    # a window narrower than the band in front of the surface still finds it; one that ends before it does not
    z2, _ = o.raycast(np.eye(3), np.zeros(3), zmin=1.45, zmax=1.6)
    assert (z2[inner] > 0).all() and np.abs(z2 - z0)[z2 > 0].max() < 0.5 * float(VS)
    z3, _ = o.raycast(np.eye(3), np.zeros(3), zmin=0.5, zmax=1.4)
    assert (z3 == 0).all()


def test_raycast_oblique_view_does_not_jump_the_band(O, pkg):
    """A plane fused head-on, rendered from up to 55 degrees off its normal and from a different distance: along such rays the
    part of the band in FRONT of the surface can be thinner than a coarse step; the walk backs up (re-walks the last coarse
    step in fine steps) whenever a coarse step ends on an existing voxel, so every ray finds the surface.
    (2 cm voxels: a pixel at this resolution is narrower than a voxel, so the fused band has no lateral gaps.)"""
    W, H = 160, 120
    K = pkg.synth.intrinsics(W, H)
    z0 = np.float32(1.5)
    vs = np.float32(0.02)
    o = O.Oracle(vs, np.float32(10) * vs, W, H, K)
    for dx in (-0.6, -0.3, 0.0, 0.3, 0.6):                            # a wide strip of the plane
        o.update(np.full((H, W), z0, np.float32), np.eye(3), np.array([dx, 0, 0], np.float32))
    v, u = np.mgrid[30:90, 40:120]
    x0 = (u - K[0, 2]) / K[0, 0]; y0 = (v - K[1, 2]) / K[1, 1]
    for deg in (0.0, 30.0, 55.0):
        a = np.deg2rad(deg)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)   # camera turned about y
        t = np.array([0, 0, z0], np.float32) - 1.2 * R[:, 2]         # its axis meets the plane point (0, 0, z0) at distance 1.2
        z, n = o.raycast(R, t, zmin=0.5, zmax=3.0)
        inner = z[30:90, 40:120]
        assert (inner > 0).all(), deg
        pts = inner[..., None] * (np.stack([x0, y0, np.ones_like(x0)], -1) @ R.T) + t
        assert np.abs(pts[..., 2] - z0).max() < 1.25 * float(vs), deg   # the rendered points lie on the plane


def test_raycast_thin_band(O, pkg):
    """Truncation of 2 voxels: the walk through empty space must not be wider than the band in front of the surface.
    (2 cm voxels: a pixel at this resolution is narrower than a voxel, so the fused band has no lateral gaps.)"""
    W, H = 96, 72
    K = pkg.synth.intrinsics(W, H)
    vs = np.float32(0.02)
    z0 = np.float32(1.2345)
    o = O.Oracle(vs, np.float32(2) * vs, W, H, K)
    o.update(np.full((H, W), z0, np.float32), np.eye(3), np.zeros(3))
    for zmin in (0.5, 0.505, 0.511, 0.517):                         # every phase of the steps relative to the band
        z, _ = o.raycast(np.eye(3), np.zeros(3), zmin=zmin)
        assert (z > 0).mean() > 0.94                                 # a few border rays leave the fused columns
        assert np.abs(z - z0)[z > 0].max() < 1.25 * float(vs)        # one sample spacing along an oblique ray: phi = dist + 1.2 g.(c - p) is not metric


# ---- exports: marching cubes (classic case tables as data) and the point cloud -------------------------------------

def _mc_tables():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "gsdf_mc_tables.h")).read()
    e = re.search(r"GSDF_MC_EDGE_TABLE\[256\] = \{(.*?)\};", txt, re.S).group(1)
    t = re.search(r"GSDF_MC_TRI_TABLE\[256 \* 16\] = \{(.*?)\};", txt, re.S).group(1)
    edge = np.array([int(v, 16) for v in re.findall(r"0x[0-9a-f]+", e)])
    tri = np.array([int(v) for v in re.findall(r"-?\d+", t)]).reshape(256, 16)
    return edge, tri


def test_mc_case_tables_are_consistent():
    """The embedded edgeTable / triTable (include/gsdf_mc_tables.h = LayeredMarchingCubesNoColor.cpp:67-352): for every
    corner pattern the flagged edges are exactly the edges whose two corners differ in sign (edge e joins corners A[e], B[e]
    of :410-549), the triangles use exactly the flagged edges, and complementary patterns cross the same edges."""
    edge, tri = _mc_tables()
    A = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]
    B = [1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7]
    assert edge.shape == (256,) and tri.shape == (256, 16)
    for c in range(256):
        mask = 0
        for e in range(12):
            if ((c >> A[e]) & 1) != ((c >> B[e]) & 1):
                mask |= 1 << e
        assert edge[c] == mask, c
        row = tri[c]
        used = row[row >= 0]
        assert len(used) % 3 == 0 and len(used) <= 15
        assert (row[len(used):] == -1).all()
        m2 = 0
        for e in used:
            m2 |= 1 << int(e)
        assert m2 == mask, c
        assert edge[255 - c] == edge[c]
    assert (tri[0] == -1).all() and (tri[255] == -1).all()


def test_oracle_marching_cubes_sphere_is_watertight(O):
    """The oracle's LayeredMarchingCubesNoColor restatement on an analytic sphere (map loaded with set_map): every mesh edge
    is shared by exactly two faces, vertices lie on the sphere within interpolation error, faces come in z-y-x sweep order."""
    N, vs, r = 20, np.float32(0.05), 0.8
    ax = np.arange(-N, N + 1, dtype=np.int32)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    keys = np.stack([x.ravel(), y.ravel(), z.ravel()], 1).astype(np.int32)
    d = (np.sqrt((keys.astype(np.float64) ** 2).sum(1)) * float(vs) - r).astype(np.float32)
    pay = np.zeros((len(keys), 5), np.float32)
    pay[:, 0] = d; pay[:, 3] = 1.0; pay[:, 4] = 1.0
    K = np.array([100, 0, 32, 0, 100, 24, 0, 0, 1], np.float32)
    o = O.Oracle(vs, np.float32(5) * vs, 64, 48, K)
    o.set_map(keys, pay)
    tris = o.extract_mesh()
    assert tris.shape[0] > 2000
    rad = np.linalg.norm(tris.reshape(-1, 3), axis=1)
    assert rad.min() > r - 0.01 and rad.max() < r + 0.01
    q = np.round(tris.astype(np.float64) * 1e5).astype(np.int64)
    edges = {}
    for f in q:
        for a in range(3):
            u, v = tuple(f[a]), tuple(f[(a + 1) % 3])
            k = (u, v) if u < v else (v, u)
            edges[k] = edges.get(k, 0) + 1
    assert all(c == 2 for c in edges.values())
    # sweep order: the lowest corner of the cube a face came from never decreases in (z, y, x)
    base = np.floor(tris.min(axis=1) / float(vs) + 1e-4).astype(np.int64)
    code = (base[:, 2] * 4096 + base[:, 1]) * 4096 + base[:, 0]
    assert (np.diff(code) >= 0).all()
    # a voxel with weight 0 at a cube corner suppresses the cube (computeLutIndex :611-618)
    pay2 = pay.copy()
    pay2[:, 4] = np.where((keys == 0).all(1) | (np.abs(keys).sum(1) % 7 == 0), 0.0, 1.0)
    o.set_map(keys[pay2[:, 4] > 0], pay2[pay2[:, 4] > 0])
    assert 0 < o.extract_mesh().shape[0] < tris.shape[0]


def test_oracle_extract_pc_plane_known_answer(O):
    """extract_pc (MapGradPixelSdf.cpp:177-220) on hand-made voxels: weight gate (>= 5), the half-voxel box test on
    dist * 1.2 g^, point = centre - dist * 1.2 g^, normal = -1.2 g^."""
    vs = np.float32(0.02)
    K = np.array([100, 0, 32, 0, 100, 24, 0, 0, 1], np.float32)
    o = O.Oracle(vs, np.float32(5) * vs, 64, 48, K)
    keys = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]], np.int32)
    pay = np.array([[0.004, 0, 0, 3.0, 6.0],      # in: |0.004 * 1.2| < 0.01
                    [0.004, 0, 0, 3.0, 4.9],      # weight < 5: out
                    [0.009, 0, 0, -2.0, 9.0],     # |0.009 * 1.2| = 0.0108 >= 0.01: out
                    [-0.008, 0, 5.0, 0, 5.0]],    # in (weight == 5 passes `< 5`), gradient along +y
                   np.float32)
    o.set_map(keys, pay)
    rows = o.extract_pc()
    assert rows.shape == (2, 6)
    g = np.float32(1.2)
    assert np.allclose(rows[0], [0, 0, 0 - np.float32(0.004) * g, 0, 0, -g], atol=1e-7)
    assert np.allclose(rows[1], [np.float32(0.06), 0 + np.float32(0.008) * g, 0, 0, -g, 0], atol=1e-7)


# ---- round 5: the restatements of absent third-party arithmetic, measured ---------------------------------------------------

def test_llt_solve6_solves_and_follows_eigens_order(O):
    """H.llt().solve(g) (RigidPointOptimizer.cpp:86) as restated from Eigen's unblocked llt_inplace + unrolled triangular solves:
    (1) it solves -- against numpy's double solve on well-conditioned tracker-like systems; (2) it is the published operation
    order -- a float32 numpy transcription of that order (pivot = a_kk - (sum of squares formed first), column = (a_ik - dot
    formed first) / pivot root, rhs_i = (rhs_i - halving-tree dot) / diagonal) reproduces it bit for bit; (3) an all-zero H
    yields NaN (SURVEY.md gotcha 9)."""
    f = np.float32
    rng = np.random.default_rng(5)

    def tree(t):
        n = len(t)
        if n == 1:
            return t[0]
        h = n // 2
        return f(tree(t[:h]) + tree(t[h:]))

    def eigen_order(H, g):
        L = H.astype(f).copy()
        for k in range(6):
            d = L[k, k]
            if k > 0:
                sq = f(L[k, 0] * L[k, 0])
                for j in range(1, k):
                    sq = f(sq + f(L[k, j] * L[k, j]))
                d = f(d - sq)
            d = f(np.sqrt(d))
            L[k, k] = d
            for i in range(k + 1, 6):
                s = L[i, k]
                if k > 0:
                    c = f(L[i, 0] * L[k, 0])
                    for j in range(1, k):
                        c = f(c + f(L[i, j] * L[k, j]))
                    s = f(s - c)
                L[i, k] = f(s / d)
        x = g.astype(f).copy()
        for i in range(6):
            s = x[i]
            if i > 0:
                s = f(s - tree([f(L[i, j] * x[j]) for j in range(i)]))
            x[i] = f(s / L[i, i])
        for i in range(5, -1, -1):
            s = x[i]
            if i < 5:
                s = f(s - tree([f(L[j, i] * x[j]) for j in range(i + 1, 6)]))
            x[i] = f(s / L[i, i])
        return x

    for _ in range(100):
        J = rng.normal(size=(60, 6)) * np.array([1, 1, 1, 3, 3, 3])
        H = (J.T @ J).astype(f)
        g = rng.normal(size=6).astype(f)
        x = O.llt_solve6(H, g)
        ref = np.linalg.solve(H.astype(np.float64), g.astype(np.float64))
        assert np.abs(x - ref).max() <= 2e-5 * np.abs(ref).max()
        assert np.array_equal(x.view(np.uint32), eigen_order(H, g).view(np.uint32))
    assert np.isnan(O.llt_solve6(np.zeros((6, 6)), np.ones(6))).all()


def test_box_filter_summation_order_measured(pkg, O):
    """NormalEstimator.h:109-114,191-193 call cv::boxFilter, whose generic path keeps RUNNING double sums (RowSum / ColumnSum);
    OpenCV is absent, so the order is restated (oracle/gsdf_oracle.cpp box_sum) and its effect MEASURED here:
      * cached planes (NormalEstimator::cache): Q = M^-1 amplifies the sums' last bits -- fresh sums change ~10 % of the Q
        floats and move normals by up to ~1e-2, enough to flip gates of MapGradPixelSdf.cpp:95,98 on a handful of pixels.  The
        definition therefore follows OpenCV's running order there (box mode (1, .), the GPU's k_ncache_rows / k_ncache_cols);
      * per-frame filters (::compute): running and fresh sums differ by ~1e-16 relative before the rounding to float; on these
        frames not one normal changes a bit, no gate flips, the maps are identical -- the definition (and the GPU's tiled
        kernel) sums freshly there (box mode (., 0)).
    "Bit-exact occupancy" in this repository means: against this reading of the absent libraries."""
    W, H, n = 320, 240, 2
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=0)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    defn = O.Oracle(vs, T, W, H, seq.K)                         # (1, 0): the definition
    run = O.Oracle(vs, T, W, H, seq.K, box_mode=(1, 1))         # OpenCV's order everywhere
    fresh = O.Oracle(vs, T, W, H, seq.K, box_mode=(0, 0))       # fresh sums everywhere (rounds 1-4)
    c_def, c_run, c_fresh = defn.normals_cache(), run.normals_cache(), fresh.normals_cache()
    assert np.array_equal(c_def.view(np.uint32), c_run.view(np.uint32))
    q_diff = int((c_def[5:].view(np.uint32) != c_fresh[5:].view(np.uint32)).sum())
    assert np.array_equal(c_def[:5].view(np.uint32), c_fresh[:5].view(np.uint32))       # the five direct planes have no sums
    assert q_diff > 0.01 * c_def[5:].size                       # the order MATTERS for Q (measured: ~10 % of the floats at 640x480)
    worst = 0.0
    for i in range(n):
        d, R, t = seq.frame(i)
        n_def, n_run, n_fresh = defn.normals(d), run.normals(d), fresh.normals(d)
        ok = np.isfinite(n_def) & np.isfinite(n_run)
        assert np.array_equal(n_def[ok].view(np.uint32), n_run[ok].view(np.uint32))     # per frame: not one bit
        ok = np.isfinite(n_def) & np.isfinite(n_fresh)
        worst = max(worst, float(np.abs(n_def - n_fresh)[ok].max()))
        assert defn.update(d, R, t) == run.update(d, R, t)      # N_upd, N_valid: no gate flips
        fresh.update(d, R, t)
    k_def, p_def = defn.export()
    k_run, p_run = run.export()
    assert np.array_equal(k_def, k_run) and np.array_equal(p_def.view(np.uint32), p_run.view(np.uint32))
    assert 1e-6 < worst < 5e-2                                  # what fresh sums in the CACHE do to normals (1e-2 at 640x480)


def test_host_side_math_header_equals_the_oracle(O, tmp_path):
    """csrc/gsdf_math.h (compiled for the host here; the same source is the tracker head's exact fallback on the device and the
    facade's SE3) against the oracle, bit for bit: gsdf_llt_solve6 (Eigen's order), gsdf_se3_exp_mul, quaternion <-> matrix."""
    import ctypes as C
    import os
    import subprocess
    from conftest import ROOT
    src = tmp_path / "m.cpp"
    src.write_text('#include "gsdf_math.h"\nextern "C" {\n'
                   'void t_llt(const float* H, const float* g, float* x) { gsdf_llt_solve6(H, g, x); }\n'
                   'void t_exp(const float* xi, float* p) { gsdf_se3_exp_mul(xi, p); }\n'
                   'void t_q2r(const float* q, float* R) { gsdf_quat_to_R(q, R); }\n'
                   'void t_r2q(const float* R, float* q) { gsdf_R_to_quat(R, q); }\n}\n')
    lib = tmp_path / "libm_t.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-I", os.path.join(ROOT, "gradient-sdf_amd", "csrc"), str(src), "-o", str(lib)])
    L = C.CDLL(str(lib))
    fp = C.POINTER(C.c_float)
    f = np.float32
    rng = np.random.default_rng(11)
    for _ in range(200):
        J = rng.normal(size=(40, 6)) * np.array([1, 1, 1, 2, 2, 2])
        H = np.ascontiguousarray((J.T @ J).astype(f)).reshape(36)
        g = rng.normal(size=6).astype(f)
        x = np.empty(6, f)
        L.t_llt(H.ctypes.data_as(fp), g.ctypes.data_as(fp), x.ctypes.data_as(fp))
        assert np.array_equal(x.view(np.uint32), O.llt_solve6(H, g).view(np.uint32))
        xi = (rng.normal(size=6) * np.array([0.01, 0.01, 0.01, 0.02, 0.02, 0.02]) * 10.0 ** rng.integers(-4, 2)).astype(f)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        p = np.concatenate([rng.normal(size=3), q]).astype(f)
        p2 = p.copy()
        L.t_exp(xi.ctypes.data_as(fp), p2.ctypes.data_as(fp))
        assert np.array_equal(p2.view(np.uint32), O.se3_exp_mul(xi, p).view(np.uint32))
        R = np.empty(9, f)
        L.t_q2r(p[3:].copy().ctypes.data_as(fp), R.ctypes.data_as(fp))
        assert np.array_equal(R.reshape(3, 3).view(np.uint32), O.quat_to_R(p[3:]).view(np.uint32))
        q2 = np.empty(4, f)
        L.t_r2q(R.ctypes.data_as(fp), q2.ctypes.data_as(fp))
        assert np.array_equal(q2.view(np.uint32), O.R_to_quat(R).view(np.uint32))
    xz = np.empty(6, f)
    Hz, gz = np.zeros(36, f), np.ones(6, f)
    L.t_llt(Hz.ctypes.data_as(fp), gz.ctypes.data_as(fp), xz.ctypes.data_as(fp))
    assert np.isnan(xz).all()


def test_tsdf_return_expression_is_the_references_mixed_precision_one(O, pkg):
    """MapGradPixelSdf.h:113-114 on 10^6 random voxels, vectorised float64 numpy transcription against the oracle, bit for bit:

        (*grad_ptr) = 1.2*v.grad.normalized();                                  // Eigen scalar * expression: 1.2 -> 1.2f
        return v.dist + 1.2*v.grad.normalized().dot(vox2float(idx) - point);    // float dot, then PLAIN C++ double * and +

    The float-only reading `dist + dot(1.2f * unit, d)` (rounds 1-5 of this repository) differs in the last bit of phi on
    about a tenth of the voxels -- the test measures that too, so the distinction it guards stays visible."""
    rng = np.random.default_rng(2024)
    n = 1_000_000
    # distinct voxel keys on a 128^3 lattice
    lin = rng.choice(128 ** 3, size=n, replace=False)
    keys = np.stack([lin % 128 - 64, (lin // 128) % 128 - 64, lin // (128 * 128) - 64], axis=1).astype(np.int32)
    pay = np.empty((n, 5), np.float32)
    pay[:, 0] = rng.uniform(-0.1, 0.1, n)                        # dist
    pay[:, 1:4] = rng.normal(0, 3, (n, 3))                       # grad (un-normalised sum of weighted normals)
    pay[:, 4] = rng.uniform(0.5, 40, n)                          # weight
    pay[:7, 1:4] = 0                                             # normalized() of a zero vector: unchanged (Eigen: only if |g|^2 > 0)
    W, H = 32, 24
    o = O.Oracle(VS, T10, W, H, pkg.synth.intrinsics(W, H))
    o.set_map(keys, pay)
    f32 = np.float32
    centre = (VS * keys.astype(np.float32)).astype(np.float32)   # vox2float: vs * float(idx)
    pts = (centre + rng.uniform(-0.0049, 0.0049, (n, 3)).astype(np.float32)).astype(np.float32)
    # only points whose voxel is the intended one (float2vox(point) == idx)
    back = np.where(f32(100.0) * pts >= 0, np.floor(f32(100.0) * pts + f32(0.5)), -np.floor(-(f32(100.0) * pts) + f32(0.5))).astype(np.int32)
    own = (back == keys).all(axis=1)
    assert own.mean() > 0.95
    dist, grad, w = o.query(pts)
    assert np.array_equal(w[own], pay[own, 4])

    g = pay[:, 1:4]
    sq = (g[:, 0] * g[:, 0] + (g[:, 1] * g[:, 1] + g[:, 2] * g[:, 2])).astype(np.float32)      # squaredNorm: x0 + (x1 + x2)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = np.where(sq[:, None] > 0, g / np.sqrt(sq)[:, None], g).astype(np.float32)
    d = (centre - pts).astype(np.float32)
    dot = (unit[:, 0] * d[:, 0] + (unit[:, 1] * d[:, 1] + unit[:, 2] * d[:, 2])).astype(np.float32)
    phi = (pay[:, 0].astype(np.float64) + np.float64(1.2) * dot.astype(np.float64)).astype(np.float32)   # :114
    gout = (f32(1.2) * unit).astype(np.float32)                                                          # :113
    assert np.array_equal(dist[own].view(np.uint32), phi[own].view(np.uint32))
    assert np.array_equal(grad[own].view(np.uint32), gout[own].view(np.uint32))

    # the float-only reading is a different function: same value to 1e-8, another last bit on ~10 % of the voxels
    phi_f = (pay[:, 0] + (gout[:, 0] * d[:, 0] + (gout[:, 1] * d[:, 1] + gout[:, 2] * d[:, 2]))).astype(np.float32)
    frac = float((phi_f[own].view(np.uint32) != phi[own].view(np.uint32)).mean())
    assert 0.02 < frac < 0.3, frac
    assert np.abs(phi_f[own] - phi[own]).max() < 1e-7


@pytest.mark.parametrize("sampling", [2, 3, 5, 1000])
def test_optimize_sampled_visits_the_strided_pixels_only(O, pkg, sampling):
    """RigidPointOptimizer::optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.cpp:62 `for (y = 0; y < h; y += sampling)
    for (x = 0; x < w; x += sampling)`: the same pixels, in the same order, as the full loop over an image whose other pixels are
    invalid (z = 0 fails the z_min gate of :64-65) -- so the two runs must agree bit for bit, pass for pass."""
    W, H = 96, 72
    seq = pkg.synth.Sequence("tum", W, H, n_frames=2, seed=4)
    vs = np.float32(0.04)
    o = O.Oracle(vs, np.float32(5) * vs, W, H, seq.K)
    d0, R0, t0 = seq.frame(0)
    o.update(d0, R0, t0)
    d1, _, _ = seq.frame(1)
    p0 = pose7_from(O, R0, t0)
    masked = np.zeros_like(d1)
    masked[::sampling, ::sampling] = d1[::sampling, ::sampling]
    ca, pa, ua, tra, ha = o.track(d1, p0, iters=6, sampling=sampling)
    cb, pb, ub, trb, hb = o.track(masked, p0, iters=6, sampling=1)
    assert ca == cb and ua == ub and np.array_equal(pa, pb) and np.array_equal(tra, trb, equal_nan=True) and np.array_equal(ha, hb)
    n_sampled = len(range(0, H, sampling)) * len(range(0, W, sampling))
    assert ha[0] <= n_sampled
    if sampling == 1000:
        assert n_sampled == 1                       # a stride beyond the image leaves pixel (0, 0)
    else:
        assert ha[0] > 0.5 * n_sampled
    # the OMP-structured variant strides the same loop (RigidPointOptimizerOmp.cpp:70)
    cc, pc, uc, _, hc = o.track(d1, p0, iters=6, sampling=sampling, omp=True)
    assert np.array_equal(hc[:1], ha[:1]) and np.abs(pc - pa).max() < 1e-4


def test_nsq_addition_order_measured(pkg, O):
    """NormalEstimator.h:104 `n_sq = 1. + x0_sq + y0_sq;` is a cv::MatExpr on Mat_<double>: lazy evaluation folds `(1. + A) + B`
    into ONE cv::addWeighted(A, 1, B, 1, gamma = 1) (MatOp::add absorbs the scalar of the first operand; MatOp_AddEx::assign
    calls addWeighted when the scalar is real and non-zero).  Three candidate orders of the two double additions:
        0  (1 + x^2) + y^2   the line read as plain doubles (rounds 1-5 of this repository)
        1  (x^2 + y^2) + 1   addWeighted's scalar loop
        2  x^2 + (y^2 + 1)   addWeighted's SIMD loop, v_fma(a, 1, v_fma(b, 1, gamma)) -- the DEFINITION (OpenCV 4.2+)
    OpenCV is absent and unpinned, so the effect is MEASURED: the five direct planes never change (1 / n_sq rounds to the same
    float), but Q = M^-1 amplifies the last bits of n_sq: ~18 % of the Q floats differ between any two orders, normals move by
    up to ~1e-2, and a few pixels per frame flip a gate of MapGradPixelSdf.cpp:95,98 -- the key sets differ.  As with the box
    filter's summation order, "bit-exact occupancy" means: against THIS reading of the absent library."""
    W, H = 320, 240
    seq = pkg.synth.Sequence("tum", W, H, n_frames=1, seed=0)
    vs = np.float32(0.02)
    T = np.float32(5) * vs
    os_ = [O.Oracle(vs, T, W, H, seq.K, nsq_order=k) for k in range(3)]
    assert np.array_equal(O.Oracle(vs, T, W, H, seq.K).normals_cache().view(np.uint32), os_[2].normals_cache().view(np.uint32))   # default = 2
    c = [o.normals_cache() for o in os_]
    d, R, t = seq.frame(0)
    n = [o.normals(d) for o in os_]
    for a, b in ((2, 0), (2, 1), (0, 1)):
        assert np.array_equal(c[a][:5].view(np.uint32), c[b][:5].view(np.uint32))            # x0, y0, x0/n2, y0/n2, 1/n2: the same floats
        q = float((c[a][5:].view(np.uint32) != c[b][5:].view(np.uint32)).mean())
        assert 0.02 < q < 0.5, (a, b, q)                                                     # the order MATTERS for Q
        ok = np.isfinite(n[a]) & np.isfinite(n[b])
        worst = float(np.abs(n[a] - n[b])[ok].max())
        assert 1e-6 < worst < 5e-2, (a, b, worst)
