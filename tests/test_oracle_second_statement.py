"""A SECOND statement of the hot path, written line by line from the reference's sources in numpy / Python float32 scalars, against
the C++ oracle (oracle/gsdf_oracle.cpp) on a small frame.  It cannot pin the oracle to the reference (nothing can, here: the
reference ships no vectors and cannot be built), but it guards the 1200-line C++ restatement against errors of
transcription: loop bounds, gates, the float loop variable, weight / truncate, the running mean, std::round, the frame counter.
The arithmetic of the absent third-party libraries follows the conventions the oracle's header states (3-term sums a + (b + c),
no FMA contraction, the float box filter summed in double) -- they are assumptions of both statements, not checked here.

  NormalEstimator::cache / compute    normals/NormalEstimator.h:81-154, 179-204
  MapGradPixelSdf::update             sdf_tracker/MapGradPixelSdf.cpp:43-122 (+ Sdf::weight / truncate, Sdf.h:72-85,
                                      float2vox / vox2float, MapGradPixelSdf.h:74-81)
  MapGradPixelSdf::tsdf / weights     sdf_tracker/MapGradPixelSdf.h:109-125
"""
import math

import numpy as np

f32 = np.float32


def sum3(a, b, c):
    return f32(a + f32(b + c))


def box_unnormalised(a, win):
    """cv::boxFilter(..., normalize=false), BORDER_REFLECT_101, sums in double (direct 2-D window sums here -- the oracle sums
    separably: the same numbers in double up to its last bits)"""
    r = win // 2
    p = np.pad(a.astype(np.float64), r, mode="reflect")
    out = np.zeros(a.shape, np.float64)
    for dy in range(win):
        for dx in range(win):
            out += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return out


def cache(K, W, H, win):
    """NormalEstimator.h:81-154, all double, results cast to float"""
    K = K.astype(np.float64)
    fx_inv, fy_inv, cx, cy = 1.0 / K[0, 0], 1.0 / K[1, 1], K[0, 2], K[1, 2]
    x0 = np.tile(np.arange(W, dtype=np.float64) - cx, (H, 1)) * fx_inv
    y0 = np.tile((np.arange(H, dtype=np.float64) - cy)[:, None], (1, W)) * fy_inv
    n_sq = x0 * x0 + (y0 * y0 + 1.0)            # :104 `1. + x0_sq + y0_sq` as cv::MatExpr evaluates it: addWeighted(x0_sq, 1, y0_sq, 1, 1), SIMD loop
    n_sq_inv = 1.0 / n_sq
    x0n, y0n = x0 * n_sq_inv, y0 * n_sq_inv
    M11 = box_unnormalised(x0 * x0 * n_sq_inv, win); M12 = box_unnormalised(x0 * y0 * n_sq_inv, win)
    M13 = box_unnormalised(x0n, win); M22 = box_unnormalised(y0 * y0 * n_sq_inv, win)
    M23 = box_unnormalised(y0n, win); M33 = box_unnormalised(n_sq_inv, win)
    det = M11 * (M22 * M33) + 2 * M12 * (M23 * M13) - (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11))
    di = 1.0 / det
    Q = [di * (M22 * M33 - M23 * M23), di * (M13 * M23 - M12 * M33), di * (M12 * M23 - M13 * M22),
         di * (M11 * M33 - M13 * M13), di * (M12 * M13 - M11 * M23), di * (M11 * M22 - M12 * M12)]
    return [a.astype(np.float32) for a in (x0, y0, x0n, y0n, n_sq_inv)] + [q.astype(np.float32) for q in Q]


def compute_normals(depth, c, win):
    """NormalEstimator.h:179-204 (float Mats; the box filter's accumulator is double, its result float)"""
    x0, y0, x0n, y0n, ninv, Q11, Q12, Q13, Q22, Q23, Q33 = c
    with np.errstate(divide="ignore", invalid="ignore"):
        z_inv = np.where(depth != 0, (f32(1) / depth).astype(np.float32), f32(0)).astype(np.float32)
        b1 = box_unnormalised((x0n * z_inv).astype(np.float32), win).astype(np.float32)
        b2 = box_unnormalised((y0n * z_inv).astype(np.float32), win).astype(np.float32)
        b3 = box_unnormalised((ninv * z_inv).astype(np.float32), win).astype(np.float32)
        nx = ((b1 * Q11 + b2 * Q12).astype(np.float32) + b3 * Q13).astype(np.float32)     # cv::Mat expression a + b + c = (a + b) + c
        ny = ((b1 * Q12 + b2 * Q22).astype(np.float32) + b3 * Q23).astype(np.float32)
        nz = ((b1 * Q13 + b2 * Q23).astype(np.float32) + b3 * Q33).astype(np.float32)
        n = np.sqrt(((nx * nx + ny * ny).astype(np.float32) + nz * nz).astype(np.float32)).astype(np.float32)
        return (nx / n).astype(np.float32), (ny / n).astype(np.float32), (nz / n).astype(np.float32)


def std_round(x):
    """std::round on a float: half away from zero (exact in double for every float below 2^52)"""
    x = float(x)
    return int(math.copysign(math.floor(abs(x) + 0.5), x))


class SecondStatement:
    """MapGradPixelSdf as the reference writes it: a dict of voxels with the RUNNING MEAN of the distance"""

    def __init__(self, voxel_size, T):
        self.vs = f32(voxel_size); self.vs_inv = f32(1.0 / float(f32(voxel_size)))       # MapGradPixelSdf.h:99-103
        self.T = f32(T); self.inv_T = f32(1.0 / float(f32(T)))                            # Sdf.h:103-107
        self.z_min, self.z_max = f32(0.5), f32(3.5)                                        # Sdf.h:67-68
        self.tsdf = {}
        self.counter = 0

    def weight(self, sdf):                                                                 # Sdf.h:76-85
        if sdf <= 0.0:
            return f32(1)
        if sdf <= self.T:
            return f32(f32(1) - f32(sdf * self.inv_T))
        return f32(0)

    def truncate(self, sdf):                                                               # Sdf.h:72-74
        return max(-self.T, min(self.T, sdf))

    def update(self, depth, x0, y0, ninv, nx, ny, nz, R, t):                               # MapGradPixelSdf.cpp:43-122
        R = R.astype(np.float32); t = t.astype(np.float32)
        factor = int(math.floor(float(f32(self.T / self.vs))))                             # :79
        H, W = depth.shape
        for m in range(H):
            for n in range(W):
                z = depth[m, n]
                if z <= self.z_min or z >= self.z_max:                                     # :87
                    continue
                xy = (x0[m, n], y0[m, n], f32(1))
                Rxy = [sum3(R[i, 0] * xy[0], R[i, 1] * xy[1], R[i, 2] * xy[2]) for i in range(3)]
                nrm = (nx[m, n], ny[m, n], nz[m, n])
                Rn = [sum3(R[i, 0] * nrm[0], R[i, 1] * nrm[1], R[i, 2] * nrm[2]) for i in range(3)]
                if float(sum3(nrm[0] * nrm[0], nrm[1] * nrm[1], nrm[2] * nrm[2])) < .1:    # :95 (float < double literal)
                    continue
                nd = sum3(nrm[0] * xy[0], nrm[1] * xy[1], nrm[2] * xy[2])
                if float(f32(f32(nd * nd) * ninv[m, n])) < .25:                            # :98
                    continue
                k = f32(-factor)
                while k <= factor:                                                         # :101, float loop variable
                    s = f32(z + f32(k * self.vs))
                    p = [f32(f32(s * Rxy[i]) + t[i]) for i in range(3)]                    # :103
                    vi = tuple(std_round(f32(self.vs_inv * p[i])) for i in range(3))      # :104
                    d = [f32(f32(self.vs * f32(vi[i])) - t[i]) for i in range(3)]
                    pz = sum3(R[0, 2] * d[0], R[1, 2] * d[1], R[2, 2] * d[2])              # row 2 of Rt
                    sdf = f32(pz - z)                                                      # :106
                    w = self.weight(sdf)
                    if w > 0:
                        v = self.tsdf.setdefault(vi, [f32(0), [f32(0), f32(0), f32(0)], f32(0)])   # dist, grad, weight
                        v[2] = f32(v[2] + w)
                        v[0] = f32(v[0] + f32(f32(f32(self.truncate(sdf) - v[0]) * w) / v[2]))       # :111
                        v[1] = [f32(v[1][i] + f32(w * Rn[i])) for i in range(3)]                    # :112
                    k = f32(k + f32(1))
        self.counter += 1


def test_second_statement_equals_the_oracle(pkg, O):
    W, H, win = 48, 36, 11
    seq = pkg.synth.Sequence("tum", W, H, n_frames=2, seed=5)
    vs, T = np.float32(0.04), np.float32(5) * np.float32(0.04)
    o = O.Oracle(vs, T, W, H, seq.K)
    st = SecondStatement(vs, T)
    c = cache(seq.K, W, H, win)
    oc = o.normals_cache()                                    # 11 planes
    for a, b in zip(c, oc):
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, float(np.abs(b).max()))    # double sums in another order, cast to float
    for i in range(2):
        d, R, t = seq.frame(i)
        n_ref = o.normals(d)
        n2 = compute_normals(d, c, win)
        m = np.isfinite(n_ref[0])
        assert np.array_equal(np.isfinite(n2[0]), m)
        for a, b in zip(n2, n_ref):
            assert np.abs(a[m] - b[m]).max() < 5e-6
        # the fusion statement is fed the ORACLE's planes and normals, so that the two are compared on identical inputs
        st.update(d, oc[0], oc[1], oc[4], n_ref[0], n_ref[1], n_ref[2], R, t)
        o.update(d, R, t)
    keys, pay = o.export()                                    # sorted by (z, y, x); payload dist, gx, gy, gz, weight
    mine = sorted(st.tsdf.items(), key=lambda kv: (kv[0][2], kv[0][1], kv[0][0]))
    assert len(mine) == len(keys) > 2000 and st.counter == o.frame_counter() == 2
    assert np.array_equal(np.array([k for k, _ in mine], np.int32), keys)          # the same voxels exist
    mp = np.array([[v[0], v[1][0], v[1][1], v[1][2], v[2]] for _, v in mine], np.float32)
    assert np.array_equal(mp.view(np.uint32), pay.view(np.uint32))                 # and hold the same bits


def normalized(v):
    """Eigen MatrixBase::normalized(): v / sqrt(|v|^2) if |v|^2 > 0 else v"""
    z = sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2])
    if z > 0:
        s = f32(np.sqrt(z))
        return [f32(v[i] / s) for i in range(3)]
    return list(v)


def tsdf_return(dist, unit, c):
    """MapGradPixelSdf.h:114 `return v.dist + 1.2*v.grad.normalized().dot(vox2float(idx) - point);` -- the float dot
    product with the UNIT gradient, times the double literal, plus dist in double, rounded once by the float return"""
    dot = sum3(unit[0] * c[0], unit[1] * c[1], unit[2] * c[2])                             # float, x0 + (x1 + x2)
    return f32(np.float64(dist) + np.float64(1.2) * np.float64(dot))


def first_pass(st, depth, K, R, t):
    """RigidPointOptimizer::optimize_sampled, the sums of its first iteration -- RigidPointOptimizer.cpp:51-84 -- with
    MapGradPixelSdf::weights / tsdf (MapGradPixelSdf.h:109-125) on the second statement's own voxel dict"""
    K = K.astype(np.float32); R = R.astype(np.float32); t = t.astype(np.float32)
    fx_inv, fy_inv, cx, cy = f32(f32(1) / K[0, 0]), f32(f32(1) / K[1, 1]), K[0, 2], K[1, 2]
    E = f32(0); g = [f32(0)] * 6; Hm = [[f32(0)] * 6 for _ in range(6)]; count = 0
    H, W = depth.shape
    for y in range(H):
        for x in range(W):
            z = depth[y, x]
            if z <= st.z_min or z >= st.z_max:                                             # :64-65
                continue
            x0 = f32(f32(f32(x) - cx) * fx_inv); y0 = f32(f32(f32(y) - cy) * fy_inv)       # :67-68
            p = (f32(x0 * z), f32(y0 * z), z)
            p = [f32(sum3(R[i, 0] * p[0], R[i, 1] * p[1], R[i, 2] * p[2]) + t[i]) for i in range(3)]   # :70
            idx = tuple(std_round(f32(st.vs_inv * p[i])) for i in range(3))
            v = st.tsdf.get(idx)
            if v is None or not (v[2] > 0):                                                # weights(): :117-125
                continue
            gn = normalized(v[1])
            gc = [f32(f32(1.2) * gn[i]) for i in range(3)]                                 # :113 scalar * Eigen expression: the literal becomes a float
            c = [f32(f32(st.vs * f32(idx[i])) - p[i]) for i in range(3)]
            phi = tsdf_return(v[0], gn, c)                                                 # :114
            E = f32(E + f32(phi * phi))
            pxg = [f32(f32(p[1] * gc[2]) - f32(p[2] * gc[1])), f32(f32(p[2] * gc[0]) - f32(p[0] * gc[2])), f32(f32(p[0] * gc[1]) - f32(p[1] * gc[0]))]
            J = gc + pxg
            g = [f32(g[i] + f32(phi * J[i])) for i in range(6)]
            Hm = [[f32(Hm[i][j] + f32(J[i] * J[j])) for j in range(6)] for i in range(6)]
            count += 1
    return E, g, Hm, count


def test_second_statement_of_the_first_tracker_pass(pkg, O):
    W, H, win = 48, 36, 11
    seq = pkg.synth.Sequence("tum", W, H, n_frames=2, seed=5)
    vs, T = np.float32(0.04), np.float32(5) * np.float32(0.04)
    o = O.Oracle(vs, T, W, H, seq.K)
    st = SecondStatement(vs, T)
    oc = o.normals_cache()
    d0, R0, t0 = seq.frame(0)
    n0 = o.normals(d0)
    p0 = np.concatenate([t0, O.R_to_quat(R0)]).astype(np.float32)
    Rq = O.quat_to_R(p0[3:])                                   # the rotation the oracle's tracker derives from the pose
    st.update(d0, oc[0], oc[1], oc[4], n0[0], n0[1], n0[2], Rq, t0)
    o.update(d0, Rq, t0)
    d1, _, _ = seq.frame(1)
    conv, pose, used, trace, hits = o.track(d1, p0, iters=1)
    E, g, Hm, count = first_pass(st, d1, seq.K, Rq, t0)
    tr = trace[0]
    assert count == int(hits[0]) == int(tr[28]) > 500
    mine = np.array([E] + g + [Hm[i][j] for i in range(6) for j in range(i, 6)], np.float32)
    assert np.array_equal(mine.view(np.uint32), tr[:28].view(np.uint32))           # E, g, H: the same bits
    xi = np.linalg.solve(np.array(Hm, np.float64), np.array(g, np.float64))        # H.llt().solve(g), in double
    assert np.abs(xi - tr[29:35]).max() <= 1e-3 * np.abs(xi).max()                 # (the float LLT of a 6x6 with condition ~1e4)


# ---- PhotoBA: getEnergy, written once more from ps_optimizer/PhotometricOptimizer.cpp:57-77 (interpolateImage), :236-260
# ---- (getIntensity), :273-321 (getEnergy) ---------------------------------------------------------------------------------

def interpolate_image(m, n, img):
    """interpolateImage(m, n, img): m indexes ROWS, n COLUMNS (the caller passes (n, m)); weights mix double and float, every
    term is rounded to float (cv::Vec3f * double), the four terms are added left to right; BGR -> RGB"""
    x, y = int(math.floor(float(m))), int(math.floor(float(n)))
    H, W = img.shape[:2]
    if (x + 1) < H and (y + 1) < W:
        w1 = (y + 1.0 - float(n)) * float(f32(m - f32(x))); w2 = (y + 1.0 - float(n)) * (x + 1.0 - float(m))
        w3 = float(f32(f32(n - f32(y)) * f32(m - f32(x))))            # float * float: the product is rounded to float (the others are double)
        w4 = float(f32(n - f32(y))) * (x + 1.0 - float(m))
        t = [f32(f32(f32(f32(w1 * float(img[x + 1, y, k])) + f32(w2 * float(img[x, y, k]))) + f32(w3 * float(img[x + 1, y + 1, k]))) +
                 f32(w4 * float(img[x, y + 1, k]))) for k in range(3)]
    elif (y + 1) < W and x >= H:
        t = [f32(f32((y + 1.0 - float(n)) * float(img[x, y, k])) + f32(float(f32(n - f32(y))) * float(img[x, y + 1, k]))) for k in range(3)]
    elif y >= W and (x + 1) < H:
        t = [f32(f32(float(f32(m - f32(x))) * float(img[x + 1, y, k])) + f32((x + 1.0 - float(m)) * float(img[x, y, k]))) for k in range(3)]
    else:
        t = [img[min(x, H - 1), min(y, W - 1), k] for k in range(3)]     # (the reference indexes out of bounds here; cannot happen behind getIntensity's test)
    return [t[2], t[1], t[0]]


def get_energy(keys, pay, vis, K, vs, images, poses, frame_idx):
    """PhotometricOptimizer::getEnergy as a double sum of its float terms (the reference adds them into one float in hash-map
    order, which leaves its own value uncertain at 1e-3: the oracle offers this order-free sum as gsdfo_ba_energy_f64)"""
    K = K.astype(np.float32); vs = f32(vs)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Himg, Wimg = images[0].shape[:2]
    E = 0.0
    n_obs = 0
    for (kx, ky, kz), p, vw in zip(keys, pay, vis):
        dist, grad = p[0], [p[1], p[2], p[3]]
        if abs(float(dist)) > float(vs):                                                   # :285
            continue
        gn = normalized(grad)
        c = [f32(vs * f32(int(kx))), f32(vs * f32(int(ky))), f32(vs * f32(int(kz)))]
        A = []
        for i, f in enumerate(frame_idx):
            if not (int(vw[f >> 5]) >> (f & 31)) & 1:                                     # :293
                continue
            R = poses[i][:3, :3].astype(np.float32); t = poses[i][:3, 3].astype(np.float32)
            d = [f32(f32(c[j] - f32(dist * gn[j])) - t[j]) for j in range(3)]              # :247
            pt = [sum3(R[0, j] * d[0], R[1, j] * d[1], R[2, j] * d[2]) for j in range(3)]  # Rt * d
            z_inv = f32(1.0 / float(pt[2]))
            m = f32(f32(f32(fx * pt[0]) * z_inv) + cx); n = f32(f32(f32(fy * pt[1]) * z_inv) + cy)
            if m < 0 or m >= Wimg or n < 0 or n >= Himg:                                   # :253
                continue
            A.append(interpolate_image(n, m, images[i]))                                   # :257: (n, m)
        if not A:
            continue
        n_obs += len(A)
        mean = [f32(0)] * 3
        for a in A:
            mean = [f32(mean[j] + a[j]) for j in range(3)]
        inv = f32(1.0 / float(f32(len(A))))
        mean = [f32(inv * mean[j]) for j in range(3)]
        for a in A:
            r = [f32(a[j] - mean[j]) for j in range(3)]
            E += float(sum3(r[0] * r[0], r[1] * r[1], r[2] * r[2]))                        # :316
    return E, n_obs


def test_second_statement_of_the_photoba_energy(pkg, O):
    W, H, n = 48, 36, 4
    seq = pkg.synth.Sequence("tum", W, H, n_frames=n, seed=5, noise=False)
    vs = np.float32(0.04); T = np.float32(5) * vs
    o = O.Oracle(vs, T, W, H, seq.K)
    for i in range(n):
        o.update(*seq.frame(i))
    imgs = np.stack([pkg.synth.render_color_bgr(seq, i) for i in range(n)])
    P = np.stack([pkg.synth.pose16(*seq.pose(i)) for i in range(n)])
    P[1:, :3, 3] += np.float32(0.01)                                 # poses off their truth: the energy is far from its minimum
    idx = np.arange(n)
    keys, pay = o.export()
    vis = o.export_vis(1)
    E2, n_obs = get_energy(keys, pay, vis, seq.K, vs, imgs, P, idx)
    ba = O.PhotoBA(o, imgs, P, idx)
    E_o = ba.energy_f64()
    assert n_obs > 1000 and E2 > 0
    assert abs(E2 - E_o) <= 1e-9 * E_o                               # the same float terms, summed in double in two orders
