/*
 * gsdf_math.h -- float32 arithmetic of the Gradient-SDF hot path with a FIXED
 * operation order, usable from device kernels and from the host facade.
 *
 * Everything here must be compiled with -ffp-contract=off (no FMA contraction)
 * and correctly rounded fp32 divide/sqrt: the voxel keys the fusion kernel
 * produces have to be bit-identical to the CPU oracle's, which restates the
 * reference's x86-64 SSE2 build (cpp/CMakeLists.txt:4).  Reference call sites
 * are cited per function (paths relative to /root/reference/cpp/include/).
 */
#ifndef GSDF_MATH_H_
#define GSDF_MATH_H_

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GSDF_HD __host__ __device__ __forceinline__
#define GSDF_UNROLL _Pragma("unroll")
#else
#define GSDF_HD inline
#define GSDF_UNROLL
#endif

struct gsdf_v3 { float x, y, z; };

/* 3-term reduction in the order of Eigen 3.4's unrolled scalar redux: x0 + (x1 + x2).
 * Used for every 3x3*3 product, dot() and squaredNorm() on the path
 * (MapGradPixelSdf.cpp:91-98,103-105; MapGradPixelSdf.h:113-114; RigidPointOptimizer.cpp:70). */
GSDF_HD float gsdf_sum3(float a, float b, float c) { return a + (b + c); }
GSDF_HD float gsdf_dot3(gsdf_v3 a, gsdf_v3 b) { return gsdf_sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
GSDF_HD gsdf_v3 gsdf_matvec(const float* R, gsdf_v3 v) {       /* R row-major 3x3 */
    gsdf_v3 r;
    r.x = gsdf_sum3(R[0] * v.x, R[1] * v.y, R[2] * v.z);
    r.y = gsdf_sum3(R[3] * v.x, R[4] * v.y, R[5] * v.z);
    r.z = gsdf_sum3(R[6] * v.x, R[7] * v.y, R[8] * v.z);
    return r;
}
GSDF_HD gsdf_v3 gsdf_cross3(gsdf_v3 a, gsdf_v3 b) {             /* RigidPointOptimizer.cpp:78 */
    gsdf_v3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
/* Eigen normalized(): n / sqrt(|n|^2) when |n|^2 > 0, else n (MapGradPixelSdf.h:113-114) */
GSDF_HD gsdf_v3 gsdf_normalized3(gsdf_v3 n) {
    const float z = gsdf_sum3(n.x * n.x, n.y * n.y, n.z * n.z);
    if (z > 0.f) {
        const float s = sqrtf(z);
        gsdf_v3 r = { n.x / s, n.y / s, n.z / s };
        return r;
    }
    return n;
}

/* The return expression of MapGradPixelSdf::tsdf -- MapGradPixelSdf.h:114:
 *     return v.dist + 1.2*v.grad.normalized().dot(vox2float(idx) - point);
 * The member calls bind before `*`: the dot product is taken in float with the UNIT gradient, the result meets the
 * double literal 1.2 (a float * double product, no Eigen expression involved, so nothing demotes the literal -- unlike
 * `1.2*v.grad.normalized()` of :113, where Eigen's scalar-times-matrix operator converts it to float), v.dist is added in
 * double and the function's float return type rounds once.  unit = grad.normalized(), d = vox2float(idx) - point.
 * Two double operations, each rounded (no fma: -ffp-contract=off). */
GSDF_HD float gsdf_tsdf_phi(float dist, gsdf_v3 unit, gsdf_v3 d) {
    return (float)((double)dist + 1.2 * (double)gsdf_dot3(unit, d));
}

/* Sdf::weight -- sdf_tracker/Sdf.h:76-85 */
GSDF_HD float gsdf_weight(float sdf, float T, float inv_T) {
    const float ramp = 1.f - sdf * inv_T;
    return sdf <= 0.f ? 1.f : (sdf <= T ? ramp : 0.f);     /* NaN -> 0, like the if-chain of the reference */
}
/* Sdf::truncate -- sdf_tracker/Sdf.h:72-74 */
GSDF_HD float gsdf_truncate(float sdf, float T) { return fmaxf(-T, fminf(T, sdf)); }

/* MapGradPixelSdf::float2vox, one component -- MapGradPixelSdf.h:74-77 (std::round = half away from zero) */
/* std::round as trunc(x + copysign(0.5 - 2^-25, x)): bit-identical to roundf for EVERY float (checked exhaustively,
 * tools/round_check.c) in 3 VALU operations; the generic lowering (trunc, |x - t| >= 0.5, copysign, add) takes 6. */
GSDF_HD float gsdf_roundf(float x) { return truncf(x + copysignf(0.49999997f, x)); }
GSDF_HD int32_t gsdf_float2vox1(float inv_vs, float p) { return (int32_t)gsdf_roundf(inv_vs * p); }

/* Eigen QuaternionBase::toRotationMatrix (SE3::rotationMatrix(), RigidPointOptimizer.cpp:53) */
GSDF_HD void gsdf_quat_to_R(const float* q /*x y z w*/, float* R) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}

/* Eigen Quaternion(Matrix3) -- used by SE3(Matrix4f) (main_scan_3d.cpp:242,252) */
GSDF_HD void gsdf_R_to_quat(const float* m, float* q /*x y z w*/) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = sqrtf(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

/* the rotation angle of xi, squared -- Sophus SO3::expAndTheta */
GSDF_HD float gsdf_se3_theta_sq(const float* xi) { return gsdf_sum3(xi[3] * xi[3], xi[4] * xi[4], xi[5] * xi[5]); }
#define GSDF_SOPHUS_EPS 1e-5f                                   /* Sophus::Constants<float>::epsilon() */

/* The part of pose7 = SE3::exp(xi) * pose7 behind the scalar coefficients (Sophus SE3::exp and the SE3 group product):
 * imag / real = the quaternion factors of SO3::expAndTheta, V = so3.matrix() when theta_small, else I + a Om + b Om^2.
 * Writes the new translation into pose7[0..2] and the product quaternion, NOT yet normalised, into qn. */
GSDF_HD void gsdf_se3_exp_mul_parts(const float* xi, float* pose7, float imag, float real, bool theta_small, float a, float b,
                                    float* qn) {
    const gsdf_v3 ups = { xi[0], xi[1], xi[2] };
    const gsdf_v3 om = { xi[3], xi[4], xi[5] };
    const float qe[4] = { imag * om.x, imag * om.y, imag * om.z, real };
    const float Om[9] = { 0.f, -om.z, om.y, om.z, 0.f, -om.x, -om.y, om.x, 0.f };
    float Om2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Om2[3 * i + j] = gsdf_sum3(Om[3 * i] * Om[j], Om[3 * i + 1] * Om[3 + j], Om[3 * i + 2] * Om[6 + j]);
    float V[9];
    if (theta_small) {
        gsdf_quat_to_R(qe, V);
    } else {
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + a * Om[i] + b * Om2[i];
    }
    const gsdf_v3 te = gsdf_matvec(V, ups);
    const float ax = qe[0], ay = qe[1], az = qe[2], aw = qe[3];
    const float bx = pose7[3], by = pose7[4], bz = pose7[5], bw = pose7[6];
    qn[3] = aw * bw - ax * bx - ay * by - az * bz;
    qn[0] = aw * bx + ax * bw + ay * bz - az * by;
    qn[1] = aw * by + ay * bw + az * bx - ax * bz;
    qn[2] = aw * bz + az * bw + ax * by - ay * bx;
    const gsdf_v3 qv = { ax, ay, az };
    const gsdf_v3 tt = { pose7[0], pose7[1], pose7[2] };
    gsdf_v3 uv = gsdf_cross3(qv, tt);
    uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
    const gsdf_v3 c2 = gsdf_cross3(qv, uv);
    pose7[0] = te.x + (tt.x + aw * uv.x + c2.x);
    pose7[1] = te.y + (tt.y + aw * uv.y + c2.y);
    pose7[2] = te.z + (tt.z + aw * uv.z + c2.z);
}

/* pose7 = SE3::exp(xi) * pose7 -- Sophus SO3::expAndTheta, SE3::exp and the SE3 group product
 * (RigidPointOptimizer.cpp:95 calls it with -xi).  pose7 = tx ty tz qx qy qz qw.
 * trig = { sinf(theta / 2), cosf(theta / 2), sinf(theta), cosf(theta) } with theta = sqrtf(theta_sq); read only when
 * theta_sq >= eps^2.  EXACT form: correctly rounded roots and divisions. */
GSDF_HD void gsdf_se3_exp_mul_trig(const float* xi, float* pose7, const float* trig) {
    const float eps = GSDF_SOPHUS_EPS;
    const float theta_sq = gsdf_se3_theta_sq(xi);
    float theta, imag, real, a = 0.f, b = 0.f;
    if (theta_sq < eps * eps) {
        theta = 0.f;
        const float theta_po4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
        real = 1.f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
    } else {
        theta = sqrtf(theta_sq);
        imag = trig[0] / theta;
        real = trig[1];
    }
    if (!(theta < eps)) {
        const float tsq = theta * theta;
        a = (1.f - trig[3]) / tsq;
        b = (theta - trig[2]) / (tsq * theta);
    }
    float qn[4];
    gsdf_se3_exp_mul_parts(xi, pose7, imag, real, theta < eps, a, b, qn);
    const float len = sqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) pose7[3 + i] = qn[i] / len;
}
GSDF_HD void gsdf_se3_exp_mul(const float* xi, float* pose7) {
    float trig[4] = { 0.f, 1.f, 0.f, 1.f };
    const float theta_sq = gsdf_se3_theta_sq(xi);
    if (!(theta_sq < GSDF_SOPHUS_EPS * GSDF_SOPHUS_EPS)) {
        const float theta = sqrtf(theta_sq), half = 0.5f * theta;
        trig[0] = sinf(half); trig[1] = cosf(half); trig[2] = sinf(theta); trig[3] = cosf(theta);
    }
    gsdf_se3_exp_mul_trig(xi, pose7, trig);
}

/* x = H^-1 g by Cholesky (Eigen H.llt().solve(g), RigidPointOptimizer.cpp:86), in the operation order of Eigen 3.4's
 * published algorithm: unblocked llt_inplace -- pivot x = A(k,k) - (sum of squares, formed first), column
 * A21 = (A21 - (A20 * A10^T, each dot product formed first)) / x -- and the completely unrolled triangular solves,
 * rhs(i) = (rhs(i) - (dot product, Eigen's halving scalar redux)) / diagonal.  H is the full symmetric 6x6, row-major.
 * A non-positive pivot stops the factorisation like Eigen's llt_inplace; the triangular solves still run (=> inf/NaN
 * for an all-zero H).  This is the EXACT form (correctly rounded divisions and roots): the host side of the ABI and the
 * rare non-positive-pivot case of the tracker's head use it; the head's common case is trk_llt_solve6_fast. */
GSDF_HD float gsdf_tree_sum(const float* t, int n) {            /* redux_novec_unroller: [0,n) = [0,n/2) + [n/2,n) */
    switch (n) {
    case 1: return t[0];
    case 2: return t[0] + t[1];
    case 3: return t[0] + (t[1] + t[2]);
    case 4: return (t[0] + t[1]) + (t[2] + t[3]);
    default: return (t[0] + t[1]) + (t[2] + (t[3] + t[4]));
    }
}
GSDF_HD void gsdf_llt_solve6(const float* Hin, const float* g, float* x) {
    float L[36];
GSDF_UNROLL
    for (int i = 0; i < 36; ++i) L[i] = Hin[i];
GSDF_UNROLL
    for (int k = 0; k < 6; ++k) {
        float d = L[6 * k + k];
        if (k > 0) {
            float sq = L[6 * k] * L[6 * k];
GSDF_UNROLL
            for (int j = 1; j < k; ++j) sq = sq + L[6 * k + j] * L[6 * k + j];
            d -= sq;
        }
        if (d <= 0.f) break;
        d = sqrtf(d);
        L[6 * k + k] = d;
GSDF_UNROLL
        for (int i = k + 1; i < 6; ++i) {
            float s = L[6 * i + k];
            if (k > 0) {
                float c = L[6 * i] * L[6 * k];
GSDF_UNROLL
                for (int j = 1; j < k; ++j) c = c + L[6 * i + j] * L[6 * k + j];
                s -= c;
            }
            L[6 * i + k] = s / d;
        }
    }
    float t[5];
GSDF_UNROLL
    for (int i = 0; i < 6; ++i) {                              /* L y = g, in place in x */
        float s = g[i];
        if (i > 0) {
GSDF_UNROLL
            for (int j = 0; j < i; ++j) t[j] = L[6 * i + j] * x[j];
            s -= gsdf_tree_sum(t, i);
        }
        x[i] = s / L[6 * i + i];
    }
GSDF_UNROLL
    for (int i = 5; i >= 0; --i) {                             /* L^T x = y */
        float s = x[i];
        if (i < 5) {
GSDF_UNROLL
            for (int j = i + 1; j < 6; ++j) t[j - i - 1] = L[6 * j + i] * x[j];
            s -= gsdf_tree_sum(t, 5 - i);
        }
        x[i] = s / L[6 * i + i];
    }
}

/* x = H^-1 b by LDL^T with symmetric diagonal pivoting (Eigen LDLT, PhotometricOptimizer.cpp:579) */
GSDF_HD void gsdf_ldlt_solve6(const float* Hin, const float* bin, float* x) {
    float A[36], bb[6];
    int perm[6];
    for (int i = 0; i < 36; ++i) A[i] = Hin[i];
    for (int i = 0; i < 6; ++i) perm[i] = i;
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        float best = fabsf(A[7 * k]);
        for (int i = k + 1; i < 6; ++i) if (fabsf(A[7 * i]) > best) { best = fabsf(A[7 * i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; ++j) { const float tmp = A[6 * k + j]; A[6 * k + j] = A[6 * piv + j]; A[6 * piv + j] = tmp; }
            for (int j = 0; j < 6; ++j) { const float tmp = A[6 * j + k]; A[6 * j + k] = A[6 * j + piv]; A[6 * j + piv] = tmp; }
            const int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
        }
        const float d = A[7 * k];
        if (d == 0.f) continue;
        for (int i = k + 1; i < 6; ++i) A[6 * i + k] /= d;
        for (int i = k + 1; i < 6; ++i)
            for (int j = k + 1; j <= i; ++j) { A[6 * i + j] -= A[6 * i + k] * d * A[6 * j + k]; A[6 * j + i] = A[6 * i + j]; }
    }
    for (int i = 0; i < 6; ++i) bb[i] = bin[perm[i]];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) bb[i] -= A[6 * i + j] * bb[j];
    for (int i = 0; i < 6; ++i) bb[i] = A[7 * i] != 0.f ? bb[i] / A[7 * i] : 0.f;
    for (int i = 5; i >= 0; --i) for (int j = i + 1; j < 6; ++j) bb[i] -= A[6 * j + i] * bb[j];
    for (int i = 0; i < 6; ++i) x[perm[i]] = bb[i];
}

#endif /* GSDF_MATH_H_ */
