/*
 * gsdf_ba.hip -- PhotoBA on the GPU: the three voxel sweeps of PhotometricOptimizer
 * (ps_optimizer/PhotometricOptimizer.cpp) over the HBM voxel table.
 *
 *   k_ba_energy  getEnergy   :273-321   sum over voxels (|dist| <= voxel size) of sum_i |A_ij - mean_j|^2
 *   k_ba_pose    solvePose   :499-590   per keyframe b_i (6) and H_i (6x6) of the photoconsistency residual
 *   k_ba_dist    solveDist   :326-388   per-voxel scalar Gauss-Newton step on the distance
 *
 * One lane = one voxel slot; every lane loops over the keyframes it is visible in (vis_ bit-vectors),
 * projects its surface point c - dist * g^ into the keyframe and samples the float BGR image
 * bilinearly.  The work is embarrassingly parallel over voxels; the only reductions are the energy and
 * the 27 numbers per keyframe, reduced per wave (DPP), per workgroup (LDS) and across workgroups in a
 * fixed order by a second small kernel, so results are deterministic.
 */
#include "gsdf_kernels.h"
#include "gsdf_math.h"
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

struct ba_img { int W, H; const float* p; };
__device__ __forceinline__ const float* ba_px(const ba_img& im, int row, int col) { return im.p + ((size_t)row * im.W + col) * 3; }

/* interpolateImage(m = row, n = col) -- :57-77 (weights in double, BGR -> RGB) */
__device__ __forceinline__ gsdf_v3 ba_interp(float m, float n, const ba_img& im) {
    const int x = (int)floorf(m), y = (int)floorf(n);
    float t[3];
    if ((x + 1) < im.H && (y + 1) < im.W) {
        const double w1 = (y + 1.0 - n) * (m - x), w2 = (y + 1.0 - n) * (x + 1.0 - m), w3 = (n - y) * (m - x), w4 = (n - y) * (x + 1.0 - m);
        const float *a = ba_px(im, x + 1, y), *b = ba_px(im, x, y), *c = ba_px(im, x + 1, y + 1), *d = ba_px(im, x, y + 1);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            t[k] = (((float)(w1 * (double)a[k]) + (float)(w2 * (double)b[k])) + (float)(w3 * (double)c[k])) + (float)(w4 * (double)d[k]);
    } else if (y >= im.W && (x + 1) < im.H) {
        const int yc = min(y, im.W - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = (float)((double)(m - x) * (double)ba_px(im, x + 1, yc)[k]) + (float)((x + 1.0 - m) * (double)ba_px(im, x, yc)[k]);
    } else {
        const float* a = ba_px(im, min(x, im.H - 1), min(y, im.W - 1));
        t[0] = a[0]; t[1] = a[1]; t[2] = a[2];
    }
    return gsdf_v3{ t[2], t[1], t[0] };
}

/* computeImageGradient(m = row, n = col, direction) -- :80-140 */
__device__ __forceinline__ gsdf_v3 ba_grad(float m, float n, const ba_img& im, int direction) {
    const int x = (int)floorf(m), y = (int)floorf(n);
    const float w01 = m - x, w11 = n - y;
    const float w00 = (float)(1.0 - w01), w10 = (float)(1.0 - w11);
    float v0[3] = { 0.f, 0.f, 0.f }, v1[3] = { 0.f, 0.f, 0.f };
    float a = 1.f, b = 0.f;
#define BA_DIFF(o, r1, c1, r0, c0) { const float *p1 = ba_px(im, r1, c1), *p0 = ba_px(im, r0, c0); o[0] = p1[0] - p0[0]; o[1] = p1[1] - p0[1]; o[2] = p1[2] - p0[2]; }
    if (direction == 0) {
        if ((x + 1) < im.H && (y + 1) < im.W) { BA_DIFF(v0, x, y + 1, x, y); BA_DIFF(v1, x + 1, y + 1, x + 1, y); a = w00; b = w01; }
        else if ((x + 1) >= im.H) { if ((y + 1) < im.W) { BA_DIFF(v0, x, y + 1, x, y); } else { BA_DIFF(v0, x, y, x, y - 1); } }
        else { BA_DIFF(v0, x, y, x, y - 1); BA_DIFF(v1, x + 1, y, x + 1, y - 1); a = w00; b = w01; }
    } else {
        if ((x + 1) < im.H && (y + 1) < im.W) { BA_DIFF(v0, x + 1, y, x, y); BA_DIFF(v1, x + 1, y + 1, x, y + 1); a = w10; b = w11; }
        else if ((x + 1) >= im.H && (y + 1) < im.W) { BA_DIFF(v0, x, y, x - 1, y); BA_DIFF(v1, x, y + 1, x - 1, y + 1); a = w10; b = w11; }
        else { if ((x + 1) < im.H) { BA_DIFF(v0, x + 1, y, x, y); } else { BA_DIFF(v0, x, y, x - 1, y); } }
    }
#undef BA_DIFF
    return gsdf_v3{ a * v0[2] + b * v1[2], a * v0[1] + b * v1[1], a * v0[0] + b * v1[0] };
}

struct ba_args {
    gsdf_table tab;
    size_t n_slots;
    const uint32_t* vis;
    int vis_words;
    int n, W, H;
    const float* images;      /* n x H x W x 3 BGR */
    const float* R;           /* n x 9 */
    const float* t;           /* n x 3 */
    const int* frame_idx;
    float fx, fy, cx, cy, vs, reg_weight;
    float trunc_sq;
    const uint32_t* gate_list;      /* nullable: slots of the voxels with |dist| <= vs, in slot order (gsdf_ba_compact) */
    const unsigned long long* gate_count;
    void* mean_cache;               /* nullable: ba_mean per entry of gate_list (see gsdf_ba_dev) */
};
/* what the first loop of getEnergy / solvePose finds for a voxel: the mean intensity over the keyframes it is seen in (already
 * scaled by 1 / Nj), their number and their set */
struct __attribute__((aligned(8))) ba_mean { float mx, my, mz; int nj; unsigned long long seen; };
static_assert(sizeof(ba_args) == sizeof(gsdf_ba_dev), "ba_args mirrors gsdf_ba_dev (the launchers memcpy one into the other)");

struct ba_voxel { float dist, w; gsdf_v3 grad, gn, c; };

__device__ __forceinline__ bool ba_load_voxel(const ba_args& a, size_t slot, ba_voxel* v) {
    const unsigned long long bk = a.tab.bkeys[slot / GSDF_BLOCK_VOX];
    if (bk == GSDF_KEY_EMPTY) return false;
    const gsdf_payload p = a.tab.vox[slot];
    if (!(p.w > 0.f)) return false;                    /* the voxel exists iff w > 0 */
    int x, y, z;
    gsdf_key_unpack(gsdf_voxel_key(bk, (uint32_t)(slot % GSDF_BLOCK_VOX)), &x, &y, &z);
    v->w = p.w; v->dist = p.s / p.w;
    v->grad = gsdf_v3{ p.gx, p.gy, p.gz };
    v->gn = gsdf_normalized3(v->grad);
    v->c = gsdf_v3{ a.vs * (float)x, a.vs * (float)y, a.vs * (float)z };
    return true;
}
__device__ __forceinline__ bool ba_visible(const ba_args& a, size_t slot, int i) {
    const int f = a.frame_idx[i];
    if (f >= 32 * a.vis_words) return false;
    return (a.vis[slot * a.vis_words + (f >> 5)] >> (f & 31)) & 1u;
}
/* LossFunction::TRUNC_L2 (loss.h:45): a keyframe whose intensity residual is too large is left out of the voxel's sums
 * (solveDist :364, solvePose :542); every other loss value behaves like L2 in the reference's code */
__device__ __forceinline__ bool ba_truncated(const ba_args& a, const gsdf_v3& A) {
    return a.trunc_sq >= 0.f && fmaxf(A.x * A.x, fmaxf(A.y * A.y, A.z * A.z)) > a.trunc_sq;
}

/* projection shared by getIntensity / computeJc / computeJdOneFrame (:165-177) */
__device__ __forceinline__ bool ba_project(const ba_args& a, const ba_voxel& v, int i, gsdf_v3* point, float* m, float* n) {
    const float* Ri = a.R + 9 * i;
    const float* ti = a.t + 3 * i;
    const gsdf_v3 d = { v.c.x - v.dist * v.gn.x - ti[0], v.c.y - v.dist * v.gn.y - ti[1], v.c.z - v.dist * v.gn.z - ti[2] };
    const gsdf_v3 p = { gsdf_sum3(Ri[0] * d.x, Ri[3] * d.y, Ri[6] * d.z), gsdf_sum3(Ri[1] * d.x, Ri[4] * d.y, Ri[7] * d.z),
                        gsdf_sum3(Ri[2] * d.x, Ri[5] * d.y, Ri[8] * d.z) };
    const float z_inv = (float)(1. / (double)p.z);
    *m = a.fx * p.x * z_inv + a.cx;
    *n = a.fy * p.y * z_inv + a.cy;
    *point = p;
    return !(*m < 0 || *m >= a.W || *n < 0 || *n >= a.H);
}
/* interpolateImage + both computeImageGradient directions at one position (row m, column n).  Away from the image border all
 * three read the SAME four pixels -- (x, y), (x+1, y), (x, y+1), (x+1, y+1) -- so they are read once (4 scattered 12-byte loads
 * instead of 12: the distance and pose sweeps are bound by the rate at which the texture-address path takes scattered
 * addresses, ~0.4 T per second at 37 M observations per millisecond); the arithmetic per output is that of ba_interp / ba_grad,
 * operation for operation.  At the border the three functions are called as they are. */
__device__ __forceinline__ void ba_sample3(float m, float n, const ba_img& im, gsdf_v3* A, gsdf_v3* g0, gsdf_v3* g1) {
    const int x = (int)floorf(m), y = (int)floorf(n);
    if (__builtin_expect(!((x + 1) < im.H && (y + 1) < im.W), 0)) {
        *A = ba_interp(m, n, im); *g0 = ba_grad(m, n, im, 0); *g1 = ba_grad(m, n, im, 1);
        return;
    }
    const float *pa = ba_px(im, x + 1, y), *pb = ba_px(im, x, y);              /* (x+1, y+1) and (x, y+1) follow them in memory */
    float a[3], b[3], c[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = pa[k]; c[k] = pa[3 + k]; b[k] = pb[k]; d[k] = pb[3 + k]; }
    const double w1 = (y + 1.0 - n) * (m - x), w2 = (y + 1.0 - n) * (x + 1.0 - m), w3 = (n - y) * (m - x), w4 = (n - y) * (x + 1.0 - m);
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        t[k] = (((float)(w1 * (double)a[k]) + (float)(w2 * (double)b[k])) + (float)(w3 * (double)c[k])) + (float)(w4 * (double)d[k]);
    *A = gsdf_v3{ t[2], t[1], t[0] };
    const float w01 = m - x, w11 = n - y;
    const float w00 = (float)(1.0 - w01), w10 = (float)(1.0 - w11);
    float u0[3], u1[3], v0[3], v1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { u0[k] = d[k] - b[k]; u1[k] = c[k] - a[k]; v0[k] = a[k] - b[k]; v1[k] = c[k] - d[k]; }
    *g0 = gsdf_v3{ w00 * u0[2] + w01 * u1[2], w00 * u0[1] + w01 * u1[1], w00 * u0[0] + w01 * u1[0] };
    *g1 = gsdf_v3{ w10 * v0[2] + w11 * v1[2], w10 * v0[1] + w11 * v1[1], w10 * v0[0] + w11 * v1[0] };
}
/* G from already sampled gradients (ba_sample3) */
__device__ __forceinline__ void ba_pi_grad_from(const ba_args& a, const gsdf_v3& p, const gsdf_v3& g0, const gsdf_v3& g1, float* G) {
    const float z_inv = (float)(1. / (double)p.z), z_inv_sq = z_inv * z_inv;
    const float pg[6] = { a.fx * z_inv, 0.f, -a.fx * p.x * z_inv_sq, 0.f, a.fy * z_inv, -a.fy * p.y * z_inv_sq };
    const float ig[6] = { g0.x, g1.x, g0.y, g1.y, g0.z, g1.z };
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) G[3 * r + c] = ig[2 * r] * pg[c] + ig[2 * r + 1] * pg[3 + c];
}
__device__ __forceinline__ void ba_image_pi_grad(const ba_args& a, const gsdf_v3& p, float m, float n, int i, float* G) {
    const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
    const float z_inv = (float)(1. / (double)p.z), z_inv_sq = z_inv * z_inv;
    const gsdf_v3 g0 = ba_grad(n, m, im, 0), g1 = ba_grad(n, m, im, 1);
    const float pg[6] = { a.fx * z_inv, 0.f, -a.fx * p.x * z_inv_sq, 0.f, a.fy * z_inv, -a.fy * p.y * z_inv_sq };
    const float ig[6] = { g0.x, g1.x, g0.y, g1.y, g0.z, g1.z };
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) G[3 * r + c] = ig[2 * r] * pg[c] + ig[2 * r + 1] * pg[3 + c];
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ba_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
template <int N>
__device__ __forceinline__ void ba_wave_sum_to_lane63(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x111, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x112, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x114, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x118, 0xf>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x142, 0xa>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ba_dpp_add<0x143, 0xc>(v[i]);
}

/* ---- getEnergy ---- */
/* block_E: [3][gridDim.x] -- the workgroup's energy, its voxels that took part (|dist| <= vs, seen by >= 1 keyframe) and their
 * observations (voxel x keyframe pairs that project into the image): the counts are the units of the sweep's algorithmic bytes */
template <bool WRITE_MEAN>
__global__ __launch_bounds__(256) void k_ba_energy(ba_args a, double* block_E) {
    __shared__ double red[3][4];
    double E = 0.0;
    unsigned int n_act = 0u, n_obs = 0u;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n_items = a.gate_list ? (size_t)*a.gate_count : a.n_slots;
    ba_mean* const mc = WRITE_MEAN ? static_cast<ba_mean*>(a.mean_cache) : nullptr;      /* (WRITE_MEAN: the gate list is in use) */
    for (size_t item = (size_t)blockIdx.x * 256 + threadIdx.x; item < n_items; item += stride) {
        const size_t slot = a.gate_list ? (size_t)a.gate_list[item] : item;
        ba_voxel v;
        bool ok = ba_load_voxel(a, slot, &v);
        ok = ok && !(fabsf(v.dist) > a.vs);                                   /* :285 */
        gsdf_v3 mean = { 0.f, 0.f, 0.f };
        int Nj = 0;
        unsigned long long seen = 0ull;
        if (ok)
            for (int i = 0; i < a.n; ++i) {
                if (!ba_visible(a, slot, i)) continue;
                gsdf_v3 p; float m, n;
                if (!ba_project(a, v, i, &p, &m, &n)) continue;
                const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
                const gsdf_v3 A = ba_interp(n, m, im);
                mean = gsdf_v3{ mean.x + A.x, mean.y + A.y, mean.z + A.z };
                ++Nj;
                seen |= 1ull << (i & 63);
            }
        if (Nj) {
            const float inv = (float)(1. / (double)(float)Nj);
            mean = gsdf_v3{ inv * mean.x, inv * mean.y, inv * mean.z };
        }
        if (WRITE_MEAN) mc[item] = ba_mean{ mean.x, mean.y, mean.z, Nj, seen };
        if (!Nj) continue;
        n_act += 1u; n_obs += (unsigned int)Nj;
        for (int i = 0; i < a.n; ++i) {                                        /* second sweep: same samples */
            if (!((seen >> (i & 63)) & 1ull)) continue;
            gsdf_v3 p; float m, n;
            ba_project(a, v, i, &p, &m, &n);
            const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
            const gsdf_v3 A = ba_interp(n, m, im);
            const gsdf_v3 r = { A.x - mean.x, A.y - mean.y, A.z - mean.z };
            E += (double)gsdf_dot3(r, r);                                      /* :316 */
        }
    }
    double cA = (double)n_act, cO = (double)n_obs;                             /* small integers: exact in any order */
    for (int off = 32; off > 0; off >>= 1) { E += __shfl_down(E, off); cA += __shfl_down(cA, off); cO += __shfl_down(cO, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = E; red[1][threadIdx.x >> 6] = cA; red[2][threadIdx.x >> 6] = cO; }
    __syncthreads();
    if (threadIdx.x < 3) block_E[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

/* ---- solveDist ---- */
/* block_cnt (nullable): [2][gridDim.x] -- the workgroup's voxels with at least one observation and their observations */
/* 512 lanes per workgroup: the sweep is a latency chain per lane (a keyframe's projection, then its twelve scattered pixel
 * reads), so it wants every wave the registers allow -- 98 VGPRs = 4 per SIMD = two workgroups of eight waves per CU, where
 * 256-lane workgroups of the same grid left it at two (1.58 ms for C5's 1.42 M voxels) */
#define BA_DIST_THREADS 512
__global__ __launch_bounds__(BA_DIST_THREADS) void k_ba_dist(ba_args a, float damping, double* block_cnt) {
    __shared__ double red[2][BA_DIST_THREADS / 64];
    unsigned int n_act = 0u, n_obs = 0u;
    const size_t stride = (size_t)gridDim.x * BA_DIST_THREADS;
    for (size_t slot = (size_t)blockIdx.x * BA_DIST_THREADS + threadIdx.x; slot < a.n_slots; slot += stride) {
        ba_voxel v;
        if (!ba_load_voxel(a, slot, &v)) continue;
        int Nj = 0;
        gsdf_v3 sA = { 0, 0, 0 }, sD = { 0, 0, 0 }, sAD = { 0, 0, 0 }, sDD = { 0, 0, 0 };
        for (int i = 0; i < a.n; ++i) {
            if (!ba_visible(a, slot, i)) continue;
            gsdf_v3 p; float m, n;
            if (!ba_project(a, v, i, &p, &m, &n)) continue;
            const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
            gsdf_v3 A, g0, g1;
            ba_sample3(n, m, im, &A, &g0, &g1);
            if (ba_truncated(a, A)) continue;                                  /* :364 */
            ++Nj;
            float G[9];
            ba_pi_grad_from(a, p, g0, g1, G);
            const float* Ri = a.R + 9 * i;
            const gsdf_v3 Rtn = { -gsdf_sum3(Ri[0] * v.grad.x, Ri[3] * v.grad.y, Ri[6] * v.grad.z),
                                  -gsdf_sum3(Ri[1] * v.grad.x, Ri[4] * v.grad.y, Ri[7] * v.grad.z),
                                  -gsdf_sum3(Ri[2] * v.grad.x, Ri[5] * v.grad.y, Ri[8] * v.grad.z) };
            const gsdf_v3 Jd = gsdf_matvec(G, Rtn);                            /* :201 */
            sA = gsdf_v3{ sA.x + A.x, sA.y + A.y, sA.z + A.z };
            sD = gsdf_v3{ sD.x + Jd.x, sD.y + Jd.y, sD.z + Jd.z };
            sAD = gsdf_v3{ sAD.x + A.x * Jd.x, sAD.y + A.y * Jd.y, sAD.z + A.z * Jd.z };
            sDD = gsdf_v3{ sDD.x + Jd.x * Jd.x, sDD.y + Jd.y * Jd.y, sDD.z + Jd.z * Jd.z };
        }
        if (!Nj) continue;
        n_act += 1u; n_obs += (unsigned int)Nj;
        const float inv_Nj = (float)(1. / (double)(float)Nj);
        float H_dd = gsdf_sum3(sDD.x, sDD.y, sDD.z) - inv_Nj * gsdf_sum3(sD.x * sD.x, sD.y * sD.y, sD.z * sD.z);
        const float b_d = gsdf_sum3(sAD.x, sAD.y, sAD.z) - inv_Nj * gsdf_sum3(sA.x * sD.x, sA.y * sD.y, sA.z * sD.z);
        H_dd += a.reg_weight * v.w;                                            /* :383 */
        if (H_dd != 0.f) {
            /* updateDist: dist -= delta (:265-268); the table stores s = dist * w */
            gsdf_payload* P = &a.tab.vox[slot];
            P->s = (v.dist - damping * b_d / H_dd) * v.w;
        }
    }
    if (block_cnt) {                                                            /* wave-uniform: a kernel argument */
        double cA = (double)n_act, cO = (double)n_obs;
        for (int off = 32; off > 0; off >>= 1) { cA += __shfl_down(cA, off); cO += __shfl_down(cO, off); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cA; red[1][threadIdx.x >> 6] = cO; }
        __syncthreads();
        if (threadIdx.x < 2) {                                                  /* (small integers: exact in any order) */
            double t = 0.0;
            for (int w = 0; w < BA_DIST_THREADS / 64; ++w) t += red[threadIdx.x][w];
            block_cnt[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
        }
    }
}

/* ---- solvePose: per-workgroup partial (b_i, H_i upper triangle) for every keyframe ---- */
#define BA_NV 27
/* BA_POSE_SLICES workgroups (blockIdx.y) share a set of voxels: each takes every BA_POSE_SLICES-th keyframe of the second loop
 * (the per-keyframe Jacobians and their 27-value wave reduction: 85 % of a wave's instructions).  The gate list holds ~10^5
 * voxels = ~2000 waves: one wave per SIMD and a chain of ~35 k dependent instructions each.  The first loop (the voxel's mean
 * intensity) is repeated by every slice, so more slices soon cost more than they hide: measured on C5 with the per-wave
 * accumulators below, 1 / 2 / 4 / 8 slices = 237 / 232 / 288 / 384 us (308 us with float atomics and no slices). */
#ifndef BA_POSE_SLICES
#define BA_POSE_SLICES 2
#endif
/* CACHED (round 6): the first loop -- the voxel's mean intensity over its keyframes, their number and set -- is what the energy
 * sweep in front of this one computed at the very same state (gsdf_ba_optimize: getEnergy always precedes solvePose; no truncating
 * loss): read from ba_mean instead of being repeated by every slice, which also lifts what kept the slice count at two. */
#ifndef BA_POSE_SLICES_CACHED
#define BA_POSE_SLICES_CACHED 4
#endif
template <bool CACHED, int SLICES>
__global__ __launch_bounds__(256) void k_ba_pose(ba_args a, float* block_part /* [gridDim.x][n][BA_NV] */) {
    const int slice = (int)blockIdx.y;
    /* one accumulator set per WAVE, [4][n][BA_NV]: lane 63 adds its wave's 27 sums of a keyframe with plain LDS read-modify-
     * writes.  (They were float atomics on one set shared by the four waves: ds_add_f32 is lane-serial, ~190 cycles per
     * instruction -- 27 of them per keyframe and wave were two thirds of this kernel.) */
    extern __shared__ float acc[];
    for (int i = threadIdx.x; i < 4 * a.n * BA_NV; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float* wacc = acc + (size_t)(threadIdx.x >> 6) * a.n * BA_NV;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n_items = a.gate_list ? (size_t)*a.gate_count : a.n_slots;
    const size_t n_iter = (n_items + stride - 1) / stride;
    for (size_t it = 0; it < n_iter; ++it) {                                   /* uniform trip count: DPP needs whole waves */
        const size_t item = it * stride + (size_t)blockIdx.x * 256 + threadIdx.x;
        const size_t slot = item < n_items ? (a.gate_list ? (size_t)a.gate_list[item] : item) : a.n_slots;
        ba_voxel v;
        bool ok = slot < a.n_slots && ba_load_voxel(a, slot, &v);
        ok = ok && !(fabsf(v.dist) > a.vs);                                    /* :509 */
        gsdf_v3 mean = { 0.f, 0.f, 0.f };
        int Nj = 0;
        unsigned long long seen = 0ull;                                        /* keyframes (<= 64) this voxel contributes to */
        if (CACHED) {
            if (ok && item < n_items) {
                const ba_mean e = static_cast<const ba_mean*>(a.mean_cache)[item];
                mean = gsdf_v3{ e.mx, e.my, e.mz }; Nj = e.nj; seen = e.seen;     /* (the mean is stored scaled by 1 / Nj) */
            }
        } else if (ok)
            for (int i = 0; i < a.n; ++i) {
                if (!ba_visible(a, slot, i)) continue;
                gsdf_v3 p; float m, n;
                if (!ba_project(a, v, i, &p, &m, &n)) continue;
                const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
                const gsdf_v3 A = ba_interp(n, m, im);
                if (ba_truncated(a, A)) continue;                              /* :542 */
                mean = gsdf_v3{ mean.x + A.x, mean.y + A.y, mean.z + A.z };
                ++Nj;
                seen |= 1ull << (i & 63);
            }
        const float inv_Nj = Nj ? (float)(1. / (double)(float)Nj) : 0.f;
        if (!CACHED) mean = gsdf_v3{ inv_Nj * mean.x, inv_Nj * mean.y, inv_Nj * mean.z };
        for (int i = slice; i < a.n; i += SLICES) {
            const bool mine = ok && Nj && ((seen >> (i & 63)) & 1ull);
            if (!__any(mine)) continue;                                        /* wave-uniform skip */
            float val[BA_NV];
#pragma unroll
            for (int k = 0; k < BA_NV; ++k) val[k] = 0.f;
            if (mine) {
                gsdf_v3 p; float m, n;
                ba_project(a, v, i, &p, &m, &n);
                const ba_img im = { a.W, a.H, a.images + (size_t)i * a.W * a.H * 3 };
                gsdf_v3 A, g0, g1;
                ba_sample3(n, m, im, &A, &g0, &g1);
                float G[9], J[18];
                ba_pi_grad_from(a, p, g0, g1, G);
                const float* Ri = a.R + 9 * i;
                const float S[9] = { 0.f, -p.z, p.y, p.z, 0.f, -p.x, -p.y, p.x, 0.f };
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {                              /* computeJc :206-233 */
                        J[6 * r + c] = -gsdf_sum3(G[3 * r] * Ri[3 * c], G[3 * r + 1] * Ri[3 * c + 1], G[3 * r + 2] * Ri[3 * c + 2]);
                        J[6 * r + 3 + c] = gsdf_sum3(G[3 * r] * S[c], G[3 * r + 1] * S[3 + c], G[3 * r + 2] * S[6 + c]);
                    }
                const gsdf_v3 r = { A.x - mean.x, A.y - mean.y, A.z - mean.z };
#pragma unroll
                for (int c = 0; c < 6; ++c) val[c] = gsdf_sum3(r.x * J[c], r.y * J[6 + c], r.z * J[12 + c]);   /* :568 */
                int q = 6;
#pragma unroll
                for (int a1 = 0; a1 < 6; ++a1)
#pragma unroll
                    for (int a2 = a1; a2 < 6; ++a2)
                        val[q++] = (1 - inv_Nj) * gsdf_sum3(J[a1] * J[a2], J[6 + a1] * J[6 + a2], J[12 + a1] * J[12 + a2]);   /* :573 */
            }
            ba_wave_sum_to_lane63(val);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < BA_NV; ++k) wacc[i * BA_NV + k] += val[k];
            }
        }
    }
    __syncthreads();
    const int nv = a.n * BA_NV;
    for (int i = threadIdx.x; i < nv; i += 256)
        if ((i / BA_NV) % SLICES == slice)                                                           /* this slice's keyframes */
            block_part[(size_t)blockIdx.x * nv + i] = (acc[i] + acc[nv + i]) + (acc[2 * nv + i] + acc[3 * nv + i]);
}
/* one WAVE per value: lane l adds the partials of workgroups l, l + 64, ... in that order, the 64 lane sums are then added in a
 * fixed tree -- a fixed order, so deterministic like the sequential loop it replaces (one lane per value walking all 512
 * partials: 124 us of dependent strided loads for 1 350 numbers) */
__global__ __launch_bounds__(256) void k_ba_pose_reduce(const float* block_part, int n_blocks, int n_vals, float* out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n_vals) return;
    float s = 0.f;
    for (int b = lane; b < n_blocks; b += 64) s += block_part[(size_t)b * n_vals + j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) out[j] = s;
}

/* the gate of getEnergy / solvePose as a predicate over slot numbers: the voxel exists and |dist| <= vs (the same s / w division
 * and comparison as the sweeps make) */
struct ba_gate_pred {
    const unsigned long long* bkeys;
    const gsdf_payload* vox;
    float vs;
    __device__ bool operator()(const uint32_t& slot) const {
        if (bkeys[slot / GSDF_BLOCK_VOX] == GSDF_KEY_EMPTY) return false;
        const float2 ws = *reinterpret_cast<const float2*>(vox + slot);        /* w, s */
        if (!(ws.x > 0.f)) return false;
        return !(fabsf(ws.y / ws.x) > vs);
    }
};
hipError_t gsdf_ba_compact(hipStream_t s, const gsdf_ba_dev& d, uint32_t* list_out, unsigned long long* count_out, void* tmp, size_t* tmp_bytes) {
    const ba_gate_pred pred{ d.tab.bkeys, d.tab.vox, d.vs };
    return rocprim::select(tmp, *tmp_bytes, rocprim::counting_iterator<uint32_t>(0u), list_out, count_out, d.n_slots, pred, s);
}

#define BA_BLOCKS 512
void gsdf_launch_ba_energy(hipStream_t s, const gsdf_ba_dev& d, double* block_E, bool write_mean_cache) {
    ba_args a; std::memcpy(&a, &d, sizeof(a));
    if (write_mean_cache && a.gate_list && a.mean_cache) hipLaunchKernelGGL(k_ba_energy<true>, dim3(BA_BLOCKS), dim3(256), 0, s, a, block_E);
    else hipLaunchKernelGGL(k_ba_energy<false>, dim3(BA_BLOCKS), dim3(256), 0, s, a, block_E);
}
void gsdf_launch_ba_dist(hipStream_t s, const gsdf_ba_dev& d, float damping, double* block_cnt) {
    ba_args a; std::memcpy(&a, &d, sizeof(a));
    hipLaunchKernelGGL(k_ba_dist, dim3(BA_BLOCKS), dim3(BA_DIST_THREADS), 0, s, a, damping, block_cnt);
}
void gsdf_launch_ba_pose(hipStream_t s, const gsdf_ba_dev& d, float* block_part, float* out, bool use_mean_cache) {
    ba_args a; std::memcpy(&a, &d, sizeof(a));
    if (use_mean_cache && a.gate_list && a.mean_cache && a.trunc_sq < 0.f)
        hipLaunchKernelGGL((k_ba_pose<true, BA_POSE_SLICES_CACHED>), dim3(BA_BLOCKS, BA_POSE_SLICES_CACHED), dim3(256), (size_t)4 * a.n * BA_NV * sizeof(float), s, a, block_part);
    else
        hipLaunchKernelGGL((k_ba_pose<false, BA_POSE_SLICES>), dim3(BA_BLOCKS, BA_POSE_SLICES), dim3(256), (size_t)4 * a.n * BA_NV * sizeof(float), s, a, block_part);
    const int n_vals = a.n * BA_NV;
    hipLaunchKernelGGL(k_ba_pose_reduce, dim3((n_vals + 3) / 4), dim3(256), 0, s, block_part, BA_BLOCKS, n_vals, out);
}
int gsdf_ba_blocks(void) { return BA_BLOCKS; }
