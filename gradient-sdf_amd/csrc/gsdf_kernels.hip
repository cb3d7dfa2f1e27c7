/*
 * gsdf_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Gradient-SDF hot path.
 *
 *   k_normals_cache  NormalEstimator::cache            normals/NormalEstimator.h:81-154
 *   k_normals        NormalEstimator::compute          normals/NormalEstimator.h:179-204
 *   k_fuse           MapGradPixelSdf::update           sdf_tracker/MapGradPixelSdf.cpp:43-122
 *   k_track_pass     RigidPointOptimizer::optimize_sampled, one Gauss-Newton pass
 *                                                      sdf_tracker/RigidPointOptimizer.cpp:51-96
 *   k_query          MapGradPixelSdf::weights / tsdf   sdf_tracker/MapGradPixelSdf.h:109-125
 *
 * Compile with -ffp-contract=off: voxel keys must be bit-identical to the CPU oracle.
 * There is no dense contraction anywhere on this path, hence no MFMA; the kernels are
 * bound by LDS/L2 atomics and HBM gathers.  Wave = 64 lanes throughout.
 */
#include "gsdf_kernels.h"
#include "gsdf_math.h"

#include <hip/hip_runtime.h>

#define FULL_MASK 0xFFFFFFFFFFFFFFFFull

/* ------------------------------------------------------------------------------------------------
 * wave-64 sum with DPP row shifts + row broadcasts (no LDS traffic); result valid in every lane.
 * ---------------------------------------------------------------------------------------------- */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);   /* row_shr:1 */
    v = dpp_add<0x112, 0xf>(v);   /* row_shr:2 */
    v = dpp_add<0x114, 0xf>(v);   /* row_shr:4 */
    v = dpp_add<0x118, 0xf>(v);   /* row_shr:8  -> lane 15 of each row holds the row sum */
    v = dpp_add<0x142, 0xa>(v);   /* row_bcast:15 into rows 1,3 */
    v = dpp_add<0x143, 0xc>(v);   /* row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum */
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int reflect101(int i, int n) {      /* cv::BORDER_REFLECT_101 */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* ------------------------------------------------------------------------------------------------
 * table clear
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_table_clear(gsdf_slot* slots, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    /* one 32-byte slot = two 16-byte stores */
    const uint4 a = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
    const uint4 b = make_uint4(0u, 0u, 0u, 0u);
    for (; i < n; i += stride) {
        uint4* p = reinterpret_cast<uint4*>(slots + i);
        p[0] = a;
        p[1] = b;
    }
}
void gsdf_launch_table_clear(hipStream_t s, gsdf_table tab, size_t n_slots) {
    hipLaunchKernelGGL(k_table_clear, dim3(2048), dim3(256), 0, s, tab.slots, n_slots);
}

/* ------------------------------------------------------------------------------------------------
 * NormalEstimator::cache -- all in double, one thread per pixel, separable summation order
 * (row sums ascending dx, then ascending dy) identical to the oracle's box_sum.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_normals_cache(int W, int H, double fx_inv, double fy_inv, double cx,
                                                       double cy, int r, float* __restrict__ out) {
    const int u = blockIdx.x * 32 + (threadIdx.x & 31);
    const int v = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (u >= W || v >= H) return;
    double M11 = 0, M12 = 0, M13 = 0, M22 = 0, M23 = 0, M33 = 0;
    for (int dy = -r; dy <= r; ++dy) {
        const int vv = reflect101(v + dy, H);
        const double y = fy_inv * ((double)vv - cy);
        const double y_sq = y * y;
        double h11 = 0, h12 = 0, h13 = 0, h22 = 0, h23 = 0, h33 = 0;
        for (int dx = -r; dx <= r; ++dx) {
            const int uu = reflect101(u + dx, W);
            const double x = fx_inv * ((double)uu - cx);
            const double x_sq = x * x, xy = x * y;
            const double n_sq = 1. + x_sq + y_sq;
            const double ni = 1. / n_sq;
            h11 += x_sq * ni; h12 += xy * ni; h13 += x * ni;
            h22 += y_sq * ni; h23 += y * ni;  h33 += ni;
        }
        M11 += h11; M12 += h12; M13 += h13; M22 += h22; M23 += h23; M33 += h33;
    }
    const double det = M11 * (M22 * M33) + 2 * (M12 * (M23 * M13)) -
                       (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11));
    const double det_inv = 1. / det;
    const double x = fx_inv * ((double)u - cx);
    const double y = fy_inv * ((double)v - cy);
    const double n_sq = 1. + x * x + y * y;
    const double ni = 1. / n_sq;
    const size_t N = (size_t)W * H, i = (size_t)v * W + u;
    out[0 * N + i] = (float)x;
    out[1 * N + i] = (float)y;
    out[2 * N + i] = (float)(x * ni);
    out[3 * N + i] = (float)(y * ni);
    out[4 * N + i] = (float)ni;
    out[5 * N + i] = (float)(det_inv * (M22 * M33 - M23 * M23));
    out[6 * N + i] = (float)(det_inv * (M13 * M23 - M12 * M33));
    out[7 * N + i] = (float)(det_inv * (M12 * M23 - M13 * M22));
    out[8 * N + i] = (float)(det_inv * (M11 * M33 - M13 * M13));
    out[9 * N + i] = (float)(det_inv * (M12 * M13 - M11 * M23));
    out[10 * N + i] = (float)(det_inv * (M11 * M22 - M12 * M12));
}
void gsdf_launch_normals_cache(hipStream_t s, int W, int H, const float* K, int win, float* planes11) {
    const double fx_inv = 1. / (double)K[0], fy_inv = 1. / (double)K[4];
    dim3 grid((W + 31) / 32, (H + 7) / 8);
    hipLaunchKernelGGL(k_normals_cache, grid, dim3(256), 0, s, W, H, fx_inv, fy_inv, (double)K[2], (double)K[5],
                       win / 2, planes11);
}

/* ------------------------------------------------------------------------------------------------
 * NormalEstimator::compute.  32x8 output tile per workgroup; the depth tile + halo is staged
 * through LDS as the three products {x0,y0,1}/n^2 * 1/z, row sums in double in LDS, then the
 * column sums, Q*b and the normalisation in registers.
 * ---------------------------------------------------------------------------------------------- */
#define NRM_TX 32
#define NRM_TY 8
#define NRM_RMAX 7
__global__ __launch_bounds__(256) void k_normals(gsdf_frame_geom g, int r, gsdf_ncache nc,
                                                 const float* __restrict__ depth, float* __restrict__ nx,
                                                 float* __restrict__ ny, float* __restrict__ nz,
                                                 const gsdf_dev_state* gate) {
    if (gate && !gate->converged) return;
    __shared__ float prod[3][NRM_TY + 2 * NRM_RMAX][NRM_TX + 2 * NRM_RMAX + 1];
    __shared__ double rows[3][NRM_TY + 2 * NRM_RMAX][NRM_TX];
    const int W = g.W, H = g.H;
    const int tx0 = blockIdx.x * NRM_TX, ty0 = blockIdx.y * NRM_TY;
    const int PW = NRM_TX + 2 * r, PH = NRM_TY + 2 * r;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < PW * PH; idx += 256) {
        const int ly = idx / PW, lx = idx - ly * PW;
        const int gy = reflect101(ty0 + ly - r, H), gx = reflect101(tx0 + lx - r, W);
        const size_t i = (size_t)gy * W + gx;
        const float z = depth[i];
        const float zi = z != 0.f ? 1.f / z : 0.f;          /* NormalEstimator.h:183-187 */
        prod[0][ly][lx] = nc.x0n[i] * zi;                   /* :191-193 */
        prod[1][ly][lx] = nc.y0n[i] * zi;
        prod[2][ly][lx] = nc.ninv[i] * zi;
    }
    __syncthreads();
    for (int idx = tid; idx < PH * NRM_TX; idx += 256) {
        const int ly = idx / NRM_TX, x = idx - ly * NRM_TX;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int dx = 0; dx <= 2 * r; ++dx) {
            s0 += (double)prod[0][ly][x + dx];
            s1 += (double)prod[1][ly][x + dx];
            s2 += (double)prod[2][ly][x + dx];
        }
        rows[0][ly][x] = s0; rows[1][ly][x] = s1; rows[2][ly][x] = s2;
    }
    __syncthreads();
    const int x = tid & (NRM_TX - 1), y = tid / NRM_TX;
    const int px = tx0 + x, py = ty0 + y;
    if (px >= W || py >= H) return;
    double b1 = 0, b2 = 0, b3 = 0;
    for (int dy = 0; dy <= 2 * r; ++dy) {
        b1 += rows[0][y + dy][x];
        b2 += rows[1][y + dy][x];
        b3 += rows[2][y + dy][x];
    }
    const size_t i = (size_t)py * W + px;
    const float c1 = (float)b1, c2 = (float)b2, c3 = (float)b3;
    const float q11 = nc.q11[i], q12 = nc.q12[i], q13 = nc.q13[i], q22 = nc.q22[i], q23 = nc.q23[i], q33 = nc.q33[i];
    const float vx = (c1 * q11 + c2 * q12) + c3 * q13;      /* :195-197 */
    const float vy = (c1 * q12 + c2 * q22) + c3 * q23;
    const float vz = (c1 * q13 + c2 * q23) + c3 * q33;
    const float n = sqrtf((vx * vx + vy * vy) + vz * vz);   /* :199 */
    nx[i] = vx / n; ny[i] = vy / n; nz[i] = vz / n;         /* :201-203 */
}
void gsdf_launch_normals(hipStream_t s, const gsdf_frame_geom& g, int win, const gsdf_ncache& nc,
                         const float* depth, float* nx, float* ny, float* nz, const gsdf_dev_state* gate) {
    dim3 grid((g.W + NRM_TX - 1) / NRM_TX, (g.H + NRM_TY - 1) / NRM_TY);
    hipLaunchKernelGGL(k_normals, grid, dim3(256), 0, s, g, win / 2, nc, depth, nx, ny, nz, gate);
}

/* ------------------------------------------------------------------------------------------------
 * MapGradPixelSdf::update -- fusion.
 *
 * One workgroup = one 16x16 pixel tile (4 waves, each an 8x8 sub-tile so that a wave's 64 rays
 * stay spatially compact).  Each lane walks its ray's 2*factor+1 samples.  Neighbouring pixels and
 * consecutive samples hit the same voxels (~10 updates per distinct voxel per frame), so updates
 * are first combined in a workgroup-private open-addressed table in LDS with LDS atomics and only
 * the distinct voxels of the tile are flushed to the HBM table (1 probe + 5 float atomics each).
 * Samples that do not fit the LDS table go to HBM directly.
 * ---------------------------------------------------------------------------------------------- */
#define FUSE_T 16
#define FUSE_LCAP 2048
#define FUSE_LPROBE 16

struct fuse_args {
    gsdf_frame_geom g;
    gsdf_ncache nc;
    const float* depth;
    const float *nx, *ny, *nz;
    gsdf_pose_arg pose;
    int use_dev_pose;
    gsdf_table tab;
    gsdf_dev_state* st;
    unsigned long long* blk_counters;   /* [n_blocks][4]: last n_upd, last n_valid, cum n_upd, cum n_valid */
};

__device__ __forceinline__ void hbm_accumulate(const gsdf_table& T, unsigned long long key, float w, float s,
                                               float gx, float gy, float gz, gsdf_dev_state* st) {
    bool inserted;
    const long long slot = gsdf_find_or_insert(T, key, &inserted);
    if (slot < 0) { atomicOr(&st->status, GSDF_STATUS_TABLE_FULL); return; }
    gsdf_slot* p = T.slots + slot;
    unsafeAtomicAdd(&p->w, w);
    unsafeAtomicAdd(&p->s, s);
    unsafeAtomicAdd(&p->gx, gx);
    unsafeAtomicAdd(&p->gy, gy);
    unsafeAtomicAdd(&p->gz, gz);
}

__global__ __launch_bounds__(256) void k_fuse(fuse_args a) {
    __shared__ unsigned long long lkey[FUSE_LCAP];
    __shared__ float lw[FUSE_LCAP], ls[FUSE_LCAP], lgx[FUSE_LCAP], lgy[FUSE_LCAP], lgz[FUSE_LCAP];
    __shared__ float red[8];
    if (a.use_dev_pose && !a.st->converged) return;        /* main_scan_3d.cpp:261: if (conv) update */
    const int tid = threadIdx.x;
    for (int i = tid; i < FUSE_LCAP; i += 256) {
        lkey[i] = GSDF_KEY_EMPTY;
        lw[i] = 0.f; ls[i] = 0.f; lgx[i] = 0.f; lgy[i] = 0.f; lgz[i] = 0.f;
    }
    float R[9], t[3];
    if (a.use_dev_pose) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = a.st->R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = a.st->pose7[i];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = a.pose.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = a.pose.t[i];
    }
    __syncthreads();

    const gsdf_frame_geom& g = a.g;
    const int wave = tid >> 6, lane = tid & 63;
    const int px = blockIdx.x * FUSE_T + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * FUSE_T + (wave >> 1) * 8 + (lane >> 3);
    bool valid = px < g.W && py < g.H;
    float z = 0.f;
    gsdf_v3 Rxy = { 0.f, 0.f, 0.f }, Rn = { 0.f, 0.f, 0.f };
    if (valid) {
        const size_t idx = (size_t)py * g.W + px;
        z = a.depth[idx];
        valid = !(z <= g.zmin || z >= g.zmax);                             /* MapGradPixelSdf.cpp:87 */
        if (valid) {
            const gsdf_v3 xy = { a.nc.x0[idx], a.nc.y0[idx], 1.f };        /* :90 */
            const gsdf_v3 n = { a.nx[idx], a.ny[idx], a.nz[idx] };         /* :92 */
            Rxy = gsdf_matvec(R, xy);                                      /* :91 */
            Rn = gsdf_matvec(R, n);                                        /* :93 */
            if ((double)gsdf_dot3(n, n) < .1) valid = false;               /* :95 */
            const float nd = gsdf_dot3(n, xy);
            if (nd * nd * a.nc.ninv[idx] < .25) valid = false;             /* :98 */
        }
    }
    float n_upd = 0.f;
    if (valid) {
        for (int kk = -g.factor; kk <= g.factor; ++kk) {                   /* :101 */
            const float s = z + (float)kk * g.vs;
            const float pxw = s * Rxy.x + t[0], pyw = s * Rxy.y + t[1], pzw = s * Rxy.z + t[2];   /* :103 */
            const int vx = gsdf_float2vox1(g.inv_vs, pxw);                 /* :104 */
            const int vy = gsdf_float2vox1(g.inv_vs, pyw);
            const int vz = gsdf_float2vox1(g.inv_vs, pzw);
            const float dx = g.vs * (float)vx - t[0], dy = g.vs * (float)vy - t[1], dz = g.vs * (float)vz - t[2];
            const float pc_z = gsdf_sum3(R[2] * dx, R[5] * dy, R[8] * dz); /* :105  (Rt row 2) */
            const float sdf = pc_z - z;                                    /* :106 */
            const float w = gsdf_weight(sdf, g.T, g.inv_T);                /* :107 */
            if (w > 0.f) {
                if (!gsdf_key_in_range(vx, vy, vz)) { atomicOr(&a.st->status, GSDF_STATUS_KEY_RANGE); continue; }
                n_upd += 1.f;
                const unsigned long long key = gsdf_key_pack(vx, vy, vz);
                const float ws = w * gsdf_truncate(sdf, g.T);              /* :111 as additive sum */
                const float wgx = w * Rn.x, wgy = w * Rn.y, wgz = w * Rn.z;   /* :112 */
                uint32_t h = (gsdf_hash(key) >> 11) & (FUSE_LCAP - 1);
                bool done = false;
                for (int p = 0; p < FUSE_LPROBE; ++p) {
                    unsigned long long cur = *(volatile unsigned long long*)&lkey[h];
                    if (cur == GSDF_KEY_EMPTY) cur = atomicCAS(&lkey[h], GSDF_KEY_EMPTY, key);
                    if (cur == GSDF_KEY_EMPTY || cur == key) {
                        atomicAdd(&lw[h], w);
                        atomicAdd(&ls[h], ws);
                        atomicAdd(&lgx[h], wgx);
                        atomicAdd(&lgy[h], wgy);
                        atomicAdd(&lgz[h], wgz);
                        done = true;
                        break;
                    }
                    h = (h + 1) & (FUSE_LCAP - 1);
                }
                if (!done) hbm_accumulate(a.tab, key, w, ws, wgx, wgy, wgz, a.st);
            }
        }
    }
    __syncthreads();
    /* flush the tile's distinct voxels to the HBM table */
    for (int i = tid; i < FUSE_LCAP; i += 256) {
        const unsigned long long key = lkey[i];
        if (key != GSDF_KEY_EMPTY) hbm_accumulate(a.tab, key, lw[i], ls[i], lgx[i], lgy[i], lgz[i], a.st);
    }
    /* per-workgroup counters (plain stores into this workgroup's own row: no hot atomics) */
    const float wu = wave_sum(n_upd), wv = wave_sum(valid ? 1.f : 0.f);
    if (lane == 0) { red[wave] = wu; red[4 + wave] = wv; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long nu = (unsigned long long)(red[0] + red[1] + red[2] + red[3]);
        const unsigned long long nv = (unsigned long long)(red[4] + red[5] + red[6] + red[7]);
        unsigned long long* c = a.blk_counters + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        c[0] = nu; c[1] = nv; c[2] += nu; c[3] += nv;
        if (blockIdx.x == 0 && blockIdx.y == 0) a.st->frames += 1;        /* :120 increase_counter() */
    }
}

void gsdf_launch_fuse(hipStream_t s, const gsdf_frame_geom& g, const gsdf_ncache& nc, const float* depth,
                         const float* nx, const float* ny, const float* nz, const gsdf_pose_arg& pose,
                         int use_dev_pose, gsdf_table tab, gsdf_dev_state* st, unsigned long long* blk_counters) {
    fuse_args a;
    a.g = g; a.nc = nc; a.depth = depth; a.nx = nx; a.ny = ny; a.nz = nz; a.pose = pose;
    a.use_dev_pose = use_dev_pose; a.tab = tab; a.st = st; a.blk_counters = blk_counters;
    dim3 grid((g.W + FUSE_T - 1) / FUSE_T, (g.H + FUSE_T - 1) / FUSE_T);
    hipLaunchKernelGGL(k_fuse, grid, dim3(256), 0, s, a);
}
int gsdf_fuse_grid_blocks(int W, int H) { return ((W + FUSE_T - 1) / FUSE_T) * ((H + FUSE_T - 1) / FUSE_T); }

/* ------------------------------------------------------------------------------------------------
 * RigidPointOptimizer::optimize_sampled -- one Gauss-Newton pass per launch.
 *
 * Every lane gathers the voxel under its back-projected pixel (one 32-byte slot), forms the
 * residual phi and the 6-vector J and accumulates the 29 normal-equation sums in registers.
 * Sums are reduced with wavefront DPP shuffles, then across the 4 waves through LDS, and each
 * workgroup stores one partial row.  The last workgroup to arrive (agent-scope release /
 * acquire around a ticket counter) adds the rows in a fixed order, solves the 6x6 system,
 * applies SE3::exp(-xi) to the device-resident pose and raises the done/converged flags, so
 * the host never has to look at an iteration.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(GSDF_TRACK_BLOCK) void k_track_begin(gsdf_dev_state* st, int max_passes, float conv_sq,
                                                                  float damping) {
    if (threadIdx.x == 0) {
        st->done = max_passes <= 0 ? 1 : 0;
        st->converged = 0;
        st->passes = 0;
        st->max_passes = max_passes;
        st->ticket = 0u;
        st->last_hits = 0.f;
        st->conv_sq = conv_sq;
        st->damping = damping;
        gsdf_quat_to_R(st->pose7 + 3, st->R);
    }
}
void gsdf_launch_track_begin(hipStream_t s, gsdf_dev_state* st, int max_passes, float conv_sq, float damping) {
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, s, st, max_passes, conv_sq, damping);
}

__global__ __launch_bounds__(GSDF_TRACK_BLOCK) void k_track_pass(gsdf_frame_geom g, const float* __restrict__ depth,
                                                                 gsdf_table tab, gsdf_dev_state* st,
                                                                 float* partials) {
    if (st->done) return;
    __shared__ float wsum[GSDF_TRACK_BLOCK / 64][32];
    __shared__ float tot[32];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = st->R[i];                          /* RigidPointOptimizer.cpp:53-54 */
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = st->pose7[i];
    const float fx_inv = 1.f / g.fx, fy_inv = 1.f / g.fy;                 /* :46-47 */

    float acc[GSDF_TRACK_NSUM];
#pragma unroll
    for (int i = 0; i < GSDF_TRACK_NSUM; ++i) acc[i] = 0.f;

    const int N = g.W * g.H;
    for (int pix = blockIdx.x * GSDF_TRACK_BLOCK + tid; pix < N; pix += gridDim.x * GSDF_TRACK_BLOCK) {
        const float z = depth[pix];
        if (z <= g.zmin || z >= g.zmax) continue;                         /* :64-65 */
        const int y = pix / g.W, x = pix - y * g.W;
        const float x0 = ((float)x - g.cx) * fx_inv;                      /* :67-68 */
        const float y0 = ((float)y - g.cy) * fy_inv;
        const gsdf_v3 pc = { x0 * z, y0 * z, z };
        const gsdf_v3 Rp = gsdf_matvec(R, pc);
        const gsdf_v3 p = { Rp.x + t[0], Rp.y + t[1], Rp.z + t[2] };     /* :70 */
        const int vx = gsdf_float2vox1(g.inv_vs, p.x), vy = gsdf_float2vox1(g.inv_vs, p.y),
                  vz = gsdf_float2vox1(g.inv_vs, p.z);
        if (!gsdf_key_in_range(vx, vy, vz)) continue;
        const gsdf_slot* sl = gsdf_find(tab, gsdf_key_pack(vx, vy, vz)); /* weights(): MapGradPixelSdf.h:117-125 */
        if (!sl) continue;
        const float w0 = sl->w;
        if (!(w0 > 0.f)) continue;                                        /* :73 */
        /* tsdf(): MapGradPixelSdf.h:109-115 */
        const gsdf_v3 gn = gsdf_normalized3(gsdf_v3{ sl->gx, sl->gy, sl->gz });
        const gsdf_v3 gr = { 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };
        const gsdf_v3 d = { g.vs * (float)vx - p.x, g.vs * (float)vy - p.y, g.vs * (float)vz - p.z };
        const float phi = sl->s / w0 + gsdf_dot3(gr, d);
        const gsdf_v3 pxg = gsdf_cross3(p, gr);                           /* :78 */
        const float J[6] = { gr.x, gr.y, gr.z, pxg.x, pxg.y, pxg.z };
        acc[0] += phi * phi;                                              /* :76 */
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[1 + i] += phi * J[i];             /* :79 */
        int q = 7;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[q++] += J[i] * J[j];          /* :80 */
        acc[28] += 1.f;                                                   /* :81 */
    }
#pragma unroll
    for (int i = 0; i < GSDF_TRACK_NSUM; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) wsum[wave][i] = v;
    }
    __syncthreads();
    if (tid < GSDF_TRACK_NSUM) {
        float v = wsum[0][tid];
#pragma unroll
        for (int w = 1; w < GSDF_TRACK_BLOCK / 64; ++w) v += wsum[w][tid];
        partials[(size_t)blockIdx.x * 32 + tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == gridDim.x - 1) ? 1 : 0;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;

    /* ---- last workgroup: fixed-order sum over workgroups, solve, pose update ---- */
    {
        const int j = tid >> 3, sub = tid & 7;
        float v = 0.f;
        if (j < GSDF_TRACK_NSUM)
            for (unsigned int b = sub; b < gridDim.x; b += 8)
                v += __hip_atomic_load(&partials[(size_t)b * 32 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        if (j < GSDF_TRACK_NSUM && sub == 0) tot[j] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float gvec[6], Hm[36];
#pragma unroll
        for (int i = 0; i < 6; ++i) gvec[i] = tot[1 + i];
        int q = 7;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 6; ++j) { Hm[6 * i + j] = tot[q]; Hm[6 * j + i] = tot[q]; ++q; }
        float xi[6];
        gsdf_llt_solve6(Hm, gvec, xi);                                    /* :86 */
        const float damping = st->damping;
        for (int i = 0; i < 6; ++i) xi[i] = damping * xi[i];
        const float nrm = gsdf_sum3(xi[0] * xi[0], xi[1] * xi[1], xi[2] * xi[2]) +
                          gsdf_sum3(xi[3] * xi[3], xi[4] * xi[4], xi[5] * xi[5]);
        const int passes = st->passes + 1;
        st->passes = passes;
        st->last_hits = tot[28];
        st->n_hit += (unsigned long long)tot[28];
        if (nrm < st->conv_sq) {                                          /* :88-91 (xi is NOT applied) */
            st->converged = 1;
            st->done = 1;
        } else {
            bool nan = false;
            for (int i = 0; i < 6; ++i) nan = nan || isnan(xi[i]);
            if (!nan) {                                                   /* :94-95 */
                float mxi[6];
                for (int i = 0; i < 6; ++i) mxi[i] = -xi[i];
                float pose[7];
                for (int i = 0; i < 7; ++i) pose[i] = st->pose7[i];
                gsdf_se3_exp_mul(mxi, pose);
                for (int i = 0; i < 7; ++i) st->pose7[i] = pose[i];
                gsdf_quat_to_R(pose + 3, st->R);
            }
            if (passes >= st->max_passes) st->done = 1;                   /* :98 return false */
        }
        st->ticket = 0u;
    }
}
void gsdf_launch_track_pass(hipStream_t s, const gsdf_frame_geom& g, const float* depth, gsdf_table tab,
                            gsdf_dev_state* st, float* partials, int n_blocks) {
    hipLaunchKernelGGL(k_track_pass, dim3(n_blocks), dim3(GSDF_TRACK_BLOCK), 0, s, g, depth, tab, st, partials);
}

/* per-frame log row: pose7, converged, passes, hits of the last pass (main_scan_3d.cpp:268-280) */
__global__ void k_frame_log(gsdf_dev_state* st, float* rows, long long max_rows) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long r = st->log_rows;
        if (r < max_rows) {
            float* o = rows + 10 * r;
            for (int i = 0; i < 7; ++i) o[i] = st->pose7[i];
            o[7] = (float)st->converged; o[8] = (float)st->passes; o[9] = st->last_hits;
        }
        st->log_rows = r + 1;
    }
}
void gsdf_launch_frame_log(hipStream_t s, gsdf_dev_state* st, float* log_rows, long long max_rows) {
    hipLaunchKernelGGL(k_frame_log, dim3(1), dim3(64), 0, s, st, log_rows, max_rows);
}

struct pose7_arg { float p[7]; };
__global__ void k_set_pose(gsdf_dev_state* st, pose7_arg a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 7; ++i) st->pose7[i] = a.p[i];
        gsdf_quat_to_R(st->pose7 + 3, st->R);
    }
}
void gsdf_launch_set_pose(hipStream_t s, gsdf_dev_state* st, const float*, const float pose7_host[7]) {
    pose7_arg a;
    for (int i = 0; i < 7; ++i) a.p[i] = pose7_host[i];
    hipLaunchKernelGGL(k_set_pose, dim3(1), dim3(64), 0, s, st, a);
}

/* ------------------------------------------------------------------------------------------------
 * export / merge / query
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_export(gsdf_table tab, size_t n_slots, unsigned long long* keys_out,
                                                float* payload_out, unsigned long long* counter, long long max_n,
                                                int raw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_slots; i += stride) {
        const gsdf_slot sl = tab.slots[i];
        if (sl.key == GSDF_KEY_EMPTY) continue;
        const unsigned long long o = atomicAdd(counter, 1ull);
        if ((long long)o >= max_n) continue;
        if (keys_out) keys_out[o] = sl.key;
        if (payload_out) {
            float* p = payload_out + 5 * o;
            if (raw) { p[0] = sl.s; p[1] = sl.gx; p[2] = sl.gy; p[3] = sl.gz; p[4] = sl.w; }
            else     { p[0] = sl.s / sl.w; p[1] = sl.gx; p[2] = sl.gy; p[3] = sl.gz; p[4] = sl.w; }
        }
    }
}
void gsdf_launch_export(hipStream_t s, gsdf_table tab, size_t n_slots, unsigned long long* keys_out,
                        float* payload_out, unsigned long long* counter, long long max_n, int raw) {
    hipLaunchKernelGGL(k_export, dim3(2048), dim3(256), 0, s, tab, n_slots, keys_out, payload_out, counter, max_n, raw);
}

__global__ __launch_bounds__(256) void k_merge_raw(gsdf_table tab, const int32_t* keys, const float* payload,
                                                   long long n, gsdf_dev_state* st) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
        if (!gsdf_key_in_range(x, y, z)) { atomicOr(&st->status, GSDF_STATUS_KEY_RANGE); continue; }
        const float* p = payload + 5 * i;
        hbm_accumulate(tab, gsdf_key_pack(x, y, z), p[4], p[0], p[1], p[2], p[3], st);
    }
}
void gsdf_launch_merge_raw(hipStream_t s, gsdf_table tab, const int32_t* keys, const float* payload, long long n,
                           gsdf_dev_state* st) {
    if (n <= 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_merge_raw, dim3(blocks), dim3(256), 0, s, tab, keys, payload, n, st);
}

/* MapGradPixelSdf::weights + ::tsdf at arbitrary points -- MapGradPixelSdf.h:109-125 */
__global__ __launch_bounds__(256) void k_query(gsdf_table tab, float vs, float inv_vs, const float* pts, long long n,
                                               float* dist, float* grad, float* w) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const gsdf_v3 p = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
        const int vx = gsdf_float2vox1(inv_vs, p.x), vy = gsdf_float2vox1(inv_vs, p.y), vz = gsdf_float2vox1(inv_vs, p.z);
        float ow = 0.f, od = 0.f;
        gsdf_v3 og = { 0.f, 0.f, 0.f };
        if (gsdf_key_in_range(vx, vy, vz)) {
            const gsdf_slot* sl = gsdf_find(tab, gsdf_key_pack(vx, vy, vz));
            if (sl) {
                ow = sl->w;
                const gsdf_v3 gn = gsdf_normalized3(gsdf_v3{ sl->gx, sl->gy, sl->gz });
                og = gsdf_v3{ 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };
                const gsdf_v3 d = { vs * (float)vx - p.x, vs * (float)vy - p.y, vs * (float)vz - p.z };
                od = sl->s / sl->w + gsdf_dot3(og, d);
            }
        }
        w[i] = ow; dist[i] = od;
        grad[3 * i] = og.x; grad[3 * i + 1] = og.y; grad[3 * i + 2] = og.z;
    }
}
void gsdf_launch_query(hipStream_t s, gsdf_table tab, float vs, float inv_vs, const float* pts, long long n,
                       float* dist, float* grad, float* w) {
    if (n <= 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_query, dim3(blocks), dim3(256), 0, s, tab, vs, inv_vs, pts, n, dist, grad, w);
}
