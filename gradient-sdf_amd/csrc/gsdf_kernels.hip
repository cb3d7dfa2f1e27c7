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

/* The path-forcing hooks of the tests (k_fuse debug bits 4, 256, 512, 8192) and the measurement switches of the
 * ablation tools (tools/fuse_ablate.py, go_count.py, track_ablate.py) exist only in builds made with
 * -DGSDF_EXPERIMENTS: libgsdf_test.so.  The production library contains none of them. */
#ifdef GSDF_EXPERIMENTS
#define GSDF_EXPERIMENT(flags, mask) (((flags) & (mask)) != 0)
/* k_fuse trace (debug bit 64): thread 0 of every workgroup stores time stamp `col` of its row (plain stores, no atomics) */
#define GSDF_TRACE(a, tr, col) do { if (GSDF_EXPERIMENT((a).debug, 64) && threadIdx.x == 0) (tr)[col] = wall_clock64(); } while (0)
#else
#define GSDF_EXPERIMENT(flags, mask) (false)
#define GSDF_TRACE(a, tr, col) do { } while (0)
#endif
#include "gsdf_math.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>

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

/* wave-64 unsigned minimum / maximum the same way (lanes without a source keep the identity); result in every lane */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin(uint32_t v) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xf, false);
    return m < v ? m : v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umax(uint32_t v) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return m > v ? m : v;
}
__device__ __forceinline__ void wave_uminmax(uint32_t& lo, uint32_t& hi) {
    lo = dpp_umin<0x111, 0xf>(lo); hi = dpp_umax<0x111, 0xf>(hi);
    lo = dpp_umin<0x112, 0xf>(lo); hi = dpp_umax<0x112, 0xf>(hi);
    lo = dpp_umin<0x114, 0xf>(lo); hi = dpp_umax<0x114, 0xf>(hi);
    lo = dpp_umin<0x118, 0xf>(lo); hi = dpp_umax<0x118, 0xf>(hi);
    lo = dpp_umin<0x142, 0xa>(lo); hi = dpp_umax<0x142, 0xa>(hi);
    lo = dpp_umin<0x143, 0xc>(lo); hi = dpp_umax<0x143, 0xc>(hi);
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
}

/* wave-64 sums of N values at once: every DPP stage is applied to all N values before the next stage, so
 * the N dependency chains interleave instead of stalling on each other; lane 63 ends up with the totals.
 * Each stage of each value is ONE instruction -- `v_add_f32_dpp v, v, v <ctrl>`: lanes without a source add 0
 * (bound_ctrl), rows outside the row mask keep their value.  Written as inline assembly because the compiler turns
 * the update_dpp builtin + add into three instructions per value and stage (clear, v_mov_dpp, packed add: 435
 * instructions for the tracker's 29 sums against 174).  The adds and their order are the same, bit for bit.
 * Hazard kept by construction: a VGPR written by a VALU instruction must not be read by a DPP instruction within the
 * next two wait states -- stages follow each other at a distance of N >= 3 instructions (volatile asm keeps its
 * order), and an s_nop separates the first stage from whatever produced the values. */
#define GSDF_DPP_STAGE(CTRL)                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; ++i)                                                \
        asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL : "+v"(v[i]))
template <int N>
__device__ __forceinline__ void wave_sum_to_lane63(float (&v)[N]) {
    static_assert(N >= 3, "DPP read-after-write distance");
    /* the s_nop names every value as an operand: whatever instruction produced v[i] is scheduled BEFORE it */
    if constexpr (N == 29) {
        asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                     "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]),
                     "+v"(v[19]), "+v"(v[20]), "+v"(v[21]), "+v"(v[22]), "+v"(v[23]), "+v"(v[24]), "+v"(v[25]), "+v"(v[26]), "+v"(v[27]), "+v"(v[28]));
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("s_nop 1" : "+v"(v[i]));
    }
    GSDF_DPP_STAGE("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GSDF_DPP_STAGE("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GSDF_DPP_STAGE("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    GSDF_DPP_STAGE("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");       /* lane 15 of each row holds the row sum */
    GSDF_DPP_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf");                 /* into rows 1, 3 */
    GSDF_DPP_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf");                 /* into rows 2, 3: lane 63 holds the wave sum */
}
#undef GSDF_DPP_STAGE

__device__ __forceinline__ int reflect101(int i, int n) {      /* cv::BORDER_REFLECT_101 */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* ------------------------------------------------------------------------------------------------
 * table clear: keys = EMPTY, payloads = 0 (so an insert never has to initialise a payload)
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_table_clear(gsdf_table tab, size_t n_blocks) {
    /* voxel records: 2 lanes per 32-byte record, one 16-byte store each; then the block keys */
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n16 = n_blocks * GSDF_BLOCK_VOX * 2;
    uint4* p = reinterpret_cast<uint4*>(tab.vox);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_blocks; i += stride) tab.bkeys[i] = GSDF_KEY_EMPTY;
    const size_t n_occ = ((size_t)gsdf_occ_mask(tab) + 1) / 32 + ((size_t)gsdf_occ2_mask(tab) + 1) / 32;      /* block filter + cell filter */
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_occ; i += stride) tab.occ[i] = 0u;
}
void gsdf_launch_table_clear(hipStream_t s, gsdf_table tab, size_t n_slots) {
    hipLaunchKernelGGL(k_table_clear, dim3(2048), dim3(256), 0, s, tab, n_slots / GSDF_BLOCK_VOX);
}
/* block filter + cell filter from the key array (the filters were zeroed by the caller): one lane per entry */
__global__ __launch_bounds__(256) void k_occ_rebuild(gsdf_table tab, size_t n_blocks) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_blocks) return;
    const unsigned long long bk = tab.bkeys[i];
    if (bk != GSDF_KEY_EMPTY) gsdf_occ_set(tab, bk);
}
void gsdf_launch_occ_rebuild(hipStream_t s, gsdf_table tab) {
    const size_t n_blocks = (size_t)tab.block_mask + 1;
    const size_t bytes = (((size_t)gsdf_occ_mask(tab) + 1) / 32 + ((size_t)gsdf_occ2_mask(tab) + 1) / 32) * sizeof(uint32_t);
    (void)hipMemsetAsync(tab.occ, 0, bytes, s);
    hipLaunchKernelGGL(k_occ_rebuild, dim3((unsigned int)((n_blocks + 255) / 256)), dim3(256), 0, s, tab, n_blocks);
}

/* ------------------------------------------------------------------------------------------------
 * NormalEstimator::cache (NormalEstimator.h:81-154) -- all in double, once per context.
 *
 * The six moment planes go through cv::boxFilter(.., normalize = false) (:109-114), whose generic path keeps RUNNING
 * sums in double (OpenCV 4 box_filter: RowSum  s += S[i+k] - S[i];  ColumnSum  s0 = SUM + Sp, SUM = s0 - Sm, down the
 * whole image).  Q = M^-1 (:116-125) amplifies the last bits of M by the condition of an 11 x 11 window's moment matrix
 * (1e8 and more): summed freshly per pixel instead, 10 % of the Q floats come out different and normals move by up to
 * 1e-2 (measured).  So the cache is built in OpenCV's order -- two sequential scans, k_ncache_rows (one
 * lane per image row and moment) and k_ncache_cols (one lane per image column, all six moments, then Q) -- identical,
 * operation for operation, to the oracle's box_sum(mode 1).  It runs once; its time (~1 ms) does not matter.
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ double ncache_moment(int m, int u, int v, double fx_inv, double fy_inv, double cx, double cy) {
    const double x = fx_inv * ((double)u - cx);                /* :94,98 */
    const double y = fy_inv * ((double)v - cy);                /* :96,100 */
    const double x_sq = x * x, y_sq = y * y;
    const double n_sq = x_sq + (y_sq + 1.);                    /* :104 as cv::MatExpr evaluates `1. + x0_sq + y0_sq`: ONE addWeighted(x0_sq, 1,
                                                                  y0_sq, 1, gamma = 1), OpenCV 4's loop v_fma(a, 1, v_fma(b, 1, gamma)) */
    const double ni = 1. / n_sq;                               /* :105 */
    switch (m) {                                               /* :106-114 */
    case 0: return x_sq * ni;
    case 1: return (x * y) * ni;
    case 2: return x * ni;
    case 3: return y_sq * ni;
    case 4: return y * ni;
    default: return ni;
    }
}
__global__ __launch_bounds__(64) void k_ncache_rows(int W, int H, double fx_inv, double fy_inv, double cx, double cy, int r,
                                                    double* __restrict__ rows /* [6][H][W] */) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= 6 * H) return;
    const int m = t / H, v = t - m * H, win = 2 * r + 1;
    double* out = rows + ((size_t)m * H + v) * W;
    /* RowSum over the BORDER_REFLECT_101-extended row: ext[i] = S[reflect101(i - r)] */
    double s = 0.0;
    for (int i = 0; i < win; ++i) s += ncache_moment(m, reflect101(i - r, W), v, fx_inv, fy_inv, cx, cy);
    out[0] = s;
    for (int i = 0; i < W - 1; ++i) {
        s += ncache_moment(m, reflect101(i + win - r, W), v, fx_inv, fy_inv, cx, cy) -
             ncache_moment(m, reflect101(i - r, W), v, fx_inv, fy_inv, cx, cy);
        out[i + 1] = s;
    }
}
__global__ __launch_bounds__(64) void k_ncache_cols(int W, int H, double fx_inv, double fy_inv, double cx, double cy, int r,
                                                    const double* __restrict__ rows, float* __restrict__ out) {
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= W) return;
    const size_t N = (size_t)W * H;
    const int win = 2 * r + 1;
    double SUM[6] = { 0, 0, 0, 0, 0, 0 };
    for (int j = 0; j < win - 1; ++j) {                        /* ColumnSum, first call: the first k - 1 rows of the window */
        const size_t row = (size_t)reflect101(j - r, H) * W + u;
#pragma unroll
        for (int m = 0; m < 6; ++m) SUM[m] += rows[m * N + row];
    }
    for (int v = 0; v < H; ++v) {
        const size_t rp = (size_t)reflect101(v + r, H) * W + u, rm = (size_t)reflect101(v - r, H) * W + u;
        double M[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const double s0 = SUM[m] + rows[m * N + rp];
            M[m] = s0;
            SUM[m] = s0 - rows[m * N + rm];
        }
        const double M11 = M[0], M12 = M[1], M13 = M[2], M22 = M[3], M23 = M[4], M33 = M[5];
        const double det = M11 * (M22 * M33) + 2 * (M12 * (M23 * M13)) -
                           (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11));      /* :116-117 */
        const double det_inv = 1. / det;                                                      /* :118 */
        const double x = fx_inv * ((double)u - cx);
        const double y = fy_inv * ((double)v - cy);
        const double n_sq = x * x + (y * y + 1.);                                             /* :104, cv::MatExpr's order (see ncache_moment) */
        const double ni = 1. / n_sq;
        const size_t i = (size_t)v * W + u;
        out[0 * N + i] = (float)x;                                                            /* :128-132 */
        out[1 * N + i] = (float)y;
        out[2 * N + i] = (float)(x * ni);
        out[3 * N + i] = (float)(y * ni);
        out[4 * N + i] = (float)ni;
        out[5 * N + i] = (float)(det_inv * (M22 * M33 - M23 * M23));                          /* :120-125 */
        out[6 * N + i] = (float)(det_inv * (M13 * M23 - M12 * M33));
        out[7 * N + i] = (float)(det_inv * (M12 * M23 - M13 * M22));
        out[8 * N + i] = (float)(det_inv * (M11 * M33 - M13 * M13));
        out[9 * N + i] = (float)(det_inv * (M12 * M13 - M11 * M23));
        out[10 * N + i] = (float)(det_inv * (M11 * M22 - M12 * M12));
    }
}
size_t gsdf_normals_cache_scratch_bytes(int W, int H) { return (size_t)6 * W * H * sizeof(double); }
void gsdf_launch_normals_cache(hipStream_t s, int W, int H, const float* K, int win, float* planes11, double* scratch) {
    const double fx_inv = 1. / (double)K[0], fy_inv = 1. / (double)K[4];
    hipLaunchKernelGGL(k_ncache_rows, dim3((6 * H + 63) / 64), dim3(64), 0, s, W, H, fx_inv, fy_inv, (double)K[2], (double)K[5],
                       win / 2, scratch);
    hipLaunchKernelGGL(k_ncache_cols, dim3((W + 63) / 64), dim3(64), 0, s, W, H, fx_inv, fy_inv, (double)K[2], (double)K[5],
                       win / 2, (const double*)scratch, planes11);
}

/* ------------------------------------------------------------------------------------------------
 * NormalEstimator::compute.  32x16 output tile per 512-thread workgroup; the depth tile + halo is staged
 * through LDS as the three products {x0,y0,1}/n^2 * 1/z, row sums in double in LDS, then the
 * column sums, Q*b and the normalisation in registers.  One tile function, two callers: k_normals
 * (GT-pose fusion, gsdf_normals_compute) and the extra workgroups of the first tracker pass of a frame
 * (k_track_pass: the normals depend on the depth only, so they are computed beside the latency-bound
 * pose iteration instead of after it).
 * ---------------------------------------------------------------------------------------------- */
#define NRM_TX 32
#define NRM_TY 16
#define NRM_THREADS 512
#define NRM_RMAX 7
struct nrm_lds {
    float prod[3][NRM_TY + 2 * NRM_RMAX][NRM_TX + 2 * NRM_RMAX + 1];
    double rows[3][NRM_TY + 2 * NRM_RMAX][NRM_TX];
    unsigned int stat[2][4];             /* per 16-pixel half of the tile: smallest / largest valid depth (float bits), valid pixels */
};
/* The gates of MapGradPixelSdf::update on one pixel (MapGradPixelSdf.cpp:87, :95, :98): does the fusion walk its ray?  ONE
 * function for the fusion kernel and for the normals stage, which tells the fusion kernel ahead of time how many pixels of each
 * of its tiles are valid and what their depth range is (gsdf_tile_stats below) -- the two must agree bit for bit. */
__device__ __forceinline__ bool gsdf_fuse_pixel_valid(float z, float x0, float y0, float ninv, float nx, float ny, float nz, float zmin, float zmax) {
    bool valid = !(z <= zmin || z >= zmax);                                     /* :87 */
    const gsdf_v3 n = { nx, ny, nz }, xy = { x0, y0, 1.f };
    if ((double)gsdf_dot3(n, n) < .1) valid = false;                            /* :95 (same comparison as the reference: NaN passes) */
    const float nd = gsdf_dot3(n, xy);
    if (nd * nd * ninv < .25) valid = false;                                    /* :98 */
    return valid;
}
/* Statistics of a frame's 16 x 16-pixel fusion tiles, written by whatever computes the frame's normals (k_normals, the normals
 * workgroups of the tracker launches, the tail of the previous fusion launch): [tile_y][tile_x][4] = smallest valid depth,
 * largest valid depth (float bits; 0x7F800000 / 0 without a valid pixel), number of valid pixels, unused.  The fusion kernel
 * makes its tile-wide decisions from them while its pixel loads are still in flight. */
struct gsdf_tile_stats {
    uint32_t* rows;                      /* nullable */
    int ntx;                             /* fusion tiles per image row */
    float zmin, zmax;                    /* Sdf::z_min_ / z_max_ */
};
template <int NT = NRM_THREADS>          /* threads of the calling workgroup (>= NRM_TX * NRM_TY) */
__device__ __forceinline__ void normals_tile(nrm_lds& S, int tile_x, int tile_y, int W, int H, int r, const gsdf_ncache& nc,
                                             const float* __restrict__ depth, float* __restrict__ nx, float* __restrict__ ny,
                                             float* __restrict__ nz, const gsdf_tile_stats ts = gsdf_tile_stats{ nullptr, 0, 0.f, 0.f }) {
    const int tx0 = tile_x * NRM_TX, ty0 = tile_y * NRM_TY;
    const int PW = NRM_TX + 2 * r, PH = NRM_TY + 2 * r;
    const int tid = threadIdx.x;
    static_assert(NT >= NRM_TX * NRM_TY, "one lane per pixel of the tile in the last stage");
    static_assert(NRM_TX == 32 && NRM_TY == 16, "two 16 x 16 fusion tiles per normals tile");
    if (tid < 2) { S.stat[tid][0] = 0x7F800000u; S.stat[tid][1] = 0u; S.stat[tid][2] = 0u; }
    for (int idx = tid; idx < PW * PH; idx += NT) {
        const int ly = idx / PW, lx = idx - ly * PW;
        const int gy = reflect101(ty0 + ly - r, H), gx = reflect101(tx0 + lx - r, W);
        const size_t i = (size_t)gy * W + gx;
        const float z = depth[i];
        const float zi = z != 0.f ? 1.f / z : 0.f;          /* NormalEstimator.h:183-187 */
        S.prod[0][ly][lx] = nc.x0n[i] * zi;                 /* :191-193 */
        S.prod[1][ly][lx] = nc.y0n[i] * zi;
        S.prod[2][ly][lx] = nc.ninv[i] * zi;
    }
    __syncthreads();
    for (int idx = tid; idx < PH * NRM_TX; idx += NT) {
        const int ly = idx / NRM_TX, x = idx - ly * NRM_TX;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int dx = 0; dx <= 2 * r; ++dx) {
            s0 += (double)S.prod[0][ly][x + dx];
            s1 += (double)S.prod[1][ly][x + dx];
            s2 += (double)S.prod[2][ly][x + dx];
        }
        S.rows[0][ly][x] = s0; S.rows[1][ly][x] = s1; S.rows[2][ly][x] = s2;
    }
    __syncthreads();
    const int x = tid & (NRM_TX - 1), y = tid / NRM_TX;
    const int px = tx0 + x, py = ty0 + y;
    const bool inside = y < NRM_TY && px < W && py < H;
    if (!inside && !ts.rows) return;
    bool valid = false;
    float z = 0.f;
    if (inside) {
        double b1 = 0, b2 = 0, b3 = 0;
        for (int dy = 0; dy <= 2 * r; ++dy) {
            b1 += S.rows[0][y + dy][x];
            b2 += S.rows[1][y + dy][x];
            b3 += S.rows[2][y + dy][x];
        }
        const size_t i = (size_t)py * W + px;
        const float c1 = (float)b1, c2 = (float)b2, c3 = (float)b3;
        const float q11 = nc.q11[i], q12 = nc.q12[i], q13 = nc.q13[i], q22 = nc.q22[i], q23 = nc.q23[i], q33 = nc.q33[i];
        const float vx = (c1 * q11 + c2 * q12) + c3 * q13;      /* :195-197 */
        const float vy = (c1 * q12 + c2 * q22) + c3 * q23;
        const float vz = (c1 * q13 + c2 * q23) + c3 * q33;
        const float n = sqrtf((vx * vx + vy * vy) + vz * vz);   /* :199 */
        const float ox = vx / n, oy = vy / n, oz = vz / n;      /* :201-203 */
        nx[i] = ox; ny[i] = oy; nz[i] = oz;
        if (ts.rows) {
            z = depth[i];
            valid = gsdf_fuse_pixel_valid(z, nc.x0[i], nc.y0[i], nc.ninv[i], ox, oy, oz, ts.zmin, ts.zmax);
        }
    }
    if (!ts.rows) return;                                   /* (uniform: a kernel argument) */
    /* the two fusion tiles of this normals tile: depth range and number of their valid pixels (every wave reduces its 64 lanes
     * with DPP, lane 0 adds them to the workgroup's six words) */
    const int lane = tid & 63;
    const int half = (lane >> 4) & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool mine = valid && half == h;
        unsigned int lo = mine ? __float_as_uint(z) : 0x7F800000u, hi = mine ? __float_as_uint(z) : 0u;
        wave_uminmax(lo, hi);
        const unsigned int cnt = (unsigned int)__popcll(__ballot(mine));
        if (lane == 0 && cnt) { atomicMin(&S.stat[h][0], lo); atomicMax(&S.stat[h][1], hi); atomicAdd(&S.stat[h][2], cnt); }
    }
    __syncthreads();
    if (tid < 2 && 2 * tile_x + tid < ts.ntx) {
        uint32_t* o = ts.rows + 4 * ((size_t)tile_y * ts.ntx + 2 * tile_x + tid);
        o[0] = S.stat[tid][0]; o[1] = S.stat[tid][1]; o[2] = S.stat[tid][2]; o[3] = 0u;
    }
}

__global__ __launch_bounds__(NRM_THREADS) void k_normals(gsdf_frame_geom g, int r, gsdf_ncache nc,
                                                         const float* __restrict__ depth, float* __restrict__ nx,
                                                         float* __restrict__ ny, float* __restrict__ nz,
                                                         unsigned int* deferred_count, gsdf_dev_state* st_rw, gsdf_tile_stats ts) {
    if (deferred_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        *deferred_count = 0u;                                  /* fresh list for k_fuse */
        if (st_rw) st_rw->frame_cur = st_rw->frames;           /* counter_ seen by every workgroup of k_fuse */
    }
    __shared__ nrm_lds S;
    normals_tile(S, blockIdx.x, blockIdx.y, g.W, g.H, r, nc, depth, nx, ny, nz, ts);
}
void gsdf_launch_normals(hipStream_t s, const gsdf_frame_geom& g, int win, const gsdf_ncache& nc,
                         const float* depth, float* nx, float* ny, float* nz,
                         unsigned int* deferred_count, gsdf_dev_state* st_rw, uint32_t* tile_stats) {
    dim3 grid((g.W + NRM_TX - 1) / NRM_TX, (g.H + NRM_TY - 1) / NRM_TY);
    const gsdf_tile_stats ts = { tile_stats, (g.W + 15) / 16, g.zmin, g.zmax };
    hipLaunchKernelGGL(k_normals, grid, dim3(NRM_THREADS), 0, s, g, win / 2, nc, depth, nx, ny, nz, deferred_count, st_rw, ts);
}

/* ------------------------------------------------------------------------------------------------
 * MapGradPixelSdf::update -- fusion.
 *
 * Work decomposition: one workgroup = one 16x16 pixel tile, 512 lanes = 8 waves.  A wave takes 32 pixels of the tile --
 * every 2nd in x, every 4th in y (wave w has phase (w & 1, w >> 1)) -- and BOTH halves of their ray walk (lanes 0-31 the
 * first half of k = -factor..factor, lanes 32-63 the second), so neighbouring lanes are >= 2 pixels or half a ray apart
 * and rarely hit one voxel in the same instruction.  Neighbouring pixels and consecutive samples do hit the same voxels
 * (~4.5 updates per distinct voxel per frame), so updates are first combined in a workgroup-private hash table in LDS;
 * only the distinct voxels of the tile are flushed to the HBM map.  Far tiles / small voxels (more distinct voxels than
 * the table holds) are walked as 2 or 4 row bands, each flushed on its own.
 *
 * What the MI355X measurements (tools/atomics_bench.hip, tools/fuse_ablate.py, tools/fuse_trace.py, profiles/) dictated:
 *  - ds_add_f32 is lane-serial (~190 cycles per wave instruction), ds_add_u64 costs 7-29: the LDS accumulators are
 *    FIXED POINT, three 64-bit words per entry that several sums share (layout below), each word in its own array so that
 *    a wave's scattered adds use all banks.  Sums are exact and order-independent; one rounding to float at the flush.
 *  - the LDS lookup dominated: 32-bit TILE-LOCAL keys (one ds_read_b128 per bucket of 4) and a LATTICE hash of the local
 *    coordinates bring it to ~1 probe per sample (a random hash of 64-bit keys needed 2.8).  ONE sample per lane is in
 *    flight (2, 3, 4 and 6 were measured slower: registers); the probe loop ends with a wave-uniform branch as soon as no
 *    lane is pending.
 *  - the walk is bound by VALU issue and LDS cycles when two workgroups share a CU, so it is written for instruction
 *    count: packed 2 x f32 arithmetic for x / y, a 3-operation round (exhaustively checked against roundf), one v_med3 for
 *    the clamp, shift-adds and 24-bit multiply-adds for keys / hash / addresses, and the weight goes to fixed point with a
 *    plain float multiply + convert (w 2^24 is an integer), the distance through the double-add trick of f2fix.
 *  - device-scope atomics run at only ~20-50 G/s chip-wide: the flush uses NONE.  Tiles hand their voxels on in colour
 *    order with plain read-modify-write (see the flush); only tiles that are too near for that, timed-out waits and LDS
 *    overflow go through a deferred list, which the LAST workgroup of the launch adds with float atomics (k_fuse_resolve
 *    is queued only while recent launches reported long lists).
 *  - two workgroups (16 waves) per CU; a third one does not pay (the CU's issue and LDS throughput are the limit, not
 *    occupancy).  The kernel exists with two table sizes (FUSE_LCAP below), chosen per launch by the host.
 * ---------------------------------------------------------------------------------------------- */
#define FUSE_T 16                        /* tile width */
#ifndef FUSE_TH
#define FUSE_TH 16                       /* tile height (a multiple of 4).  20: a 640x480 frame is 960 tiles = 1.88 dispatch rounds of the
                                            512 workgroup slots instead of 2.34 (DESIGN.md "Dispatch rounds") */
#endif
#define FUSE_ZSPLIT 2                    /* slices of the ray walk a lane pair shares: the two halves of a wave */
#define FUSE_NWAVES (FUSE_TH / 2)        /* a wave takes 32 pixels: 16 x TH / 32 */
#define FUSE_THREADS (64 * FUSE_NWAVES)
#define FUSE_RP (FUSE_TH / 4)            /* row phases of the spread mapping: wave w walks rows RP * (ly & 3) + (w >> 1) */
#define FUSE_NPASS_MAX (FUSE_NWAVES / 2) /* most bands a tile is split into (4 rows each) */
static_assert(FUSE_TH % 4 == 0 && FUSE_TH >= 8 && FUSE_TH <= 32, "tile height");
/* LDS table entries.  The kernel exists in two sizes (template parameter LCAP, FUSE_LCAP below is that parameter):
 *   2048  512 buckets of 4, 57 KB with the accumulators: near scenes;
 *   2560  640 buckets, 72 KB (still two workgroups per CU): keeps far tiles (2.5-3 m at 640x480 / 1 cm), which the small
 *         table has to walk in two bands, in one; near tiles use its first 2048 entries.  Fusions of frames with far
 *         geometry 68 -> 63 us, of near scenes 62 -> 63 us (the bigger table clear competes with the CU's other walk).
 * The host picks per launch from what the previous fusions reported (tiles that did not fit the small table). */
#define FUSE_LCAP LCAP
#ifndef FUSE_LCAP_NEAR
#define FUSE_LCAP_NEAR 2048
#endif
#ifndef FUSE_LCAP_FAR
#define FUSE_LCAP_FAR 2560
#endif
#ifndef FUSE_POLL_SLEEP
#define FUSE_POLL_SLEEP 8                  /* x 64 cycles between two looks at the neighbours' flags */
#endif
#ifndef FUSE_BSLOTS
#define FUSE_BSLOTS 4                    /* keys per bucket of the LDS table = one ds_read_b128 per probe (2 per bucket: half the read and
                                            compare work, 1.03 probes per sample on typical tiles -- but dense tiles then cluster: long probe
                                            chains, 20 k samples per frame on the deferred route, 68 us instead of 61) */
#endif
#define FUSE_NB (FUSE_LCAP / FUSE_BSLOTS)
/* A tile whose voxels fit uses only the first FUSE_LCAP_SMALL entries (512 buckets: the bucket index is a mask, and the flush
 * has a slot less per lane to look at); the full table is for far tiles, which would otherwise be walked in two bands */
#define FUSE_LCAP_SMALL FUSE_LCAP_NEAR
#define FUSE_DUAL (FUSE_LCAP == FUSE_LCAP_FAR && FUSE_LCAP_FAR != FUSE_LCAP_NEAR && FUSE_BSLOTS == 4)
#define FUSE_LKEY_EMPTY 0xFFFFFFFFu      /* LDS keys are 32-bit: voxel coordinates relative to the tile origin, 10 bits each */
#define FUSE_LKEY_DEFER 0x80000000u      /* flush: the entry goes to the deferred list, low 31 bits = voxel record index */
#define FUSE_LPROBE (48 / FUSE_BSLOTS)   /* buckets probed before a sample takes the deferred route */
#ifndef FUSE_BOUNDS
#define FUSE_BOUNDS __launch_bounds__(FUSE_THREADS, FUSE_OCC)
#endif
#ifndef FUSE_OCC
#define FUSE_OCC (FUSE_THREADS / 128)     /* waves per SIMD the register allocation must allow: 2 workgroups per CU */
#endif
/* LDS accumulators are fixed point (exact, order-independent integer adds) in THREE 64-bit words per entry, each in its
 * own array of 8-byte entries (a wave's scattered adds then use all banks; the walk is bound by LDS cycles).  Several
 * sums share a word and are added as ONE integer; the flush separates them again (a negative low field has borrowed 1
 * from the field above it).  A ray's samples are >= one voxel apart, so at most 2 of them round to one voxel: a voxel
 * collects <= 2 x 256 terms of a tile.
 *  - acc[0]  bits 0..50: sum of w * truncated sdf, signed, 2^-40.  dist = s / w must hold to 1e-4 also for a voxel whose
 *            only sample has a weight of 2^-24: the sum keeps every bit of its float terms (|x| >= 2^-17 exactly, below
 *            that to 2^-40); 512 terms of |x| <= T < 2 m stay below 2^50 (gsdf_create checks T).
 *            bits 51..61: sum of the LOW 2 bits of the z terms of the normal (512 x 3 < 2^11).
 *  - acc[1]  bits 0..33: sum of w in units of 2^-24.  w = 1 - sdf / T is a multiple of 2^-24 (the subtraction is exact
 *            for sdf / T >= 1/2 and rounded to 2^-24 below), so the weight sum is EXACT (512 x 2^24 = 2^33) and a voxel
 *            exists (w > 0) exactly when the reference creates it.
 *            bits 34..63: sum of the z terms of the normal without their low 2 bits (floor), signed.
 *  - acc[2]  bits 0..31 / 32..63: sums of the x / y terms of the normal, signed.
 *  Normal terms w (R n) are truncated to 2^-21 (4.8e-7; |sum| <= 512 < 2^10): the tracker NORMALISES the stored
 *  gradient, so voxels of small weight need the resolution (2^-19 for z was tried: pose parity lost after 3 passes). */
#define FUSE_FIX_G 2097152.0f            /* 2^21 */
#define FUSE_FIX_G_INV 4.76837158203125e-07f
#define FUSE_FIX_W 16777216.0f           /* 2^24 */
#define FUSE_FIX_W_INV 5.9604644775390625e-08f
#ifndef FUSE_SPREAD
#define FUSE_SPREAD 1                    /* spread lane -> (pixel, slice) mapping, see k_fuse */
#endif
#define FUSE_NSTAT (FUSE_SPREAD ? FUSE_NWAVES : 4) /* waves of a workgroup that hold distinct pixels */
/* tile statistics from the normals stage (normals tiles are 32 x 16 pixels = two fusion tiles): the standard tile shape only;
 * other shapes (build experiments) reduce them from their own pixels */
#ifndef FUSE_STATS_AHEAD
#define FUSE_STATS_AHEAD (FUSE_T == 16 && FUSE_TH == 16)
#endif
/* Hand-placed wave priorities inside k_fuse (s_setprio 0..3 at the kernel's start, in front of the walk, in front of the flush):
 * build parameters for the experiment only, default none.  The library is compiled with -amdgpu-set-wave-priority, which keeps a
 * wave at priority 3 until its last batch of loads is issued (in k_fuse: up to the flush's record loads) and at 0 behind that:
 * +2 % frames/s; every hand-placed ranking of the phases on top of it lost that gain again ("Kernels"). */
#ifndef FUSE_PRIO_START
#define FUSE_PRIO_START -1
#endif
#ifndef FUSE_PRIO_WALK
#define FUSE_PRIO_WALK -1
#endif
#ifndef FUSE_PRIO_FLUSH
#define FUSE_PRIO_FLUSH -1
#endif
#define FUSE_SETPRIO(n) do { if ((n) >= 0) __builtin_amdgcn_s_setprio((n) < 0 ? 0 : (n)); } while (0)
/* The flush's record accesses as LANE PAIRS: a 16-byte sc1 store is a 32-byte fabric write whatever it is next to, but two
 * adjacent lanes of ONE instruction that write the two halves of a record make one write of it (1.0x, not 2 x 0.75:
 * profiles/r03_write_calibration.txt); loads likewise.  Lanes 2j and 2j+1 therefore serve each other: instruction 1 moves the
 * even lane's record (even lane: bytes 0-15, odd lane: bytes 16-31), instruction 2 the odd lane's; the halves change lanes
 * through DPP quad permutes.  0 = every lane moves its own record with two instructions (rounds 2-4). */
#ifndef FUSE_PAIRED_IO
#define FUSE_PAIRED_IO 1
#endif
template <int QP>
__device__ __forceinline__ uint32_t fuse_quad(uint32_t v) {      /* v of the lane quad_perm QP names, all lanes active */
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, QP, 0xf, 0xf, false);
}
#define FUSE_QP_EVEN 0xA0                /* quad_perm [0,0,2,2]: the even lane of my pair */
#define FUSE_QP_ODD 0xF5                 /* quad_perm [1,1,3,3]: the odd lane of my pair */
#define FUSE_QP_SWAP 0xB1                /* quad_perm [1,0,3,2]: my partner */
typedef uint32_t gsdf_u32x4 __attribute__((ext_vector_type(4)));
typedef float gsdf_f2 __attribute__((ext_vector_type(2)));           /* packed f32 arithmetic (v_pk_*_f32): two results per issue slot */
/* a * b + c with a, b < 2^24 (b uniform): full rate, where the 32-bit v_mul_lo_u32 is quarter rate */
__device__ __forceinline__ uint32_t gsdf_mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

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
    gsdf_deferred* deferred;            /* list of contributions to voxels owned by another tile */
    unsigned int* deferred_count;
    unsigned int deferred_cap;
    unsigned int tag;                   /* serial of this fusion launch, never 0: value of a published tile flag */
    unsigned int* tile_flags;           /* [nty][ntx]: tag of the last launch in which the tile finished its flush */
    int ntx, nty;                       /* tiles per image row / column */
    const uint32_t* tile_order;         /* workgroup -> tile (x | y << 16): colour-major, and workgroup b (which runs on
                                           XCD b % 8) gets a tile of image stripe b % 8, so neighbouring tiles share an L2 */
    uint32_t* vis;                      /* optional per-voxel frame bit-vectors (vis_, MapGradPixelSdf.h:70); nullable */
    int vis_words;
    int debug;                          /* path-forcing / measurement switches: only read by builds with -DGSDF_EXPERIMENTS */
    unsigned int* ticket;               /* arrivals of finished workgroups; the last one resolves the deferred list and resets it */
    float* log_rows;                    /* nullable: per-frame log of the Scan3D loop (use_dev_pose) */
    long long max_rows;
    int resolve_follows;                /* a k_fuse_resolve launch is queued behind this one: leave long lists to it */
    unsigned int* host_note;            /* nullable, pinned host word: length of this launch's deferred list */
    /* Workgroups beyond the tiles (GT-pose fusion of resident frames, gsdf_update_dev): NormalEstimator::compute of the NEXT
     * frame, one 32x16 tile each.  They are the last of the grid, so they run in the launch's tail -- 1200 fusion workgroups
     * on 512 slots leave the last round a third full -- instead of as a 10 us launch in front of the next fusion. */
    int n_tiles;                        /* fusion workgroups of this launch */
    const float* nrm_depth;             /* nullable: depth image of the next frame */
    float *nrm_x, *nrm_y, *nrm_z;       /* its normal planes (the other set) */
    uint32_t* nrm_stats;                /* ... and its tile statistics (gsdf_tile_stats) */
    const uint32_t* tile_stats;         /* THIS frame's tile statistics, written with its normals */
    int nrm_r, nrm_ntx;                 /* window radius, normals tiles per image row */
    gsdf_fuse_head hd;                  /* k_fuse<.., HEAD>: the closing head of optimize() this launch performs first (k = 0: none) */
    unsigned int nrm_token;             /* tracked frames: what the normals role leaves in st->nrm_token when it ran (never 0) */
};
#define FUSE_RESOLVE_INLINE 8192u       /* deferred entries the last workgroup adds itself even when a resolve launch follows */

/* workgroups of fewer than 512 threads (8-row tiles, a build experiment) cannot take the normals role: the launcher then
 * computes the next frame's normals with k_normals in front of the fusion */
#define FUSE_CARRIES_NORMALS (FUSE_THREADS >= NRM_TX * NRM_TY)
template <int LCAP>
struct fuse_lds {
    uint32_t key[FUSE_LCAP] __attribute__((aligned(16)));
    unsigned long long acc[3][FUSE_LCAP] __attribute__((aligned(16)));  /* see above: s | w + gz | gx + gy */
    unsigned int cnt[2][FUSE_NWAVES];       /* per wave: samples with w > 0, valid pixels */
    unsigned int n_defer, defer_base;
    unsigned int st_min[FUSE_NSTAT], st_max[FUSE_NSTAT];  /* per wave: smallest / largest valid depth (float bits) */
    float st_cnt[FUSE_NSTAT];                             /* per wave: valid pixels */
    int dec[8];                         /* the tile-wide decisions, made by wave 0: n_pass, big, ox, oy, oz, range_ok */
    unsigned int ordered;               /* flush with plain read-modify-write (1) or through the deferred list (0) */
    unsigned int any_defer, is_last;    /* this workgroup appended to the deferred list / is the last one to finish */
    const float* plane[7];              /* depth, x0, y0, 1/n2, nx, ny, nz: read from here by the band reloads, so the
                                           seven pointers do not stay in scalar registers across the ray walk */
};

/* float -> signed 2^-40 fixed point for |x| < 2048: x + 1.5*2^12 in double has its last mantissa bit at
 * 2^-40, so the bit pattern minus that of 1.5*2^12 IS the fixed-point value (exact for |x| >= 2^-17,
 * nearest 2^-40 below that).  Two double-rate VALU ops + a 64-bit subtract. */
__device__ __forceinline__ unsigned long long f2fix(float x) {
    const double d = (double)x + 6144.0;
    return (unsigned long long)(__double_as_longlong(d) - 0x40B8000000000000ll);
}
__device__ __forceinline__ float fix2f(unsigned long long v) {
    return __ll2float_rn((long long)v) * 9.094947017729282e-13f;   /* 2^-40 */
}
/* one sample's contribution as the three words of the accumulator layout above */
__device__ __forceinline__ void fuse_pack(uint32_t wi, unsigned long long s_fix, int gx, int gy, int gz,
                                          unsigned long long& a0, unsigned long long& a1, unsigned long long& a2) {
    a0 = s_fix + ((unsigned long long)((uint32_t)gz & 3u) << 51);
    a1 = (unsigned long long)wi | ((unsigned long long)((uint32_t)gz & ~3u) << 32);       /* floor(gz / 4) 2^34: no carry between the words */
    a2 = (unsigned long long)(uint32_t)gx | ((unsigned long long)(uint32_t)(gy + (gx >> 31)) << 32);
}
/* the sums of one LDS entry as floats (one rounding each, of the exact fixed-point sum) */
struct fuse_sums { float w, s, gx, gy, gz; };
__device__ __forceinline__ fuse_sums fuse_unpack(unsigned long long a0, unsigned long long a1, unsigned long long a2) {
    fuse_sums r;
    const long long sv = (long long)(a0 << 13) >> 13;                 /* sign-extended bits 0..50 */
    const int gz_lo = (int)((a0 - (unsigned long long)sv) >> 51);
    r.s = __ll2float_rn(sv) * 9.094947017729282e-13f;                 /* 2^-40 */
    r.w = __ull2float_rn(a1 & 0x3FFFFFFFFull) * FUSE_FIX_W_INV;
    r.gz = (float)((int)((long long)a1 >> 34) * 4 + gz_lo) * FUSE_FIX_G_INV;
    const int gx = (int)(uint32_t)a2;
    const int gy = (int)(uint32_t)(a2 >> 32) - (gx >> 31);           /* a negative low field borrowed 1 */
    r.gx = (float)gx * FUSE_FIX_G_INV;
    r.gy = (float)gy * FUSE_FIX_G_INV;
    return r;
}

/* additive update with float atomics: merge / resolve kernels */
__device__ __forceinline__ void hbm_accumulate(const gsdf_table& T, unsigned long long key, float w, float s,
                                               float gx, float gy, float gz, gsdf_dev_state* st) {
    gsdf_payload* p = gsdf_find_or_insert(T, key);
    if (!p) { atomicOr(&st->status, GSDF_STATUS_TABLE_FULL); return; }
    unsafeAtomicAdd(&p->w, w);
    unsafeAtomicAdd(&p->s, s);
    unsafeAtomicAdd(&p->gx, gx);
    unsafeAtomicAdd(&p->gy, gy);
    unsafeAtomicAdd(&p->gz, gz);
}

__device__ __forceinline__ void defer_append(const fuse_args& a, gsdf_payload* p, float w, float s, float gx, float gy,
                                             float gz) {
    const unsigned int i = atomicAdd(a.deferred_count, 1u);
    if (i >= a.deferred_cap) { atomicOr(&a.st->status, GSDF_STATUS_TABLE_FULL); return; }
    gsdf_deferred d;
    d.p = p; d.w = w; d.s = s; d.gx = gx; d.gy = gy; d.gz = gz; d.pad = 0u;
    a.deferred[i] = d;
}

/* vis_[vi]: resize(counter_), push_back(true) -- MapGradPixelSdf.cpp:113-115: set bit `frame` of the voxel */
__device__ __forceinline__ void vis_mark(const fuse_args& a, const gsdf_payload* p, long long frame) {
    if (!a.vis || frame >= 32ll * a.vis_words) return;
    const size_t slot = (size_t)(p - a.tab.vox);
    atomicOr(&a.vis[slot * a.vis_words + (frame >> 5)], 1u << (frame & 31));
}

template <int LCAP>
__device__ __forceinline__ void fuse_lds_clear(fuse_lds<LCAP>& L, int tid) {
    gsdf_u32x4* k4 = reinterpret_cast<gsdf_u32x4*>(L.key);
    gsdf_u32x4* a4 = reinterpret_cast<gsdf_u32x4*>(L.acc);
    for (int i = tid; i < FUSE_LCAP / 4; i += FUSE_THREADS) k4[i] = gsdf_u32x4{ FUSE_LKEY_EMPTY, FUSE_LKEY_EMPTY, FUSE_LKEY_EMPTY, FUSE_LKEY_EMPTY };
    for (int i = tid; i < 3 * FUSE_LCAP / 2; i += FUSE_THREADS) a4[i] = gsdf_u32x4{ 0u, 0u, 0u, 0u };
}

/* per-frame log row of the Scan3D loop: pose7, converged, passes, hits of the last pass (main_scan_3d.cpp:268-280) */
__device__ __forceinline__ void fuse_log_row(const fuse_args& a) {
    if (!a.log_rows) return;
    gsdf_dev_state* st = a.st;
    const long long r = st->log_rows;
    if (r < a.max_rows) {
        float* o = a.log_rows + 10 * r;
        for (int i = 0; i < 7; ++i) o[i] = st->pose7[i];
        o[7] = (float)st->converged; o[8] = (float)st->passes; o[9] = st->last_hits;
    }
    st->log_rows = r + 1;
}

/* (the tracker's head, defined with k_track_pass below) */
__device__ __forceinline__ void trk_solve_update(const float* tot, float damping, float conv_sq, int passes, int max_passes,
                                                 int no_solve, bool exact_solve, float pose[7], int* done, int* converged);

/* k_fuse<.., HEAD>: the head of tracker launch a.hd.k (see k_fuse).  On return (pose, done, conv) are the state behind that head:
 * computed by the `solver` wave (workgroup 0's first) -- or, when optimize() had ended in an earlier launch, read by it from a.st,
 * which nobody has written since -- and taken from the three tagged chunks the solver publishes by EVERY other wave of the launch.
 *
 * Round 6, found by the eight-process exchange test (a frame without its log row, 1 run in 3): the non-solver waves used to look
 * at st->trk[(k - 1) & 1].done with a plain load first ("ended before this launch: the state in a.st is final") and then trusted
 * the plain loads of st->done / converged / R / pose7 from the top of the kernel.  But the solver itself sets that `done` word
 * (sticky, for launches queued beyond the end) and rewrites st->done / pose7 in THIS launch, and the per-XCD L2s are not coherent
 * for plain accesses: a workgroup of a later dispatch round could see the NEW `done` word (its line re-fetched) beside the OLD
 * st->done = 0 (its line still cached from an earlier workgroup of the same XCD) and leave without fusing its tile -- no ticket, no
 * last workgroup, no log row.  Now nothing that the solver writes is read plainly by anybody else in the launch. */
__device__ __forceinline__ void fuse_head(const fuse_args& a, int tid, bool solver, float (&pose)[7], int& hdone, int& hconv) {
    const gsdf_trk_buf& in = a.st->trk[(a.hd.k - 1) & 1];
    const int hlane = tid & 63;
    int hpasses = 0;
    bool have = false;
    if (!solver) {
        /* the three chunks, one per lane 0..2, until all carry this launch's tag (for workgroups of the later dispatch rounds
         * they are there at the first look) */
        unsigned long long t0 = wall_clock64();
        gsdf_u32x4 ch = { 0u, 0u, 0u, 0u };
        for (;;) {
            if (hlane < 3) {
                const unsigned int* q = a.st->fh + 4 * hlane;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ch) : "v"(q) : "memory");
            }
            if (__all(hlane >= 3 || ch.x == a.tag)) { have = true; break; }
            if (wall_clock64() - t0 > 200000ull) {
                /* 2 ms at 100 MHz without the solver's chunks (a GPU shared with other processes): perform the head here -- unless
                 * the `done` word of the input state is set: then optimize() ended earlier or the solver has just ended it, and in
                 * both cases its chunks are the only consistent statement of the result: keep polling */
                if (!__hip_atomic_load(const_cast<int*>(&in.done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                t0 = wall_clock64();
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (have) {
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)ch.y, 0);
            hdone = (int)(f & 1u); hconv = (int)((f >> 1) & 1u);
            pose[0] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.z, 0));
            pose[1] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.w, 0));
            pose[2] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.y, 1));
            pose[3] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.z, 1));
            pose[4] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.w, 1));
            pose[5] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.y, 2));
            pose[6] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)ch.z, 2));
            return;
        }
    }
    /* (the solver wave is the first of the launch: whatever it reads plainly was written by earlier kernels) */
    const bool ended_before = solver && in.done != 0;
    if (ended_before) {
        gsdf_dev_state* st = a.st;
        hdone = st->done; hconv = st->converged; hpasses = st->passes;
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = st->pose7[i];
    } else {
        /* the head itself, as in k_track_pass: lane v < 29 adds the group sums of value v in increasing order, readlane hands the
         * totals to every lane, the solve runs in every lane alike.  (A wave whose wait expired: rows, in.pose7 and in.passes are
         * not written in this launch.) */
        const double* acc_prev = a.hd.rows + (size_t)a.hd.rot_prev * GSDF_TRACK_ROWSET;
        double gs = 0.0;
        if (hlane < GSDF_TRACK_NSUM) {
            double part[GSDF_TRACK_GROUPS];
#pragma unroll
            for (int grp = 0; grp < GSDF_TRACK_GROUPS; ++grp) part[grp] = acc_prev[grp * 32 + hlane];
            gs = part[0];
#pragma unroll
            for (int grp = 1; grp < GSDF_TRACK_GROUPS; ++grp) gs += part[grp];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = in.pose7[i];
        hpasses = in.passes + 1;
        const float totv = (float)gs;
        float tot[GSDF_TRACK_NSUM];
#pragma unroll
        for (int i = 0; i < GSDF_TRACK_NSUM; ++i) tot[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(totv), i));
        trk_solve_update(tot, a.hd.damping, a.hd.conv_sq, hpasses, a.hd.max_passes, GSDF_EXPERIMENT(a.hd.debug, 1),
                         GSDF_EXPERIMENT(a.hd.debug, 4), pose, &hdone, &hconv);
        if (solver && hlane == 0) {
            /* published as k_track_pass's workgroup 0 publishes its head (see there) ... */
            gsdf_dev_state* st = a.st;
            gsdf_trk_buf& o = st->trk[a.hd.k & 1];
#pragma unroll
            for (int i = 0; i < 7; ++i) { o.pose7[i] = pose[i]; st->pose7[i] = pose[i]; }
            o.done = hdone; o.converged = hconv; o.passes = hpasses;
            if (hdone) __hip_atomic_store(&st->trk[(a.hd.k - 1) & 1].done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gsdf_quat_to_R(pose + 3, st->R);
            st->converged = hconv;
            st->done = hdone;
            st->passes = hpasses;
            st->last_hits = tot[28];
            st->n_hit += (unsigned long long)tot[28];
            if (a.hd.progress)
                __hip_atomic_store(&a.hd.progress[0], (a.hd.serial << 16) | (hdone ? 0x8000u : 0u) | (unsigned int)(hpasses & 0x7FFF),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (solver && hlane == 0) {
        /* ... and for every other wave of THIS launch: the tagged chunks (also when optimize() had ended before the launch) */
        const gsdf_u32x4 c0 = { a.tag, (uint32_t)hdone | ((uint32_t)hconv << 1), __float_as_uint(pose[0]), __float_as_uint(pose[1]) };
        const gsdf_u32x4 c1 = { a.tag, __float_as_uint(pose[2]), __float_as_uint(pose[3]), __float_as_uint(pose[4]) };
        const gsdf_u32x4 c2 = { a.tag, __float_as_uint(pose[5]), __float_as_uint(pose[6]), (uint32_t)hpasses };
        unsigned int* q = a.st->fh;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                     "global_store_dwordx4 %0, %3, off offset:32 sc1" :: "v"(q), "v"(c0), "v"(c1), "v"(c2) : "memory");
        /* the plain stores above are read by the launch's LAST workgroup (frame log), possibly on another XCD */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
}

/* NEXT_NORMALS: the instantiation whose launches carry the normals workgroups of the next frame (GT-pose runs, and tracked frames
 * whose successor was named with gsdf_hint_next_depth_dev).
 * HEAD (round 6): the frame's first gated fusion launch stands in for the LAST tracker launch of the first batch -- its head, which
 * finishes pass hd.k - 1 (reduce the group sums, solve, stop test, pose update).  For the usual frame that head is the closing one
 * of optimize(): a launch (~2 us) and a kernel boundary (1.4 us) less per frame.  Workgroup 0's first wave performs it -- the same
 * arithmetic as k_track_pass's head, trk_solve_update -- and publishes the result as k_track_pass's workgroup 0 would (st->trk,
 * st->pose7 / R / done / converged, the host's progress word) plus three tagged 16-byte chunks (gsdf_dev_state::fh); every other
 * wave of the launch polls those chunks (agent scope; one look once they are there), so the solve runs once per launch and not
 * once per workgroup.  Workgroup 0 is dispatched first and waits for nobody; the wait of the others is bounded, and a wave whose
 * wait expires performs the head itself (bit-identical).  If optimize() has not ended with that head, every workgroup leaves,
 * and the host's next batch starts with a tracker launch that skips its head (gsdf_track_params::head_done). */
template <int LCAP, bool NEXT_NORMALS, bool HEAD>
__global__ FUSE_BOUNDS void k_fuse(fuse_args a) {
    __shared__ fuse_lds<LCAP> L;
    const int tid = threadIdx.x;
    FUSE_SETPRIO(FUSE_PRIO_START);
    if constexpr (NEXT_NORMALS && FUSE_CARRIES_NORMALS) {
    if ((int)blockIdx.x >= a.n_tiles) {                       /* the next frame's normals, in the tail of this launch */
        static_assert(sizeof(nrm_lds) <= sizeof(fuse_lds<LCAP>), "the normals tile works in the fusion table's LDS");
        const int t = (int)blockIdx.x - a.n_tiles;
        if (a.use_dev_pose) {
            /* a tracked frame (gsdf_hint_next_depth_dev): only beside a fusion that RUNS -- there the tail is idle; a launch whose
             * gate is closed would run these tiles alone, in the open (14-19 us, measured).  The next frame's tracker launches
             * carry the tiles as riders as ever; they look at the token below and leave when the work is done. */
            int gd = a.st->done, gc = a.st->converged;
            if constexpr (HEAD) {
                float pose[7];
                int hd_ = 0, hc_ = 0;
                fuse_head(a, tid, false, pose, hd_, hc_);
                gd = __builtin_amdgcn_readfirstlane(hd_); gc = __builtin_amdgcn_readfirstlane(hc_);
            }
            if (!(gd && gc)) return;
            if (t == 0 && tid == 0) a.st->nrm_token = a.nrm_token;
        }
        normals_tile<FUSE_THREADS>(*reinterpret_cast<nrm_lds*>(&L), t % a.nrm_ntx, t / a.nrm_ntx, a.g.W, a.g.H, a.nrm_r, a.nc, a.nrm_depth, a.nrm_x,
                     a.nrm_y, a.nrm_z, gsdf_tile_stats{ a.nrm_stats, (a.g.W + 15) / 16, a.g.zmin, a.g.zmax });
        return;
    }
    }
    /* main_scan_3d.cpp:261: if (conv) update.  The launch may have been queued before optimize() ended (the host
     * issues it behind every batch of passes): it runs only once the pose iteration is done AND converged; the
     * launch that finds it done but NOT converged only writes the frame's log row. */
    /* everything the prologue needs from memory is requested at once (one scalar round trip, not three in a row):
     * the gate, the device pose, the workgroup's tile */
    const gsdf_dev_state* st_in = a.st;
    int done = st_in->done, conv = st_in->converged;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = st_in->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = st_in->pose7[i];
    const uint32_t tile_id = a.tile_order[blockIdx.x];
    if constexpr (HEAD) {
        float pose[7];
        int hdone = 0, hconv = 0;
        fuse_head(a, tid, blockIdx.x == 0 && tid < 64, pose, hdone, hconv);
        /* (done, conv, R, t) of this launch are the head's statement, never the plain loads above: see fuse_head */
        done = __builtin_amdgcn_readfirstlane(hdone); conv = __builtin_amdgcn_readfirstlane(hconv);
        float Rh[9];
        gsdf_quat_to_R(pose + 3, Rh);                                    /* st->R is computed the same way: identical bits */
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(Rh[i])));
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pose[i])));
    }
    if (a.use_dev_pose) {
        if (!(done && conv)) {
            if (done && blockIdx.x == 0 && tid == 0) fuse_log_row(a);
            return;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = a.pose.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = a.pose.t[i];
    }
    const unsigned long long T0 = GSDF_EXPERIMENT(a.debug, 128) ? wall_clock64() : 0ull;
    unsigned long long* tr = nullptr;                         /* trace row of this workgroup (test build, debug bit 64) */
    if (GSDF_EXPERIMENT(a.debug, 64)) {
        tr = reinterpret_cast<unsigned long long*>(st_in->dbg[23]) + 16 * (size_t)blockIdx.x;
        if (tid == 0) {
            tr[0] = wall_clock64();
            tr[14] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |              /* HW_ID */
                     ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);       /* XCC_ID */
            tr[15] = tile_id;
        }
    }

    const gsdf_frame_geom& g = a.g;
    const long long frame_cur = a.vis ? a.st->frame_cur : 0;   /* Sdf::counter_ of this update (snapshot by k_normals) */
    const int wave = tid >> 6, lane = tid & 63;
    const int zslice = wave >> 2;
    (void)zslice;
    const int lx = lane & 7, ly = lane >> 3;
    const int tile_x = (int)(tile_id & 0xFFFFu), tile_y = (int)(tile_id >> 16);
#if FUSE_STATS_AHEAD
    /* The tile's depth range and its number of valid pixels come from the normals stage of this frame (gsdf_tile_stats): three
     * words, requested now.  Wave 0 makes the tile-wide decisions from them while the seven pixel planes are still on their way
     * -- they used to be reduced from the arrived pixels (DPP, LDS, a barrier) and the decisions (~200 dependent instructions
     * of one wave) followed behind that: 1.9 us of every tile's prologue, exposed whenever all workgroups of the chip start
     * together (the first dispatch round) or run alone (the last). */
    const uint32_t* tsr = a.tile_stats + 4 * ((size_t)tile_y * a.ntx + tile_x);
    const unsigned int ts_zmin = tsr[0], ts_zmax = tsr[1], ts_nval = tsr[2];
#endif
    bool valid = false;
    float z = 0.f;
    gsdf_v3 Rxy = { 0.f, 0.f, 0.f }, Rn = { 0.f, 0.f, 0.f };   /* Rn carries the fixed-point scale: w * Rn is the scaled term */
    auto finish_pixel = [&](bool inside, float zr, float x0r, float y0r, float ninv, float nxr, float nyr, float nzr) {
        valid = inside;
        z = 0.f;
        if (inside) {
            z = zr;
            const gsdf_v3 xy = { x0r, y0r, 1.f };                          /* :90 */
            const gsdf_v3 n = { nxr, nyr, nzr };                           /* :92 */
            valid = gsdf_fuse_pixel_valid(z, x0r, y0r, ninv, nxr, nyr, nzr, g.zmin, g.zmax);   /* MapGradPixelSdf.cpp:87, :95, :98 */
            Rxy = gsdf_matvec(R, xy);                                      /* :91 */
            const gsdf_v3 rn = gsdf_matvec(R, n);                          /* :93 */
            Rn = gsdf_v3{ rn.x * FUSE_FIX_G, rn.y * FUSE_FIX_G, rn.z * FUSE_FIX_G };         /* exact (power of two) */
        }
    };
    auto load_pixel = [&](int px, int py, const float* dp, const float* x0p, const float* y0p, const float* nip,
                          const float* nxp, const float* nyp, const float* nzp) {
        const bool inside = px < g.W && py < g.H;
        float r[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (inside) {
            /* all seven loads are issued together: one memory round trip, not two */
            const size_t idx = (size_t)py * g.W + px;
            r[0] = dp[idx]; r[1] = x0p[idx]; r[2] = y0p[idx]; r[3] = nip[idx]; r[4] = nxp[idx]; r[5] = nyp[idx]; r[6] = nzp[idx];
        }
        finish_pixel(inside, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
    };
    /* the common case is one band: its pixels are requested now; the table clear and everything of the tile-wide
     * decisions that does not depend on the pixels (corner rays of the tile's frustum) run while the loads are in flight */
#if FUSE_SPREAD
    /* Lanes of a wave are spread out, because lanes that hit the same voxel in one instruction serialise in the LDS
     * atomics: a wave takes 32 pixels (every 2nd in x, every (4 / bands)-th in y: wave w has phase (w & 1, w >> 1)) and
     * BOTH slices of their ray walk (lanes 0-31 / 32-63), so neighbouring lanes are >= 2 pixels or half a ray apart. */
    static_assert(FUSE_ZSPLIT == 2, "the spread mapping splits a ray between the two halves of a wave");
    const int px0 = tile_x * FUSE_T + 2 * lx + (wave & 1), py0 = tile_y * FUSE_TH + FUSE_RP * (ly & 3) + (wave >> 1);
#else
    static_assert(FUSE_TH == 16, "the compact mapping exists for 16 x 16 tiles only");
    const int px0 = tile_x * FUSE_T + (wave & 1) * 8 + lx, py0 = tile_y * FUSE_T + ((wave >> 1) & 1) * 8 + ly;
#endif
    float raw[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    const bool inside0 = px0 < g.W && py0 < g.H;
    if (inside0) {
        const size_t idx = (size_t)py0 * g.W + px0;
        raw[0] = a.depth[idx]; raw[1] = a.nc.x0[idx]; raw[2] = a.nc.y0[idx]; raw[3] = a.nc.ninv[idx];
        raw[4] = a.nx[idx]; raw[5] = a.ny[idx]; raw[6] = a.nz[idx];
    }
    /* meanwhile: empty table */
    fuse_lds_clear(L, tid);
    if (GSDF_EXPERIMENT(a.debug, 128)) {                    /* phase timer: all seven pixel loads have arrived */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) atomicAdd(&a.st->dbg[5], wall_clock64() - T0);
    }
    if (tid == 0) {
        L.n_defer = 0u; L.defer_base = 0u; L.any_defer = 0u;
        L.plane[0] = a.depth; L.plane[1] = a.nc.x0; L.plane[2] = a.nc.y0; L.plane[3] = a.nc.ninv;
        L.plane[4] = a.nx; L.plane[5] = a.ny; L.plane[6] = a.nz;
    }
    /* extreme components of the rays through the tile's corners (plus a pixel of margin): R (x0, y0, 1) */
    float dmin[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, dmax[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (wave == 0) {                                                    /* the tile-wide decisions below are wave 0's job */
        const float rfx = __frcp_rn(g.fx), rfy = __frcp_rn(g.fy);       /* a bounding box with margin: no parity item */
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float u = (float)(tile_x * FUSE_T + ((c & 1) ? FUSE_T : -1)), v = (float)(tile_y * FUSE_TH + ((c & 2) ? FUSE_TH : -1));
            const gsdf_v3 d = gsdf_matvec(R, gsdf_v3{ (u - g.cx) * rfx, (v - g.cy) * rfy, 1.f });
            dmin[0] = fminf(dmin[0], d.x); dmin[1] = fminf(dmin[1], d.y); dmin[2] = fminf(dmin[2], d.z);
            dmax[0] = fmaxf(dmax[0], d.x); dmax[1] = fmaxf(dmax[1], d.y); dmax[2] = fmaxf(dmax[2], d.z);
        }
    }
#if !FUSE_STATS_AHEAD
    if (GSDF_EXPERIMENT(a.debug, 64)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GSDF_TRACE(a, tr, 11); }   /* pixel loads arrived */
    finish_pixel(inside0, raw[0], raw[1], raw[2], raw[3], raw[4], raw[5], raw[6]);
    /* depth range of the tile and the number of valid pixels: one entry per wave that holds distinct pixels */
    if (FUSE_SPREAD || zslice == 0) {
        unsigned int zb = valid ? __float_as_uint(z) : 0x7F800000u, zt = valid ? __float_as_uint(z) : 0u;
        wave_uminmax(zb, zt);                                         /* DPP: no LDS round trips */
        /* spread mapping: both halves of a wave hold the same 32 pixels */
        const float cnt = (float)__popcll(FUSE_SPREAD ? (__ballot(valid) & 0xFFFFFFFFull) : __ballot(valid));
        if (lane == 0) { L.st_min[wave] = zb; L.st_max[wave] = zt; L.st_cnt[wave] = cnt; }
    }
    __syncthreads();
    GSDF_TRACE(a, tr, 12);                                            /* tile statistics reduced */
#endif
    const int nk_all = 2 * g.factor + 1;
    const int colour = (tile_x & 1) + 2 * (tile_y & 1);
    /* Wave 0 derives the tile-wide decisions from the per-wave entries and hands them to the others through LDS: ~200 uniform
     * instructions that all eight waves used to execute -- 7 % of the kernel's instruction issue, which is what bounds it. */
    int n_pass = 1;
    int big = 0;                                                    /* the tile (each of its bands) uses the full LDS table */
    int ox = 0, oy = 0, oz = 0;
    bool range_ok = false;
    if (wave == 0) {
#if FUSE_STATS_AHEAD
        const unsigned int zmin_bits = ts_zmin, zmax_bits = ts_zmax;
        const float n_valid = (float)ts_nval;
#else
        unsigned int zmin_bits = L.st_min[0], zmax_bits = L.st_max[0];
        float n_valid = L.st_cnt[0];
#pragma unroll
        for (int i = 1; i < FUSE_NSTAT; ++i) {
            zmin_bits = min(zmin_bits, L.st_min[i]); zmax_bits = max(zmax_bits, L.st_max[i]);
            n_valid += L.st_cnt[i];                                   /* small integers: exact in any order */
        }
#endif
        /* (1) of the flush comment below: may this tile write its voxels itself? */
        const float D = 1.7421f * g.vs;
        const float s_min = __uint_as_float(zmin_bits) - (float)g.factor * g.vs - 2.f * D;
        const float x_lo = (float)(tile_x * FUSE_T) - g.cx, x_hi = x_lo + (float)(FUSE_T - 1);
        const float y_lo = (float)(tile_y * FUSE_TH) - g.cy, y_hi = y_lo + (float)(FUSE_TH - 1);
        const float xm = fmaxf(fabsf(x_lo), fabsf(x_hi)) + 1.f, ym = fmaxf(fabsf(y_lo), fabsf(y_hi)) + 1.f;
        const float gap = (float)FUSE_T + 0.5f, gap_y = (float)FUSE_TH + 0.5f;     /* true gaps: one pixel more */
        bool ordered = s_min > 0.f && D * (g.fx + xm) <= gap * s_min && D * (g.fy + ym) <= gap_y * s_min;
        if (!(n_valid > 0.f)) ordered = false;                          /* nothing to write: the flag is published at once (below) */
        if (GSDF_EXPERIMENT(a.debug, 4)) ordered = false;
        if (tid == 0) {
            L.ordered = ordered ? 1u : 0u;                            /* read after the ray walk's barrier */
            /* a tile that writes nothing itself has nothing to hand over: publish at once */
            if (!ordered) __hip_atomic_store(a.tile_flags + (size_t)tile_y * a.ntx + tile_x, a.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        /* Distinct voxels the tile will touch: the volume its rays sweep (pixels x samples / pixels per voxel face) plus
         * half a voxel around that prism (a side of the tile is `side` voxels wide at depth zf).  Calibrated against
         * counted voxels on the bench stream: counted / estimate = 0.93 median, never above 1 for tiles that matter.  Far
         * tiles (or small voxels) would overflow the LDS table -- and a table filled beyond ~80 % probes long and drops
         * samples into the deferred list -- so they are walked as 2 or 4 row bands, each flushed on its own. */
        const float zf = __uint_as_float(zmax_bits);
        const float rzf = __builtin_amdgcn_rcpf(zf);                    /* an estimate: hardware reciprocals will do */
        const float ppr = (g.fx * g.vs * rzf) * (g.fy * g.vs * rzf);    /* pixels per voxel face at the far end */
        const float side = (float)FUSE_T * __builtin_amdgcn_rsqf(ppr), side_y = (float)FUSE_TH * __builtin_amdgcn_rsqf(ppr);
        const float samples = n_valid * (float)nk_all;
        const float est = fminf(samples, samples * __builtin_amdgcn_rcpf(ppr) + (side * side_y + (side + side_y) * (float)nk_all) * __builtin_amdgcn_sqrtf(n_valid * (1.f / (float)(FUSE_T * FUSE_TH))));
        n_pass = est <= 0.8f * FUSE_LCAP ? 1 : (est <= 1.6f * FUSE_LCAP ? 2 : FUSE_NPASS_MAX);
        if (GSDF_EXPERIMENT(a.debug, 256)) n_pass = 1;
        if (GSDF_EXPERIMENT(a.debug, 512)) n_pass = FUSE_NPASS_MAX;
        /* a tile without a valid pixel (background: two thirds of the tiles of a sphere frame) walks and flushes nothing: no band
         * at all -- it used to go through an empty walk and an empty flush, ~7 us of a workgroup slot */
        if (!(n_valid > 0.f)) n_pass = 0;
        big = FUSE_DUAL ? (est > 0.8f * FUSE_LCAP_SMALL * (float)n_pass ? 1 : 0) : 0;
        big = __builtin_amdgcn_readfirstlane(big);
        /* tiles the small table cannot hold in one band: what the host chooses the next launches' table size by */
        if (tid == 0 && est > 0.8f * FUSE_LCAP_SMALL) atomicAdd(&a.st->far_tiles, 1u);
        n_pass = __builtin_amdgcn_readfirstlane(n_pass);
        /* origin of the tile-local voxel coordinates: the world bounding box of the tile's frustum chunk
         * (4 corner rays x the two ends of the sampled depth range) minus a margin; a sample whose voxel is not
         * within 1024 cells of it (huge depth range at tiny voxels) takes the deferred route instead */
        const float s_lo = __uint_as_float(zmin_bits) - ((float)g.factor + 1.f) * g.vs;
        const float s_hi = __uint_as_float(zmax_bits) + ((float)g.factor + 1.f) * g.vs;
        float mn[3];                                                    /* a bilinear form is smallest at a vertex of its box */
#pragma unroll
        for (int i = 0; i < 3; ++i) mn[i] = fminf(fminf(s_lo * dmin[i], s_hi * dmin[i]), fminf(s_lo * dmax[i], s_hi * dmax[i]));
        const bool any_valid = zmax_bits != 0u;
        /* clamped so that the integer conversion is defined whatever the pose; a clamped origin fails range_ok */
        const float lim = 2.0e6f;
        ox = any_valid ? (int)fminf(lim, fmaxf(-lim, floorf((mn[0] + t[0]) * g.inv_vs))) - 3 : 0;
        oy = any_valid ? (int)fminf(lim, fmaxf(-lim, floorf((mn[1] + t[1]) * g.inv_vs))) - 3 : 0;
        oz = any_valid ? (int)fminf(lim, fmaxf(-lim, floorf((mn[2] + t[2]) * g.inv_vs))) - 3 : 0;
        ox = __builtin_amdgcn_readfirstlane(ox); oy = __builtin_amdgcn_readfirstlane(oy); oz = __builtin_amdgcn_readfirstlane(oz);
        /* every voxel with a local key is packable (21 bits per biased axis): the per-sample range test of the
         * global key is only needed on the deferred route */
        range_ok = ox >= -GSDF_KEY_OFF && ox + 1023 < GSDF_KEY_OFF && oy >= -GSDF_KEY_OFF && oy + 1023 < GSDF_KEY_OFF &&
                   oz >= -GSDF_KEY_OFF && oz + 1023 < GSDF_KEY_OFF;
        if (lane == 0) { L.dec[0] = n_pass; L.dec[1] = big; L.dec[2] = ox; L.dec[3] = oy; L.dec[4] = oz; L.dec[5] = range_ok ? 1 : 0; }
    }
#if FUSE_STATS_AHEAD
    /* (wave 0 has decided while the loads were in flight; now everybody takes its pixel) */
    if (GSDF_EXPERIMENT(a.debug, 64)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GSDF_TRACE(a, tr, 11); }   /* (wave 0: decisions made and) pixel loads arrived */
    finish_pixel(inside0, raw[0], raw[1], raw[2], raw[3], raw[4], raw[5], raw[6]);
    GSDF_TRACE(a, tr, 12);
#endif
    __syncthreads();
    n_pass = __builtin_amdgcn_readfirstlane(L.dec[0]); big = __builtin_amdgcn_readfirstlane(L.dec[1]);
    ox = __builtin_amdgcn_readfirstlane(L.dec[2]); oy = __builtin_amdgcn_readfirstlane(L.dec[3]); oz = __builtin_amdgcn_readfirstlane(L.dec[4]);
    range_ok = __builtin_amdgcn_readfirstlane(L.dec[5]) != 0;
    unsigned int n_upd_w = 0u, n_val_w = 0u;                          /* wave-uniform counters */
    unsigned int dbg_go = 0u, dbg_full = 0u, dbg_lost = 0u;
    unsigned long long T1 = GSDF_EXPERIMENT(a.debug, 128) ? wall_clock64() : 0ull;
    GSDF_TRACE(a, tr, 1);                                             /* prologue done */
    if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) atomicAdd(&a.st->dbg[4], T1 - T0);
    for (int pass = 0; pass < n_pass; ++pass) {
    /* lanes -> (pixel of the band, slice of the ray walk).  One band: as loaded above, FUSE_ZSPLIT slices.  Two bands
     * of 16x8 pixels: 2 waves (8x8 each) per slice, twice the slices.  Four bands of 16x4: 1 wave per slice. */
#if FUSE_SPREAD
    /* a band of 256 / n_pass pixels = G groups of 32 (one per wave % G), walked in 2 * n_pass slices: the two halves
     * of a wave take slices 2 (wave / G) and 2 (wave / G) + 1 */
    const int G = FUSE_NWAVES / n_pass, grp = wave % G;              /* n_pass is 1, 2 or FUSE_NPASS_MAX */
    const int zs = (lane >> 5) + 2 * (wave / G);
    int bx = 2 * lx + (grp & 1), by = (G / 2) * (ly & 3) + (grp >> 1);
    if (G & 1) {                                                      /* an odd number of groups (TH = 20, two bands): row-major pixels */
        const int q = grp * 32 + (lane & 31);
        bx = q & (FUSE_T - 1); by = q / FUSE_T;
    }
#else
    const int per = FUSE_THREADS / (FUSE_ZSPLIT * n_pass);
    const int q = tid % per, zs = tid / per;
    const int bx = n_pass == 4 ? (q & 15) : ((q >> 6) & 1) * 8 + (q & 7);
    const int by = n_pass == 4 ? (q >> 4) : (q >> 7) * 8 + ((q >> 3) & 7);
#endif
    if (n_pass > 1)
        load_pixel(tile_x * FUSE_T + bx, tile_y * FUSE_TH + pass * (FUSE_TH / n_pass) + by,
                   L.plane[0], L.plane[1], L.plane[2], L.plane[3], L.plane[4], L.plane[5], L.plane[6]);
    n_val_w += (unsigned int)__popcll(__ballot(valid && zs == 0));    /* every pixel is held by one lane per slice */
    /* this slice's share of the ray walk k = -factor..factor (:101); the order is free (sums) */
    const int k_lo = -g.factor + (zs * nk_all) / (FUSE_ZSPLIT * n_pass);
    const int k_hi = -g.factor + ((zs + 1) * nk_all) / (FUSE_ZSPLIT * n_pass) - 1;
    const int nk_lane = k_hi - k_lo + 1;
    int nk = nk_lane;                                                  /* loop count of the wave: its longest slice */
#if FUSE_SPREAD
    nk = max(nk, __shfl_xor(nk, 32));
#endif
    nk = __builtin_amdgcn_readfirstlane(nk);
    FUSE_SETPRIO(FUSE_PRIO_WALK);                    /* (build experiment, default none: see FUSE_PRIO_*) */
    if (nk > 0 && __any(valid)) {                    /* waves without any valid pixel skip the walk */
        /* One sample per lane and iteration (2, 3, 4 or 6 samples in flight were measured slower: registers).  The loop
         * is bound by VALU issue when both workgroups of a CU walk, so it is written for instruction count: x and y go
         * through packed (2 x f32) instructions, the clamp is one v_med3, keys / hash / LDS addresses use shift-adds and
         * 24-bit multiply-adds (v_mul_lo_u32 is quarter rate), the weight goes to fixed point without the double trick. */
        const gsdf_f2 Rxy2 = { Rxy.x, Rxy.y }, t2 = { t[0], t[1] }, Rz2 = { R[2], R[5] }, Rn2 = { Rn.x, Rn.y };
        const uint32_t nb_used = (FUSE_DUAL && !big) ? (uint32_t)(FUSE_LCAP_SMALL / FUSE_BSLOTS) : (uint32_t)FUSE_NB;   /* buckets of this tile's table */
        float kf = (float)k_lo;                                       /* small integers: exact */
        for (int c0 = 0; c0 < nk; ++c0, kf += 1.f) {
            /* 1. the sample */
            const float sd = z + kf * g.vs;
            const gsdf_f2 pxy = sd * Rxy2 + t2;                       /* :103 */
            const float pzw = sd * Rxy.z + t[2];
            /* float2vox (:104): std::round; the rounded value is kept as a float too -- (float)vi == the rounded
             * float for every index that fits an int, so vox2float needs no second conversion */
            const gsdf_f2 qxy = g.inv_vs * pxy;
            const float rx = gsdf_roundf(qxy.x), ry = gsdf_roundf(qxy.y), rz = gsdf_roundf(g.inv_vs * pzw);
            const int vx = (int)rx, vy = (int)ry, vz = (int)rz;
            const gsdf_f2 rxy = { rx, ry };
            const gsdf_f2 dxy = g.vs * rxy - t2;
            const float dz = g.vs * rz - t[2];
            const gsdf_f2 cxy = Rz2 * dxy;
            const float pc_z = gsdf_sum3(cxy.x, cxy.y, R[8] * dz);    /* :105  (Rt row 2) */
            const float sdf = pc_z - z;                                /* :106 */
            /* :107 Sdf::weight: 1 for sdf <= 0 (there 1 - sdf / T >= 1), the ramp up to T, 0 beyond (and for NaN) */
            const float w = sdf <= g.T ? __builtin_amdgcn_fmed3f(1.f - sdf * g.inv_T, -__builtin_inff(), 1.f) : 0.f;
            const bool act = valid && w > 0.f && c0 < nk_lane;
            n_upd_w += (unsigned int)__popcll(__ballot(act));
            /* tile-local key: 10 bits per axis relative to the tile origin */
            const uint32_t lx3 = (uint32_t)(vx - ox), ly3 = (uint32_t)(vy - oy), lz3 = (uint32_t)(vz - oz);
            const bool local = range_ok && (lx3 | ly3 | lz3) < 1024u;
            const uint32_t key = lx3 + (ly3 << 10) + (lz3 << 20);
            /* LDS bucket: a LATTICE hash, not a random one.  A tile's voxels are a compact oblique prism;
             * x + 98 y + 143 z (mod 512) sends any two voxels closer than ~8.6 cells to different buckets
             * (best 3-D lattice for this modulus, found by search), so buckets fill evenly, rarely
             * overflow, and the distinct voxels of one wave instruction never compete for a bucket: counted on
             * tiles of the bench stream, 1.004 probes per sample.  (The HBM table keeps the full 64-bit finaliser.) */
            static_assert((FUSE_BSLOTS == 2 && FUSE_NB == 1024) || (FUSE_BSLOTS == 4 && (FUSE_NB == 512 || FUSE_NB == 480 || FUSE_NB == 640 || FUSE_NB == 256 || FUSE_NB == 320)),
                          "lattice constants exist for 1024 x 2, 512 x 4, 480 x 4, 640 x 4 (16-row tiles) and 256 x 4, 320 x 4 (8-row tiles)");
            uint32_t bk;
            if (FUSE_BSLOTS == 2) bk = gsdf_mad_u24(lz3, 75u, gsdf_mad_u24(ly3, 86u, lx3)) & 1023u;
            else if (FUSE_NB == 640 && (big || !FUSE_DUAL)) {          /* x + 253 y + 541 z (mod 640): min distance 9.3; 1.000 probes per sample on the densest tiles */
                const uint32_t hx = gsdf_mad_u24(lz3, 541u, gsdf_mad_u24(ly3, 253u, lx3));         /* < 2^20 for local keys */
                bk = hx - 640u * __umulhi(hx, 6710887u);                /* exact for hx < 2^21 */
                bk = bk < 640u ? bk : 0u;                               /* keys outside the local range: any bucket, never used */
            }
            else if (FUSE_NB == 320 && (big || !FUSE_DUAL)) {          /* x + 299 y + 271 z (mod 320): min distance 7.3 (8-row tiles, far) */
                const uint32_t hx = gsdf_mad_u24(lz3, 271u, gsdf_mad_u24(ly3, 299u, lx3));         /* < 2^20 for local keys */
                bk = hx - 320u * __umulhi(hx, 13421773u);               /* exact for hx < 2^25 */
                bk = bk < 320u ? bk : 0u;
            }
            else if (FUSE_NB == 480) {                                 /* x + 313 y + 195 z (mod 480): min distance 8.1; 1.12 probes per sample on the densest tiles */
                const uint32_t hx = gsdf_mad_u24(lz3, 195u, gsdf_mad_u24(ly3, 313u, lx3));         /* < 2^20 for local keys */
                bk = hx - 480u * __umulhi(hx, 8947849u);                /* exact for hx < 2^20 */
                bk = bk < 480u ? bk : 0u;                               /* keys outside the local range: any bucket, never used */
            }
            else if (FUSE_LCAP_SMALL == 1024) bk = gsdf_mad_u24(lz3, 73u, gsdf_mad_u24(ly3, 136u, lx3)) & 255u;   /* x + 136 y + 73 z (mod 256): min distance 6.9 (8-row tiles) */
            else bk = gsdf_mad_u24(lz3, 143u, gsdf_mad_u24(ly3, 98u, lx3)) & 511u;
            /* 2.-4. look the voxel up in the LDS table: one read per bucket; the first slot that holds the key or is empty
             *    decides (used slots are a prefix: entries are never removed and inserts take the first empty slot), at
             *    most one CAS per probe.  Plain LDS reads: a stale EMPTY is resolved by the CAS. */
            int slot = -1;
            bool pend = act && local && !GSDF_EXPERIMENT(a.debug, 2);
            if (GSDF_EXPERIMENT(a.debug, 32)) { if (act) slot = (int)(FUSE_BSLOTS * bk + (key & (FUSE_BSLOTS - 1))); pend = false; }   /* experiment: no lookup */
            for (int probe = 0; probe < FUSE_LPROBE && pend; ++probe) {       /* a divergent loop: lanes leave it as they find their slot */
                ++dbg_go;
                int pos;
                bool hit;
                if (FUSE_BSLOTS == 2) {
                    const uint2 k2 = *reinterpret_cast<const uint2*>(&L.key[2 * bk]);
                    const bool h0 = k2.x == key, h1 = k2.y == key;
                    const bool m0 = h0 | (k2.x == FUSE_LKEY_EMPTY), m1 = h1 | (k2.y == FUSE_LKEY_EMPTY);
                    pos = m1 ? 1 : -1;
                    pos = m0 ? 0 : pos;
                    hit = h0 | h1;
                } else {
                    const uint4 k4 = *reinterpret_cast<const uint4*>(&L.key[4 * bk]);
                    const bool h0 = k4.x == key, h1 = k4.y == key, h2 = k4.z == key, h3 = k4.w == key;
                    const bool m0 = h0 | (k4.x == FUSE_LKEY_EMPTY), m1 = h1 | (k4.y == FUSE_LKEY_EMPTY),
                               m2 = h2 | (k4.z == FUSE_LKEY_EMPTY), m3 = h3 | (k4.w == FUSE_LKEY_EMPTY);
                    pos = m3 ? 3 : -1;
                    pos = m2 ? 2 : pos; pos = m1 ? 1 : pos; pos = m0 ? 0 : pos;
                    hit = h0 | h1 | h2 | h3;
                }
                const int at = (int)((uint32_t)FUSE_BSLOTS * bk) + pos;
                const bool try_cas = pend && !hit && pos >= 0;
                if (pend && hit) { slot = at; pend = false; }
                if (pend && pos < 0) bk = bk + 1u == nb_used ? 0u : bk + 1u;                                 /* bucket full of others */
                if (GSDF_EXPERIMENT(a.debug, 128) && __any(pend && pos < 0)) ++dbg_full;
                if (try_cas) {
                    const uint32_t old = atomicCAS(&L.key[at], FUSE_LKEY_EMPTY, key);
                    if (old == FUSE_LKEY_EMPTY || old == key) { slot = at; pend = false; }
                }
                if (GSDF_EXPERIMENT(a.debug, 128) && __any(try_cas && pend)) ++dbg_lost;
                /* a lost CAS (slot taken by another voxel) re-reads the same bucket in the next probe */
            }
            /* 5. accumulate (integer adds: exact and order-independent) */
            if (!act || GSDF_EXPERIMENT(a.debug, 2) || GSDF_EXPERIMENT(a.debug, 16)) continue;
            /* w 2^24 is an integer (see the accumulator layout) */
            const uint32_t wi = (uint32_t)(w * FUSE_FIX_W);
            const unsigned long long qs = f2fix(w * __builtin_amdgcn_fmed3f(sdf, -g.T, g.T));   /* :111 as additive sum; Sdf::truncate */
            const gsdf_f2 gxy = w * Rn2;
            const int qg0 = (int)gxy.x, qg1 = (int)gxy.y, qg2 = (int)(w * Rn.z);                /* :112 (truncating) */
            unsigned long long q0, qwz, qgxy;
            fuse_pack(wi, qs, qg0, qg1, qg2, q0, qwz, qgxy);
            if (slot >= 0) {
                atomicAdd(&L.acc[0][slot], q0);
                atomicAdd(&L.acc[1][slot], qwz);
                atomicAdd(&L.acc[2][slot], qgxy);
            } else {
                /* LDS table full for this voxel, or voxel outside the local key range: deferred list */
                if (!gsdf_key_in_range(vx, vy, vz)) { atomicOr(&a.st->status, GSDF_STATUS_KEY_RANGE); continue; }
                gsdf_payload* p = gsdf_find_or_insert(a.tab, gsdf_key_pack(vx, vy, vz));
                if (!p) atomicOr(&a.st->status, GSDF_STATUS_TABLE_FULL);
                else {
                    const fuse_sums d = fuse_unpack(q0, qwz, qgxy);
                    defer_append(a, p, d.w, d.s, d.gx, d.gy, d.gz);
                    L.any_defer = 1u;
                    vis_mark(a, p, frame_cur);
                }
            }
        }
    }
    FUSE_SETPRIO(FUSE_PRIO_FLUSH);                   /* (build experiment, default none) */
    if (GSDF_EXPERIMENT(a.debug, 128) && lane == 0) {
        atomicAdd(&a.st->n_hit, (unsigned long long)dbg_go);
        atomicAdd(&a.st->dbg[0], (unsigned long long)dbg_full); atomicAdd(&a.st->dbg[1], (unsigned long long)dbg_lost);
        dbg_go = dbg_full = dbg_lost = 0u;
    }
    __syncthreads();
    if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T2 = wall_clock64(); atomicAdd(&a.st->dbg[2], T2 - T1); T1 = T2; }
    if (pass == 0) GSDF_TRACE(a, tr, 2);                              /* ray walk done (first band) */
    /* Flush the tile's distinct voxels: read-modify-write of the HBM payload with NO atomics.
     *
     * Mutual exclusion between tiles comes from two facts:
     *  (1) far tiles cannot meet.  Two samples s1 d1, s2 d2 (d = (x0, y0, 1)) that round to one voxel are
     *      < D = sqrt(3) vs apart, hence |s1 - s2| < D and |x1 - x2| < D (1 + |x1|) / s2.  If every sample
     *      of this tile is deep enough that this bound stays below FUSE_T + 1 pixels, only the 8 adjacent
     *      tiles can touch this tile's voxels ("ordered" tile).  A tile that is too near for that sends
     *      all its contributions through the deferred list (float atomics in k_fuse_resolve, after the launch).
     *  (2) adjacent tiles take turns.  Tiles are coloured by the parity of (tile_x, tile_y); an ordered tile
     *      waits until its adjacent tiles of LOWER colour have published their flag for this launch.
     *      Workgroups are numbered colour-major, so whatever a tile waits for was dispatched before it.
     *      (Dispatch order is only a performance assumption: the wait is bounded, and a tile whose wait
     *      times out defers its contributions instead.)
     * Hand-off between workgroups (other CUs, other XCDs: neither L1 nor the per-XCD L2s are coherent for
     * plain accesses): voxel records are read and written with agent-scope (sc1) 16-byte accesses, every
     * storing wave drains its stores, then one lane publishes the tile's flag with an agent-scope store.
     * Block keys are insert-only, so plain (possibly stale) key loads can only show EMPTY and the CAS
     * settles it.  Each lane owns FUSE_LCAP/FUSE_THREADS LDS slots and drives them through the stages
     * together, so the dependent HBM round trips (bucket keys -> payload) of its entries overlap. */
    unsigned int* const my_flag = a.tile_flags + (size_t)tile_y * a.ntx + tile_x;
    const unsigned int tag = a.tag;
    if (!GSDF_EXPERIMENT(a.debug, 1)) {
        const gsdf_table& tab = a.tab;
        constexpr int NE = (FUSE_LCAP + FUSE_THREADS - 1) / FUSE_THREADS;
        constexpr uint32_t NOREC = 0xFFFFFFFFu;
        unsigned long long bkey[NE], k0[NE];
        uint32_t home[NE];
        uint32_t rec[NE];                   /* the entry's voxel record: index in the block, then index in the map; NOREC = no entry */
        uint32_t have = 0u;                 /* bit e: slot e of this lane holds a voxel */
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            /* back from the tile-local key to the packed voxel key of the HBM map */
            const int i = tid + FUSE_THREADS * e;
            const uint32_t lk = (i < FUSE_LCAP && (big || !FUSE_DUAL || i < FUSE_LCAP_SMALL)) ? L.key[i] : FUSE_LKEY_EMPTY;
            const unsigned long long ek = gsdf_key_pack(ox + (int)(lk & 1023u), oy + (int)((lk >> 10) & 1023u), oz + (int)(lk >> 20));
            bkey[e] = gsdf_block_key(ek);
            home[e] = gsdf_hash(bkey[e]) & tab.block_mask;
            rec[e] = lk == FUSE_LKEY_EMPTY ? NOREC : gsdf_block_local(ek);
            if (lk != FUSE_LKEY_EMPTY) have |= 1u << e;
            k0[e] = GSDF_KEY_EMPTY;
        }
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if ((have >> e) & 1u) k0[e] = tab.bkeys[home[e]];                   /* 512 KB of block keys: L2 hits */
        unsigned long long TF = GSDF_EXPERIMENT(a.debug, 128) ? wall_clock64() : 0ull;
        if (pass == 0) GSDF_TRACE(a, tr, 3);                          /* keys unpacked, home-entry loads issued */
        /* Meanwhile one wave looks after the adjacent tiles of lower colour (lane j watches neighbour j).  Their flags have
         * normally been published long ago (they were dispatched a colour earlier), so the first look is only REQUESTED here and
         * examined behind this wave's own block lookups: the load's round trip (an agent-scope access, ~0.5 us) runs under the
         * lookups' instead of in front of them -- wave 0 was the wave every flush waited for at the barrier below.  Only when a
         * flag is not there yet does the wave poll. */
        bool need = false;
        const unsigned int* flag = my_flag;
        unsigned int first_look = 0u;
        const bool watcher = wave == 0 && L.ordered && pass == 0;
        if (watcher) {
            if (lane < 8) {
                const int j = lane < 4 ? lane : lane + 1;             /* 3x3 neighbourhood without the centre */
                const int nx = tile_x + (j % 3) - 1, ny = tile_y + (j / 3) - 1;
                if (nx >= 0 && nx < a.ntx && ny >= 0 && ny < a.nty && (nx & 1) + 2 * (ny & 1) < colour && !GSDF_EXPERIMENT(a.debug, 8)) {
                    need = true;
                    flag = a.tile_flags + (size_t)ny * a.ntx + nx;
                }
            }
            first_look = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pass == 0) GSDF_TRACE(a, tr, 4);                          /* (wave 0) the flags are requested */
        {
            /* the blocks of this lane's entries, looked up (and inserted) together: the probe chains overlap */
            int blk[NE];
            gsdf_block_lookup_n<NE, true>(tab, bkey, home, k0, have, blk);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (!((have >> e) & 1u)) continue;
                if (blk[e] < 0) { atomicOr(&a.st->status, GSDF_STATUS_TABLE_FULL); rec[e] = NOREC; L.key[tid + FUSE_THREADS * e] = FUSE_LKEY_EMPTY; }
                else rec[e] += (uint32_t)blk[e] * GSDF_BLOCK_VOX;                  /* < 2^31 records (gsdf_create) */
            }
        }
        if (pass == 0) GSDF_TRACE(a, tr, 5);                          /* wave 0: its blocks looked up */
        if (watcher) {
            const unsigned long long t0 = wall_clock64();
            bool ok = !need || first_look == tag;
            for (;;) {
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(FUSE_POLL_SLEEP);
                /* 2 ms at 100 MHz: give up, defer instead (the test build can make every wait expire at once) */
                if (wall_clock64() - t0 > 200000ull || GSDF_EXPERIMENT(a.debug, 8192)) {
                    if (lane == 0) { L.ordered = 0u; atomicAdd(&a.st->fuse_timeouts, 1u); }
                    break;
                }
                if (!ok) ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
            }
        }
        __syncthreads();                                              /* the wait above is over (or timed out) */
        if (pass == 0) GSDF_TRACE(a, tr, 6);                          /* every wave has its blocks */
        if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T = wall_clock64(); atomicAdd(&a.st->dbg[8 + 4 * colour + 0], T - TF); TF = T; }
        const bool ordered = L.ordered != 0u;
        if (ordered) {
            /* a record is 32 bytes, 32-byte aligned: two 16-byte agent-scope (sc1) accesses each way -- narrower
             * write-through stores cost one fabric write each (3 x 8 B measured 2x slower).  hipcc does not
             * count asm memory operations: the wait statement below names every destination register.
             * FRAGILE (found in round 6, tools/experiments/k_fuse_flush_compact_r06.patch): because each load is issued under an
             * `if`, its destination is merged with the zero it holds otherwise, and in a restructured flush the compiler placed
             * that merge (v_mov copies of the destination) BETWEEN the load and the wait -- reading registers with a load in
             * flight (memory fault on dense tiles).  This form compiles to loads followed by the wait with nothing in between
             * (checked in the ISA: `global_load_dwordx4 ... sc1` x 2 NE, `s_waitcnt vmcnt(0)`); anyone who moves this code
             * should look at the ISA again or issue loads and wait as ONE asm statement, as that patch does. */
            gsdf_u32x4 ra[NE], rb[NE];
            /* (the 2560-entry table with the normals role is at the register limit: its five entries per lane stay unpaired) */
            constexpr bool PAIRED = FUSE_PAIRED_IO && !(NE > 4 && NEXT_NORMALS);
            /* ra[e]: the half this lane moves of the EVEN lane's record of entry e (even lane: bytes 0-15, odd lane: 16-31),
             * rb[e]: its half of the ODD lane's record.  The exchanges run with all lanes active; only the memory instructions
             * are predicated (a pair moves a record if the lane that owns it has one). */
            const bool odd = (lane & 1) != 0;
            uint32_t rec_a[NE], rec_b[NE];                   /* record index of the even / the odd lane's entry e (NOREC = none) */
#pragma unroll
            for (int e = 0; e < NE; ++e) { rec_a[e] = fuse_quad<FUSE_QP_EVEN>(rec[e]); rec_b[e] = fuse_quad<FUSE_QP_ODD>(rec[e]); }
            auto half_of = [&](uint32_t r) { return reinterpret_cast<const char*>(tab.vox + r) + (odd ? 16 : 0); };
            if constexpr (PAIRED) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                ra[e] = gsdf_u32x4{ 0u, 0u, 0u, 0u }; rb[e] = ra[e];
                if (rec_a[e] != NOREC) { const char* q = half_of(rec_a[e]); asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ra[e]) : "v"(q) : "memory"); }
                if (rec_b[e] != NOREC) { const char* q = half_of(rec_b[e]); asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rb[e]) : "v"(q) : "memory"); }
            }
            } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                ra[e] = gsdf_u32x4{ 0u, 0u, 0u, 0u }; rb[e] = ra[e];
                if (rec[e] == NOREC) continue;
                const gsdf_payload* pe = tab.vox + rec[e];
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ra[e]) : "v"(pe) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1" : "=v"(rb[e]) : "v"(pe) : "memory");
            }
            }
            static_assert(NE >= 2 && NE <= 5, "the wait statement names NE x 2 destination registers");
            if constexpr (NE == 5)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(ra[0]), "+v"(rb[0]), "+v"(ra[1]), "+v"(rb[1]), "+v"(ra[2]), "+v"(rb[2]), "+v"(ra[3]), "+v"(rb[3]), "+v"(ra[4]), "+v"(rb[4]) :: "memory");
            else if constexpr (NE == 4)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(ra[0]), "+v"(rb[0]), "+v"(ra[1]), "+v"(rb[1]), "+v"(ra[2]), "+v"(rb[2]), "+v"(ra[3]), "+v"(rb[3]) :: "memory");
            else if constexpr (NE == 3)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]), "+v"(ra[1]), "+v"(rb[1]), "+v"(ra[2]), "+v"(rb[2]) :: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]), "+v"(ra[1]), "+v"(rb[1]) :: "memory");
            if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T = wall_clock64(); atomicAdd(&a.st->dbg[8 + 4 * colour + 1], T - TF); TF = T; }
            if (pass == 0) GSDF_TRACE(a, tr, 7);                      /* wave 0: records arrived */
            if constexpr (PAIRED) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int i = tid + FUSE_THREADS * e;
                const bool mine = rec[e] != NOREC;
                fuse_sums d = { 0.f, 0.f, 0.f, 0.f, 0.f };
                if (mine) d = fuse_unpack(L.acc[0][i], L.acc[1][i], L.acc[2][i]);
                /* this lane's own record as it was: bytes 0-15 are ra (even lane) or the partner's rb (odd lane), the first word
                 * of bytes 16-31 (gz) the partner's ra.x (even lane) or rb.x (odd lane) */
                const uint32_t p0 = fuse_quad<FUSE_QP_SWAP>(rb[e].x), p1 = fuse_quad<FUSE_QP_SWAP>(rb[e].y),
                               p2 = fuse_quad<FUSE_QP_SWAP>(rb[e].z), p3 = fuse_quad<FUSE_QP_SWAP>(rb[e].w), q0 = fuse_quad<FUSE_QP_SWAP>(ra[e].x);
                const uint32_t o0 = odd ? p0 : ra[e].x, o1 = odd ? p1 : ra[e].y, o2 = odd ? p2 : ra[e].z, o3 = odd ? p3 : ra[e].w;
                const uint32_t o4 = odd ? rb[e].x : q0;
                gsdf_u32x4 oa;
                oa.x = __float_as_uint(__uint_as_float(o0) + d.w);
                oa.y = __float_as_uint(__uint_as_float(o1) + d.s);
                oa.z = __float_as_uint(__uint_as_float(o2) + d.gx);
                oa.w = __float_as_uint(__uint_as_float(o3) + d.gy);
                const uint32_t ogz = __float_as_uint(__uint_as_float(o4) + d.gz);
                /* what this lane stores: of the even lane's record (sa) and of the odd lane's record (sb) */
                const uint32_t e_gz = fuse_quad<FUSE_QP_SWAP>(ogz);
                gsdf_u32x4 sa, sb;
                sa.x = odd ? e_gz : oa.x; sa.y = odd ? tag : oa.y; sa.z = odd ? 0u : oa.z; sa.w = odd ? 0u : oa.w;
                const uint32_t x0 = fuse_quad<FUSE_QP_SWAP>(oa.x), x1 = fuse_quad<FUSE_QP_SWAP>(oa.y), x2 = fuse_quad<FUSE_QP_SWAP>(oa.z), x3 = fuse_quad<FUSE_QP_SWAP>(oa.w);
                sb.x = odd ? ogz : x0; sb.y = odd ? tag : x1; sb.z = odd ? 0u : x2; sb.w = odd ? 0u : x3;
                if (rec_a[e] != NOREC) { const char* q = half_of(rec_a[e]); asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(sa) : "memory"); }
                if (rec_b[e] != NOREC) { const char* q = half_of(rec_b[e]); asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(sb) : "memory"); }
                if (mine) vis_mark(a, tab.vox + rec[e], frame_cur);
            }
            } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (rec[e] == NOREC) continue;
                const int i = tid + FUSE_THREADS * e;
                const fuse_sums d = fuse_unpack(L.acc[0][i], L.acc[1][i], L.acc[2][i]);
                gsdf_payload* pe = tab.vox + rec[e];
                gsdf_u32x4 oa, ob;
                oa.x = __float_as_uint(__uint_as_float(ra[e].x) + d.w);
                oa.y = __float_as_uint(__uint_as_float(ra[e].y) + d.s);
                oa.z = __float_as_uint(__uint_as_float(ra[e].z) + d.gx);
                oa.w = __float_as_uint(__uint_as_float(ra[e].w) + d.gy);
                ob.x = __float_as_uint(__uint_as_float(rb[e].x) + d.gz);
                ob.y = tag; ob.z = 0u; ob.w = 0u;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(pe), "v"(oa) : "memory");
                asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\ts_nop 1" :: "v"(pe), "v"(ob) : "memory");
                vis_mark(a, pe, frame_cur);
            }
            }
            /* hand the voxels on: every storing wave drains its stores, then ONE lane publishes the flag.  (Tiles of
             * the highest colour drain too although nobody waits for their flag: the deferred contributions are
             * added by the last workgroup of this launch, after every tile's stores.) */
            if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T = wall_clock64(); atomicAdd(&a.st->dbg[8 + 4 * colour + 2], T - TF); TF = T; }
            if (pass == 0) GSDF_TRACE(a, tr, 8);                      /* wave 0: stores issued */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && pass + 1 == n_pass) __hip_atomic_store(my_flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T = wall_clock64(); atomicAdd(&a.st->dbg[8 + 4 * colour + 3], T - TF); TF = T; }
            if (pass == 0) GSDF_TRACE(a, tr, 9);                      /* all stores drained, flag published */
        } else {
            /* near tile (or timed-out wait): everything goes through the deferred list */
            unsigned int my_defer = 0u;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (rec[e] == NOREC) continue;
                L.key[tid + FUSE_THREADS * e] = FUSE_LKEY_DEFER | rec[e];
                vis_mark(a, tab.vox + rec[e], frame_cur);
                ++my_defer;
            }
            if (my_defer) atomicAdd(&L.n_defer, my_defer);
            __syncthreads();
            if (L.n_defer) {
                if (tid == 0) { L.defer_base = atomicAdd(a.deferred_count, L.n_defer); L.any_defer = 1u; }   /* one global atomic per workgroup */
                __syncthreads();
                if (tid == 0) L.n_defer = 0u;
                __syncthreads();
                for (int i = tid; i < FUSE_LCAP; i += FUSE_THREADS) {
                    const uint32_t key = L.key[i];
                    if (key == FUSE_LKEY_EMPTY) continue;
                    const unsigned int o = L.defer_base + atomicAdd(&L.n_defer, 1u);
                    if (o >= a.deferred_cap) { atomicOr(&a.st->status, GSDF_STATUS_TABLE_FULL); continue; }
                    const fuse_sums u = fuse_unpack(L.acc[0][i], L.acc[1][i], L.acc[2][i]);
                    gsdf_deferred d;
                    d.p = tab.vox + (key & ~FUSE_LKEY_DEFER);
                    d.w = u.w; d.s = u.s;
                    d.gx = u.gx; d.gy = u.gy; d.gz = u.gz;
                    d.pad = 0u;
                    a.deferred[o] = d;
                }
            }
            /* a timed-out tile still has to release the tiles that wait for it (it wrote nothing itself) */
            if (tid == 0 && pass + 1 == n_pass) __hip_atomic_store(my_flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (tid == 0) {
        __hip_atomic_store(my_flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (GSDF_EXPERIMENT(a.debug, 128) && tid == 0) { const unsigned long long T3 = wall_clock64(); atomicAdd(&a.st->dbg[3], T3 - T1); T1 = T3; }
    if (pass + 1 < n_pass) {                                          /* next band: start from an empty table */
        __syncthreads();
        fuse_lds_clear(L, tid);
        if (tid == 0) L.n_defer = 0u;
        __syncthreads();
    }
    }   /* pass */
    GSDF_TRACE(a, tr, 10);                                            /* all bands done */
    if (GSDF_EXPERIMENT(a.debug, 64) && tid == 0) tr[13] = (unsigned long long)n_pass;
    /* per-workgroup counters (plain stores into this workgroup's own row: no hot atomics) */
    if (lane == 0) { L.cnt[0][wave] = n_upd_w; L.cnt[1][wave] = n_val_w; }
    /* every wave drains what it stored -- deferred-list entries among it -- before the barrier behind which thread 0 releases
     * them to the last workgroup of the launch (the record stores were drained above already; normally nothing is pending) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned long long nu = 0ull, nv = 0ull;
#pragma unroll
        for (int i = 0; i < FUSE_NWAVES; ++i) { nu += L.cnt[0][i]; nv += L.cnt[1][i]; }
        unsigned long long* c = a.blk_counters + 4 * ((size_t)tile_y * a.ntx + tile_x);
        c[0] = nu; c[1] = nv;
        if (nu | nv) { c[2] += nu; c[3] += nv; }                      /* (an empty tile does not wait for its totals to arrive) */
        if (tile_x == 0 && tile_y == 0) {                             /* :120 increase_counter() */
            /* read by the launch's LAST workgroup below, which may sit on another XCD: a plain store would wait in this XCD's L2
             * until the kernel ends (the per-XCD L2s are not coherent for plain accesses) and the snapshot for the next update()
             * would, rarely, be the OLD counter -- vis_ bits of the next frame one position too low (seen once in 30 runs of the
             * eight-process exchange test).  An agent-scope atomic, released before this workgroup takes its ticket. */
            __hip_atomic_fetch_add(&a.st->frames, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        /* Arrival.  The deferred contributions (near tiles, LDS overflow, timed-out waits: normally a handful) are added
         * by whichever workgroup finishes last, so the common case needs no second launch.  A workgroup that appended to
         * the list releases its entries first (cdna_hip_programming.md G16: stores -> barrier -> one agent-scope release
         * -> drained -> atomic); every workgroup has drained its record stores above. */
        if (L.any_defer) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        L.is_last = atomicAdd(a.ticket, 1u) + 1u == (NEXT_NORMALS ? (unsigned int)a.n_tiles : gridDim.x) ? 1u : 0u;
        if (L.is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!L.is_last) return;
    unsigned int n = __hip_atomic_load(a.deferred_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    n = n < a.deferred_cap ? n : a.deferred_cap;
    const bool mine = !(a.resolve_follows && n > FUSE_RESOLVE_INLINE);
    if (mine)
        for (unsigned int i = tid; i < n; i += FUSE_THREADS) {
            const gsdf_deferred d = a.deferred[i];
            unsafeAtomicAdd(&d.p->w, d.w);
            unsafeAtomicAdd(&d.p->s, d.s);
            unsafeAtomicAdd(&d.p->gx, d.gx);
            unsafeAtomicAdd(&d.p->gy, d.gy);
            unsafeAtomicAdd(&d.p->gz, d.gz);
        }
    if (tid == 0) {
        if (mine) { a.st->n_deferred += n; __hip_atomic_store(a.deferred_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        a.st->last_deferred = n;
        /* the host decides by it, without waiting, whether the next launches get a k_fuse_resolve behind them */
        if (a.host_note) {
            __hip_atomic_store(a.host_note, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_note + 1, __hip_atomic_load(&a.st->far_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&a.st->far_tiles, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        /* Sdf::counter_ as the NEXT update() will see it (this launch's increment included: tile (0, 0) arrived before this
         * workgroup read the ticket).  The normals stage used to take this snapshot; in GT-pose mode it now runs on its own
         * stream beside this kernel and must not touch state the fusion uses. */
        a.st->frame_cur = __hip_atomic_load(&a.st->frames, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.use_dev_pose) fuse_log_row(a);
    }
}

/* Adds a LONG deferred list after k_fuse (scenes with many near tiles: every contribution of such a tile is deferred).
 * Queued only while the previous launches reported long lists (gsdf_dev_state::last_deferred, read by the host without
 * waiting); short lists are added by the last workgroup of k_fuse itself, which leaves count = 0 here. */
__global__ __launch_bounds__(256) void k_fuse_resolve(const gsdf_deferred* list, unsigned int* count, unsigned int cap,
                                                       const gsdf_dev_state* gate, gsdf_dev_state* st, unsigned int* ticket) {
    if (gate && !(gate->done && gate->converged)) return;
    unsigned int n = *count;
    n = n < cap ? n : cap;
    if (n == 0u) return;
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const gsdf_deferred d = list[i];
        unsafeAtomicAdd(&d.p->w, d.w);
        unsafeAtomicAdd(&d.p->s, d.s);
        unsafeAtomicAdd(&d.p->gx, d.gx);
        unsafeAtomicAdd(&d.p->gy, d.gy);
        unsafeAtomicAdd(&d.p->gz, d.gz);
    }
    /* the last block to finish counts the list and empties it for the next fusion (every block has read `count` by then) */
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket, 1u) + 1u == gridDim.x) {
        st->n_deferred += n;
        __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

void gsdf_launch_fuse(hipStream_t s, const gsdf_frame_geom& g, const gsdf_ncache& nc, const float* depth,
                      const float* nx, const float* ny, const float* nz, const gsdf_pose_arg& pose,
                      int use_dev_pose, gsdf_table tab, gsdf_dev_state* st, unsigned long long* blk_counters,
                      gsdf_deferred* deferred, unsigned int* deferred_count, unsigned int deferred_cap,
                      unsigned int tag, unsigned int* tile_flags, const uint32_t* tile_order, float* log_rows,
                      long long max_rows, uint32_t* vis, int vis_words, int debug, unsigned int* ticket, int resolve_follows,
                      unsigned int* host_note, int far_table, const gsdf_fuse_head* head, const float* next_depth, float* next_nx,
                      float* next_ny, float* next_nz, int win, const uint32_t* tile_stats, uint32_t* next_tile_stats,
                      unsigned int next_token) {
    fuse_args a;
    a.tile_stats = tile_stats; a.nrm_stats = next_tile_stats;
    a.host_note = host_note;
    a.ticket = ticket; a.log_rows = use_dev_pose ? log_rows : nullptr; a.max_rows = max_rows; a.resolve_follows = resolve_follows;
    a.vis = vis; a.vis_words = vis_words;
    a.debug = debug;
    a.g = g; a.nc = nc; a.depth = depth; a.nx = nx; a.ny = ny; a.nz = nz; a.pose = pose;
    a.use_dev_pose = use_dev_pose; a.tab = tab; a.st = st; a.blk_counters = blk_counters;
    a.deferred = deferred; a.deferred_count = deferred_count; a.deferred_cap = deferred_cap; a.tag = tag;
    const int ntx = (g.W + FUSE_T - 1) / FUSE_T, nty = (g.H + FUSE_TH - 1) / FUSE_TH;
    gsdf_dev_state* gate = use_dev_pose ? st : nullptr;
    a.tile_flags = tile_flags; a.ntx = ntx; a.nty = nty; a.tile_order = tile_order;
    const int n = ntx * nty;
    a.n_tiles = n;
    a.nrm_depth = next_depth; a.nrm_x = next_nx; a.nrm_y = next_ny; a.nrm_z = next_nz;
    a.nrm_r = win / 2; a.nrm_ntx = (g.W + NRM_TX - 1) / NRM_TX;
    int extra = next_depth ? a.nrm_ntx * ((g.H + NRM_TY - 1) / NRM_TY) : 0;
    if (extra && !FUSE_CARRIES_NORMALS) {
        gsdf_launch_normals(s, g, win, nc, next_depth, next_nx, next_ny, next_nz, nullptr, nullptr, next_tile_stats);
        extra = 0;
    }
    std::memset(&a.hd, 0, sizeof(a.hd));
    a.nrm_token = next_token;
    if (head && use_dev_pose && head->k > 0) {
        a.hd = *head;
        if (extra) {
            if (far_table) hipLaunchKernelGGL((k_fuse<FUSE_LCAP_FAR, true, true>), dim3(n + extra), dim3(FUSE_THREADS), 0, s, a);
            else hipLaunchKernelGGL((k_fuse<FUSE_LCAP_NEAR, true, true>), dim3(n + extra), dim3(FUSE_THREADS), 0, s, a);
        } else if (far_table) hipLaunchKernelGGL((k_fuse<FUSE_LCAP_FAR, false, true>), dim3(n), dim3(FUSE_THREADS), 0, s, a);
        else hipLaunchKernelGGL((k_fuse<FUSE_LCAP_NEAR, false, true>), dim3(n), dim3(FUSE_THREADS), 0, s, a);
    } else if (extra) {
        if (far_table) hipLaunchKernelGGL((k_fuse<FUSE_LCAP_FAR, true, false>), dim3(n + extra), dim3(FUSE_THREADS), 0, s, a);
        else hipLaunchKernelGGL((k_fuse<FUSE_LCAP_NEAR, true, false>), dim3(n + extra), dim3(FUSE_THREADS), 0, s, a);
    } else if (far_table) hipLaunchKernelGGL((k_fuse<FUSE_LCAP_FAR, false, false>), dim3(n), dim3(FUSE_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_fuse<FUSE_LCAP_NEAR, false, false>), dim3(n), dim3(FUSE_THREADS), GSDF_EXPERIMENT(debug, 4096) ? 81920 : 0, s, a);   /* experiment: 1 workgroup per CU */
    if (resolve_follows)
        hipLaunchKernelGGL(k_fuse_resolve, dim3(512), dim3(256), 0, s, deferred, deferred_count, deferred_cap, gate, st, ticket + 1);
}
int gsdf_fuse_grid_blocks(int W, int H) { return ((W + FUSE_T - 1) / FUSE_T) * ((H + FUSE_TH - 1) / FUSE_TH); }
/* Launch order of the fusion tiles.  Colour-major (colour = parity of tile x, y): a tile only ever waits for tiles
 * of lower colour, which were dispatched before it.  Within that, workgroup b -- which the hardware places on XCD
 * b % 8 (observed; a performance assumption only) -- takes its tile from vertical image stripe b % 8, so the tiles
 * that share voxel records and block keys meet in one XCD's L2. */
void gsdf_fuse_tile_order(int W, int H, uint32_t* order) {
    const int ntx = (W + FUSE_T - 1) / FUSE_T, nty = (H + FUSE_TH - 1) / FUSE_TH;
    std::vector<uint32_t> lists[4][8];
    for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx)
            lists[(tx & 1) + 2 * (ty & 1)][(int)((long long)tx * 8 / ntx)].push_back((uint32_t)tx | ((uint32_t)ty << 16));
    size_t next[4][8] = {};
    int b = 0;
    for (int c = 0; c < 4; ++c) {
        size_t left = 0;
        for (int r = 0; r < 8; ++r) left += lists[c][r].size();
        for (; left > 0; --left, ++b) {
            int r = b % 8;
            if (next[c][r] >= lists[c][r].size()) {           /* stripe exhausted: take from the fullest one */
                size_t best = 0;
                for (int q = 0; q < 8; ++q) {
                    const size_t rem = lists[c][q].size() - next[c][q];
                    if (rem > best) { best = rem; r = q; }
                }
            }
            order[b] = lists[c][r][next[c][r]++];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * RigidPointOptimizer::optimize_sampled -- one Gauss-Newton pass per launch.
 *
 * Every lane gathers the voxel under its back-projected pixel (one 32-byte slot), forms the
 * residual phi and the 6-vector J and accumulates the 29 normal-equation sums in registers.
 * Sums are reduced with wavefront DPP shuffles, then across the 4 waves through LDS, and each
 * workgroup stores one partial row.  The NEXT launch starts with a "head" in which every workgroup
 * adds those rows in the same fixed order, solves the 6x6 system and applies SE3::exp(-xi)
 * (bit-identical everywhere), so a launch contains no inter-workgroup synchronisation at all: the
 * kernel boundary is the only barrier, rows and tracker state are double-buffered by pass parity,
 * and workgroup 0 publishes pose / done / converged for the host and the fusion kernels.
 * ---------------------------------------------------------------------------------------------- */
/* RigidOptimizer with num_iterations_ <= 0: optimize() returns false without touching the pose */
__global__ void k_track_none(gsdf_dev_state* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->done = 1; st->converged = 0; st->passes = 0; st->last_hits = 0.f; }
}
void gsdf_launch_track_none(hipStream_t s, gsdf_dev_state* st) { hipLaunchKernelGGL(k_track_none, dim3(1), dim3(64), 0, s, st); }

#ifndef TRK_PPT
#define TRK_PPT 3          /* pixels per lane handled as one batch: independent gathers in flight */
#endif
#ifndef TRK_CHUNK          /* pixels of one workgroup's batch in k_track_pass: half of its waves TRK_PPT per lane, the others one less */
#define TRK_CHUNK ((GSDF_TRACK_BLOCK / 128) * 64 * (2 * TRK_PPT - 1))
#endif

/* One Gauss-Newton step from the 29 sums (RigidPointOptimizer.cpp:86-98): solve, test, apply.  `passes` counts
 * this pass.  Identical arithmetic wherever it runs (every workgroup computes it redundantly).
 *
 * The head's solve is executed by ONE wave whose 64 lanes all do the same scalar work, at the top of every launch with a
 * cold instruction cache: through round 4 it reproduced gsdf_llt_solve6 / gsdf_se3_exp_mul bit for bit (6 correctly
 * rounded square roots, 27 divisions, four full-range sinf / cosf: ~1 200 dependent instructions, 2.3-2.9 us of a 9.5 us
 * pass) -- exactness that bought no parity, because its INPUTS already differ from the oracle's in their last bits
 * (pairwise float / f64 group sums here, one sequential float sum there).  The common case is now the same algorithm in
 * <= 1-ulp forms (trk_llt_solve6_fast, trk_se3_exp_mul_fast below): hardware reciprocal / reciprocal square root with one
 * Newton step through FMA, FMA dot products, and sin / cos as short polynomials for rotation steps below 0.1 rad.  The
 * exact forms stay for what the fast ones do not cover: a non-positive pivot (Eigen stops the factorisation there and
 * solves on what it has), theta^2 below Sophus' epsilon (its series branch) and rotation steps of 0.1 rad or more. */
__device__ __forceinline__ float trk_lane_value(float v, int lane_const) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const));
}
/* 1 / x and 1 / sqrt(x): v_rcp_f32 / v_rsq_f32 (1 ulp) + one Newton step in FMA arithmetic => below 1 ulp */
__device__ __forceinline__ float trk_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.f), r, r);
}
__device__ __forceinline__ float trk_rsq(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-(x * r), 0.5f * r, 0.5f);      /* (1 - x r^2) / 2 */
    return __builtin_fmaf(r, e, r);
}
/* H.llt().solve(g) (RigidPointOptimizer.cpp:86) in the structure of Eigen's unblocked llt_inplace and unrolled triangular
 * solves -- pivot = A(k,k) - (sum of squares), column = (A21 - A20 * A10^T) / pivot root, rhs(i) = (rhs(i) - dot) / diagonal
 * -- with the reciprocal root of the pivot kept in the diagonal, so that every division is a multiplication.
 * Returns false (x untouched) when a pivot is not positive: the caller takes the exact path then. */
__device__ __forceinline__ bool trk_llt_solve6_fast(const float* Hm, const float* g, float* x) {
    float L[36];                                               /* lower triangle; L(k,k) holds 1 / sqrt(pivot) */
    bool positive = true;                                      /* every pivot so far was > 0 (or NaN) */
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < k; ++j) sq = __builtin_fmaf(L[6 * k + j], L[6 * k + j], sq);
        const float d = Hm[6 * k + k] - sq;
        positive = positive && !(d <= 0.f);
        const float ri = trk_rsq(d);
        L[6 * k + k] = ri;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            float c = 0.f;
#pragma unroll
            for (int j = 0; j < k; ++j) c = __builtin_fmaf(L[6 * i + j], L[6 * k + j], c);
            L[6 * i + k] = (Hm[6 * i + k] - c) * ri;
        }
    }
    if (!positive) return false;
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < i; ++j) c = __builtin_fmaf(L[6 * i + j], y[j], c);
        y[i] = (g[i] - c) * L[6 * i + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float c = 0.f;
#pragma unroll
        for (int j = i + 1; j < 6; ++j) c = __builtin_fmaf(L[6 * j + i], x[j], c);
        x[i] = (y[i] - c) * L[6 * i + i];
    }
    return true;
}
/* The exact forms, spread over lanes WITHOUT changing a single operation or its order (bit-identical to gsdf_llt_solve6 /
 * gsdf_se3_exp_mul in every lane).  Rare paths since round 5. */
__device__ __forceinline__ void trk_se3_exp_mul_wave(const float* xi, float* pose7) {
    float trig[4] = { 0.f, 1.f, 0.f, 1.f };
    const float theta_sq = gsdf_se3_theta_sq(xi);
    if (!(theta_sq < GSDF_SOPHUS_EPS * GSDF_SOPHUS_EPS)) {
        const float theta = sqrtf(theta_sq), half = 0.5f * theta;
        const float arg = (threadIdx.x & 1u) ? theta : half;   /* sinf / cosf of theta / 2 and of theta evaluated once: odd lanes take theta */
        const float sn = sinf(arg), cs = cosf(arg);
        trig[0] = trk_lane_value(sn, 0); trig[1] = trk_lane_value(cs, 0);
        trig[2] = trk_lane_value(sn, 1); trig[3] = trk_lane_value(cs, 1);
    }
    gsdf_se3_exp_mul_trig(xi, pose7, trig);
}
/* sin / cos for |x| < 0.1: Taylor polynomials whose first omitted terms are below 3e-14 / 3e-13 relative, in FMA Horner form */
__device__ __forceinline__ float trk_sin_small(float x, float x2) {
    float p = __builtin_fmaf(x2, -1.f / 5040.f, 1.f / 120.f);
    p = __builtin_fmaf(x2, p, -1.f / 6.f);
    return __builtin_fmaf(x, x2 * p, x);
}
__device__ __forceinline__ float trk_cos_small(float x2) {
    float p = __builtin_fmaf(x2, -1.f / 720.f, 1.f / 24.f);
    p = __builtin_fmaf(x2, p, -0.5f);
    return __builtin_fmaf(x2, p, 1.f);
}
/* pose7 = SE3::exp(xi) * pose7 with Sophus' formulas as written (gsdf_se3_exp_mul_trig) in <= 1-ulp forms, for
 * eps^2 <= theta^2 < 0.01; everything else goes through the exact path. */
__device__ __forceinline__ void trk_se3_exp_mul_fast(const float* xi, float* pose7) {
    const float theta_sq = gsdf_se3_theta_sq(xi);
    if (__builtin_expect(!(theta_sq >= GSDF_SOPHUS_EPS * GSDF_SOPHUS_EPS && theta_sq < 0.01f), 0)) {
        trk_se3_exp_mul_wave(xi, pose7);
        return;
    }
    const float rth = trk_rsq(theta_sq);                       /* 1 / theta */
    const float theta = theta_sq * rth, half = 0.5f * theta;
    const float tsq = theta * theta, hsq = half * half;
    const float rth2 = rth * rth;
    const float imag = trk_sin_small(half, hsq) * rth;         /* sin(theta / 2) / theta */
    const float real = trk_cos_small(hsq);
    const float a = (1.f - trk_cos_small(tsq)) * rth2;         /* (1 - cos theta) / theta^2 */
    const float b = (theta - trk_sin_small(theta, tsq)) * (rth2 * rth);   /* (theta - sin theta) / theta^3 */
    float qn[4];
    gsdf_se3_exp_mul_parts(xi, pose7, imag, real, false, a, b, qn);
    const float rl = trk_rsq(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) pose7[3 + i] = qn[i] * rl;
}

/* called by a full wave (all 64 lanes active, same arguments in every lane) */
__device__ __forceinline__ void trk_solve_update(const float* tot, float damping, float conv_sq, int passes, int max_passes,
                                                 int no_solve, bool exact_solve /* test build: the round-4 arithmetic */,
                                                 float pose[7], int* done, int* converged) {
    float gvec[6], Hm[36];
#pragma unroll
    for (int i = 0; i < 6; ++i) gvec[i] = tot[1 + i];
    int q = 7;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { Hm[6 * i + j] = tot[q]; Hm[6 * j + i] = tot[q]; ++q; }
    float xi[6];
    if (no_solve) { for (int i = 0; i < 6; ++i) xi[i] = 1.f; }            /* experiment switch */
    else if (exact_solve || __builtin_expect(!trk_llt_solve6_fast(Hm, gvec, xi), 0))
        gsdf_llt_solve6(Hm, gvec, xi);                                    /* RigidPointOptimizer.cpp:86 */
#pragma unroll
    for (int i = 0; i < 6; ++i) xi[i] = damping * xi[i];
    const float nrm = gsdf_sum3(xi[0] * xi[0], xi[1] * xi[1], xi[2] * xi[2]) +
                      gsdf_sum3(xi[3] * xi[3], xi[4] * xi[4], xi[5] * xi[5]);
    *done = 0; *converged = 0;
    if (nrm < conv_sq) {                                                  /* :88-91 (xi is NOT applied) */
        *converged = 1;
        *done = 1;
    } else {
        bool nan = false;
#pragma unroll
        for (int i = 0; i < 6; ++i) nan = nan || isnan(xi[i]);
        if (!nan && !no_solve) {                                          /* :94-95 */
            float mxi[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) mxi[i] = -xi[i];
            if (exact_solve) trk_se3_exp_mul_wave(mxi, pose);
            else trk_se3_exp_mul_fast(mxi, pose);
        }
        if (passes >= max_passes) *done = 1;                              /* :98 return false */
    }
}

/* Gather + normal-equation sums of one pass for this lane's pixels: back-project, voxel lookup (block key from the
 * L2-resident key array, then the 32-byte record), residual, Jacobian.  The lane's pixels of one batch are pix0 + j * pix_stride
 * (j < PPT: independent gathers in flight), the next batch follows batch_stride pixels further.
 * z_first (nullable): depth of the first batch, already in registers. */
template <int PPT>
__device__ __forceinline__ void trk_gather(const gsdf_frame_geom& g, const gsdf_table& tab, const float* __restrict__ depth,
                                           const float* z_first, const float pose[7], int pix0, int pix_stride, int batch_stride,
                                           float (&acc)[GSDF_TRACK_NSUM], unsigned long long* wave_stamp = nullptr,
                                           const int xy_scale = 1) {
    /* wave_stamp (test build, tools/track_waves.py): one word per wave = gather ticks | ticks until the block lookups are done |
     * pixels that passed the z gate | pixels with a voxel, 16 bits each (first batch of pixels) */
    const unsigned long long ws_t0 = wave_stamp ? wall_clock64() : 0ull;
    float R[9];
    gsdf_quat_to_R(pose + 3, R);                                          /* RigidPointOptimizer.cpp:53-54 */
    const float t[3] = { pose[0], pose[1], pose[2] };
    const float fx_inv = 1.f / g.fx, fy_inv = 1.f / g.fy;                 /* :46-47 */
    const int N = g.W * g.H;
    const uint32_t uW = (uint32_t)g.W;
    const uint32_t step_y = (uint32_t)pix_stride / uW, step_x = (uint32_t)pix_stride - step_y * uW;
    for (int base = pix0; base < N; base += batch_stride) {
        /* stage A: depth */
        float z[PPT];
        bool ok[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int pix = base + j * pix_stride;
            ok[j] = pix < N;
            z[j] = (z_first && base == pix0) ? z_first[j] : (ok[j] ? depth[pix] : 0.f);
        }
        /* stage B: back-project, voxel key, block key at the home entry (L2-resident key array) */
        gsdf_v3 p[PPT];
        int vx[PPT], vy[PPT], vz[PPT];
        unsigned long long key[PPT], bkey[PPT], k0[PPT];
        uint32_t home[PPT];
        /* (x, y) of the lane's pixels: one division for the first, the others follow by the (uniform) stride */
        uint32_t px, py = (uint32_t)base / uW;
        px = (uint32_t)base - py * uW;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            ok[j] = ok[j] && !(z[j] <= g.zmin || z[j] >= g.zmax);         /* :64-65 */
            /* optimize_sampled's `y += sampling`, `x += sampling` (RigidPointOptimizer.cpp:62): g.W x g.H is then the grid of
             * sampled pixels, `depth` their compacted image, and the pixel coordinate the grid index times the stride
             * (xy_scale is the literal 1 in the unsampled instantiation: folded away) */
            const int y = (int)py * xy_scale, x = (int)px * xy_scale;
            px += step_x; py += step_y;
            if (px >= uW) { px -= uW; ++py; }
            const float x0 = ((float)x - g.cx) * fx_inv;                  /* :67-68 */
            const float y0 = ((float)y - g.cy) * fy_inv;
            const gsdf_v3 pc = { x0 * z[j], y0 * z[j], z[j] };
            const gsdf_v3 Rp = gsdf_matvec(R, pc);
            p[j] = gsdf_v3{ Rp.x + t[0], Rp.y + t[1], Rp.z + t[2] };      /* :70 */
            vx[j] = gsdf_float2vox1(g.inv_vs, p[j].x);
            vy[j] = gsdf_float2vox1(g.inv_vs, p[j].y);
            vz[j] = gsdf_float2vox1(g.inv_vs, p[j].z);
            ok[j] = ok[j] && gsdf_key_in_range(vx[j], vy[j], vz[j]);
            key[j] = gsdf_key_pack(vx[j], vy[j], vz[j]);
            bkey[j] = gsdf_block_key(key[j]);
            home[j] = gsdf_hash(bkey[j]) & tab.block_mask;
            k0[j] = ok[j] ? tab.bkeys[home[j]] : GSDF_KEY_EMPTY;
        }
        /* stage C: the voxel record (neighbouring pixels share lines: 4 x-adjacent voxels per line) */
        const gsdf_payload* P[PPT];
        float2 pa[PPT], pb[PPT], pc2[PPT];
        int blk[PPT];
        {
            uint32_t want = 0u;
#pragma unroll
            for (int j = 0; j < PPT; ++j) want |= ok[j] ? 1u << j : 0u;
            gsdf_block_lookup_n<PPT, false>(tab, bkey, home, k0, want, blk);   /* the pixels' probe chains overlap */
        }
        unsigned long long ws_lookup = 0ull, ws_ok = 0ull;
        if (wave_stamp && base == pix0) {
            ws_lookup = wall_clock64() - ws_t0;
#pragma unroll
            for (int j = 0; j < PPT; ++j) ws_ok += (unsigned long long)__popcll(__ballot(ok[j]));
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            P[j] = blk[j] >= 0 ? tab.vox + ((size_t)blk[j] * GSDF_BLOCK_VOX + gsdf_block_local(key[j])) : nullptr;
            if (P[j]) {
                const float2* q = reinterpret_cast<const float2*>(P[j]);
                pa[j] = q[0]; pb[j] = q[1]; pc2[j] = q[2];
            } else {
                pa[j] = make_float2(0.f, 0.f); pb[j] = pa[j]; pc2[j] = pa[j];
            }
        }
        /* stage D: residual, Jacobian, normal-equation sums */
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float w0 = pa[j].x;                                     /* weights(): MapGradPixelSdf.h:117-125 */
            if (!(w0 > 0.f)) continue;                                    /* :73 */
            /* tsdf(): MapGradPixelSdf.h:109-115 */
            const gsdf_v3 gn = gsdf_normalized3(gsdf_v3{ pb[j].x, pb[j].y, pc2[j].x });
            const gsdf_v3 gr = { 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };
            const gsdf_v3 d = { g.vs * (float)vx[j] - p[j].x, g.vs * (float)vy[j] - p[j].y, g.vs * (float)vz[j] - p[j].z };
            const float phi = gsdf_tsdf_phi(pa[j].y / w0, gn, d);         /* :114 (double literal, see gsdf_math.h) */
            const gsdf_v3 pxg = gsdf_cross3(p[j], gr);                    /* :78 */
            const float J[6] = { gr.x, gr.y, gr.z, pxg.x, pxg.y, pxg.z };
            acc[0] += phi * phi;                                          /* :76 */
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[1 + i] += phi * J[i];         /* :79 */
            int q = 7;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int jj = i; jj < 6; ++jj) acc[q++] += J[i] * J[jj];  /* :80 */
            acc[28] += 1.f;                                               /* :81 */
        }
        if (wave_stamp && base == pix0) {
            unsigned long long hits = 0ull;
#pragma unroll
            for (int j = 0; j < PPT; ++j) hits += (unsigned long long)__popcll(__ballot(pa[j].x > 0.f));
            const unsigned long long tot = wall_clock64() - ws_t0;
            if ((threadIdx.x & 63) == 0)
                *wave_stamp = (tot & 0xFFFFull) | ((ws_lookup & 0xFFFFull) << 16) | ((ws_ok & 0xFFFFull) << 32) | ((hits & 0xFFFFull) << 48);
        }
    }
}

static_assert(GSDF_TRACK_BLOCK == NRM_THREADS, "the normals tiles of the first pass run in tracker-sized workgroups");
/* SAMPLED: optimize_sampled(depth, K, sampling > 1) -- the public stride argument of RigidPointOptimizer.h:65.  g.W x g.H is
 * the grid of sampled pixels (ceil(W / s) x ceil(H / s)), `depth` its compacted image (k_subsample), tp.sampling the stride.
 * A second instantiation so that the sampling-1 kernel of the frame loop stays instruction for instruction what it was. */
template <bool SAMPLED>
__global__ __launch_bounds__(GSDF_TRACK_BLOCK, 4) void k_track_pass(gsdf_frame_geom g, const float* __restrict__ depth,
                                                                 gsdf_table tab, gsdf_dev_state* st,
                                                                 double* rows, gsdf_track_params tp, gsdf_normals_job nj) {
    /* Workgroups beyond the tracker's own (first pass of a frame in the Scan3D loop only): one normals tile each
     * (NormalEstimator::compute of THIS depth frame for the update() that follows a converged optimize(),
     * main_scan_3d.cpp:258-263).  They need the depth only, fill the chip beside the latency-bound pass, and use
     * dynamic LDS so that the later passes of the frame stay small. */
    if ((int)blockIdx.x >= tp.n_track_blocks) {
        extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
        /* (gsdf_hint_next_depth_dev: the previous frame's fusion launch computed these normals in its tail if it ran -- then it
         * left this frame's token) */
        if (nj.token && st->nrm_token == nj.token) return;
        const int t = (int)blockIdx.x - tp.n_track_blocks + nj.tile_first;
        if (t == 0 && threadIdx.x == 0) {
            *nj.deferred_count = 0u;                            /* fresh list for the k_fuse of this frame */
            st->frame_cur = st->frames;                         /* counter_ seen by every workgroup of k_fuse */
        }
        normals_tile(*reinterpret_cast<nrm_lds*>(dyn_lds), t % nj.ntx, t / nj.ntx, g.W, g.H, nj.r, nj.nc, depth, nj.nx, nj.ny, nj.nz,
                     gsdf_tile_stats{ nj.stats, (g.W + 15) / 16, g.zmin, g.zmax });
        return;
    }
    /* test build, debug bit 64 of the tracker flags: time stamps of thread 0 in rows 2048 + pass * 512 + workgroup of the
     * trace buffer (rows 0..2047 belong to k_fuse), tools/track_trace.py */
    unsigned long long* trk_tr = nullptr;
    if (GSDF_EXPERIMENT(tp.debug, 64) && blockIdx.x < 512u && tp.pass_index < 12) {
        trk_tr = reinterpret_cast<unsigned long long*>(st->dbg[23]) + 16 * (2048 + (size_t)tp.pass_index * 512 + blockIdx.x);
        if (threadIdx.x == 0) trk_tr[0] = wall_clock64();
    }
    __shared__ float wsum[GSDF_TRACK_BLOCK / 64][32];
    __shared__ float sh_pose[8];
    __shared__ int sh_done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = tp.pass_index;
    float pose[7];
    /* Partial sums of a pass: GSDF_TRACK_GROUPS groups x 32 doubles (29 used); workgroup b adds its sums to group
     * b % GROUPS with f64 atomics, so the next launch's head reads 4 KB instead of one 128-byte row per workgroup (32 KB at 256
     * workgroups -- that re-reduction was ~25 % of a pass).  Three buffers rotate with the launch number (which
     * runs on across optimize() calls, tp.rot): launch j accumulates into buffer j % 3, reads j - 1 and clears
     * j + 1, which nothing has touched since launch j - 2 read it.  Every launch does the clearing first, also
     * the ones that return early. */
    /* the depth of this lane's (first) pixels does not depend on the pose: requested now, it arrives under the head */
    /* Pixels -> lanes.  A workgroup takes chunks of TRK_CHUNK = 1280 consecutive pixels (chunk b, b + workgroups, ...): waves
     * 0-3 three rows of 64 per lane-set (3 pixels per lane), waves 4-7 two.  Wave w and wave w + 4 share a SIMD, and the gather
     * is paced by the SIMD's instruction issue: 3 + 2 pixels on every SIMD of every CU, where 3 pixels per lane in all eight waves
     * left a third of the workgroups with 6 per SIMD and the others with 4 (307 200 pixels are 2.34 per lane). */
    const bool trk_heavy = wave < GSDF_TRACK_BLOCK / 128;
    const int trk_pix0 = (int)blockIdx.x * TRK_CHUNK + (trk_heavy ? wave * (64 * TRK_PPT) : (GSDF_TRACK_BLOCK / 128) * 64 * TRK_PPT + (wave - GSDF_TRACK_BLOCK / 128) * (64 * (TRK_PPT - 1))) + lane;
    const int trk_batch = tp.n_track_blocks * TRK_CHUNK;
    float z_pre[TRK_PPT];
    {
        const int N = g.W * g.H;
#pragma unroll
        for (int j = 0; j < TRK_PPT; ++j) {
            const int pix = trk_pix0 + j * 64;
            z_pre[j] = (pix < N && (trk_heavy || j < TRK_PPT - 1)) ? depth[pix] : 0.f;
        }
    }
    double* acc_cur = rows + (size_t)(tp.rot % 3u) * GSDF_TRACK_ROWSET;
    const double* acc_prev = rows + (size_t)((tp.rot + 2u) % 3u) * GSDF_TRACK_ROWSET;
    if (blockIdx.x == 0) {
        double* nxt = rows + (size_t)((tp.rot + 1u) % 3u) * GSDF_TRACK_ROWSET;
        for (int i = threadIdx.x; i < GSDF_TRACK_ROWSET; i += GSDF_TRACK_BLOCK) nxt[i] = 0.0;
    }

    if (k > 0 && tp.head_done) {
        /* the head of this launch was performed by the fusion launch in front of it (k_fuse<.., HEAD>, which found that optimize()
         * had not ended): its result is in st->trk[k & 1], written in an earlier kernel */
        const gsdf_trk_buf& cur = st->trk[k & 1];
        if (cur.done) return;
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = cur.pose7[i];
    } else if (k == 0) {
        /* a new optimize(): the pose is RigidOptimizer::pose_ (kept in st->pose7 between frames) */
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = st->pose7[i];
        if (blockIdx.x == 0 && tid == 0) {
            gsdf_trk_buf& o = st->trk[0];
#pragma unroll
            for (int i = 0; i < 7; ++i) o.pose7[i] = pose[i];
            o.done = 0; o.converged = 0; o.passes = 0;
            st->done = 0; st->converged = 0;                               /* what the gated fusion launches of this frame read */
        }
    } else {
        /* ---- head: every workgroup reduces the rows of pass k-1 in the same fixed order, solves the
         * 6x6 system and applies the update -- bit-identical everywhere, so no workgroup has to wait for
         * another one inside a launch; workgroup 0 publishes the result for the next launch / the host ----
         * Wave 0 does it alone: lane v < 29 adds the GSDF_TRACK_GROUPS group sums of value v in increasing order (all
         * loads in flight at once, issued before the state words are looked at: a launch that finds `done` set wasted
         * them, and is rare), `v_readlane` hands the 29 totals to every lane, the solve runs without an LDS stage; the
         * other waves meet it at ONE barrier (two LDS stages behind three barriers before). */
        const gsdf_trk_buf& in = st->trk[(k - 1) & 1];
        if (GSDF_EXPERIMENT(tp.debug, 8) && wave != 0) {
            /* experiment (test build, tracker debug bit 8; tools/track_waves.py warm): while wave 0 solves, the other waves run the
             * gather of their pixels with the OLD pose and throw the sums away -- a pass moves the pose by a fraction of a voxel, so
             * the key-array entries and most record lines of the real gather are then in this XCD's L2 (every launch starts with the
             * L2 invalidated: FETCH ~ algorithmic bytes).  Does the real gather get shorter? */
            float pose_old[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) pose_old[i] = in.pose7[i];
            float dummy[GSDF_TRACK_NSUM];
#pragma unroll
            for (int i = 0; i < GSDF_TRACK_NSUM; ++i) dummy[i] = 0.f;
            const int xy_scale_w = SAMPLED ? tp.sampling : 1;
            if (trk_heavy) trk_gather<TRK_PPT>(g, tab, depth, z_pre, pose_old, trk_pix0, 64, trk_batch, dummy, nullptr, xy_scale_w);
            else trk_gather<TRK_PPT - 1>(g, tab, depth, z_pre, pose_old, trk_pix0, 64, trk_batch, dummy, nullptr, xy_scale_w);
            if (dummy[28] == -1.f) st->dbg[20] = 1ull;                    /* (never: a count; keeps the loads alive) */
        }
        if (wave == 0) {
            double gs = 0.0;
            if (lane < GSDF_TRACK_NSUM) {
                double part[GSDF_TRACK_GROUPS];
#pragma unroll
                for (int grp = 0; grp < GSDF_TRACK_GROUPS; ++grp) part[grp] = acc_prev[grp * 32 + lane];
                gs = part[0];
#pragma unroll
                for (int grp = 1; grp < GSDF_TRACK_GROUPS; ++grp) gs += part[grp];
            }
            const int in_done = in.done;
#pragma unroll
            for (int i = 0; i < 7; ++i) pose[i] = in.pose7[i];
            const int passes = in.passes + 1;
            if (trk_tr && threadIdx.x == 0) { trk_tr[4] = wall_clock64() + (unsigned long long)(gs != gs) + (unsigned long long)(passes < 0); }   /* sums + state arrived */
            int done = 1, converged = 0;
            if (!in_done) {                                               /* else: this optimize() already ended */
                const float totv = (float)gs;
                float tot[GSDF_TRACK_NSUM];                               /* the 29 sums, in every lane */
#pragma unroll
                for (int i = 0; i < GSDF_TRACK_NSUM; ++i) tot[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(totv), i));
                trk_solve_update(tot, tp.damping, tp.conv_sq, passes, tp.max_passes, GSDF_EXPERIMENT(tp.debug, 1), GSDF_EXPERIMENT(tp.debug, 4), pose, &done, &converged);
                if (trk_tr && threadIdx.x == 0) trk_tr[5] = wall_clock64() + (unsigned long long)(pose[0] != pose[0]);               /* solved */
                if (blockIdx.x == 0 && tid == 0) {
                    gsdf_trk_buf& o = st->trk[k & 1];
#pragma unroll
                    for (int i = 0; i < 7; ++i) { o.pose7[i] = pose[i]; st->pose7[i] = pose[i]; }
                    o.done = done; o.converged = converged; o.passes = passes;
                    /* make `done` sticky in the other parity as well: launches that the host queued beyond the
                     * end of this optimize() must not take the older buffer for live state and redo the step
                     * (workgroups of THIS launch that still read it return early, which is what they do anyway) */
                    if (done) st->trk[(k - 1) & 1].done = 1;
                    gsdf_quat_to_R(pose + 3, st->R);
                    st->converged = converged;
                    st->done = done;
                    st->passes = passes;
                    st->last_hits = tot[28];
                    st->n_hit += (unsigned long long)tot[28];
                    /* progress for the host's adaptive pass issue (pinned host memory, system scope) */
                    if (tp.progress)             /* one word, so the host reads a consistent (passes, done) pair */
                        __hip_atomic_store(&tp.progress[0], (tp.serial << 16) | (done ? 0x8000u : 0u) | (unsigned int)(passes & 0x7FFF),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 7; ++i) sh_pose[i] = pose[i];
                sh_done = done;
            }
        }
        __syncthreads();
        if (sh_done) return;
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = sh_pose[i];
    }
    if (trk_tr && threadIdx.x == 0) trk_tr[1] = wall_clock64();                         /* head done */
    if (k >= tp.max_passes || GSDF_EXPERIMENT(tp.debug, 2)) return;                     /* head-only launch */

    /* ---- gather + normal-equation sums of pass k with the current pose ---- */
    float acc[GSDF_TRACK_NSUM];
#pragma unroll
    for (int i = 0; i < GSDF_TRACK_NSUM; ++i) acc[i] = 0.f;
    const int xy_scale = SAMPLED ? tp.sampling : 1;
    if (trk_heavy) trk_gather<TRK_PPT>(g, tab, depth, z_pre, pose, trk_pix0, 64, trk_batch, acc, trk_tr ? trk_tr + 8 + wave : nullptr, xy_scale);
    else trk_gather<TRK_PPT - 1>(g, tab, depth, z_pre, pose, trk_pix0, 64, trk_batch, acc, trk_tr ? trk_tr + 8 + wave : nullptr, xy_scale);
    if (trk_tr && threadIdx.x == 0) trk_tr[2] = wall_clock64();                         /* wave 0: gather done */
    /* every wave reduces its sums as soon as its own gather is done (wsum is used here only): ONE barrier per pass tail */
    wave_sum_to_lane63(acc);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < GSDF_TRACK_NSUM; ++i) wsum[wave][i] = acc[i];
    }
    if (trk_tr && threadIdx.x == 0) trk_tr[6] = wall_clock64();                         /* wave 0: its sums are in LDS */
    __syncthreads();
    if (trk_tr && threadIdx.x == 0) trk_tr[7] = wall_clock64();                         /* every wave's sums are in LDS */
    if (tid < 32) {
        float v = 0.f;
        if (tid < GSDF_TRACK_NSUM) {
            v = wsum[0][tid];
#pragma unroll
            for (int w = 1; w < GSDF_TRACK_BLOCK / 64; ++w) v += wsum[w][tid];
        }
        /* the workgroup's sums (float, fixed order) join its group; f64 adds of ~256 terms stay far below float
         * resolution whatever their order */
        if (tid < GSDF_TRACK_NSUM) unsafeAtomicAdd(&acc_cur[(blockIdx.x % GSDF_TRACK_GROUPS) * 32 + tid], (double)v);
    }
    if (trk_tr && threadIdx.x == 0) trk_tr[3] = wall_clock64();
}
int gsdf_normals_tiles(int W, int H) { return ((W + NRM_TX - 1) / NRM_TX) * ((H + NRM_TY - 1) / NRM_TY); }
void gsdf_launch_track_pass(hipStream_t s, const gsdf_frame_geom& g, const float* depth, gsdf_table tab,
                            gsdf_dev_state* st, double* partials, int n_blocks, const gsdf_track_params& tp_in,
                            const gsdf_normals_job* normals) {
    gsdf_track_params tp = tp_in;
    /* one chunk of TRK_CHUNK pixels per workgroup while the grid allows it (640 x 480: 240 workgroups), more by looping */
    {                                   /* ... and then the same number of chunks for every workgroup (1280 x 960: 480 x 2, not 512 x 1.9) */
        const int chunks = std::max(1, (g.W * g.H + TRK_CHUNK - 1) / TRK_CHUNK), cap = std::max(1, n_blocks);
        const int per = (chunks + cap - 1) / cap;
        n_blocks = (chunks + per - 1) / per;
    }
    /* the launch behind the last pass only finishes it (head: reduce, solve, publish): one workgroup does -- workgroup 0 is the
     * one that publishes; the others would solve the same system and return */
    if (tp.pass_index >= tp.max_passes && !normals) n_blocks = 1;
    tp.n_track_blocks = n_blocks;
    gsdf_normals_job nj;
    memset(&nj, 0, sizeof(nj));
    int extra = 0;
    size_t dyn = 0;
    if (normals) {
        nj = *normals;
        nj.ntx = (g.W + NRM_TX - 1) / NRM_TX;
        const int rest = nj.ntx * ((g.H + NRM_TY - 1) / NRM_TY) - nj.tile_first;
        extra = std::max(0, nj.tile_count > 0 ? std::min(nj.tile_count, rest) : rest);
        dyn = extra ? sizeof(nrm_lds) : 0;
    }
    if (tp.sampling > 1)
        hipLaunchKernelGGL(k_track_pass<true>, dim3(n_blocks + extra), dim3(GSDF_TRACK_BLOCK), dyn, s, g, depth, tab, st, partials, tp, nj);
    else
        hipLaunchKernelGGL(k_track_pass<false>, dim3(n_blocks + extra), dim3(GSDF_TRACK_BLOCK), dyn, s, g, depth, tab, st, partials, tp, nj);
}

/* the sampled pixels of optimize_sampled (RigidPointOptimizer.cpp:62: y = 0, s, 2s, ... < H; x likewise), compacted row-major */
__global__ __launch_bounds__(256) void k_subsample(const float* __restrict__ depth, int W, int s, int Ws, int Ns, float* __restrict__ out) {
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= Ns) return;
    const int qy = q / Ws, qx = q - qy * Ws;
    out[q] = depth[(size_t)(qy * s) * W + (size_t)qx * s];
}
void gsdf_launch_subsample(hipStream_t st, const float* depth, int W, int H, int s, float* out) {
    const int Ws = (W + s - 1) / s, Hs = (H + s - 1) / s, Ns = Ws * Hs;
    hipLaunchKernelGGL(k_subsample, dim3((Ns + 255) / 256), dim3(256), 0, st, depth, W, s, Ws, Ns, out);
}

/* ------------------------------------------------------------------------------------------------
 * RigidPointOptimizer::optimize_sampled as ONE launch: k_track_all.
 *
 * The per-pass launches above pay, per Gauss-Newton pass, the launch gap (1.4 us), the head's memory round trip for the
 * group sums (1.5-2 us), the f64 atomics of the previous pass (1.5 us) and a gather from cold L2s (the per-XCD L2s are
 * invalidated at every kernel boundary).  Here the 256 workgroups (one per CU, co-resident) stay for the whole optimize():
 *  - a lane keeps the depth of its pixels in registers; voxel records and block keys stay in the XCD's L2 between passes;
 *  - the exchange of a pass uses NO atomics and NO separate flag: every workgroup stores its 29 sums as ONE ROW of ten
 *    16-byte chunks {sum, sum, sum, tag} (tag = optimize() serial and pass number; one wave instruction, agent scope) and
 *    then reads ALL rows (5 chunks per lane) until every chunk carries the tag of this pass.  A 16-byte chunk is written
 *    and read whole, so a chunk with the right tag holds the right sums: no ordering between data and flag is needed, which
 *    is what made the atomics-and-ticket variants of rounds 1-2 slower than relaunching (each ordering point is a 1.5 us
 *    round trip).  One hop: the last workgroup's store -> everybody's next poll.
 *  - every workgroup then adds the rows in the same fixed order (double), solves the 6x6 system and updates the pose:
 *    bit-identical everywhere, so no result has to be broadcast.  Rows are double-buffered by pass parity (a workgroup can
 *    be at most one pass ahead of the slowest one).
 * The wait for rows is bounded (the workgroups must be co-resident: guaranteed on an otherwise idle GPU, 256 workgroups
 * of 512 lanes on 256 CUs; another process on the same GPU can break it): a workgroup that waits longer than 50 ms raises
 * the abort word, everybody leaves, and the sticky status bit GSDF_STATUS_TRACK_ABORT makes gsdf_sync fail loudly.
 * The normals of the frame are computed by extra workgroups of the same launch, as in the per-pass path.
 * ---------------------------------------------------------------------------------------------- */
#define TRK_ROW_CHUNKS 10                 /* 29 sums + 1 spare in chunks of 3 + tag */
__device__ __forceinline__ gsdf_u32x4 trk_load_chunk(const gsdf_u32x4* p) {
    gsdf_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MAXB>       /* workgroups the row exchange is sized for: a lane reads MAXB * 10 / 512 chunks per poll */
__global__ __launch_bounds__(GSDF_TRACK_BLOCK) void k_track_all(gsdf_frame_geom g, const float* __restrict__ depth, gsdf_table tab,
                                                                gsdf_dev_state* st, gsdf_u32x4* rows /* [2][n_track_blocks][TRK_ROW_CHUNKS] */,
                                                                unsigned int* abort_word, gsdf_track_params tp, gsdf_normals_job nj) {
    if ((int)blockIdx.x >= tp.n_track_blocks) {
        extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
        const int t = (int)blockIdx.x - tp.n_track_blocks;
        if (t == 0 && threadIdx.x == 0) {
            *nj.deferred_count = 0u;                            /* fresh list for the k_fuse of this frame */
            st->frame_cur = st->frames;                         /* counter_ seen by every workgroup of k_fuse */
        }
        normals_tile(*reinterpret_cast<nrm_lds*>(dyn_lds), t % nj.ntx, t / nj.ntx, g.W, g.H, nj.r, nj.nc, depth, nj.nx, nj.ny, nj.nz,
                     gsdf_tile_stats{ nj.stats, (g.W + 15) / 16, g.zmin, g.zmax });
        return;
    }
    constexpr int NW = GSDF_TRACK_BLOCK / 64;
    __shared__ float wsum[NW][32];
    __shared__ double part[16][32];
    /* the rows of a pass as floats, [n_track_blocks][30]: in the dynamic LDS region (the normals workgroups of this launch
     * use the same region for their tile; static LDS would be charged to both roles and keep them from sharing a CU) */
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds_t[];
    float (*rowv)[3 * TRK_ROW_CHUNKS] = reinterpret_cast<float (*)[3 * TRK_ROW_CHUNKS]>(dyn_lds_t);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = tp.n_track_blocks;
    const int n_chunks = nb * TRK_ROW_CHUNKS;
    unsigned long long* trk_tr = nullptr;
    if (GSDF_EXPERIMENT(tp.debug, 64) && blockIdx.x < 512u)
        trk_tr = reinterpret_cast<unsigned long long*>(st->dbg[23]) + 16 * (2048 + (size_t)blockIdx.x);
    /* the depth of this lane's (first) pixels: loaded once, kept across the passes */
    float z_pre[TRK_PPT];
    {
        const int N = g.W * g.H, base0 = (int)blockIdx.x * GSDF_TRACK_BLOCK + tid, nthreads = nb * GSDF_TRACK_BLOCK;
#pragma unroll
        for (int j = 0; j < TRK_PPT; ++j) {
            const int pix = base0 + j * nthreads;
            z_pre[j] = pix < N ? depth[pix] : 0.f;
        }
    }
    float pose[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = st->pose7[i];                    /* RigidOptimizer::pose_ (kept in st->pose7 between frames) */
    int done = 0, converged = 0, passes = 0;
    float hits = 0.f;
    unsigned long long hit_total = 0ull;
    bool aborted = false;
    for (int k = 0; k < tp.max_passes && !done; ++k) {
        if (trk_tr && tid == 0 && k < 12) trk_tr[16 * 512 * k + 0] = wall_clock64();
        /* ---- gather + normal-equation sums of pass k with the current pose ---- */
        float acc[GSDF_TRACK_NSUM];
#pragma unroll
        for (int i = 0; i < GSDF_TRACK_NSUM; ++i) acc[i] = 0.f;
        trk_gather<TRK_PPT>(g, tab, depth, z_pre, pose, blockIdx.x * GSDF_TRACK_BLOCK + tid, nb * GSDF_TRACK_BLOCK, TRK_PPT * nb * GSDF_TRACK_BLOCK, acc);
        if (trk_tr && tid == 0 && k < 12) trk_tr[16 * 512 * k + 1] = wall_clock64();        /* wave 0: gather done */
        wave_sum_to_lane63(acc);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < GSDF_TRACK_NSUM; ++i) wsum[wave][i] = acc[i];
        }
        __syncthreads();
        /* ---- this workgroup's row: ten chunks {3 sums, tag}, one store instruction ---- */
        const uint32_t tag = (tp.serial << 8) | (uint32_t)(k + 1);
        gsdf_u32x4* buf = rows + (size_t)(k & 1) * (size_t)n_chunks;
        if (tid < TRK_ROW_CHUNKS) {
            float v3[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = 3 * tid + q;
                float v = 0.f;
                if (i < GSDF_TRACK_NSUM) {
                    v = wsum[0][i];
#pragma unroll
                    for (int w = 1; w < NW; ++w) v += wsum[w][i];           /* float, fixed order */
                }
                v3[q] = v;
            }
            const gsdf_u32x4 c = { __float_as_uint(v3[0]), __float_as_uint(v3[1]), __float_as_uint(v3[2]), tag };
            gsdf_u32x4* dst = buf + (size_t)blockIdx.x * TRK_ROW_CHUNKS + tid;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(c) : "memory");
        }
        if (trk_tr && tid == 0 && k < 12) trk_tr[16 * 512 * k + 2] = wall_clock64();        /* row stored */
        /* ---- all rows of pass k: poll until every chunk this lane reads carries the tag ---- */
        constexpr int PER = (MAXB * TRK_ROW_CHUNKS + GSDF_TRACK_BLOCK - 1) / GSDF_TRACK_BLOCK;   /* chunks per lane at most */
        gsdf_u32x4 ch[PER];
        const unsigned long long t0 = wall_clock64();
        unsigned int rounds = 0u;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = tid + GSDF_TRACK_BLOCK * i;
                ch[i] = gsdf_u32x4{ 0u, 0u, 0u, tag };
                if (c < n_chunks) ch[i] = trk_load_chunk(buf + c);
            }
            /* hipcc does not track the loads issued from inline asm: each wait statement names a destination register, so
             * no use of it can be scheduled before the data has arrived (the first wait drains all, the others cost nothing) */
#pragma unroll
            for (int i = 0; i < PER; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ch[i]) :: "memory");
#pragma unroll
            for (int i = 0; i < PER; ++i) ok = ok && ch[i].w == tag;
            ++rounds;
            if (__syncthreads_and(ok ? 1 : 0)) break;
            /* bounded: co-residency is a property of the launch environment, not of this code */
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tp.serial || wall_clock64() - t0 > 5000000ull) {
                aborted = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (aborted) {
            if (tid == 0) {
                __hip_atomic_store(abort_word, tp.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(&st->status, GSDF_STATUS_TRACK_ABORT);
            }
            break;
        }
        if (trk_tr && tid == 0 && k < 12) { trk_tr[16 * 512 * k + 3] = wall_clock64(); trk_tr[16 * 512 * k + 6] = rounds; }   /* rows complete */
        /* ---- the same fixed-order sum everywhere: 16 groups of rows in double, then the groups ---- */
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + GSDF_TRACK_BLOCK * i;
            if (c < n_chunks) {
                const int r = c / TRK_ROW_CHUNKS, q = c - r * TRK_ROW_CHUNKS;
                rowv[r][3 * q] = __uint_as_float(ch[i].x); rowv[r][3 * q + 1] = __uint_as_float(ch[i].y); rowv[r][3 * q + 2] = __uint_as_float(ch[i].z);
            }
        }
        __syncthreads();
        {
            const int v = tid & 31, grp = tid >> 5;                           /* 16 groups x 32 values */
            const int per = (nb + 15) / 16;
            double sgrp = 0.0;
            if (v < 3 * TRK_ROW_CHUNKS)
                for (int r = grp * per; r < (grp + 1) * per && r < nb; ++r) sgrp += (double)rowv[r][v];
            part[grp][v] = sgrp;
        }
        __syncthreads();
        float tot[GSDF_TRACK_NSUM];
        {
            double gs = 0.0;
            if (lane < GSDF_TRACK_NSUM) {
                gs = part[0][lane];
#pragma unroll
                for (int grp = 1; grp < 16; ++grp) gs += part[grp][lane];
            }
            const float totv = (float)gs;
#pragma unroll
            for (int i = 0; i < GSDF_TRACK_NSUM; ++i) tot[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(totv), i));
        }
        /* ---- one Gauss-Newton step (every wave of every workgroup computes it: identical bits) ---- */
        passes = k + 1;
        trk_solve_update(tot, tp.damping, tp.conv_sq, passes, tp.max_passes, GSDF_EXPERIMENT(tp.debug, 1), GSDF_EXPERIMENT(tp.debug, 4), pose, &done, &converged);
        hits = tot[28];
        hit_total += (unsigned long long)tot[28];
        if (trk_tr && tid == 0 && k < 12) trk_tr[16 * 512 * k + 4] = wall_clock64();        /* solved */
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (aborted) { done = 1; converged = 0; }
#pragma unroll
        for (int i = 0; i < 7; ++i) st->pose7[i] = pose[i];
        gsdf_quat_to_R(pose + 3, st->R);
        st->converged = converged;
        st->done = 1;
        st->passes = passes;
        st->last_hits = hits;
        st->n_hit += hit_total;
        st->trk[0].done = 1; st->trk[1].done = 1;
        if (tp.progress)
            __hip_atomic_store(&tp.progress[0], (tp.serial << 16) | 0x8000u | (unsigned int)(passes & 0x7FFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void gsdf_launch_track_all(hipStream_t s, const gsdf_frame_geom& g, const float* depth, gsdf_table tab, gsdf_dev_state* st,
                           void* rows, unsigned int* abort_word, int n_blocks, const gsdf_track_params& tp_in,
                           const gsdf_normals_job* normals) {
    gsdf_track_params tp = tp_in;
    tp.n_track_blocks = n_blocks;
    gsdf_normals_job nj;
    memset(&nj, 0, sizeof(nj));
    int extra = 0;
    size_t dyn = 0;
    if (normals) {
        nj = *normals;
        nj.ntx = (g.W + NRM_TX - 1) / NRM_TX;
        extra = nj.ntx * ((g.H + NRM_TY - 1) / NRM_TY);
        dyn = sizeof(nrm_lds);
    }
    dyn = dyn > (size_t)n_blocks * 3 * TRK_ROW_CHUNKS * sizeof(float) ? dyn : (size_t)n_blocks * 3 * TRK_ROW_CHUNKS * sizeof(float);
    if (n_blocks <= GSDF_TRACK_MAXBLK)
        hipLaunchKernelGGL(k_track_all<GSDF_TRACK_MAXBLK>, dim3(n_blocks + extra), dim3(GSDF_TRACK_BLOCK), dyn, s, g, depth, tab, st,
                           reinterpret_cast<gsdf_u32x4*>(rows), abort_word, tp, nj);
    else
        hipLaunchKernelGGL(k_track_all<2 * GSDF_TRACK_MAXBLK>, dim3(n_blocks + extra), dim3(GSDF_TRACK_BLOCK), dyn, s, g, depth, tab, st,
                           reinterpret_cast<gsdf_u32x4*>(rows), abort_word, tp, nj);
}
size_t gsdf_track_all_rows_bytes(int n_blocks) { return (size_t)2 * (size_t)n_blocks * TRK_ROW_CHUNKS * sizeof(gsdf_u32x4); }

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
                                                int raw, const uint32_t* vis, int vis_words, uint32_t* vis_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_slots; i += stride) {
        const unsigned long long bk = tab.bkeys[i / GSDF_BLOCK_VOX];      /* one block per wavefront */
        if (bk == GSDF_KEY_EMPTY) continue;
        const gsdf_payload sl = tab.vox[i];
        if (!(sl.w > 0.f)) continue;                                      /* the voxel exists iff w > 0 */
        const unsigned long long key = gsdf_voxel_key(bk, (uint32_t)(i % GSDF_BLOCK_VOX));
        const unsigned long long o = atomicAdd(counter, 1ull);
        if ((long long)o >= max_n) continue;
        if (keys_out) keys_out[o] = key;
        if (payload_out) {
            float* p = payload_out + 5 * o;
            if (raw) { p[0] = sl.s; p[1] = sl.gx; p[2] = sl.gy; p[3] = sl.gz; p[4] = sl.w; }
            else     { p[0] = sl.s / sl.w; p[1] = sl.gx; p[2] = sl.gy; p[3] = sl.gz; p[4] = sl.w; }
        }
        if (vis_out && vis)
            for (int w = 0; w < vis_words; ++w) vis_out[o * vis_words + w] = vis[i * vis_words + w];
    }
}
void gsdf_launch_export(hipStream_t s, gsdf_table tab, size_t n_slots, unsigned long long* keys_out,
                        float* payload_out, unsigned long long* counter, long long max_n, int raw,
                        const uint32_t* vis, int vis_words, uint32_t* vis_out) {
    hipLaunchKernelGGL(k_export, dim3(2048), dim3(256), 0, s, tab, n_slots, keys_out, payload_out, counter, max_n, raw,
                       vis, vis_words, vis_out);
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
/* unsorted compaction of (key as int32 x3, raw sums s,gx,gy,gz,w) into caller-provided device buffers */
__global__ __launch_bounds__(256) void k_export_raw(gsdf_table tab, size_t n_slots, int32_t* keys_out, float* payload_out,
                                                    unsigned long long* counter, long long max_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_slots; i += stride) {
        const unsigned long long bk = tab.bkeys[i / GSDF_BLOCK_VOX];
        if (bk == GSDF_KEY_EMPTY) continue;
        const gsdf_payload sl = tab.vox[i];
        if (!(sl.w > 0.f)) continue;
        const unsigned long long key = gsdf_voxel_key(bk, (uint32_t)(i % GSDF_BLOCK_VOX));
        const unsigned long long o = atomicAdd(counter, 1ull);
        if ((long long)o >= max_n) continue;
        int x, y, z;
        gsdf_key_unpack(key, &x, &y, &z);
        keys_out[3 * o] = x; keys_out[3 * o + 1] = y; keys_out[3 * o + 2] = z;
        float* p = payload_out + 5 * o;
        p[0] = sl.s; p[1] = sl.gx; p[2] = sl.gy; p[3] = sl.gz; p[4] = sl.w;
    }
}
void gsdf_launch_export_raw(hipStream_t s, gsdf_table tab, size_t n_slots, int32_t* keys_out, float* payload_out,
                            unsigned long long* counter, long long max_n) {
    hipLaunchKernelGGL(k_export_raw, dim3(2048), dim3(256), 0, s, tab, n_slots, keys_out, payload_out, counter, max_n);
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
            const gsdf_payload* sl = gsdf_find(tab, gsdf_key_pack(vx, vy, vz));
            if (sl && sl->w > 0.f) {
                ow = sl->w;
                const gsdf_v3 gn = gsdf_normalized3(gsdf_v3{ sl->gx, sl->gy, sl->gz });
                og = gsdf_v3{ 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };
                const gsdf_v3 d = { vs * (float)vx - p.x, vs * (float)vy - p.y, vs * (float)vz - p.z };
                od = gsdf_tsdf_phi(sl->s / sl->w, gn, d);                                  /* :114 */
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

/* tsdf_.at(idx) for n voxel indices -- MapGradPixelSdf.h:127-129 (getSdf): the stored SdfVoxel (dist, raw gradient sum, weight) */
__global__ __launch_bounds__(256) void k_get_voxels(gsdf_table tab, const int32_t* keys, long long n, float* payload, int32_t* found) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int x = keys[3 * i], y = keys[3 * i + 1], z = keys[3 * i + 2];
        float o[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
        int32_t f = 0;
        if (gsdf_key_in_range(x, y, z)) {
            const gsdf_payload* sl = gsdf_find(tab, gsdf_key_pack(x, y, z));
            if (sl && sl->w > 0.f) { f = 1; o[0] = sl->s / sl->w; o[1] = sl->gx; o[2] = sl->gy; o[3] = sl->gz; o[4] = sl->w; }
        }
        for (int k = 0; k < 5; ++k) payload[5 * i + k] = o[k];
        found[i] = f;
    }
}
void gsdf_launch_get_voxels(hipStream_t s, gsdf_table tab, const int32_t* keys, long long n, float* payload, int32_t* found) {
    if (n <= 0) return;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_get_voxels, dim3(blocks), dim3(256), 0, s, tab, keys, n, payload, found);
}

/* ------------------------------------------------------------------------------------------------
 * Voxel-hash raycaster (BASELINE.json north_star; absent from the reference): defined
 * on top of weights()/tsdf() -- MapGradPixelSdf.h:109-125 -- and the tracker's back-projection
 * (RigidPointOptimizer.cpp:46-47,67-70): p(s) = s R (x0, y0, 1) + t; coarse steps of min(4, factor - 1) voxels of depth
 * (at least 1) while the voxel under p(s) is missing -- re-walked in fine steps when they end on an existing voxel -- and
 * 1-voxel steps inside the band; hit = first sign change phi_prev < 0 <= phi of two consecutive existing samples (the SDF
 * is negative in front of a surface); depth by linear interpolation, normal = R^T grad/|grad| of the sample behind the
 * surface.  The test infrastructure holds the CPU statement of the same definition.
 *
 * Design (measured on the bench map, tools/raycast_bench.py + tools/raycast_pmc.sh).  A ray is ~45 coarse samples
 * through empty space and ~15 fine ones in the band.  The sample-at-a-time walk of rounds 1-2 (k_raycast_v1, kept in the test
 * build) pays a block-key probe -- key packing, a 64-bit hash, a chain of dependent loads to the first empty entry -- for
 * every one of them: 129 wave instructions per sample.  The kernel is bound by VALU issue as much as by latency (a wave64
 * instruction occupies its SIMD for 4 cycles: 35 M wave instructions = 57 us of the 121 us), so the design is about NOT
 * computing, in three levels:
 *  1. cell filter (gsdf_table::occ2, one hashed bit per 32^3-voxel cell that holds a block, 8 KB): a sample in an empty cell
 *     is missing, and so is every further sample up to the cell's far face -- the ray's exit distance is three multiplies,
 *     the skipped samples cost one float add each (s advances by the same repeated additions as in the definition, so the
 *     positions stay bit-identical);
 *  2. block filter (gsdf_table::occ, one hashed bit per existing block): in a cell that holds blocks a sample whose block
 *     bit is clear is missing -- one load, no key packing, no hash, no probe chain;
 *  3. only samples whose block bit is set are looked up (probe + 32-byte record), the filter word and the home entry of the
 *     key array requested together.
 *  Same samples, same arithmetic, same results bit for bit as the sample-at-a-time walk.  A wave is an 8x8-pixel patch: its
 *  64 rays cross the same one or two blocks at every depth, so a wave instruction's loads coalesce into a few requests and
 *  its lanes enter the band at about the same depth.  Few registers: all 4800 waves of a 640x480 render are resident at once.
 *  Measured and rejected: looking RC_B samples ahead per lane (their lookups issued together, the state machine consuming
 *  them while the predicted positions hold): 8x fewer dependent round trips, but 20 % MORE instructions (speculative work
 *  beyond a hit or a change of step), 118-136 registers (two rounds of waves) -- 138-200 us against 121.
 *  Round 4, the same idea without the registers: once half of a wave's rays have ended (the slowest waves spend 40 % of their
 *  iterations like that, profiles/r04_raycast_helper_lanes.txt), 2 / 4 / 8 lanes per remaining ray look up its next positions
 *  and the ray takes the results in order through ds_bpermute -- bit-identical output, the slowest waves' iterations 67 -> 54,
 *  and the launch no shorter (125 us with the second loop in the kernel, helping or not, against 119 without it): a band
 *  sample's own step (normalisation, divisions: ~150 dependent instructions) is as long as its lookup.
 * Counters (for the roofline entry): samples the definition evaluated and records it read, one row per workgroup.
 * ---------------------------------------------------------------------------------------------- */
struct rc_ray_state {
    float s, fine_until, s_coarse_from, phi_prev, s_prev, out_z;
    gsdf_v3 out_n;
    bool prev_ok, done;
};
/* one step of the definition with the sample at st.s already looked up: p = its position, (vx, vy, vz) its voxel,
 * w0 = the voxel's weight (0: missing), (sd, gx, gy, gz) its record */
__device__ __forceinline__ void rc_step(rc_ray_state& st, const gsdf_v3& p, int vx, int vy, int vz, const gsdf_pose_arg& pose, float vs,
                                        float fine, float coarse, float w0, float sd, float gx, float gy, float gz) {
    const float* R = pose.R;
    const float s = st.s;
    if (w0 > 0.f && st.s_coarse_from >= 0.f) {                /* entered the band by a coarse step: walk that stretch again */
        st.fine_until = s;
        st.s = st.s_coarse_from + fine;
        st.s_coarse_from = -1.f;
        st.prev_ok = false;
        return;
    }
    if (w0 > 0.f) {
        const gsdf_v3 gn = gsdf_normalized3(gsdf_v3{ gx, gy, gz });
        const gsdf_v3 c = { vs * (float)vx - p.x, vs * (float)vy - p.y, vs * (float)vz - p.z };
        const float phi = gsdf_tsdf_phi(sd / w0, gn, c);                                   /* MapGradPixelSdf.h:114 */
        if (st.prev_ok && st.phi_prev < 0.f && phi >= 0.f) {   /* the stored SDF is negative in front of the surface */
            st.out_z = st.s_prev + (s - st.s_prev) * (st.phi_prev / (st.phi_prev - phi));
            st.out_n = gsdf_v3{ gsdf_sum3(R[0] * gn.x, R[3] * gn.y, R[6] * gn.z), gsdf_sum3(R[1] * gn.x, R[4] * gn.y, R[7] * gn.z),
                                gsdf_sum3(R[2] * gn.x, R[5] * gn.y, R[8] * gn.z) };
            st.done = true;
            return;
        }
        st.prev_ok = true; st.phi_prev = phi; st.s_prev = s;
        st.s = s + fine;
    } else if (st.prev_ok && s - st.s_prev < 1.5f * fine) {
        st.s = s + fine;                                       /* one missing sample inside the band is bridged */
    } else {
        st.prev_ok = false;
        if (s < st.fine_until) st.s = s + fine;
        else { st.s_coarse_from = s; st.s = s + coarse; }
    }
}

__global__ __launch_bounds__(256) void k_raycast(gsdf_table tab, float vs, float inv_vs, int factor, int W, int H, float fx, float fy,
                                                  float cx, float cy, gsdf_pose_arg pose, float zmin, float zmax,
                                                  float* __restrict__ depth, float* __restrict__ normals,
                                                  unsigned long long* __restrict__ wg_counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t_begin = wall_clock64();
    const int u = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7), v = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool live = u < W && v < H;
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
    const float x0 = ((float)u - cx) * fx_inv, y0 = ((float)v - cy) * fy_inv;
    const gsdf_v3 d = gsdf_matvec(pose.R, gsdf_v3{ x0, y0, 1.f });
    const float fine = vs, coarse = (float)(factor < 2 ? 1 : (factor > 5 ? 4 : factor - 1)) * vs;
    /* for the cell skip: depth per voxel of travel along each axis, and coarse steps per unit of depth (estimates with
     * margin, no parity items) */
    const float rdx = vs * __builtin_amdgcn_rcpf(d.x), rdy = vs * __builtin_amdgcn_rcpf(d.y), rdz = vs * __builtin_amdgcn_rcpf(d.z);
    const float inv_coarse = __builtin_amdgcn_rcpf(coarse);
    constexpr int CELL = 1 << GSDF_CELL_SHIFT;
    rc_ray_state st;
    st.s = zmin; st.fine_until = zmin; st.s_coarse_from = -1.f; st.phi_prev = 0.f; st.s_prev = 0.f; st.out_z = 0.f;
    st.out_n = gsdf_v3{ 0.f, 0.f, 0.f };
    st.prev_ok = false; st.done = !live;
    unsigned int n_samp = 0u, n_rec = 0u;
    unsigned int it_fast = 0u, it_slow = 0u;                          /* wave-uniform loop counts: cell skips taken / all iterations */
#ifdef GSDF_EXPERIMENTS
    unsigned int it_le32 = 0u, it_le16 = 0u, it_le8 = 0u, lanes_sum = 0u;
#endif
    for (;;) {
        const bool active = !st.done && st.s < zmax;
        if (!__any(active)) break;
        ++it_slow;
#ifdef GSDF_EXPERIMENTS
        {   /* how full the wave is, iteration by iteration (tools/raycast_bench.py): iterations with <= 32 / 16 / 8 rays left */
            const unsigned int na = (unsigned int)__popcll(__ballot(active));
            it_le32 += na <= 32u; it_le16 += na <= 16u; it_le8 += na <= 8u; lanes_sum += na;
        }
#endif
        const float s = st.s;
        const gsdf_v3 p = { s * d.x + pose.t[0], s * d.y + pose.t[1], s * d.z + pose.t[2] };
        const float qx = inv_vs * p.x, qy = inv_vs * p.y, qz = inv_vs * p.z;
        const int vx = (int)gsdf_roundf(qx), vy = (int)gsdf_roundf(qy), vz = (int)gsdf_roundf(qz);           /* gsdf_float2vox1 */
        const bool inr = gsdf_key_in_range(vx, vy, vz);
        /* in empty space a missing sample is followed by a coarse step */
        const bool in_coarse = active && (st.s_coarse_from >= 0.f || (!st.prev_ok && !(s < st.fine_until)));
        /* (Measured and rejected: lanes in empty space first, lanes already in a band waiting for them, so that an iteration is
         * either the short empty-space step or the full band step -- rays that run along a surface in occupied cells make
         * their whole wave wait for hundreds of single steps: 302 us.) */
        const bool mine = active;
        bool missing = !inr;
        int n_skip = 0;
        float w0 = 0.f, sd = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        if (mine) {
            /* 1. + 2. cell filter and block filter, both words requested together (in a band only the block filter) */
            const uint32_t cb = inr && in_coarse ? gsdf_occ2_bit_vox(tab, vx, vy, vz) : 0u;
            const uint32_t ob = inr ? gsdf_occ_bit_vox(tab, vx, vy, vz) : 0u;
            const uint32_t cw = inr && in_coarse ? gsdf_occ2(tab)[cb >> 5] : 0xFFFFFFFFu;
            const uint32_t ow = inr ? tab.occ[ob >> 5] : 0u;
            /* in a band the sample most likely exists: the home entry of the key array is requested with the filter word (one
             * round trip less on the chain that decides the render's duration); in empty space it would be a wasted hash */
            const bool early = inr && !in_coarse;
            const unsigned long long key = gsdf_key_pack(vx, vy, vz);
            const unsigned long long bkey = gsdf_block_key(key);
            uint32_t home = 0u;
            unsigned long long k0 = GSDF_KEY_EMPTY;
            if (early) { home = gsdf_hash(bkey) & tab.block_mask; k0 = tab.bkeys[home]; }
            if (in_coarse && inr && !gsdf_occ_test(cw, cb)) {
                missing = true;
                /* every sample whose position stays >= half a voxel inside this cell is missing too: depth to the first face */
                const float lx = (float)((((vx + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT) << GSDF_CELL_SHIFT) - GSDF_KEY_OFF);
                const float ly = (float)((((vy + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT) << GSDF_CELL_SHIFT) - GSDF_KEY_OFF);
                const float lz = (float)((((vz + GSDF_KEY_OFF) >> GSDF_CELL_SHIFT) << GSDF_CELL_SHIFT) - GSDF_KEY_OFF);
                const float hi = (float)CELL - 1.5f;
                const float tx = d.x > 0.f ? (lx + hi - qx) * rdx : (d.x < 0.f ? (lx + 0.5f - qx) * rdx : 3.0e38f);
                const float ty = d.y > 0.f ? (ly + hi - qy) * rdy : (d.y < 0.f ? (ly + 0.5f - qy) * rdy : 3.0e38f);
                const float tz = d.z > 0.f ? (lz + hi - qz) * rdz : (d.z < 0.f ? (lz + 0.5f - qz) * rdz : 3.0e38f);
                const float te = fminf(fminf(tx, ty), tz);
                n_skip = te > 0.f ? (int)fminf(te * inv_coarse, 64.f) : 0;
            }
            if (!gsdf_occ_test(ow, ob)) missing = true;
            if (!missing) {
                /* 3. the lookup proper: probe of the key array + the 32-byte record */
                if (!early) { home = gsdf_hash(bkey) & tab.block_mask; k0 = tab.bkeys[home]; }
                const int blk = gsdf_block_find(tab, bkey, home, k0);
                if (blk >= 0) {
                    const float2* q = reinterpret_cast<const float2*>(tab.vox + ((size_t)blk * GSDF_BLOCK_VOX + gsdf_block_local(key)));
                    const float2 a = q[0], b = q[1], c = q[2];
                    w0 = a.x; sd = a.y; gx = b.x; gy = b.y; gz = c.x;
                    ++n_rec;
                }
            }
            ++n_samp;
            if (in_coarse && !(w0 > 0.f)) {
                /* missing, in empty space: a coarse step -- and one for every further sample known to be missing */
                st.prev_ok = false;
                st.s_coarse_from = s;
                st.s = s + coarse;
                for (int k = 0; k < n_skip && st.s < zmax; ++k) { st.s_coarse_from = st.s; st.s = st.s + coarse; ++n_samp; }
            } else {
                rc_step(st, p, vx, vy, vz, pose, vs, fine, coarse, w0, sd, gx, gy, gz);
            }
        }
        if (n_skip > 0) ++it_fast;
    }
    if (live) {
        const size_t i = (size_t)v * W + u;
        depth[i] = st.out_z;
        if (normals) { normals[i] = st.out_n.x; normals[(size_t)W * H + i] = st.out_n.y; normals[2 * (size_t)W * H + i] = st.out_n.z; }
    }
    /* counters of the roofline entry: one row per workgroup, plain read-modify-write by its thread 0 (launches are ordered
     * on the stream).  NOT atomics on one word: 2 same-address atomics per wave serialised the whole kernel (measured: +60 us) */
    __shared__ float rc_cnt[4][2];
    __shared__ unsigned int rc_it[4];
    const float ts = wave_sum((float)n_samp), tr = wave_sum((float)n_rec);           /* < 2^24 per wave: exact */
    if (lane == 0) { rc_cnt[wave][0] = ts; rc_cnt[wave][1] = tr; rc_it[wave] = it_slow; }
#ifdef GSDF_EXPERIMENTS
    __shared__ unsigned long long rc_fill[4];
    if (lane == 0) rc_fill[wave] = (unsigned long long)it_le32 | ((unsigned long long)it_le16 << 12) | ((unsigned long long)it_le8 << 24) |
                                   ((unsigned long long)lanes_sum << 36);
#endif
    __syncthreads();
    if (wg_counts && threadIdx.x == 0) {
        unsigned long long* row = wg_counts + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        row[0] += (unsigned long long)(rc_cnt[0][0] + rc_cnt[1][0] + rc_cnt[2][0] + rc_cnt[3][0]);
        row[1] += (unsigned long long)(rc_cnt[0][1] + rc_cnt[1][1] + rc_cnt[2][1] + rc_cnt[3][1]);
        row[2] += it_fast; row[3] += it_slow;                           /* lane 0 of wave 0: cell skips / loop iterations */
        row[4] = t_begin; row[5] = wall_clock64();                      /* life of the workgroup, 100 MHz ticks (last launch) */
        row[6] = max(max(rc_it[0], rc_it[1]), max(rc_it[2], rc_it[3]));   /* loop iterations of its slowest wave (last launch) */
#ifdef GSDF_EXPERIMENTS
        int sw = 0;
        for (int i = 1; i < 4; ++i) if (rc_it[i] > rc_it[sw]) sw = i;
        row[7] = rc_fill[sw];                                           /* fill of that wave: le32 | le16 << 12 | le8 << 24 | lane sum << 36 */
#endif
    }
}

#ifdef GSDF_EXPERIMENTS
/* the sample-at-a-time walk of rounds 1 and 2 (one lane per pixel, one dependent lookup per step): kept for the before /
 * after measurement of tools/raycast_bench.py (debug bit 16384) */
__global__ __launch_bounds__(256) void k_raycast_v1(gsdf_table tab, float vs, float inv_vs, int factor, int W, int H, float fx, float fy,
                                                     float cx, float cy, gsdf_pose_arg pose, float zmin, float zmax,
                                                     float* __restrict__ depth, float* __restrict__ normals) {
    const int u = blockIdx.x * 16 + (threadIdx.x & 15), v = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (u >= W || v >= H) return;
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
    const float x0 = ((float)u - cx) * fx_inv, y0 = ((float)v - cy) * fy_inv;
    const gsdf_v3 d = gsdf_matvec(pose.R, gsdf_v3{ x0, y0, 1.f });
    const float fine = vs, coarse = (float)(factor < 2 ? 1 : (factor > 5 ? 4 : factor - 1)) * vs;
    rc_ray_state st;
    st.s = zmin; st.fine_until = zmin; st.s_coarse_from = -1.f; st.phi_prev = 0.f; st.s_prev = 0.f; st.out_z = 0.f;
    st.out_n = gsdf_v3{ 0.f, 0.f, 0.f };
    st.prev_ok = false; st.done = false;
    while (!st.done && st.s < zmax) {
        const gsdf_v3 p = { st.s * d.x + pose.t[0], st.s * d.y + pose.t[1], st.s * d.z + pose.t[2] };
        const int vx = gsdf_float2vox1(inv_vs, p.x), vy = gsdf_float2vox1(inv_vs, p.y), vz = gsdf_float2vox1(inv_vs, p.z);
        const gsdf_payload* sl = gsdf_key_in_range(vx, vy, vz) ? gsdf_find(tab, gsdf_key_pack(vx, vy, vz)) : nullptr;
        float2 ws = make_float2(0.f, 0.f), gxy = ws, gz_ = ws;
        if (sl) { const float2* q = reinterpret_cast<const float2*>(sl); ws = q[0]; gxy = q[1]; gz_ = q[2]; }
        rc_step(st, p, vx, vy, vz, pose, vs, fine, coarse, ws.x, ws.y, gxy.x, gxy.y, gz_.x);
    }
    const size_t i = (size_t)v * W + u;
    depth[i] = st.out_z;
    if (normals) { normals[i] = st.out_n.x; normals[(size_t)W * H + i] = st.out_n.y; normals[2 * (size_t)W * H + i] = st.out_n.z; }
}
#endif
void gsdf_launch_raycast(hipStream_t s, gsdf_table tab, float vs, float inv_vs, int factor, int W, int H, const float K[9],
                         const gsdf_pose_arg& pose, float zmin, float zmax, float* depth, float* normals, unsigned long long* wg_counts,
                         int debug) {
#ifdef GSDF_EXPERIMENTS
    if (debug & 16384) {
        hipLaunchKernelGGL(k_raycast_v1, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, s, tab, vs, inv_vs, factor, W, H, K[0], K[4],
                           K[2], K[5], pose, zmin, zmax, depth, normals);
        return;
    }
#endif
    (void)debug;
    hipLaunchKernelGGL(k_raycast, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, s, tab, vs, inv_vs, factor, W, H, K[0], K[4], K[2],
                       K[5], pose, zmin, zmax, depth, normals, wg_counts);
}

/* ------------------------------------------------------------------------------------------------
 * Iso-surface extraction on the device: LayeredMarchingCubesNoColor::computeIsoSurface
 * (mesh/LayeredMarchingCubesNoColor.cpp:354-712) without the layer buffers -- one lane per voxel record; a lane
 * whose voxel exists is the (0,0,0) corner of a cube, gathers the other 7 corners through the block map
 * (computeLutIndex :593-639: a cube is skipped when any corner has weight 0), interpolates the crossing edges
 * (interpolate :642-662: 1e-7 guards, double mu, clamp) and appends its non-degenerate triangles (:686-712)
 * with a sort key (voxel key in z-y-x order, triangle number): the host sorts them into the reference's sweep
 * order.  The 256 x 16 triangle table comes from the caller.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_mesh_bbox(gsdf_table tab, size_t n_slots, int* mn /* [3], preset to INT_MAX */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int m[3] = { 2147483647, 2147483647, 2147483647 };
    for (; i < n_slots; i += stride) {
        const unsigned long long bk = tab.bkeys[i / GSDF_BLOCK_VOX];
        if (bk == GSDF_KEY_EMPTY) continue;
        if (!(tab.vox[i].w > 0.f)) continue;
        int x, y, z;
        gsdf_key_unpack(gsdf_voxel_key(bk, (uint32_t)(i % GSDF_BLOCK_VOX)), &x, &y, &z);
        m[0] = x < m[0] ? x : m[0]; m[1] = y < m[1] ? y : m[1]; m[2] = z < m[2] ? z : m[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int v = m[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(v, o); v = other < v ? other : v; }
        if ((threadIdx.x & 63) == 0 && v != 2147483647) atomicMin(&mn[a], v);
    }
}

__device__ __forceinline__ gsdf_v3 mesh_interpolate(float t0, float t1, gsdf_v3 v0, gsdf_v3 v1, float iso) {
    if (fabs((double)(iso - t0)) < 1e-7) return v0;                        /* :645-650 (float difference, double compare) */
    if (fabs((double)(iso - t1)) < 1e-7) return v1;
    if (fabs((double)(t0 - t1)) < 1e-7) return v0;
    double mu = (double)((iso - t0) / (t1 - t0));
    if (mu > 1.0) mu = 1.0; else if (mu < 0) mu = 0.0;
    gsdf_v3 v;
    v.x = (float)((double)v0.x + mu * (double)(v1.x - v0.x));
    v.y = (float)((double)v0.y + mu * (double)(v1.y - v0.y));
    v.z = (float)((double)v0.z + mu * (double)(v1.z - v0.z));
    return v;
}

__global__ __launch_bounds__(256) void k_mesh(gsdf_table tab, size_t n_slots, float vs, float iso, const int* __restrict__ mn,
                                               const signed char* __restrict__ tri_table, float* __restrict__ tris,
                                               unsigned long long* __restrict__ keys, unsigned long long* counter, long long max_tris) {
    /* corner c -> (dx, dy, dz), numbering of computeLutIndex (:599-606); edge e -> its two corners */
    const int CORNER[8][3] = { { 1, 1, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 0, 1, 0 }, { 1, 1, 1 }, { 1, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 } };
    const int EDGE[12][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 0 }, { 4, 5 }, { 5, 6 }, { 6, 7 }, { 7, 4 }, { 0, 4 }, { 1, 5 }, { 2, 6 }, { 3, 7 } };
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int m0 = mn[0], m1 = mn[1], m2 = mn[2];
    const float o0 = -(float)m0 * vs, o1 = -(float)m1 * vs, o2 = -(float)m2 * vs;     /* origin_ (:377) */
    for (; i < n_slots; i += stride) {
        const unsigned long long bk = tab.bkeys[i / GSDF_BLOCK_VOX];
        if (bk == GSDF_KEY_EMPTY) continue;
        const gsdf_payload self = tab.vox[i];
        if (!(self.w > 0.f)) continue;
        int x, y, z;
        gsdf_key_unpack(gsdf_voxel_key(bk, (uint32_t)(i % GSDF_BLOCK_VOX)), &x, &y, &z);
        float d[8];
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c == 2) { d[c] = self.s / self.w; continue; }
            const int cx = x + CORNER[c][0], cy = y + CORNER[c][1], cz = z + CORNER[c][2];
            const gsdf_payload* q = gsdf_key_in_range(cx, cy, cz) ? gsdf_find(tab, gsdf_key_pack(cx, cy, cz)) : nullptr;
            float w = 0.f, sd = 0.f;
            if (q) { const float2 ws = *reinterpret_cast<const float2*>(q); w = ws.x; sd = ws.y; }
            if (!(w > 0.f)) ok = false;
            d[c] = ok ? sd / w : 0.f;
        }
        if (!ok) continue;
        int idx = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) if (d[c] > iso) idx |= 1 << c;
        if (idx == 0 || idx == 255) continue;
        const signed char* t = tri_table + 16 * idx;
        for (int k = 0; k < 15 && t[k] >= 0; k += 3) {
            gsdf_v3 p[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int e = t[k + v], a = EDGE[e][0], b = EDGE[e][1];
                const gsdf_v3 wa = { (float)(x + CORNER[a][0] - m0) * vs - o0, (float)(y + CORNER[a][1] - m1) * vs - o1,
                                     (float)(z + CORNER[a][2] - m2) * vs - o2 };
                const gsdf_v3 wb = { (float)(x + CORNER[b][0] - m0) * vs - o0, (float)(y + CORNER[b][1] - m1) * vs - o1,
                                     (float)(z + CORNER[b][2] - m2) * vs - o2 };
                p[v] = mesh_interpolate(d[a], d[b], wa, wb, iso);
            }
            auto same = [](const gsdf_v3& a, const gsdf_v3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
            if (same(p[0], p[1]) || same(p[0], p[2]) || same(p[1], p[2])) continue;      /* computeTriangles (:686-712) */
            const unsigned long long o = atomicAdd(counter, 1ull);
            if ((long long)o >= max_tris) continue;
            float* out = tris + 9 * o;
#pragma unroll
            for (int v = 0; v < 3; ++v) { out[3 * v] = p[v].x; out[3 * v + 1] = p[v].y; out[3 * v + 2] = p[v].z; }
            /* sweep order of the reference: z, then y, then x (relative to the bounding-box minimum, 20 bits each),
             * then the triangle number within the cube */
            keys[o] = ((((unsigned long long)(uint32_t)(z - m2) << 40) | ((unsigned long long)(uint32_t)(y - m1) << 20) |
                        (unsigned long long)(uint32_t)(x - m0)) << 3) | (unsigned long long)(k / 3);
        }
    }
}
void gsdf_launch_mesh(hipStream_t s, gsdf_table tab, size_t n_slots, float vs, float iso, int* mn_dev, const signed char* tri_table_dev,
                      float* tris_dev, unsigned long long* keys_dev, unsigned long long* counter, long long max_tris) {
    hipLaunchKernelGGL(k_mesh_bbox, dim3(1024), dim3(256), 0, s, tab, n_slots, mn_dev);
    hipLaunchKernelGGL(k_mesh, dim3(2048), dim3(256), 0, s, tab, n_slots, vs, iso, mn_dev, tri_table_dev, tris_dev, keys_dev, counter,
                       max_tris);
}

/* ------------------------------------------------------------------------------------------------
 * Dense block exchange for the frame-sharded fusion: the ranks agree on the union of their block
 * keys, every rank packs its raw sums (w, s, gx, gy, gz per voxel; zeros where it has nothing) for those blocks
 * into one dense buffer, RCCL all-reduces the buffer (sum), and every rank stores the result.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void k_block_keys(gsdf_table tab, size_t n_blocks, unsigned long long* out,
                                                     unsigned long long* counter, long long max_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_blocks; i += stride) {
        const unsigned long long bk = tab.bkeys[i];
        if (bk == GSDF_KEY_EMPTY) continue;
        const unsigned long long o = atomicAdd(counter, 1ull);
        if ((long long)o < max_n) out[o] = bk;
    }
}
/* one 64-lane wavefront per block: lane = voxel of the block */
__global__ __launch_bounds__(256) void k_pack_blocks(gsdf_table tab, const unsigned long long* __restrict__ keys, long long n,
                                                      float* __restrict__ dense) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = keys[b];
    const uint32_t h = gsdf_hash(bk) & tab.block_mask;
    const int blk = gsdf_block_find(tab, bk, h, tab.bkeys[h]);
    float v[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
    if (blk >= 0) {
        const gsdf_payload p = tab.vox[(size_t)blk * GSDF_BLOCK_VOX + lane];
        v[0] = p.w; v[1] = p.s; v[2] = p.gx; v[3] = p.gy; v[4] = p.gz;
    }
    float* o = dense + ((size_t)b * GSDF_BLOCK_VOX + lane) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = v[k];
}
__global__ __launch_bounds__(256) void k_unpack_blocks(gsdf_table tab, const unsigned long long* __restrict__ keys, long long n,
                                                        const float* __restrict__ dense, gsdf_dev_state* st) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = keys[b];
    const uint32_t h = gsdf_hash(bk) & tab.block_mask;
    int blk = 0;
    if (lane == 0) blk = gsdf_block_find_or_insert(tab, bk, h, tab.bkeys[h]);
    blk = __shfl(blk, 0);
    if (blk < 0) { if (lane == 0) atomicOr(&st->status, GSDF_STATUS_TABLE_FULL); return; }
    const float* v = dense + ((size_t)b * GSDF_BLOCK_VOX + lane) * 5;
    gsdf_payload p;
    p.w = v[0]; p.s = v[1]; p.gx = v[2]; p.gy = v[3]; p.gz = v[4]; p.aux = 0u; p.pad[0] = 0u; p.pad[1] = 0u;
    tab.vox[(size_t)blk * GSDF_BLOCK_VOX + lane] = p;
}
/* vis_ of the exchange (MapGradPixelSdf.cpp:113-115: bit f of a voxel = "updated by integrated frame f"): every rank numbers
 * its own frames from 0, so its bit-vectors are shifted by the frames of the ranks before it (contiguous frame shards in rank
 * order); the shifted vectors of different ranks have no bit in common, so their SUM as unsigned words is their OR.
 * One wavefront per block of the union, lane = voxel, `vw` words per voxel. */
__global__ __launch_bounds__(256) void k_pack_vis(gsdf_table tab, const uint32_t* __restrict__ vis, int vw, long long bit_offset,
                                                   const unsigned long long* __restrict__ keys, long long n, uint32_t* __restrict__ dense) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = keys[b];
    const uint32_t h = gsdf_hash(bk) & tab.block_mask;
    const int blk = gsdf_block_find(tab, bk, h, tab.bkeys[h]);
    uint32_t* o = dense + ((size_t)b * GSDF_BLOCK_VOX + lane) * vw;
    const uint32_t* in = blk >= 0 ? vis + ((size_t)blk * GSDF_BLOCK_VOX + lane) * vw : nullptr;
    const int q = (int)(bit_offset >> 5), sh = (int)(bit_offset & 31);
    for (int w = 0; w < vw; ++w) {
        uint32_t v = 0u;
        if (in) {
            if (w - q >= 0) v = in[w - q] << sh;
            if (sh && w - q - 1 >= 0) v |= in[w - q - 1] >> (32 - sh);
        }
        o[w] = v;
    }
}
__global__ __launch_bounds__(256) void k_unpack_vis(gsdf_table tab, uint32_t* __restrict__ vis, int vw,
                                                     const unsigned long long* __restrict__ keys, long long n,
                                                     const uint32_t* __restrict__ dense) {
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long bk = keys[b];
    const uint32_t h = gsdf_hash(bk) & tab.block_mask;
    const int blk = gsdf_block_find(tab, bk, h, tab.bkeys[h]);       /* inserted by k_unpack_blocks */
    if (blk < 0) return;
    const uint32_t* in = dense + ((size_t)b * GSDF_BLOCK_VOX + lane) * vw;
    uint32_t* o = vis + ((size_t)blk * GSDF_BLOCK_VOX + lane) * vw;
    for (int w = 0; w < vw; ++w) o[w] = in[w];
}
/* Sdf::counter_ after the exchange = frames integrated by all ranks */
__global__ void k_set_frames(gsdf_dev_state* st, long long frames) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->frames = frames; st->frame_cur = frames; }
}
void gsdf_launch_pack_vis(hipStream_t s, gsdf_table tab, const uint32_t* vis, int vw, long long bit_offset,
                          const unsigned long long* keys, long long n, uint32_t* dense) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack_vis, dim3((unsigned int)((n + 3) / 4)), dim3(256), 0, s, tab, vis, vw, bit_offset, keys, n, dense);
}
void gsdf_launch_unpack_vis(hipStream_t s, gsdf_table tab, uint32_t* vis, int vw, const unsigned long long* keys, long long n,
                            const uint32_t* dense) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpack_vis, dim3((unsigned int)((n + 3) / 4)), dim3(256), 0, s, tab, vis, vw, keys, n, dense);
}
void gsdf_launch_set_frames(hipStream_t s, gsdf_dev_state* st, long long frames) {
    hipLaunchKernelGGL(k_set_frames, dim3(1), dim3(64), 0, s, st, frames);
}
void gsdf_launch_block_keys(hipStream_t s, gsdf_table tab, size_t n_blocks, unsigned long long* out, unsigned long long* counter,
                            long long max_n) {
    hipLaunchKernelGGL(k_block_keys, dim3(256), dim3(256), 0, s, tab, n_blocks, out, counter, max_n);
}
void gsdf_launch_pack_blocks(hipStream_t s, gsdf_table tab, const unsigned long long* keys, long long n, float* dense) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned int)((n + 3) / 4)), dim3(256), 0, s, tab, keys, n, dense);
}
void gsdf_launch_unpack_blocks(hipStream_t s, gsdf_table tab, const unsigned long long* keys, long long n, const float* dense,
                               gsdf_dev_state* st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpack_blocks, dim3((unsigned int)((n + 3) / 4)), dim3(256), 0, s, tab, keys, n, dense, st);
}
