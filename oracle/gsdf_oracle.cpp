/*
 * gsdf_oracle.cpp -- CPU ORACLE (test infrastructure, see gsdf_oracle.h).
 *
 * Dependency-free C++17 restatement of the reference's SERIAL hot path
 * (c-sommer/gradient-sdf).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/cpp/include/).  Build with
 *   g++ -O2 -ffp-contract=off -fopenmp        (see oracle/Makefile)
 * which mirrors the reference's RelWithDebInfo x86-64 build (SSE2 scalar float
 * arithmetic, no FMA contraction; cpp/CMakeLists.txt:4,60).
 *
 * PARITY UNPINNED by the reference (no tests / fixtures exist there).  The float
 * operation orders written here are the definition the HIP kernels copy:
 *   - 3-term reductions (3x3*3 products, dot, squaredNorm) are  x0 + (x1 + x2),
 *     the order produced by Eigen 3.4's unrolled scalar redux (redux_novec_unroller
 *     splits [0,3) into [0,1) and [1,3));  Eigen itself is absent here.
 *   - box filter = separable double-precision sums, row pass (ascending dx) then
 *     column pass (ascending dy), BORDER_REFLECT_101, result rounded to float.
 *     OpenCV's boxFilter accumulates in double (sumType CV_64F) with RUNNING sums (RowSum:
 *     s += S[i+k] - S[i]; ColumnSum: SUM += Sp, SUM -= Sm down the image).  The cached planes
 *     of NormalEstimator::cache follow that order (box_sum mode 1: the matrix inverse behind
 *     them amplifies the sums' last bits); the per-frame filters of ::compute form a FRESH sum
 *     of their 11 inputs per output (mode 0), which is NOT OpenCV's order but gives the same
 *     floats (the two differ by ~1e-16 relative before the rounding to float; counted in
 *     tests/test_oracle_known_answers.py).
 *   - H.llt().solve(g): Eigen's unblocked llt_inplace + unrolled triangular solves in
 *     their published operation order (see llt_solve6); the packetisation of Eigen's
 *     reductions under the reference's flags is not knowable here.
 * "Bit-exact occupancy" everywhere in this repository therefore means bit-exact against
 * THIS reading of the absent libraries' arithmetic.
 */
#include "gsdf_oracle.h"
#include "../include/gsdf_mc_tables.h"   /* constant data: the classic marching-cubes case tables */

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <limits>
#include <array>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V3 { float x, y, z; };

/* Eigen 3.4 scalar redux of 3 terms: x0 + (x1 + x2) */
static inline float sum3(float a, float b, float c) { return a + (b + c); }
static inline float dot3(V3 a, V3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 matvec(const float R[9], V3 v) {           /* R row-major */
    V3 r;
    r.x = sum3(R[0] * v.x, R[1] * v.y, R[2] * v.z);
    r.y = sum3(R[3] * v.x, R[4] * v.y, R[5] * v.z);
    r.z = sum3(R[6] * v.x, R[7] * v.y, R[8] * v.z);
    return r;
}
static inline V3 cross3(V3 a, V3 b) {                        /* Eigen cross3 */
    V3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
/* Eigen MatrixBase::normalized(): n / sqrt(|n|^2) if |n|^2 > 0 else n */
static inline V3 normalized3(V3 n) {
    float z = sum3(n.x * n.x, n.y * n.y, n.z * n.z);
    if (z > 0.f) {
        float s = std::sqrt(z);
        V3 r = { n.x / s, n.y / s, n.z / s };
        return r;
    }
    return n;
}

/* SdfVoxel -- sdf_voxel/SdfVoxel.h:45-57 (20-byte payload, zero-initialised) */
struct SdfVoxel {
    float dist = 0.f;
    float grad[3] = { 0.f, 0.f, 0.f };
    float weight = 0.f;
};

struct Key {
    int32_t x, y, z;
    bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
};
/* hash is NOT a parity item: only container iteration order depends on it */
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = (uint64_t)(uint32_t)k.x * 0x9E3779B97F4A7C15ull;
        h ^= (uint64_t)(uint32_t)k.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
        h ^= (uint64_t)(uint32_t)k.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
        h ^= h >> 29;
        return (size_t)h;
    }
};

static inline int reflect101(int i, int n) {                 /* cv::BORDER_REFLECT_101 */
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* unnormalised box sum, double accumulation, separable (row then column).
 * mode 0: every output is a fresh sum of its 11 (win) inputs, ascending offset.
 * mode 1: the summation order of OpenCV 4's generic FilterEngine
 *   path for cv::boxFilter(.., normalize=false) on CV_32F / CV_64F (imgproc/src/box_filter.simd.hpp; sumType CV_64F):
 *   RowSum      s = S[0] + .. + S[k-1];  D[0] = s;  then  s += S[i+k] - S[i];  D[i+1] = s        (running, per row)
 *   ColumnSum   SUM = rows[0] + .. + rows[k-2];  per output row  s0 = SUM + Sp;  D = s0;  SUM = s0 - Sm   (running,
 *               down the whole image: the filter object keeps SUM between FilterEngine::proceed calls)
 *   over the BORDER_REFLECT_101-extended rows / row list.  Whether a reference build takes this path (and not IPP's or an
 *   OpenCL one) depends on how its OpenCV was built -- not knowable here; the library is absent.
 * DEFINITION the HIP kernels copy: the one-off cached planes (NormalEstimator::cache) use mode 1 -- Q = M^-1 amplifies the
 * last bits of the sums by 1e8 and more, the order matters there (10 % of the Q floats, normals up to 1e-2) -- and the
 * per-frame filters of NormalEstimator::compute use mode 0, which gives the same floats as mode 1 there (a 1e-16 difference
 * in double only shows in rare double roundings; counted by tests/test_oracle_known_answers.py). */
template <typename Tin>
static void box_sum(const Tin* src, double* dst, int W, int H, int win, int mode = 0) {
    const int r = win / 2;
    std::vector<double> rows((size_t)W * H);
    if (mode == 1) {
        std::vector<double> ext((size_t)W + 2 * r), SUM((size_t)W);
        for (int y = 0; y < H; ++y) {
            for (int i = 0; i < W + 2 * r; ++i) ext[i] = (double)src[(size_t)y * W + reflect101(i - r, W)];
            double s = 0.0;
            for (int i = 0; i < win; ++i) s += ext[i];
            rows[(size_t)y * W] = s;
            for (int i = 0; i < W - 1; ++i) {
                s += ext[i + win] - ext[i];
                rows[(size_t)y * W + i + 1] = s;
            }
        }
        std::fill(SUM.begin(), SUM.end(), 0.0);
        for (int j = 0; j < win - 1; ++j) {
            const double* Sp = &rows[(size_t)reflect101(j - r, H) * W];
            for (int x = 0; x < W; ++x) SUM[x] += Sp[x];
        }
        for (int y = 0; y < H; ++y) {
            const double* Sp = &rows[(size_t)reflect101(y + r, H) * W];
            const double* Sm = &rows[(size_t)reflect101(y - r, H) * W];
            for (int x = 0; x < W; ++x) {
                const double s0 = SUM[x] + Sp[x];
                dst[(size_t)y * W + x] = s0;
                SUM[x] = s0 - Sm[x];
            }
        }
        return;
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s = 0.0;
            for (int dx = -r; dx <= r; ++dx)
                s += (double)src[(size_t)y * W + reflect101(x + dx, W)];
            rows[(size_t)y * W + x] = s;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s = 0.0;
            for (int dy = -r; dy <= r; ++dy)
                s += rows[(size_t)reflect101(y + dy, H) * W + x];
            dst[(size_t)y * W + x] = s;
        }
}

} // namespace

struct gsdfo {
    /* Sdf.h:63-68 */
    float T_, inv_T_;
    size_t counter_ = 0;
    float z_min_ = 0.5f, z_max_ = 3.5f;
    /* MapGradPixelSdf.h:61-70 */
    float voxel_size_, voxel_size_inv_;
    std::unordered_map<Key, SdfVoxel, KeyHash> tsdf_;
    std::unordered_map<Key, std::vector<bool>, KeyHash> vis_;
    /* NormalEstimator.h:55-64 */
    int W = 0, H = 0, win = 0;
    std::vector<float> x0_, y0_, x0n_, y0n_, ninv_, Q11_, Q12_, Q13_, Q22_, Q23_, Q33_;
    int threads = 4;
    /* summation order of the box filters (see box_sum): the cached planes follow OpenCV's running sums, the per-frame
     * filters sum freshly (measured to give the same floats); the other settings exist for measuring the difference */
    int box_mode_cache = 1, box_mode_frame = 0;
    int nsq_order = 2;           /* NormalEstimator.h:104 `1. + x0_sq + y0_sq` through cv::MatExpr: see gsdfo_set_nsq_order */

    /* Sdf::truncate -- Sdf.h:72-74 */
    float truncate(float sdf) const { return std::max(-T_, std::min(T_, sdf)); }
    /* Sdf::weight -- Sdf.h:76-85 */
    float weight(float sdf) const {
        float w = 0.f;
        if (sdf <= 0.) w = 1.f;
        else if (sdf <= T_) w = 1.f - sdf * inv_T_;
        return w;
    }
    /* MapGradPixelSdf::float2vox -- MapGradPixelSdf.h:74-77 (std::round: half away from zero) */
    Key float2vox(V3 p) const {
        Key k;
        k.x = (int32_t)std::round(voxel_size_inv_ * p.x);
        k.y = (int32_t)std::round(voxel_size_inv_ * p.y);
        k.z = (int32_t)std::round(voxel_size_inv_ * p.z);
        return k;
    }
    /* MapGradPixelSdf::vox2float -- MapGradPixelSdf.h:79-81 */
    V3 vox2float(Key k) const {
        V3 r = { voxel_size_ * (float)k.x, voxel_size_ * (float)k.y, voxel_size_ * (float)k.z };
        return r;
    }
};

extern "C" {

gsdfo* gsdfo_create(float voxel_size, float T) {
    gsdfo* o = new gsdfo();
    o->T_ = T;
    o->inv_T_ = (float)(1. / T);                      /* Sdf.h:103-107 */
    o->voxel_size_ = voxel_size;
    o->voxel_size_inv_ = (float)(1. / voxel_size);    /* MapGradPixelSdf.h:99-103 */
    return o;
}
void gsdfo_destroy(gsdfo* o) { delete o; }
void gsdfo_set_zrange(gsdfo* o, float zmin, float zmax) { o->z_min_ = zmin; o->z_max_ = zmax; }
void gsdfo_set_threads(gsdfo* o, int threads) { o->threads = threads > 0 ? threads : 1; }
/* NormalEstimator.h:104 `n_sq = 1. + x0_sq + y0_sq` is a cv::MatExpr: OpenCV folds `(1. + A) + B` into ONE
 * addWeighted(A, 1, B, 1, gamma = 1), whose scalar loop evaluates (A*1 + B*1) + gamma and whose SIMD loop (4.x, CV_SIMD_64F)
 * fma(A, 1, fma(B, 1, gamma)) -- three candidate orders of two double additions, decided inside an absent, unpinned library:
 *   0 (1 + x^2) + y^2   what the source line would mean for plain doubles; NOT what a lazy MatExpr evaluates
 *   1 (x^2 + y^2) + 1   addWeighted, scalar loop (OpenCV <= 4.1, builds without CV_SIMD_64F, the last N % lanes elements)
 *   2 x^2 + (y^2 + 1)   addWeighted, SIMD loop (OpenCV 4.2+: op_add_weighted<double>, v_fma(a, alpha, v_fma(b, beta, gamma));
 *                       multiplying by 1 is exact, so a fused and an unfused v_fma give the same doubles) -- the DEFINITION
 *                       (round 6; the reference's README asks for OpenCV 4 and its Docker image follows a current 4.x), also
 *                       k_ncache_rows / k_ncache_cols on the GPU.  The choice decides last bits of Q = M^-1: 18 % of its floats,
 *                       normals up to 1e-2, a few gate flips per frame (tests/test_oracle_known_answers.py measures all three).
 * The vector loop's scalar tail (N % lanes elements in order 1) is not modelled: N is a multiple of 4 in every BASELINE config. */
void gsdfo_set_nsq_order(gsdfo* o, int order) { o->nsq_order = order >= 0 && order <= 2 ? order : 2; }
void gsdfo_set_box_mode(gsdfo* o, int cache_mode, int frame_mode) { o->box_mode_cache = cache_mode == 1; o->box_mode_frame = frame_mode == 1; }

/* NormalEstimator::cache -- NormalEstimator.h:81-154 (all in double, then cast to float) */
int gsdfo_normals_init(gsdfo* o, int W, int H, const float K[9], int win) {
    if (W <= 0 || H <= 0 || win <= 0 || !(win & 1)) return -1;
    o->W = W; o->H = H; o->win = win;
    const size_t N = (size_t)W * H;
    const double fx_inv = 1. / (double)K[0];
    const double fy_inv = 1. / (double)K[4];
    const double cx = (double)K[2];
    const double cy = (double)K[5];
    std::vector<double> x0(N), y0(N), ninv(N), x0n(N), y0n(N);
    std::vector<double> m11(N), m12(N), m13(N), m22(N), m23(N), m33(N);
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const size_t i = (size_t)v * W + u;
            const double x = fx_inv * ((double)u - cx);      /* :94,98 */
            const double y = fy_inv * ((double)v - cy);      /* :96,100 */
            const double x_sq = x * x, y_sq = y * y, xy = x * y;
            const double n_sq = o->nsq_order == 1 ? (x_sq + y_sq) + 1. : o->nsq_order == 2 ? x_sq + (y_sq + 1.)
                                                                         : 1. + x_sq + y_sq;            /* :104 (gsdfo_set_nsq_order) */
            const double ni = 1. / n_sq;                     /* :105 */
            x0[i] = x; y0[i] = y; ninv[i] = ni;
            x0n[i] = x * ni; y0n[i] = y * ni;                /* :106-107 */
            m11[i] = x_sq * ni; m12[i] = xy * ni; m22[i] = y_sq * ni;   /* :109,110,112 */
        }
    std::vector<double> M11(N), M12(N), M13(N), M22(N), M23(N), M33(N);
    box_sum(m11.data(), M11.data(), W, H, win, o->box_mode_cache);              /* :109-114 */
    box_sum(m12.data(), M12.data(), W, H, win, o->box_mode_cache);
    box_sum(x0n.data(), M13.data(), W, H, win, o->box_mode_cache);
    box_sum(m22.data(), M22.data(), W, H, win, o->box_mode_cache);
    box_sum(y0n.data(), M23.data(), W, H, win, o->box_mode_cache);
    box_sum(ninv.data(), M33.data(), W, H, win, o->box_mode_cache);
    o->x0_.resize(N); o->y0_.resize(N); o->x0n_.resize(N); o->y0n_.resize(N); o->ninv_.resize(N);
    o->Q11_.resize(N); o->Q12_.resize(N); o->Q13_.resize(N); o->Q22_.resize(N); o->Q23_.resize(N); o->Q33_.resize(N);
    for (size_t i = 0; i < N; ++i) {
        const double a11 = M11[i], a12 = M12[i], a13 = M13[i], a22 = M22[i], a23 = M23[i], a33 = M33[i];
        /* :116-118 */
        const double det = a11 * (a22 * a33) + 2 * (a12 * (a23 * a13))
                         - (a13 * (a13 * a22) + a12 * (a12 * a33) + a23 * (a23 * a11));
        const double det_inv = 1. / det;
        o->Q11_[i] = (float)(det_inv * (a22 * a33 - a23 * a23));   /* :120-125 */
        o->Q12_[i] = (float)(det_inv * (a13 * a23 - a12 * a33));
        o->Q13_[i] = (float)(det_inv * (a12 * a23 - a13 * a22));
        o->Q22_[i] = (float)(det_inv * (a11 * a33 - a13 * a13));
        o->Q23_[i] = (float)(det_inv * (a12 * a13 - a11 * a23));
        o->Q33_[i] = (float)(det_inv * (a11 * a22 - a12 * a12));
        o->x0_[i] = (float)x0[i]; o->y0_[i] = (float)y0[i];        /* :128-132 */
        o->x0n_[i] = (float)x0n[i]; o->y0n_[i] = (float)y0n[i]; o->ninv_[i] = (float)ninv[i];
    }
    return 0;
}

void gsdfo_normals_cache(const gsdfo* o, float* p) {
    const size_t N = (size_t)o->W * o->H;
    const std::vector<float>* v[11] = { &o->x0_, &o->y0_, &o->x0n_, &o->y0n_, &o->ninv_,
        &o->Q11_, &o->Q12_, &o->Q13_, &o->Q22_, &o->Q23_, &o->Q33_ };
    for (int k = 0; k < 11; ++k) std::memcpy(p + k * N, v[k]->data(), N * sizeof(float));
}

/* NormalEstimator::compute -- NormalEstimator.h:179-204 */
void gsdfo_normals_compute(const gsdfo* o, const float* depth, float* nx, float* ny, float* nz) {
    const int W = o->W, H = o->H;
    const size_t N = (size_t)W * H;
    std::vector<float> p1(N), p2(N), p3(N);
    for (size_t i = 0; i < N; ++i) {
        const float zi = depth[i] != 0.f ? 1.f / depth[i] : 0.f;   /* :183-187 */
        p1[i] = o->x0n_[i] * zi;                                   /* :191-193 (.mul) */
        p2[i] = o->y0n_[i] * zi;
        p3[i] = o->ninv_[i] * zi;
    }
    std::vector<double> b1(N), b2(N), b3(N);
    box_sum(p1.data(), b1.data(), W, H, o->win, o->box_mode_frame);
    box_sum(p2.data(), b2.data(), W, H, o->win, o->box_mode_frame);
    box_sum(p3.data(), b3.data(), W, H, o->win, o->box_mode_frame);
    for (size_t i = 0; i < N; ++i) {
        const float c1 = (float)b1[i], c2 = (float)b2[i], c3 = (float)b3[i];
        const float x = (c1 * o->Q11_[i] + c2 * o->Q12_[i]) + c3 * o->Q13_[i];   /* :195-197 */
        const float y = (c1 * o->Q12_[i] + c2 * o->Q22_[i]) + c3 * o->Q23_[i];
        const float z = (c1 * o->Q13_[i] + c2 * o->Q23_[i]) + c3 * o->Q33_[i];
        const float n = std::sqrt((x * x + y * y) + z * z);                      /* :199 */
        nx[i] = x / n; ny[i] = y / n; nz[i] = z / n;                             /* :201-203, IEEE: 0/0 = NaN */
    }
}

/* MapGradPixelSdf::update -- MapGradPixelSdf.cpp:43-122 (omp: MapGradPixelSdfOmp.cpp:44-130).
 * cv::medianBlur at :53 is dead work (result never read) and is not restated. */
int64_t gsdfo_update(gsdfo* o, const float* depth, const float R[9], const float t[3],
                     int omp, int64_t* n_valid_out) {
    const int W = o->W, H = o->H;
    const size_t N = (size_t)W * H;
    std::vector<float> nx(N), ny(N), nz(N);
    gsdfo_normals_compute(o, depth, nx.data(), ny.data(), nz.data());   /* :60 */

    const float z_min = o->z_min_, z_max = o->z_max_;
    const float vs = o->voxel_size_;
    const V3 tv = { t[0], t[1], t[2] };
    const int factor = (int)std::floor(o->T_ / vs);                     /* :79 */
    int64_t n_upd = 0, n_valid = 0;

    auto body = [&](size_t idx, int64_t& upd, int64_t& valid, bool locked) {
        const float z = depth[idx];
        if (z <= z_min || z >= z_max) return;                            /* :87 */
        const V3 xy_hom = { o->x0_[idx], o->y0_[idx], 1.f };            /* :90 */
        const V3 R_xy_hom = matvec(R, xy_hom);                          /* :91 */
        const V3 normal = { nx[idx], ny[idx], nz[idx] };                /* :92 */
        const V3 Rn = matvec(R, normal);                                /* :93 */
        if ((double)dot3(normal, normal) < .1) return;                  /* :95 */
        const float nd = dot3(normal, xy_hom);
        if (nd * nd * o->ninv_[idx] < .25) return;                      /* :98 */
        ++valid;
        for (float k = (float)-factor; k <= (float)factor; ++k) {       /* :101 */
            const float s = z + k * vs;
            V3 point = { s * R_xy_hom.x + tv.x, s * R_xy_hom.y + tv.y, s * R_xy_hom.z + tv.z };  /* :103 */
            const Key vi = o->float2vox(point);                         /* :104 */
            const V3 c = o->vox2float(vi);
            const V3 d = { c.x - tv.x, c.y - tv.y, c.z - tv.z };
            /* point = Rt * d, only component 2 is used (:105-106); Rt(2,j) = R(j,2) */
            const float pz = sum3(R[2] * d.x, R[5] * d.y, R[8] * d.z);
            const float sdf = pz - z;
            const float w = o->weight(sdf);                             /* :107 */
            if (w > 0) {
                auto apply = [&]() {
                    SdfVoxel& v = o->tsdf_[vi];                         /* :109 */
                    v.weight += w;                                      /* :110 */
                    v.dist += (o->truncate(sdf) - v.dist) * w / v.weight;   /* :111 */
                    v.grad[0] += w * Rn.x;                              /* :112 */
                    v.grad[1] += w * Rn.y;
                    v.grad[2] += w * Rn.z;
                    std::vector<bool>& vis = o->vis_[vi];               /* :113-115 */
                    vis.resize(o->counter_);
                    vis.push_back(true);
                };
                if (locked) {
#ifdef _OPENMP
#pragma omp critical(gsdfo_fuse)
#endif
                    apply();
                } else apply();
                ++upd;
            }
        }
    };

    if (omp) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(o->threads) reduction(+ : n_upd, n_valid) schedule(static)
#endif
        for (int64_t i = 0; i < (int64_t)N; ++i) body((size_t)i, n_upd, n_valid, true);
    } else {
        for (size_t i = 0; i < N; ++i) body(i, n_upd, n_valid, false);   /* :81 row-major */
    }
    ++o->counter_;                                                       /* :120 */
    if (n_valid_out) *n_valid_out = n_valid;
    return n_upd;
}

int64_t gsdfo_count(const gsdfo* o) { return (int64_t)o->tsdf_.size(); }
int64_t gsdfo_frame_counter(const gsdfo* o) { return (int64_t)o->counter_; }

static std::vector<Key> sorted_keys(const gsdfo* o) {
    std::vector<Key> ks;
    ks.reserve(o->tsdf_.size());
    for (const auto& kv : o->tsdf_) ks.push_back(kv.first);
    std::sort(ks.begin(), ks.end(), [](const Key& a, const Key& b) {
        if (a.z != b.z) return a.z < b.z;
        if (a.y != b.y) return a.y < b.y;
        return a.x < b.x;
    });
    return ks;
}

void gsdfo_export(const gsdfo* o, int32_t* keys, float* payload) {
    std::vector<Key> ks = sorted_keys(o);
    for (size_t i = 0; i < ks.size(); ++i) {
        const SdfVoxel& v = o->tsdf_.at(ks[i]);
        keys[3 * i + 0] = ks[i].x; keys[3 * i + 1] = ks[i].y; keys[3 * i + 2] = ks[i].z;
        payload[5 * i + 0] = v.dist;
        payload[5 * i + 1] = v.grad[0]; payload[5 * i + 2] = v.grad[1]; payload[5 * i + 3] = v.grad[2];
        payload[5 * i + 4] = v.weight;
    }
}

void gsdfo_export_vis(const gsdfo* o, uint32_t* words, int wpv) {
    std::vector<Key> ks = sorted_keys(o);
    for (size_t i = 0; i < ks.size(); ++i) {
        for (int w = 0; w < wpv; ++w) words[i * wpv + w] = 0u;
        auto it = o->vis_.find(ks[i]);
        if (it == o->vis_.end()) continue;
        const std::vector<bool>& b = it->second;
        for (size_t f = 0; f < b.size() && f < (size_t)wpv * 32; ++f)
            if (b[f]) words[i * wpv + f / 32] |= 1u << (f % 32);
    }
}

/* test plumbing: replace the map by the given (key, SdfVoxel) pairs, so that the export / query / mesh restatements can
 * be run on exactly the voxel values another implementation produced (payload = dist, gx, gy, gz, weight) */
void gsdfo_set_map(gsdfo* o, const int32_t* keys, const float* payload, int64_t n) {
    o->tsdf_.clear();
    o->vis_.clear();
    for (int64_t i = 0; i < n; ++i) {
        SdfVoxel v;
        v.dist = payload[5 * i];
        v.grad[0] = payload[5 * i + 1]; v.grad[1] = payload[5 * i + 2]; v.grad[2] = payload[5 * i + 3];
        v.weight = payload[5 * i + 4];
        o->tsdf_[Key{ keys[3 * i], keys[3 * i + 1], keys[3 * i + 2] }] = v;
    }
}

/* test plumbing: overwrite the SdfVoxel of EXISTING voxels (vis_ and the key set stay): lets the PhotoBA restatement run on
 * exactly the voxel values another implementation fused.  Returns the number of keys that were not in the map. */
int64_t gsdfo_set_payload(gsdfo* o, const int32_t* keys, const float* payload, int64_t n) {
    int64_t missing = 0;
    for (int64_t i = 0; i < n; ++i) {
        auto it = o->tsdf_.find(Key{ keys[3 * i], keys[3 * i + 1], keys[3 * i + 2] });
        if (it == o->tsdf_.end()) { ++missing; continue; }
        it->second.dist = payload[5 * i];
        it->second.grad[0] = payload[5 * i + 1]; it->second.grad[1] = payload[5 * i + 2]; it->second.grad[2] = payload[5 * i + 3];
        it->second.weight = payload[5 * i + 4];
    }
    return missing;
}

/* ---- exports: MapGradPixelSdf::extract_pc and LayeredMarchingCubesNoColor -------------------------------- */

/* MapGradPixelSdf::extract_pc -- MapGradPixelSdf.cpp:177-220.  Rows: x y z nx ny nz; voxels are visited in
 * (z,y,x) order (the reference's phmap order is unknowable).  rows6 == NULL only counts. */
int64_t gsdfo_extract_pc(const gsdfo* o, float* rows6) {
    const float voxel_size_2 = (float)(.5 * o->voxel_size_);                        /* :179 */
    int64_t n = 0;
    for (const Key& k : sorted_keys(o)) {
        const SdfVoxel& v = o->tsdf_.at(k);
        if (v.weight < 5) continue;                                                 /* :184-185 */
        const V3 gn = normalized3(V3{ v.grad[0], v.grad[1], v.grad[2] });
        const V3 g = { 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };                     /* :186 */
        const V3 d = { v.dist * g.x, v.dist * g.y, v.dist * g.z };                  /* :187 */
        if (std::fabs(d.x) < voxel_size_2 && std::fabs(d.y) < voxel_size_2 && std::fabs(d.z) < voxel_size_2) {   /* :188 */
            if (rows6) {
                const V3 c = o->vox2float(k);
                float* r = rows6 + 6 * n;
                r[0] = c.x - d.x; r[1] = c.y - d.y; r[2] = c.z - d.z;               /* :191 */
                r[3] = -g.x; r[4] = -g.y; r[5] = -g.z;                              /* :192 */
            }
            ++n;
        }
    }
    return n;
}

namespace {
/* LayeredMarchingCubesNoColor -- mesh/LayeredMarchingCubesNoColor.cpp:354-712, on the oracle's map.
 * State and method names follow the reference; the case tables are the constant data of
 * include/gsdf_mc_tables.h (= :67-352). */
struct LayeredMC {
    const gsdfo* o;
    float vs;
    int min_[3], dim_[3];
    float origin_[3];
    size_t areaXY_;
    std::vector<float> tsdf_, weights_;
    std::vector<float> tris;                                  /* 9 floats per face, in emission order */

    /* copyLayer -- :565-587: a missing voxel only clears the weight; its tsdf entry stays stale */
    void copyLayer(int z) {
        for (int y = 0; y < dim_[1]; ++y)
            for (int x = 0; x < dim_[0]; ++x) {
                const size_t off = ((size_t)(z % 2) * dim_[1] + y) * dim_[0] + x;
                auto it = o->tsdf_.find(Key{ x + min_[0], y + min_[1], z + min_[2] });
                if (it != o->tsdf_.end()) { weights_[off] = it->second.weight; tsdf_[off] = it->second.dist; }
                else weights_[off] = 0;
            }
    }
    /* computeLutIndex -- :593-639 */
    int computeLutIndex(int i, int j, int k, float iso) const {
        const size_t offZ = (size_t)(k % 2) * areaXY_, offZp = areaXY_ - offZ, offY = (size_t)dim_[0];
        const size_t off[8] = { offZ + (j + 1) * offY + (i + 1), offZ + j * offY + (i + 1), offZ + j * offY + i, offZ + (j + 1) * offY + i,
                                offZp + (j + 1) * offY + (i + 1), offZp + j * offY + (i + 1), offZp + j * offY + i, offZp + (j + 1) * offY + i };
        for (int c = 0; c < 8; ++c) if (weights_[off[c]] == 0.0f) return 0;
        int cubeIdx = 0;
        for (int c = 0; c < 8; ++c) if (tsdf_[off[c]] > iso) cubeIdx |= 1 << c;
        return cubeIdx;
    }
    /* voxelToWorld -- :715-719 */
    V3 voxelToWorld(int i, int j, int k) const {
        return V3{ (float)i * vs - origin_[0], (float)j * vs - origin_[1], (float)k * vs - origin_[2] };
    }
    /* interpolate -- :642-662 (float differences, double comparisons, double mu) */
    static V3 interpolate(float tsdf0, float tsdf1, V3 val0, V3 val1, float iso) {
        if (std::fabs(iso - tsdf0) < 1e-7) return val0;
        if (std::fabs(iso - tsdf1) < 1e-7) return val1;
        if (std::fabs(tsdf0 - tsdf1) < 1e-7) return val0;
        double mu = (iso - tsdf0) / (tsdf1 - tsdf0);
        if (mu > 1.0) mu = 1.0;
        else if (mu < 0) mu = 0.0;
        V3 val;
        val.x = (float)(val0.x + mu * (val1.x - val0.x));
        val.y = (float)(val0.y + mu * (val1.y - val0.y));
        val.z = (float)(val0.z + mu * (val1.z - val0.z));
        return val;
    }
    /* getVertex -- :665-672 */
    V3 getVertex(int i1, int j1, int k1, int i2, int j2, int k2, float iso) const {
        const float v1 = tsdf_[(size_t)(k1 % 2) * dim_[0] * dim_[1] + (size_t)j1 * dim_[0] + i1];
        const float v2 = tsdf_[(size_t)(k2 % 2) * dim_[0] * dim_[1] + (size_t)j2 * dim_[0] + i2];
        return interpolate(v1, v2, voxelToWorld(i1, j1, k1), voxelToWorld(i2, j2, k2), iso);
    }
    /* computeTriangles -- :675-704: no vertex sharing; a face with two equal corners is dropped */
    void computeTriangles(int cubeIndex, const V3 edgePoints[12]) {
        const int8_t* t = GSDF_MC_TRI_TABLE + 16 * cubeIndex;
        auto ne = [](V3 a, V3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; };
        for (int i = 0; t[i] != -1; i += 3) {
            const V3 p1 = edgePoints[t[i]], p2 = edgePoints[t[i + 1]], p3 = edgePoints[t[i + 2]];
            if (ne(p1, p2) && ne(p1, p3) && ne(p2, p3)) {
                const float f[9] = { p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p3.x, p3.y, p3.z };
                tris.insert(tris.end(), f, f + 9);
            }
        }
    }
    /* computeIsoSurface -- :354-561 */
    bool computeIsoSurface(float iso) {
        if (o->tsdf_.empty()) return false;
        int mx[3];
        for (int a = 0; a < 3; ++a) { min_[a] = std::numeric_limits<int>::max(); mx[a] = std::numeric_limits<int>::min(); }
        for (const auto& kv : o->tsdf_) {
            const int c[3] = { kv.first.x, kv.first.y, kv.first.z };
            for (int a = 0; a < 3; ++a) { min_[a] = std::min(min_[a], c[a]); mx[a] = std::max(mx[a], c[a]); }
        }
        for (int a = 0; a < 3; ++a) { origin_[a] = -(float)min_[a] * vs; dim_[a] = mx[a] - min_[a] + 1; }   /* :376-378 */
        areaXY_ = (size_t)dim_[0] * dim_[1];
        tsdf_.assign(2 * areaXY_, 0.f);
        weights_.assign(2 * areaXY_, 0.f);
        /* edge e joins corner A[e] to corner B[e] (offsets of :410-549, in that order) */
        static const int C[8][3] = { { 1, 1, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 0, 1, 0 }, { 1, 1, 1 }, { 1, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 } };
        static const int A[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3 }, B[12] = { 1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7 };
        V3 edgePoints[12] = {};
        copyLayer(0);
        for (int z = 0; z < dim_[2] - 1; ++z) {
            copyLayer(z + 1);
            for (int y = 0; y < dim_[1] - 1; ++y)
                for (int x = 0; x < dim_[0] - 1; ++x) {
                    const int cubeindex = computeLutIndex(x, y, z, iso);
                    if (cubeindex == 0 || cubeindex == 255) continue;
                    for (int e = 0; e < 12; ++e)
                        if (GSDF_MC_EDGE_TABLE[cubeindex] & (1 << e))
                            edgePoints[e] = getVertex(x + C[A[e]][0], y + C[A[e]][1], z + C[A[e]][2],
                                                      x + C[B[e]][0], y + C[B[e]][1], z + C[B[e]][2], iso);
                    computeTriangles(cubeindex, edgePoints);
                }
        }
        return true;
    }
};
} // namespace

/* MapGradPixelSdf::extract_mesh -- MapGradPixelSdf.cpp:124-175 (lmc.computeIsoSurface(&tsdf_), iso 0 by default).
 * tris9 == NULL (or max_tris too small) only counts; returns the number of faces. */
int64_t gsdfo_extract_mesh(const gsdfo* o, float iso, float* tris9, int64_t max_tris) {
    LayeredMC mc;
    mc.o = o; mc.vs = o->voxel_size_;
    if (!mc.computeIsoSurface(iso)) return 0;
    const int64_t n = (int64_t)(mc.tris.size() / 9);
    if (tris9 && n <= max_tris) std::memcpy(tris9, mc.tris.data(), mc.tris.size() * sizeof(float));
    return n;
}

/* MapGradPixelSdf::weights -- MapGradPixelSdf.h:117-125 */
static inline float oracle_weights(const gsdfo* o, V3 p, const SdfVoxel** vout, Key* kout) {
    const Key idx = o->float2vox(p);
    auto it = o->tsdf_.find(idx);
    if (it != o->tsdf_.end()) { *vout = &it->second; *kout = idx; return it->second.weight; }
    *vout = nullptr;
    return 0.f;
}
/* MapGradPixelSdf::tsdf -- MapGradPixelSdf.h:109-115 */
static inline float oracle_tsdf(const gsdfo* o, const SdfVoxel& v, Key idx, V3 p, V3* grad) {
    const V3 gn = normalized3(V3{ v.grad[0], v.grad[1], v.grad[2] });
    /* :113 `1.2*v.grad.normalized()`: scalar * Eigen expression -- Eigen's operator converts the double literal to the
     * expression's scalar type (float) before the product */
    if (grad) *grad = V3{ 1.2f * gn.x, 1.2f * gn.y, 1.2f * gn.z };
    const V3 c = o->vox2float(idx);
    const V3 d = { c.x - p.x, c.y - p.y, c.z - p.z };
    /* :114 `v.dist + 1.2*v.grad.normalized().dot(c - point)`: the member calls bind before `*`, so .dot() is taken in
     * float with the UNIT gradient and returns a float; 1.2 * float and v.dist + (...) are plain C++ double arithmetic,
     * and the float return type rounds once.  (Built with -ffp-contract=off: two double roundings, no fma.) */
    return (float)((double)v.dist + 1.2 * (double)dot3(gn, d));
}

void gsdfo_query(const gsdfo* o, const float* pts, int64_t n, float* dist, float* grad, float* w) {
    for (int64_t i = 0; i < n; ++i) {
        const V3 p = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
        const SdfVoxel* v; Key k;
        const float w0 = oracle_weights(o, p, &v, &k);
        w[i] = w0;
        V3 g = { 0.f, 0.f, 0.f };
        float phi = 0.f;
        if (v) phi = oracle_tsdf(o, *v, k, p, &g);
        dist[i] = phi;
        grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
    }
}

/* ---- voxel-hash raycaster ----------------------------------------------------------------------
 * BASELINE.json's north_star names a "voxel-hash raycaster"; the reference has none, so
 * there is nothing to restate: this is the DEFINITION the HIP kernel k_raycast is checked against,
 * built only from the reference's point query weights()/tsdf() (MapGradPixelSdf.h:109-125) and the
 * tracker's back-projection (RigidPointOptimizer.cpp:46-47,67-70).  PARITY UNPINNED (self-defined).
 *
 * Pixel (u,v): d = R (x0, y0, 1), p(s) = s d + t, s = camera depth.  March s from zmin.  While the voxel under p(s) is
 * missing: COARSE steps of min(4, factor - 1) voxels of depth (at least 1; from a fusing pose the samples then coincide with
 * the fused ones).  The band of existing voxels around a surface is >= ~6 voxels thick along its normal (2 factor + 1 voxels
 * along the fusing rays, which the normal gate keeps within 72.5 degrees of the normal), so a coarse step (<= 5 voxels of
 * 3-D length at the image corners) cannot jump it -- but it can jump its FRONT part and land behind the surface.  Hence: the first existing voxel found after a coarse step sends the
 * walk back to the start of that step, and the stretch is walked again in FINE steps (1 voxel of depth, also across
 * missing voxels).  Inside the band: fine steps.  A hit is the first sign change phi_prev < 0 <= phi (the reference's SDF
 * is negative in front of the surface) between two existing samples that are consecutive or have ONE missing sample
 * between them (fused pixel columns leave lateral gaps when a pixel is wider than a voxel); depth = linear interpolation of s,
 * normal = R^T grad/|grad| of the sample behind the surface (camera frame, like NormalEstimator). */
void gsdfo_raycast(const gsdfo* o, const float K[9], const float R[9], const float t[3], int W, int H,
                   float zmin, float zmax, float* depth, float* normals) {
    const float fx_inv = 1.f / K[0], fy_inv = 1.f / K[4], cx = K[2], cy = K[5];
    const int factor = (int)std::floor(o->T_ / o->voxel_size_);          /* band half-width in voxels (MapGradPixelSdf.cpp:79) */
    const float fine = o->voxel_size_, coarse = (float)std::max(1, std::min(4, factor - 1)) * o->voxel_size_;
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const float x0 = ((float)u - cx) * fx_inv, y0 = ((float)v - cy) * fy_inv;
            const V3 d = matvec(R, V3{ x0, y0, 1.f });
            float out_z = 0.f;
            V3 out_n = { 0.f, 0.f, 0.f };
            bool prev_ok = false;
            float phi_prev = 0.f, s_prev = 0.f;
            float fine_until = zmin;                   /* below this depth missing voxels are crossed in fine steps */
            float s_coarse_from = -1.f;                /* start of the coarse step that led to s (< 0: s was reached by a fine step) */
            for (float s = zmin; s < zmax;) {
                const V3 p = { s * d.x + t[0], s * d.y + t[1], s * d.z + t[2] };
                const SdfVoxel* vox; Key idx;
                const float w0 = oracle_weights(o, p, &vox, &idx);
                if (w0 > 0.f && s_coarse_from >= 0.f) {        /* entered the band by a coarse step: walk that stretch again */
                    fine_until = s;
                    s = s_coarse_from + fine;
                    s_coarse_from = -1.f;
                    prev_ok = false;
                    continue;
                }
                if (w0 > 0.f) {
                    V3 g;
                    const float phi = oracle_tsdf(o, *vox, idx, p, &g);
                    if (prev_ok && phi_prev < 0.f && phi >= 0.f) {   /* the stored SDF is negative in front of the surface */
                        out_z = s_prev + (s - s_prev) * (phi_prev / (phi_prev - phi));
                        const V3 gn = normalized3(V3{ vox->grad[0], vox->grad[1], vox->grad[2] });
                        out_n = V3{ sum3(R[0] * gn.x, R[3] * gn.y, R[6] * gn.z), sum3(R[1] * gn.x, R[4] * gn.y, R[7] * gn.z),
                                    sum3(R[2] * gn.x, R[5] * gn.y, R[8] * gn.z) };
                        break;
                    }
                    prev_ok = true; phi_prev = phi; s_prev = s;
                    s += fine;
                } else if (prev_ok && s - s_prev < 1.5f * fine) {
                    s += fine;                                  /* ONE missing sample inside the band is bridged (lateral gaps
                                                                   between fused pixel columns): the previous sample stays */
                } else {
                    prev_ok = false;
                    if (s < fine_until) s += fine;
                    else { s_coarse_from = s; s += coarse; }
                }
            }
            const size_t i = (size_t)v * W + u;
            depth[i] = out_z;
            if (normals) { normals[i] = out_n.x; normals[(size_t)W * H + i] = out_n.y; normals[2 * (size_t)W * H + i] = out_n.z; }
        }
}

/* ---- Eigen / Sophus restatements (third-party code absent, unpinned) ---------------------- */

/* Eigen::QuaternionBase::toRotationMatrix */
void gsdfo_quat_to_R(const float q[4], float R[9]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}

/* Eigen quaternionbase_assign_impl<Matrix3> (rotation matrix -> quaternion) */
void gsdfo_R_to_quat(const float m[9], float q[4]) {
    float t = m[0] + m[4] + m[8];
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f);
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
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

/* Sophus: SO3::expAndTheta, SE3::exp, SE3 group product (quaternion product + normalisation,
 * translation t1 + q1*t2 via Eigen's _transformVector).  pose = exp(xi) * pose. */
void gsdfo_se3_exp_mul(const float xi[6], float pose7[7]) {
    const float eps = 1e-5f;                                   /* Sophus::Constants<float>::epsilon() */
    const V3 ups = { xi[0], xi[1], xi[2] };
    const V3 om = { xi[3], xi[4], xi[5] };
    const float theta_sq = sum3(om.x * om.x, om.y * om.y, om.z * om.z);
    float theta, imag, real;
    if (theta_sq < eps * eps) {
        theta = 0.f;
        const float theta_po4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
        real = 1.f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
    } else {
        theta = std::sqrt(theta_sq);
        const float half = 0.5f * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    const float qe[4] = { imag * om.x, imag * om.y, imag * om.z, real };   /* x y z w */
    /* V matrix */
    float Om[9] = { 0.f, -om.z, om.y, om.z, 0.f, -om.x, -om.y, om.x, 0.f };
    float Om2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Om2[3 * i + j] = sum3(Om[3 * i] * Om[j], Om[3 * i + 1] * Om[3 + j], Om[3 * i + 2] * Om[6 + j]);
    float V[9];
    if (theta < eps) {
        gsdfo_quat_to_R(qe, V);                                /* V = so3.matrix() */
    } else {
        const float tsq = theta * theta;
        const float a = (1.f - std::cos(theta)) / tsq;
        const float b = (theta - std::sin(theta)) / (tsq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + a * Om[i] + b * Om2[i];
    }
    const V3 te = matvec(V, ups);
    /* group product: (qe, te) * (q, t) */
    const float ax = qe[0], ay = qe[1], az = qe[2], aw = qe[3];
    const float bx = pose7[3], by = pose7[4], bz = pose7[5], bw = pose7[6];
    float qn[4];
    qn[3] = aw * bw - ax * bx - ay * by - az * bz;
    qn[0] = aw * bx + ax * bw + ay * bz - az * by;
    qn[1] = aw * by + ay * bw + az * bx - ax * bz;
    qn[2] = aw * bz + az * bw + ax * by - ay * bx;
    const float len = std::sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) qn[i] /= len;
    /* t_new = te + qe * t  (uv = qv x t; uv += uv; t + w*uv + qv x uv) */
    const V3 qv = { ax, ay, az };
    const V3 tt = { pose7[0], pose7[1], pose7[2] };
    V3 uv = cross3(qv, tt);
    uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
    const V3 c2 = cross3(qv, uv);
    pose7[0] = te.x + (tt.x + aw * uv.x + c2.x);
    pose7[1] = te.y + (tt.y + aw * uv.y + c2.y);
    pose7[2] = te.z + (tt.z + aw * uv.z + c2.z);
    pose7[3] = qn[0]; pose7[4] = qn[1]; pose7[5] = qn[2]; pose7[6] = qn[3];
}

/* 6x6 LLT solve: H.llt().solve(g) (RigidPointOptimizer.cpp:86), restated from Eigen 3.4's PUBLISHED algorithm
 * (Eigen itself is absent here; file names are Eigen's):
 *  - factorisation = llt_inplace<float, Lower>::unblocked (Cholesky/LLT.h; the blocked form hands sizes < 32 to it):
 *      x = A(k,k);  if (k > 0) x -= A10.squaredNorm();      the sum of squares is formed FIRST, then one subtraction
 *      if (x <= 0) return k;  A(k,k) = x = sqrt(x);
 *      if (k > 0 && rs > 0) A21.noalias() -= A20 * A10.adjoint();   a gemv: res_i += (-1) * (sum_j A(i,j) * A(k,j)), the dot
 *                                                                   product formed first from 0, columns ascending
 *      if (rs > 0) A21 /= x;                                        a division per element
 *    A10 is a strided row of a column-major matrix (no packet access) => its redux is the sequential scalar one.
 *  - solves = triangular_solver_unroller (SolveTriangular.h; a 6-vector right-hand side is <= 8 => CompleteUnrolling):
 *      rhs(i) -= (row segment of the factor . rhs segment).sum();   the dot product first, then one subtraction
 *      rhs(i) /= diagonal;
 *    the fixed-size .sum() is Eigen's unrolled scalar redux, which halves the range recursively
 *    (redux_novec_unroller: [0,n) -> [0,n/2) + [n/2,n)) -- the convention already used for the 3-term sums above.
 * NOT knowable without the library and the reference's compiler flags: whether the contiguous segments of the upper
 * solve (columns of L) are summed through SSE packets instead; this statement uses the scalar tree for both solves.
 * On a non-positive pivot Eigen stops the factorisation and solve() still runs on what is there
 * (=> inf/NaN for an all-zero H, SURVEY.md gotcha 9). */
static float tree_sum(const float* t, int n) {           /* redux_novec_unroller<.., Start, Length> */
    if (n == 1) return t[0];
    const int half = n / 2;
    return tree_sum(t, half) + tree_sum(t + half, n - half);
}
static void llt_solve6(const float Hin[36], const float g[6], float x[6]) {
    float L[36];
    std::memcpy(L, Hin, sizeof(L));
    for (int k = 0; k < 6; ++k) {
        float d = L[6 * k + k];
        if (k > 0) {
            float sq = L[6 * k] * L[6 * k];
            for (int j = 1; j < k; ++j) sq = sq + L[6 * k + j] * L[6 * k + j];
            d -= sq;
        }
        if (d <= 0.f) break;
        d = std::sqrt(d);
        L[6 * k + k] = d;
        for (int i = k + 1; i < 6; ++i) {
            float s = L[6 * i + k];
            if (k > 0) {
                float c = L[6 * i] * L[6 * k];
                for (int j = 1; j < k; ++j) c = c + L[6 * i + j] * L[6 * k + j];
                s -= c;
            }
            L[6 * i + k] = s / d;
        }
    }
    float t[5];
    for (int i = 0; i < 6; ++i) {          /* L y = g, in place in x */
        float s = g[i];
        if (i > 0) {
            for (int j = 0; j < i; ++j) t[j] = L[6 * i + j] * x[j];
            s -= tree_sum(t, i);
        }
        x[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {         /* L^T x = y */
        float s = x[i];
        if (i < 5) {
            for (int j = i + 1; j < 6; ++j) t[j - i - 1] = L[6 * j + i] * x[j];
            s -= tree_sum(t, 5 - i);
        }
        x[i] = s / L[6 * i + i];
    }
}

void gsdfo_llt_solve6(const float H36[36], const float g[6], float x[6]) { llt_solve6(H36, g, x); }

/* RigidPointOptimizer::optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.cpp:40-99; sampling = the public stride
 * argument of RigidPointOptimizer.h:65 (`y += sampling`, `x += sampling`, .cpp:62; optimize() passes 1, .h:71) */
int gsdfo_track(gsdfo* o, const float* depth, const float K[9], float pose7[7],
                int num_iterations, float conv_threshold, float damping, int sampling,
                int omp, int* iters_used, float* trace, int64_t* hits) {
    if (sampling < 1) return -1;                                      /* size_t 0: the reference's loops would never end */
    const float z_min = o->z_min_, z_max = o->z_max_;
    const int w = o->W, h = o->H;
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;                 /* :46-47 */
    const float conv_sq = conv_threshold * conv_threshold;            /* RigidOptimizer.h:72 */
    int used = 0;
    for (int k = 0; k < num_iterations; ++k) {                        /* :51 */
        float R[9];
        gsdfo_quat_to_R(pose7 + 3, R);                                /* :53 pose_.rotationMatrix() */
        const V3 t = { pose7[0], pose7[1], pose7[2] };
        float E = 0.f;
        float g[6] = { 0 }, Hm[36] = { 0 };
        size_t counter = 0;

        auto pixel = [&](int x, int y, float& E_, float* g_, float* H_, size_t& c_) {
            const float z = depth[(size_t)y * w + x];
            if (z <= z_min || z >= z_max) return;                     /* :64-65 */
            const float x0 = ((float)x - cx) * fx_inv;                /* :67-68 */
            const float y0 = ((float)y - cy) * fy_inv;
            V3 p = { x0 * z, y0 * z, z };
            const V3 Rp = matvec(R, p);
            p = V3{ Rp.x + t.x, Rp.y + t.y, Rp.z + t.z };             /* :70 */
            const SdfVoxel* v; Key idx;
            const float w0 = oracle_weights(o, p, &v, &idx);          /* :72 */
            if (w0 > 0) {
                V3 gc;
                const float phi0 = oracle_tsdf(o, *v, idx, p, &gc);   /* :75 */
                E_ += phi0 * phi0;                                    /* :76 */
                const V3 pxg = cross3(p, gc);
                const float J[6] = { gc.x, gc.y, gc.z, pxg.x, pxg.y, pxg.z };   /* :77-78 */
                for (int i = 0; i < 6; ++i) g_[i] += phi0 * J[i];     /* :79 */
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) H_[6 * i + j] += J[i] * J[j];   /* :80 */
                ++c_;
            }
        };

        if (omp) {
#ifdef _OPENMP
            const int nt = o->threads;
            std::vector<float> Es(nt, 0.f), gs(6 * nt, 0.f), Hs(36 * nt, 0.f);
            std::vector<size_t> cs(nt, 0);
#pragma omp parallel num_threads(nt)
            {
                const int tid = omp_get_thread_num();
                float E_ = 0.f, g_[6] = { 0 }, H_[36] = { 0 };
                size_t c_ = 0;
#pragma omp for schedule(static)
                for (int y = 0; y < h; y += sampling)                 /* Omp.cpp:70: the parallel for is over the strided y */
                    for (int x = 0; x < w; x += sampling) pixel(x, y, E_, g_, H_, c_);
                Es[tid] = E_; cs[tid] = c_;
                std::memcpy(&gs[6 * tid], g_, sizeof(g_));
                std::memcpy(&Hs[36 * tid], H_, sizeof(H_));
            }
            for (int tdx = 0; tdx < nt; ++tdx) {
                E += Es[tdx]; counter += cs[tdx];
                for (int i = 0; i < 6; ++i) g[i] += gs[6 * tdx + i];
                for (int i = 0; i < 36; ++i) Hm[i] += Hs[36 * tdx + i];
            }
#else
            for (int y = 0; y < h; y += sampling) for (int x = 0; x < w; x += sampling) pixel(x, y, E, g, Hm, counter);
#endif
        } else {
            for (int y = 0; y < h; y += sampling) for (int x = 0; x < w; x += sampling) pixel(x, y, E, g, Hm, counter);   /* :62 */
        }

        float xi[6];
        llt_solve6(Hm, g, xi);                                        /* :86 */
        for (int i = 0; i < 6; ++i) xi[i] = damping * xi[i];
        float nrm = 0.f;                                              /* Eigen redux of 6: (x0+(x1+x2)) + (x3+(x4+x5)) */
        nrm = sum3(xi[0] * xi[0], xi[1] * xi[1], xi[2] * xi[2]) + sum3(xi[3] * xi[3], xi[4] * xi[4], xi[5] * xi[5]);
        if (trace) {
            float* tr = trace + 36 * k;
            tr[0] = E;
            for (int i = 0; i < 6; ++i) tr[1 + i] = g[i];
            int q = 7;
            for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) tr[q++] = Hm[6 * i + j];
            tr[28] = (float)counter;
            for (int i = 0; i < 6; ++i) tr[29 + i] = xi[i];
            tr[35] = nrm;
        }
        if (hits) hits[k] = (int64_t)counter;
        used = k + 1;
        if (nrm < conv_sq) {                                          /* :88-91: xi NOT applied */
            if (iters_used) *iters_used = used;   /* passes executed; the reference prints k */
            return 1;
        }
        bool nan = false;
        for (int i = 0; i < 6; ++i) if (std::isnan(xi[i])) nan = true;
        if (!nan) {                                                   /* :94-95 pose_ = SE3::exp(-xi) * pose_ */
            float mxi[6];
            for (int i = 0; i < 6; ++i) mxi[i] = -xi[i];
            gsdfo_se3_exp_mul(mxi, pose7);
        }
    }
    if (iters_used) *iters_used = used;
    return 0;                                                         /* :98 */
}

} // extern "C"

/* ==================================================================================================
 * PhotoBA oracle: PhotometricOptimizer restated (ps_optimizer/PhotometricOptimizer.cpp), L2 loss path
 * (the default CAUCHY setting never enters the TRUNC_L2 branches :364,:542; gsdfo_ba_set_loss selects them).
 * ================================================================================================== */
namespace {

struct Img { int W, H; const float* p; };                    /* BGR float, row-major */
static inline const float* px(const Img& im, int row, int col) { return im.p + ((size_t)row * im.W + col) * 3; }

/* interpolateImage(m, n, img) -- :57-77: m = row coordinate, n = column coordinate; weights in double,
 * each weighted pixel rounded to float (cv::Vec3f * double), summed in float; BGR -> RGB */
static V3 interpolate_image(float m, float n, const Img& im) {
    const int x = (int)std::floor(m), y = (int)std::floor(n);
    float t[3];
    auto wpx = [](double w, const float* q, int c) { return (float)(w * (double)q[c]); };
    if ((x + 1) < im.H && (y + 1) < im.W) {
        const double w1 = (y + 1.0 - n) * (m - x), w2 = (y + 1.0 - n) * (x + 1.0 - m), w3 = (n - y) * (m - x), w4 = (n - y) * (x + 1.0 - m);
        for (int c = 0; c < 3; ++c)
            t[c] = ((wpx(w1, px(im, x + 1, y), c) + wpx(w2, px(im, x, y), c)) + wpx(w3, px(im, x + 1, y + 1), c)) + wpx(w4, px(im, x, y + 1), c);
    } else if ((y + 1) < im.W && x >= im.H) {                 /* unreachable for in-bounds m (kept for fidelity) */
        for (int c = 0; c < 3; ++c) t[c] = wpx(y + 1.0 - n, px(im, std::min(x, im.H - 1), y), c) + wpx(n - y, px(im, std::min(x, im.H - 1), y + 1), c);
    } else if (y >= im.W && (x + 1) < im.H) {
        for (int c = 0; c < 3; ++c) t[c] = wpx(m - x, px(im, x + 1, std::min(y, im.W - 1)), c) + wpx(x + 1.0 - m, px(im, x, std::min(y, im.W - 1)), c);
    } else {
        for (int c = 0; c < 3; ++c) t[c] = px(im, std::min(x, im.H - 1), std::min(y, im.W - 1))[c];
    }
    return V3{ t[2], t[1], t[0] };
}

/* computeImageGradient(m, n, img, direction) -- :80-140 (finite differences, bilinear in the other axis) */
static V3 image_gradient(float m, float n, const Img& im, int direction) {
    const int x = (int)std::floor(m), y = (int)std::floor(n);
    const float w01 = m - x, w11 = n - y;
    const float w00 = (float)(1.0 - w01), w10 = (float)(1.0 - w11);
    float v0[3] = { 0, 0, 0 }, v1[3] = { 0, 0, 0 };
    auto diff = [&](float* o, int r1, int c1, int r0, int c0) { for (int c = 0; c < 3; ++c) o[c] = px(im, r1, c1)[c] - px(im, r0, c0)[c]; };
    auto comb = [&](float a, float b) { return V3{ a * v0[2] + b * v1[2], a * v0[1] + b * v1[1], a * v0[0] + b * v1[0] }; };
    if (direction == 0) {
        if ((x + 1) < im.H && (y + 1) < im.W) { diff(v0, x, y + 1, x, y); diff(v1, x + 1, y + 1, x + 1, y); return comb(w00, w01); }
        else if ((x + 1) >= im.H) {
            if ((y + 1) < im.W) diff(v0, x, y + 1, x, y); else diff(v0, x, y, x, y - 1);      /* reference reads past the row end here */
            return V3{ v0[2], v0[1], v0[0] };
        } else { diff(v0, x, y, x, y - 1); diff(v1, x + 1, y, x + 1, y - 1); return comb(w00, w01); }
    } else {
        if ((x + 1) < im.H && (y + 1) < im.W) { diff(v0, x + 1, y, x, y); diff(v1, x + 1, y + 1, x, y + 1); return comb(w10, w11); }
        else if ((x + 1) >= im.H && (y + 1) < im.W) { diff(v0, x, y, x - 1, y); diff(v1, x, y + 1, x - 1, y + 1); return comb(w10, w11); }
        else {
            if ((x + 1) < im.H) diff(v0, x + 1, y, x, y); else diff(v0, x, y, x - 1, y);      /* reference reads past the last row here */
            return V3{ v0[2], v0[1], v0[0] };
        }
    }
}

} // namespace

struct gsdfo_ba {
    gsdfo* o;
    float fx, fy, cx, cy;
    int n, W, H;
    std::vector<float> images;
    std::vector<float> R;          /* n x 9 row-major */
    std::vector<float> t;          /* n x 3 */
    std::vector<int> frame_idx;
    float reg_weight;
    int loss = 1;                /* LossFunction (loss.h:39-46): 0 L2, 1 CAUCHY (OptSettings default), 2 HUBER, 3 TUKEY, 4 TRUNC_L2 */
    float lambda_sq = 0.25f;     /* OptSettings::lambda_sq, lambda = 0.5 (PhotometricOptimizer.h:62-63) */
    /* the only place the loss enters: solveDist :364 and solvePose :542 skip a keyframe whose residual is too large */
    bool truncated(const V3& a) const { return loss == 4 && std::max(a.x * a.x, std::max(a.y * a.y, a.z * a.z)) > lambda_sq; }
    std::vector<Key> order;        /* voxel visiting order (z,y,x) */

    Img img(int i) const { return Img{ W, H, images.data() + (size_t)i * W * H * 3 }; }
    bool visible(const Key& k, int i) const {
        auto it = o->vis_.find(k);
        if (it == o->vis_.end()) return false;
        const std::vector<bool>& v = it->second;
        return !((int)v.size() <= frame_idx[i] || !v[frame_idx[i]]);      /* :289 */
    }
    /* the common projection of getIntensity / computeJc / computeJdOneFrame (:165-177) */
    bool project(const Key& idx, const SdfVoxel& vox, int i, V3* point, float* m, float* n) const {
        const float* Ri = &R[9 * i];
        const V3 gn = normalized3(V3{ vox.grad[0], vox.grad[1], vox.grad[2] });
        const V3 c = o->vox2float(idx);
        const V3 d = { c.x - vox.dist * gn.x - t[3 * i], c.y - vox.dist * gn.y - t[3 * i + 1], c.z - vox.dist * gn.z - t[3 * i + 2] };
        /* Rt * d */
        const V3 p = { sum3(Ri[0] * d.x, Ri[3] * d.y, Ri[6] * d.z), sum3(Ri[1] * d.x, Ri[4] * d.y, Ri[7] * d.z),
                       sum3(Ri[2] * d.x, Ri[5] * d.y, Ri[8] * d.z) };
        const float z_inv = (float)(1. / p.z);
        *m = fx * p.x * z_inv + cx;
        *n = fy * p.y * z_inv + cy;
        *point = p;
        return !(*m < 0 || *m >= W || *n < 0 || *n >= H);
    }
    bool intensity(const Key& idx, const SdfVoxel& vox, int i, V3* A) const {       /* getIntensity :238-260 */
        V3 p; float m, n;
        if (!project(idx, vox, i, &p, &m, &n)) return false;
        *A = interpolate_image(n, m, img(i));
        return true;
    }
    /* image_grad (3x2) * pi_grad (2x3) as a 3x3 row-major matrix (:188-199, :221-231) */
    bool image_pi_grad(const Key& idx, const SdfVoxel& vox, int i, V3* point, float G[9]) const {
        V3 p; float m, n;
        if (!project(idx, vox, i, &p, &m, &n)) return false;
        const float z_inv = (float)(1. / p.z), z_inv_sq = z_inv * z_inv;
        const V3 g0 = image_gradient(n, m, img(i), 0), g1 = image_gradient(n, m, img(i), 1);
        const float pg[6] = { fx * z_inv, 0.f, -fx * p.x * z_inv_sq, 0.f, fy * z_inv, -fy * p.y * z_inv_sq };
        const float ig[6] = { g0.x, g1.x, g0.y, g1.y, g0.z, g1.z };          /* 3x2 row-major */
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) G[3 * r + c] = ig[2 * r] * pg[c] + ig[2 * r + 1] * pg[3 + c];
        *point = p;
        return true;
    }
};

extern "C" {

gsdfo_ba* gsdfo_ba_create(gsdfo* o, const float K[9], int n, int W, int H, const float* images_bgr, const float* poses16,
                          const int* frame_idx, float reg_weight) {
    gsdfo_ba* b = new gsdfo_ba();
    b->o = o; b->fx = K[0]; b->fy = K[4]; b->cx = K[2]; b->cy = K[5];
    b->n = n; b->W = W; b->H = H; b->reg_weight = reg_weight;
    b->images.assign(images_bgr, images_bgr + (size_t)n * W * H * 3);
    b->R.resize(9 * n); b->t.resize(3 * n);
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) b->R[9 * i + 3 * r + c] = poses16[16 * i + 4 * r + c]; b->t[3 * i + r] = poses16[16 * i + 4 * r + 3]; }
    b->frame_idx.assign(frame_idx, frame_idx + n);
    b->order = sorted_keys(o);
    return b;
}
void gsdfo_ba_destroy(gsdfo_ba* b) { delete b; }

/* getEnergy -- :273-321.  The per-voxel terms are the reference's float arithmetic; the reference adds them up in ONE float
 * in the iteration order of its hash map (unknowable), which for ~10^6 terms leaves its own value uncertain at the 1e-3 level
 * (terms below half an ulp of the running sum vanish).  gsdfo_ba_energy does the same in (z,y,x) order; gsdfo_ba_energy_f64
 * adds the identical terms in double -- the order-independent value the GPU sweep is compared with. */
static double ba_energy_impl(gsdfo_ba* b, float* E_float) {
    float E = 0.f;
    double E64 = 0.0;
    std::vector<V3> A;
    for (const Key& idx : b->order) {
        const SdfVoxel& vox = b->o->tsdf_.at(idx);
        if (std::fabs(vox.dist) > b->o->voxel_size_) continue;
        A.clear();
        V3 mean = { 0, 0, 0 };
        for (int i = 0; i < b->n; ++i) {
            if (!b->visible(idx, i)) continue;
            V3 a;
            if (!b->intensity(idx, vox, i, &a)) continue;
            mean = V3{ mean.x + a.x, mean.y + a.y, mean.z + a.z };
            A.push_back(a);
        }
        const float inv = (float)(1. / (float)A.size());
        mean = V3{ inv * mean.x, inv * mean.y, inv * mean.z };
        for (const V3& a : A) {
            const V3 r = { a.x - mean.x, a.y - mean.y, a.z - mean.z };
            const float term = dot3(r, r);
            E += term;
            E64 += (double)term;
        }
    }
    if (E_float) *E_float = E;
    return E64;
}
void gsdfo_ba_set_loss(gsdfo_ba* b, int loss, float lambda) { b->loss = loss; b->lambda_sq = lambda * lambda; }
float gsdfo_ba_energy(gsdfo_ba* b) {
    float E = 0.f;
    ba_energy_impl(b, &E);
    return E;
}
double gsdfo_ba_energy_f64(gsdfo_ba* b) { return ba_energy_impl(b, nullptr); }

/* solveDist -- :326-388 */
void gsdfo_ba_solve_dist(gsdfo_ba* b, float damping) {
    for (const Key& idx : b->order) {
        SdfVoxel& vox = b->o->tsdf_.at(idx);
        size_t Nj = 0;
        V3 sA = { 0, 0, 0 }, sD = { 0, 0, 0 }, sAD = { 0, 0, 0 }, sDD = { 0, 0, 0 };
        for (int i = 0; i < b->n; ++i) {
            if (!b->visible(idx, i)) continue;
            V3 a;
            if (!b->intensity(idx, vox, i, &a)) continue;
            if (b->truncated(a)) continue;                                              /* :364 */
            ++Nj;
            /* computeJdOneFrame :160-203: Jd = image_grad * pi_grad * (-Rt * grad) */
            V3 p; float G[9];
            b->image_pi_grad(idx, vox, i, &p, G);
            const float* Ri = &b->R[9 * i];
            const V3 Rtn = { -sum3(Ri[0] * vox.grad[0], Ri[3] * vox.grad[1], Ri[6] * vox.grad[2]),
                             -sum3(Ri[1] * vox.grad[0], Ri[4] * vox.grad[1], Ri[7] * vox.grad[2]),
                             -sum3(Ri[2] * vox.grad[0], Ri[5] * vox.grad[1], Ri[8] * vox.grad[2]) };
            const V3 Jd = matvec(G, Rtn);
            sA = V3{ sA.x + a.x, sA.y + a.y, sA.z + a.z };
            sD = V3{ sD.x + Jd.x, sD.y + Jd.y, sD.z + Jd.z };
            sAD = V3{ sAD.x + a.x * Jd.x, sAD.y + a.y * Jd.y, sAD.z + a.z * Jd.z };
            sDD = V3{ sDD.x + Jd.x * Jd.x, sDD.y + Jd.y * Jd.y, sDD.z + Jd.z * Jd.z };
        }
        if (Nj == 0) continue;
        const float inv_Nj = (float)(1. / (float)Nj);
        float H_dd = sum3(sDD.x, sDD.y, sDD.z) - inv_Nj * sum3(sD.x * sD.x, sD.y * sD.y, sD.z * sD.z);
        const float b_d = sum3(sAD.x, sAD.y, sAD.z) - inv_Nj * sum3(sA.x * sD.x, sA.y * sD.y, sA.z * sD.z);
        H_dd += b->reg_weight * vox.weight;
        if (H_dd != 0) vox.dist -= damping * b_d / H_dd;                  /* updateDist :265-268 */
    }
}

/* 6x6 solve by LDL^T with symmetric (diagonal) pivoting, Eigen's LDLT algorithm restated */
static void ldlt_solve6(const float Hin[36], const float bin[6], float x[6]) {
    float A[36]; int perm[6]; float bb[6];
    std::memcpy(A, Hin, sizeof(A));
    for (int i = 0; i < 6; ++i) perm[i] = i;
    for (int k = 0; k < 6; ++k) {
        int piv = k; float best = std::fabs(A[7 * k]);
        for (int i = k + 1; i < 6; ++i) if (std::fabs(A[7 * i]) > best) { best = std::fabs(A[7 * i]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; ++j) std::swap(A[6 * k + j], A[6 * piv + j]);
            for (int j = 0; j < 6; ++j) std::swap(A[6 * j + k], A[6 * j + piv]);
            std::swap(perm[k], perm[piv]);
        }
        const float d = A[7 * k];
        if (d == 0.f) continue;
        for (int i = k + 1; i < 6; ++i) A[6 * i + k] /= d;
        for (int i = k + 1; i < 6; ++i)
            for (int j = k + 1; j <= i; ++j) { A[6 * i + j] -= A[6 * i + k] * d * A[6 * j + k]; A[6 * j + i] = A[6 * i + j]; }
    }
    for (int i = 0; i < 6; ++i) bb[i] = bin[perm[i]];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) bb[i] -= A[6 * i + j] * bb[j];      /* L y = P b */
    for (int i = 0; i < 6; ++i) bb[i] = A[7 * i] != 0.f ? bb[i] / A[7 * i] : 0.f;                /* D z = y (Eigen: 0 for tiny pivots) */
    for (int i = 5; i >= 0; --i) for (int j = i + 1; j < 6; ++j) bb[i] -= A[6 * j + i] * bb[j];  /* L^T w = z */
    for (int i = 0; i < 6; ++i) x[perm[i]] = bb[i];
}

/* solvePose -- :499-590 */
void gsdfo_ba_solve_pose(gsdfo_ba* b, float) {
    const int n = b->n;
    std::vector<float> Hs(36 * (size_t)n, 0.f), bs(6 * (size_t)n, 0.f);
    std::vector<int> vi; std::vector<V3> vA; std::vector<std::array<float, 18>> vJ;
    for (const Key& idx : b->order) {
        const SdfVoxel& vox = b->o->tsdf_.at(idx);
        if (std::fabs(vox.dist) > b->o->voxel_size_) continue;
        vi.clear(); vA.clear(); vJ.clear();
        V3 mean = { 0, 0, 0 };
        for (int i = 0; i < n; ++i) {
            if (!b->visible(idx, i)) continue;
            V3 a, p; float G[9];
            if (!b->intensity(idx, vox, i, &a) || !b->image_pi_grad(idx, vox, i, &p, G)) continue;
            if (b->truncated(a)) continue;                                              /* :542 */
            /* computeJc :206-233: Jc = [ -G * Rt , G * skew(point) ]  (3x6) */
            const float* Ri = &b->R[9 * i];
            std::array<float, 18> J;
            const float S[9] = { 0.f, -p.z, p.y, p.z, 0.f, -p.x, -p.y, p.x, 0.f };
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    J[6 * r + c] = -sum3(G[3 * r] * Ri[3 * c], G[3 * r + 1] * Ri[3 * c + 1], G[3 * r + 2] * Ri[3 * c + 2]);   /* (G Rt)(r,c) = sum_k G(r,k) R(c,k) */
                    J[6 * r + 3 + c] = sum3(G[3 * r] * S[c], G[3 * r + 1] * S[3 + c], G[3 * r + 2] * S[6 + c]);
                }
            vi.push_back(i); vA.push_back(a); vJ.push_back(J);
            mean = V3{ mean.x + a.x, mean.y + a.y, mean.z + a.z };
        }
        const size_t Nj = vi.size();
        if (Nj == 0) continue;
        const float inv_Nj = (float)(1. / (float)Nj);
        mean = V3{ inv_Nj * mean.x, inv_Nj * mean.y, inv_Nj * mean.z };
        for (size_t k = 0; k < Nj; ++k) {
            const int i1 = vi[k];
            const V3 r = { vA[k].x - mean.x, vA[k].y - mean.y, vA[k].z - mean.z };
            const std::array<float, 18>& J = vJ[k];
            for (int c = 0; c < 6; ++c) bs[6 * i1 + c] += sum3(r.x * J[c], r.y * J[6 + c], r.z * J[12 + c]);
            for (int a1 = 0; a1 < 6; ++a1)
                for (int a2 = 0; a2 < 6; ++a2)
                    Hs[36 * i1 + 6 * a1 + a2] += (1 - inv_Nj) * sum3(J[a1] * J[a2], J[6 + a1] * J[6 + a2], J[12 + a1] * J[12 + a2]);
        }
    }
    for (int i = 0; i < n; ++i) {
        float dp[6];
        ldlt_solve6(&Hs[36 * i], &bs[6 * i], dp);
        bool nan = false;
        for (int k = 0; k < 6; ++k) nan = nan || std::isnan(dp[k]);
        if (nan) continue;
        for (int k = 0; k < 3; ++k) b->t[3 * i + k] -= dp[k];                              /* :587 */
        /* R = R * SO3::exp(-omega).matrix()  (:588) */
        float pose[7] = { 0, 0, 0, 0, 0, 0, 1 };
        const float xi[6] = { 0, 0, 0, -dp[3], -dp[4], -dp[5] };
        gsdfo_se3_exp_mul(xi, pose);
        float E[9], Rn[9];
        gsdfo_quat_to_R(pose + 3, E);
        const float* Ri = &b->R[9 * i];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rn[3 * r + c] = sum3(Ri[3 * r] * E[c], Ri[3 * r + 1] * E[3 + c], Ri[3 * r + 2] * E[6 + c]);
        std::memcpy(&b->R[9 * i], Rn, sizeof(Rn));
    }
}

/* optimize -- :611-662 */
int gsdfo_ba_optimize(gsdfo_ba* b, int max_it, float* energies, int* n_energies) {
    int ne = 0;
    float E = gsdfo_ba_energy(b);
    energies[ne++] = E;
    for (int iter = 0; iter < max_it; ++iter) {
        gsdfo_ba_solve_pose(b, 1.0f);
        const float E_pose = gsdfo_ba_energy(b);
        energies[ne++] = E_pose;
        gsdfo_ba_solve_dist(b, 1.0f);
        E = gsdfo_ba_energy(b);
        energies[ne++] = E;
        const float rel = std::fabs(E_pose - E) / E_pose;
        if (rel < 0.0005f) { *n_energies = ne; return 1; }
        if (E_pose < E) { *n_energies = ne; return 0; }
    }
    *n_energies = ne;
    return 0;
}

void gsdfo_ba_get_poses(const gsdfo_ba* b, float* P) {
    for (int i = 0; i < b->n; ++i) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) P[16 * i + 4 * r + c] = b->R[9 * i + 3 * r + c]; P[16 * i + 4 * r + 3] = b->t[3 * i + r]; }
        P[16 * i + 12] = P[16 * i + 13] = P[16 * i + 14] = 0.f; P[16 * i + 15] = 1.f;
    }
}

} // extern "C"
