/*
 * gsdf_oracle.h -- C API of the CPU ORACLE for the Gradient-SDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the
 * reference's serial code path (c-sommer/gradient-sdf, cpp/include/...).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline -- never as the product.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and
 * cannot be compiled here (Eigen, Sophus, OpenCV, phmap, CLI11, Boost absent and
 * unpinned), so this oracle is pinned only by analytic
 * known-answer tests (tests/test_oracle_known_answers.py) and by fixtures it
 * generated itself (tests/golden/).  Where arithmetic lives in the absent
 * third-party code (Eigen 3.4 products/reductions, Sophus SE3::exp, OpenCV 4
 * boxFilter/divide/sqrt) the published algorithm is restated and the exact
 * float operation order chosen here DEFINES parity for the HIP path.
 *
 * Conventions (same as the product C-ABI, include/gsdf.h):
 *   depth  : float32, row-major H x W, metres, 0 = invalid
 *   K      : row-major 3x3 float (fx 0 cx; 0 fy cy; 0 0 1)
 *   R, t   : camera->world, p_w = R p_c + t, R row-major 3x3
 *   pose7  : tx ty tz qx qy qz qw  (TUM order; Sophus SE3 state = unit quat + t)
 */
#ifndef GSDF_ORACLE_H_
#define GSDF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsdfo gsdfo;

/* MapGradPixelSdf(voxel_size, T)  -- MapGradPixelSdf.h:99-103, Sdf.h:103-107 */
gsdfo* gsdfo_create(float voxel_size, float trunc_dist);
void   gsdfo_destroy(gsdfo* o);
/* Sdf::set_zmin / set_zmax -- Sdf.h:123-129 (defaults 0.5 / 3.5, Sdf.h:67-68) */
void   gsdfo_set_zrange(gsdfo* o, float zmin, float zmax);
/* threads used by the *_omp variants (fusion: parallel for + critical, as
 * MapGradPixelSdfOmp.cpp:82,112; tracker: reduction, RigidPointOptimizerOmp.cpp:68-69) */
void   gsdfo_set_threads(gsdfo* o, int threads);
/* box-filter summation order (0 = a fresh double sum per output, 1 = OpenCV's running sums: RowSum / ColumnSum of the
 * generic FilterEngine path), separately for the cached planes (default 1) and the per-frame filters (default 0, same
 * floats as 1 there).  For measuring the differences; takes effect at the next gsdfo_normals_init / _compute. */
void   gsdfo_set_box_mode(gsdfo* o, int cache_mode, int frame_mode);
/* order of the two double additions of NormalEstimator.h:104 (cv::MatExpr -> one addWeighted): 2 = OpenCV 4's SIMD loop,
 * x^2 + (y^2 + 1) (the definition); 1 = its scalar loop, (x^2 + y^2) + 1; 0 = (1 + x^2) + y^2, the line read as plain doubles.
 * 0 / 1 exist to measure what the choice changes.  Before gsdfo_normals_init. */
void   gsdfo_set_nsq_order(gsdfo* o, int order);

/* cv::NormalEstimator<float>(W, H, K, Size(win,win)) -> cache()  -- NormalEstimator.h:81-165 */
int    gsdfo_normals_init(gsdfo* o, int W, int H, const float K[9], int win);
/* the 11 cached float planes, each W*H: x0,y0,x0/n2,y0/n2,1/n2,Q11,Q12,Q13,Q22,Q23,Q33 */
void   gsdfo_normals_cache(const gsdfo* o, float* planes11);
/* NormalEstimator::compute -- NormalEstimator.h:179-204 */
void   gsdfo_normals_compute(const gsdfo* o, const float* depth, float* nx, float* ny, float* nz);

/* MapGradPixelSdf::update -- MapGradPixelSdf.cpp:43-122.  omp=1 runs the
 * MapGradPixelSdfOmp.cpp structure (identical arithmetic). Returns N_upd =
 * number of (pixel,k) samples with w>0; *n_valid = pixels passing all gates. */
int64_t gsdfo_update(gsdfo* o, const float* depth, const float R[9], const float t[3],
                     int omp, int64_t* n_valid);

int64_t gsdfo_count(const gsdfo* o);
int64_t gsdfo_frame_counter(const gsdfo* o);
/* export sorted by (z, y, x): keys int32[n][3] (x,y,z), payload float[n][5] = dist,gx,gy,gz,weight */
void   gsdfo_export(const gsdfo* o, int32_t* keys, float* payload);
/* vis_ bit-vectors (MapGradPixelSdf.cpp:113-115) as ceil(frames/32) uint32 words per voxel, same order as export */
void   gsdfo_export_vis(const gsdfo* o, uint32_t* words, int words_per_voxel);

/* test plumbing: replace the map by n (key, SdfVoxel) pairs (keys int32[n][3], payload float[n][5] = dist,gx,gy,gz,weight),
 * so that the restatements below can run on exactly the voxel values another implementation produced */
void   gsdfo_set_map(gsdfo* o, const int32_t* keys, const float* payload, int64_t n);
/* test plumbing: overwrite the payload of EXISTING voxels, keeping vis_ and the key set; returns the keys not found */
int64_t gsdfo_set_payload(gsdfo* o, const int32_t* keys, const float* payload, int64_t n);
/* MapGradPixelSdf::extract_pc -- MapGradPixelSdf.cpp:177-220: rows x y z nx ny nz of the voxels with weight >= 5 whose
 * surface point lies inside the voxel, voxels visited in (z,y,x) order.  rows6 == NULL only counts.  Returns the row count. */
int64_t gsdfo_extract_pc(const gsdfo* o, float* rows6);
/* MapGradPixelSdf::extract_mesh -> LayeredMarchingCubesNoColor::computeIsoSurface (mesh/LayeredMarchingCubesNoColor.cpp:
 * 354-712) with the classic case tables (include/gsdf_mc_tables.h = :67-352): 9 floats per face in the z-y-x sweep
 * order, no vertex sharing.  Returns the face count; tris9 is filled only when max_tris suffices. */
int64_t gsdfo_extract_mesh(const gsdfo* o, float iso, float* tris9, int64_t max_tris);

/* MapGradPixelSdf::weights / ::tsdf at n points -- MapGradPixelSdf.h:109-125.
 * w[i]=0 when the voxel is absent (then dist/grad are 0; the reference would throw from .at()). */
void   gsdfo_query(const gsdfo* o, const float* pts, int64_t n, float* dist, float* grad, float* w);

/* Voxel-hash raycaster (named in BASELINE.json's north_star, ABSENT from the reference):
 * self-defined on weights()/tsdf() (MapGradPixelSdf.h:109-125), see gsdf_oracle.cpp.  depth: H*W camera-z
 * (0 = no hit); normals (nullable): 3 planes H*W, camera frame.  PARITY UNPINNED. */
void   gsdfo_raycast(const gsdfo* o, const float K[9], const float R[9], const float t[3], int W, int H,
                     float zmin, float zmax, float* depth, float* normals);

/* RigidPointOptimizer::optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.cpp:40-99 (sampling >= 1: pixel stride, :62).
 * pose7 in/out.  trace (optional, may be NULL): per executed iteration 36 floats =
 * E, g[6], H upper-tri[21] row-major, count, xi[6], |xi|^2.  hits (optional) = N_hit per iteration.
 * Returns 1 if converged (RigidPointOptimizer.cpp:88-91), else 0. omp=1: threaded reduction. */
int    gsdfo_track(gsdfo* o, const float* depth, const float K[9], float pose7[7],
                   int num_iterations, float conv_threshold, float damping, int sampling,
                   int omp, int* iters_used, float* trace, int64_t* hits);

/* SE3 helpers restating Eigen / Sophus (used by tests and the facade parity tests) */
void   gsdfo_quat_to_R(const float q_xyzw[4], float R[9]);       /* Eigen Quaternion::toRotationMatrix */
void   gsdfo_R_to_quat(const float R[9], float q_xyzw[4]);       /* Eigen Quaternion(Matrix3) */
void   gsdfo_se3_exp_mul(const float xi[6], float pose7[7]);     /* pose = SE3::exp(xi) * pose */
void   gsdfo_llt_solve6(const float H36[36], const float g[6], float x[6]);   /* H.llt().solve(g), Eigen's order */

#ifdef __cplusplus
}
#endif
#endif /* GSDF_ORACLE_H_ */

/* ---- PhotoBA: PhotometricOptimizer (ps_optimizer/PhotometricOptimizer.cpp) on the oracle's map ----------
 * images: float32 BGR in [0,1], n x H x W x 3 (cv::Mat CV_32FC3, ImageLoader.h:198-216);
 * poses16: n row-major 4x4 camera->world (key_poses, main_photo_ba.cpp:251);
 * frame_idx: integrated-frame index of each keyframe = bit tested in vis_ (PhotometricOptimizer.cpp:289).
 * Voxels are visited in (z,y,x) order (the reference's phmap order is unknowable). */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct gsdfo_ba gsdfo_ba;
gsdfo_ba* gsdfo_ba_create(gsdfo* o, const float K[9], int n, int W, int H, const float* images_bgr,
                          const float* poses16, const int* frame_idx, float reg_weight);
void   gsdfo_ba_destroy(gsdfo_ba* b);
/* OptSettings::loss / lambda (PhotometricOptimizer.h:54-57, loss.h:39-46): only TRUNC_L2 (4) changes anything, as in the reference */
void   gsdfo_ba_set_loss(gsdfo_ba* b, int loss, float lambda);
float  gsdfo_ba_energy(gsdfo_ba* b);                       /* getEnergy        :273-321 (float sum, (z,y,x) order) */
double gsdfo_ba_energy_f64(gsdfo_ba* b);                   /* the same float terms added in double (order-independent) */
void   gsdfo_ba_solve_pose(gsdfo_ba* b, float damping);    /* solvePose        :499-590 */
void   gsdfo_ba_solve_dist(gsdfo_ba* b, float damping);    /* solveDist        :326-388 */
/* optimize() :611-662.  energies receives E0, then E after every pose / dist step (<= 2*max_it+1 values).
 * Returns 1 = converged (rel. change < 5e-4), 0 = diverged or max_it reached. */
int    gsdfo_ba_optimize(gsdfo_ba* b, int max_it, float* energies, int* n_energies);
void   gsdfo_ba_get_poses(const gsdfo_ba* b, float* poses16);
#ifdef __cplusplus
}
#endif
