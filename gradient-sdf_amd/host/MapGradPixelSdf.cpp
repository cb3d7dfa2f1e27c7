/* MapGradPixelSdf facade: every method forwards to the C-ABI (include/gsdf.h). */
#include "MapGradPixelSdf.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>

#include "exports.h"

void MapGradPixelSdf::check(int rc, const char* what) const {
    if (rc != GSDF_OK) throw std::runtime_error(std::string(what) + ": " + gsdf_last_error());
}

MapGradPixelSdf::MapGradPixelSdf(float voxel_size, float T, int capacity_log2, int device, int max_capacity_log2)
    : voxel_size_(voxel_size), T_(T) {
    check(gsdf_create(&ctx_, voxel_size, T, capacity_log2, device), "gsdf_create");
    /* tsdf_ of the reference grows as the scan does (MapGradPixelSdf.h:65-68): so does this map, up to max_capacity_log2 */
    if (max_capacity_log2 > capacity_log2) check(gsdf_set_auto_grow(ctx_, max_capacity_log2), "gsdf_set_auto_grow");
}

MapGradPixelSdf::~MapGradPixelSdf() { gsdf_destroy(ctx_); }

void MapGradPixelSdf::ensure_frame(const DepthImage& depth, const Mat3f& K, NormalEstimator* NEst) {
    if (frame_w_ == depth.cols && frame_h_ == depth.rows) return;
    const int win = NEst ? NEst->window : 11;
    check(gsdf_normals_init(ctx_, depth.cols, depth.rows, NEst ? NEst->K.data() : K.data(), win), "gsdf_normals_init");
    frame_w_ = depth.cols; frame_h_ = depth.rows;
}

void MapGradPixelSdf::prepare(int W, int H, const Mat3f& K, NormalEstimator* NEst) {
    if (frame_w_ == W && frame_h_ == H) return;
    check(gsdf_normals_init(ctx_, W, H, NEst ? NEst->K.data() : K.data(), NEst ? NEst->window : 11), "gsdf_normals_init");
    frame_w_ = W; frame_h_ = H;
}

void MapGradPixelSdf::update(const ColorImage&, const DepthImage& depth, const Mat3f K, const SE3& pose,
                             NormalEstimator* NEst) {
    if (!NEst) {   /* MapGradPixelSdf.cpp:55-58 */
        std::cerr << "No normal estimation possible - cannot update SDF volume!" << std::endl;
        return;
    }
    ensure_frame(depth, K, NEst);
    const Mat3f R = pose.rotationMatrix();           /* :62 */
    const Vec3f t = pose.translation();              /* :64 */
    check(gsdf_update(ctx_, depth.data(), R.data(), t.data()), "gsdf_update");
    std::cout << "Current frame counter: " << frame_counter() << std::endl;   /* :121 */
}

float MapGradPixelSdf::weights(Vec3f point) const {
    float d, g[3], w;
    check(gsdf_query(ctx_, point.data(), 1, &d, g, &w), "gsdf_query");
    return w;
}

float MapGradPixelSdf::tsdf(Vec3f point, Vec3f* grad_ptr) const {
    float d, g[3], w;
    check(gsdf_query(ctx_, point.data(), 1, &d, g, &w), "gsdf_query");
    if (!(w > 0.f)) throw std::out_of_range("MapGradPixelSdf::tsdf: voxel not in map");   /* tsdf_.at(idx) */
    if (grad_ptr) *grad_ptr = Vec3f(g[0], g[1], g[2]);
    return d;
}

void MapGradPixelSdf::raycast(const Mat3f& K, const SE3& pose, int W, int H, float* depth_out, float* normals_out) const {
    const Mat3f R = pose.rotationMatrix();
    const Vec3f t = pose.translation();
    check(gsdf_raycast(ctx_, K.m, R.m, t.data(), W, H, zmin_, zmax_, depth_out, normals_out), "gsdf_raycast");
}

int64_t MapGradPixelSdf::size() const {
    int64_t n = 0;
    check(gsdf_count(ctx_, &n), "gsdf_count");
    return n;
}

int64_t MapGradPixelSdf::frame_counter() const {
    gsdf_stats st;
    check(gsdf_get_stats(ctx_, &st), "gsdf_get_stats");
    return st.frames;
}

void MapGradPixelSdf::export_arrays(std::vector<int32_t>& keys, std::vector<float>& payload) const {
    const int64_t n = size();
    keys.resize((size_t)n * 3);
    payload.resize((size_t)n * 5);
    int64_t got = 0;
    if (n) check(gsdf_export(ctx_, keys.data(), payload.data(), n, &got, 1, 0), "gsdf_export");
}

SdfLrMap MapGradPixelSdf::get_tsdf() const {
    std::vector<int32_t> k;
    std::vector<float> p;
    export_arrays(k, p);
    SdfLrMap m;
    m.reserve(k.size() / 3);
    for (size_t i = 0; i < k.size() / 3; ++i) {
        SdfVoxel v;
        v.dist = p[5 * i];
        v.grad = Vec3f(p[5 * i + 1], p[5 * i + 2], p[5 * i + 3]);
        v.weight = p[5 * i + 4];
        m.emplace(Vec3i(k[3 * i], k[3 * i + 1], k[3 * i + 2]), v);
    }
    return m;
}

/* getSdf -- MapGradPixelSdf.h:127-129: tsdf_.at(idx), i.e. the stored voxel (dist, RAW gradient sum, weight) */
SdfVoxel MapGradPixelSdf::getSdf(Vec3i idx) const {
    const int32_t key[3] = { idx[0], idx[1], idx[2] };
    float p[5];
    int32_t found = 0;
    check(gsdf_get_voxels(ctx_, key, 1, p, &found), "gsdf_get_voxels");
    if (!found) throw std::out_of_range("MapGradPixelSdf::getSdf: voxel not in map");   /* .at() */
    SdfVoxel v;
    v.dist = p[0]; v.grad = Vec3f(p[1], p[2], p[3]); v.weight = p[4];
    return v;
}

/* extract_pc / save_sdf / extract_mesh -- MapGradPixelSdf.cpp:124-296: the writers live in exports.cpp (also reachable
 * from C through capi_host.cpp for hosts that hold a bare gsdf_ctx) */
bool MapGradPixelSdf::extract_pc(std::string filename) { return gsdf_exports::write_cloud_ply(ctx_, voxel_size_, filename, nullptr); }
bool MapGradPixelSdf::save_sdf(std::string filename) { return gsdf_exports::write_sdf_txt(ctx_, voxel_size_, filename); }
bool MapGradPixelSdf::extract_mesh(std::string filename) { return gsdf_exports::write_mesh_ply(ctx_, voxel_size_, filename, nullptr); }
