/* MapGradPixelSdf facade: every method forwards to the C-ABI (include/gsdf.h). */
#include "MapGradPixelSdf.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>

#include "MarchingCubes.h"

void MapGradPixelSdf::check(int rc, const char* what) const {
    if (rc != GSDF_OK) throw std::runtime_error(std::string(what) + ": " + gsdf_last_error());
}

MapGradPixelSdf::MapGradPixelSdf(float voxel_size, float T, int capacity_log2, int device)
    : voxel_size_(voxel_size), T_(T) {
    check(gsdf_create(&ctx_, voxel_size, T, capacity_log2, device), "gsdf_create");
}

MapGradPixelSdf::~MapGradPixelSdf() { gsdf_destroy(ctx_); }

void MapGradPixelSdf::ensure_frame(const DepthImage& depth, const Mat3f& K, NormalEstimator* NEst) {
    if (frame_w_ == depth.cols && frame_h_ == depth.rows) return;
    const int win = NEst ? NEst->window : 11;
    check(gsdf_normals_init(ctx_, depth.cols, depth.rows, NEst ? NEst->K.data() : K.data(), win), "gsdf_normals_init");
    frame_w_ = depth.cols; frame_h_ = depth.rows;
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

SdfVoxel MapGradPixelSdf::getSdf(Vec3i idx) const {
    const Vec3f c(voxel_size_ * (float)idx[0], voxel_size_ * (float)idx[1], voxel_size_ * (float)idx[2]);
    float d, g[3], w;
    check(gsdf_query(ctx_, c.data(), 1, &d, g, &w), "gsdf_query");
    if (!(w > 0.f)) throw std::out_of_range("MapGradPixelSdf::getSdf: voxel not in map");
    SdfVoxel v;            /* at the voxel centre phi == dist; the raw gradient is in the export */
    v.dist = d; v.grad = Vec3f(g[0], g[1], g[2]); v.weight = w;
    return v;
}

/* extract_pc -- MapGradPixelSdf.cpp:177-220: voxels with weight >= 5 whose surface point
 * c - dist * 1.2 g^ lies inside the voxel; normal = -1.2 g^.  Rows follow the sorted export. */
bool MapGradPixelSdf::extract_pc(std::string filename) {
    std::vector<int32_t> k;
    std::vector<float> p;
    export_arrays(k, p);
    const float voxel_size_2 = .5f * voxel_size_;
    std::vector<std::array<float, 6>> pts;
    for (size_t i = 0; i < k.size() / 3; ++i) {
        if (p[5 * i + 4] < 5) continue;
        const Vec3f gn = Vec3f(p[5 * i + 1], p[5 * i + 2], p[5 * i + 3]).normalized();
        const Vec3f g = 1.2f * gn;
        const Vec3f d = p[5 * i] * g;
        if (std::fabs(d[0]) < voxel_size_2 && std::fabs(d[1]) < voxel_size_2 && std::fabs(d[2]) < voxel_size_2) {
            const Vec3f c(voxel_size_ * (float)k[3 * i], voxel_size_ * (float)k[3 * i + 1], voxel_size_ * (float)k[3 * i + 2]);
            const Vec3f q = c - d;
            pts.push_back({ q[0], q[1], q[2], -g[0], -g[1], -g[2] });
        }
    }
    std::ofstream f(filename.c_str());
    if (!f.is_open()) return false;
    f << "ply\nformat ascii 1.0\nelement vertex " << pts.size() << "\n"
      << "property float x\nproperty float y\nproperty float z\n"
      << "property float nx\nproperty float ny\nproperty float nz\nend_header\n";
    for (const auto& q : pts) f << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << " " << q[4] << " " << q[5] << "\n";
    return true;
}

/* save_sdf -- MapGradPixelSdf.cpp:222-296: sparse "lin_idx value" text files + grid info. */
bool MapGradPixelSdf::save_sdf(std::string filename) {
    std::vector<int32_t> k;
    std::vector<float> p;
    export_arrays(k, p);
    int mn[3] = { std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), std::numeric_limits<int>::max() };
    int mx[3] = { std::numeric_limits<int>::min(), std::numeric_limits<int>::min(), std::numeric_limits<int>::min() };
    for (size_t i = 0; i < k.size() / 3; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], k[3 * i + a]); mx[a] = std::max(mx[a], k[3 * i + a]); }
    const int dim[3] = { mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1 };
    std::ofstream grid((filename + "_grid_info.txt").c_str());
    if (!grid.is_open()) { std::cerr << "couldn't save grid_info file!" << std::endl; return false; }
    grid << "voxel size: " << voxel_size_ << "\n"
         << "voxel dim: " << dim[0] << " " << dim[1] << " " << dim[2] << "\n"
         << "voxel min: " << mn[0] << " " << mn[1] << " " << mn[2] << "\n"
         << "voxel max: " << mx[0] << " " << mx[1] << " " << mx[2] << "\n";
    std::ofstream fd((filename + "_sdf_d.txt").c_str()), fw((filename + "_sdf_weight.txt").c_str());
    std::ofstream f0((filename + "_sdf_n0.txt").c_str()), f1((filename + "_sdf_n1.txt").c_str()), f2((filename + "_sdf_n2.txt").c_str());
    if (!fd.is_open() || !fw.is_open() || !f0.is_open() || !f1.is_open() || !f2.is_open()) {
        std::cerr << "couldn't save sdf or sdf weight file!" << std::endl;
        return false;
    }
    for (size_t i = 0; i < k.size() / 3; ++i) {
        const int lin = dim[0] * dim[1] * (k[3 * i + 2] - mn[2]) + dim[0] * (k[3 * i + 1] - mn[1]) + k[3 * i] - mn[0];
        fd << lin << " " << p[5 * i] << "\n";
        fw << lin << " " << p[5 * i + 4] << "\n";
        f0 << lin << " " << p[5 * i + 1] << "\n";
        f1 << lin << " " << p[5 * i + 2] << "\n";
        f2 << lin << " " << p[5 * i + 3] << "\n";
    }
    return true;
}

/* extract_mesh -- MapGradPixelSdf.cpp:124-175 -> LayeredMarchingCubesNoColor on the exported map */
bool MapGradPixelSdf::extract_mesh(std::string filename) {
    /* marching cubes on the device (gsdf_extract_mesh, tri_table = NULL: the reference's triTable); the triangle list
     * equals LayeredMarchingCubesNoColor::computeIsoSurface's, in its order (tests/test_gpu_parity.py vs the oracle) */
    int64_t n = 0;
    check(gsdf_extract_mesh(ctx_, 0.f, nullptr, nullptr, 0, &n), "gsdf_extract_mesh");
    if (n <= 0) return false;
    std::vector<float> tris((size_t)n * 9);
    check(gsdf_extract_mesh(ctx_, 0.f, nullptr, tris.data(), n, &n), "gsdf_extract_mesh");
    MarchingCubes mc(voxel_size_);
    mc.setTriangles(tris.data(), (size_t)n);
    return mc.savePly(filename);
}
