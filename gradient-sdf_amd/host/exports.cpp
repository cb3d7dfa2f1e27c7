/* exports.cpp -- see exports.h.  Rows / voxels follow the (z,y,x)-sorted export (the reference's phmap order is unknowable). */
#include "exports.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>
#include <vector>

#include "MarchingCubes.h"
#include "mat.h"

namespace {
bool export_arrays(gsdf_ctx* ctx, std::vector<int32_t>& keys, std::vector<float>& payload) {
    int64_t n = 0;
    if (gsdf_count(ctx, &n) != GSDF_OK) return false;
    keys.resize((size_t)n * 3);
    payload.resize((size_t)n * 5);
    int64_t got = 0;
    return n == 0 || gsdf_export(ctx, keys.data(), payload.data(), n, &got, 1, 0) == GSDF_OK;
}
}

namespace gsdf_exports {

/* extract_pc -- MapGradPixelSdf.cpp:177-220: voxels with weight >= 5 whose surface point c - dist * 1.2 g^ lies inside
 * the voxel; normal = -1.2 g^. */
bool write_cloud_ply(gsdf_ctx* ctx, float voxel_size, const std::string& filename, long* n_rows) {
    std::vector<int32_t> k;
    std::vector<float> p;
    if (!export_arrays(ctx, k, p)) return false;
    const float voxel_size_2 = (float)(.5 * voxel_size);                 /* :179 */
    std::vector<std::array<float, 6>> pts;
    for (size_t i = 0; i < k.size() / 3; ++i) {
        if (p[5 * i + 4] < 5) continue;                                  /* :184-185 */
        const Vec3f gn = Vec3f(p[5 * i + 1], p[5 * i + 2], p[5 * i + 3]).normalized();
        const Vec3f g = 1.2f * gn;                                       /* :186 */
        const Vec3f d = p[5 * i] * g;                                    /* :187 */
        if (std::fabs(d[0]) < voxel_size_2 && std::fabs(d[1]) < voxel_size_2 && std::fabs(d[2]) < voxel_size_2) {
            const Vec3f c(voxel_size * (float)k[3 * i], voxel_size * (float)k[3 * i + 1], voxel_size * (float)k[3 * i + 2]);
            const Vec3f q = c - d;                                       /* :191 */
            pts.push_back({ q[0], q[1], q[2], -g[0], -g[1], -g[2] });    /* :192 */
        }
    }
    if (n_rows) *n_rows = (long)pts.size();
    std::ofstream f(filename.c_str());
    if (!f.is_open()) return false;
    f << "ply\nformat ascii 1.0\nelement vertex " << pts.size() << "\n"
      << "property float x\nproperty float y\nproperty float z\n"
      << "property float nx\nproperty float ny\nproperty float nz\nend_header\n";
    for (const auto& q : pts) f << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << " " << q[4] << " " << q[5] << "\n";
    return true;
}

/* save_sdf -- MapGradPixelSdf.cpp:222-296: sparse "lin_idx value" text files + grid info. */
bool write_sdf_txt(gsdf_ctx* ctx, float voxel_size, const std::string& filename) {
    std::vector<int32_t> k;
    std::vector<float> p;
    if (!export_arrays(ctx, k, p)) return false;
    int mn[3] = { std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), std::numeric_limits<int>::max() };
    int mx[3] = { std::numeric_limits<int>::min(), std::numeric_limits<int>::min(), std::numeric_limits<int>::min() };
    for (size_t i = 0; i < k.size() / 3; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], k[3 * i + a]); mx[a] = std::max(mx[a], k[3 * i + a]); }
    const int dim[3] = { mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1 };
    std::ofstream grid((filename + "_grid_info.txt").c_str());
    if (!grid.is_open()) { std::cerr << "couldn't save grid_info file!" << std::endl; return false; }
    grid << "voxel size: " << voxel_size << "\n"
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

/* extract_mesh -- MapGradPixelSdf.cpp:124-175 -> LayeredMarchingCubesNoColor::computeIsoSurface + savePly.  Marching cubes
 * runs on the device (gsdf_extract_mesh, tri_table = NULL: the reference's triTable); the triangle list equals
 * computeIsoSurface's, in its order (tests/test_gpu_parity.py, against the oracle's restatement). */
bool write_mesh_ply(gsdf_ctx* ctx, float voxel_size, const std::string& filename, long* n_faces) {
    int64_t n = 0;
    if (gsdf_extract_mesh(ctx, 0.f, nullptr, nullptr, 0, &n) != GSDF_OK || n <= 0) return false;
    std::vector<float> tris((size_t)n * 9);
    if (gsdf_extract_mesh(ctx, 0.f, nullptr, tris.data(), n, &n) != GSDF_OK) return false;
    MarchingCubes mc(voxel_size);
    mc.setTriangles(tris.data(), (size_t)n);
    if (n_faces) *n_faces = (long)mc.faces().size();
    return mc.savePly(filename);
}

} // namespace gsdf_exports
