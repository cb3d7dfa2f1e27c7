/* C wrappers around the host-side exports so that non-C++ hosts (the Python runner of the
 * frame-sharded config C4) can write the same artefacts as Scan3D from an existing gsdf_ctx:
 * mesh (MapGradPixelSdf.cpp:124-175), point cloud (:177-220), sdf text files (:222-296). */
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/gsdf.h"
#include "MarchingCubes.h"
#include "exports.h"

extern "C" {

/* returns the number of faces written, or -1 (marching cubes on the device, gsdf_extract_mesh) */
long gsdf_host_extract_mesh(gsdf_ctx* ctx, float voxel_size, const char* path) {
    long n = 0;
    return gsdf_exports::write_mesh_ply(ctx, voxel_size, path, &n) ? n : -1;
}
/* MapGradPixelSdf::extract_pc: returns the number of points written, or -1 */
long gsdf_host_extract_pc(gsdf_ctx* ctx, float voxel_size, const char* path) {
    long n = 0;
    return gsdf_exports::write_cloud_ply(ctx, voxel_size, path, &n) ? n : -1;
}
/* MapGradPixelSdf::save_sdf: 0 on success */
int gsdf_host_save_sdf(gsdf_ctx* ctx, float voxel_size, const char* path_prefix) {
    return gsdf_exports::write_sdf_txt(ctx, voxel_size, path_prefix) ? 0 : -1;
}

/* test hook: the device mesh against the host sweep over the exported map (MarchingCubes::computeIsoSurface).
 * Returns the number of triangles when both lists are bit-identical, -1 on an API error, -(2 + i) when
 * triangle i differs (or the counts differ at i). */
long gsdf_host_mesh_check(gsdf_ctx* ctx, float voxel_size) {
    int8_t table[256 * 16];
    MarchingCubes::fill_table(table);
    int64_t n = 0;
    if (gsdf_extract_mesh(ctx, 0.f, table, nullptr, 0, &n) != GSDF_OK) return -1;
    std::vector<float> tris((size_t)std::max<int64_t>(n, 1) * 9);
    if (n > 0 && gsdf_extract_mesh(ctx, 0.f, table, tris.data(), n, &n) != GSDF_OK) return -1;
    int64_t nv = 0;
    if (gsdf_count(ctx, &nv) != GSDF_OK || nv <= 0) return n == 0 ? 0 : -2;
    std::vector<int32_t> k((size_t)nv * 3);
    std::vector<float> p((size_t)nv * 5);
    if (gsdf_export(ctx, k.data(), p.data(), nv, &nv, 1, 0) != GSDF_OK) return -1;
    MarchingCubes mc(voxel_size);
    mc.computeIsoSurface(k, p, 0.f);
    const size_t nh = mc.faces().size();
    for (size_t i = 0; i < std::min<size_t>(nh, (size_t)n); ++i)
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a)
                if (mc.vertices()[3 * i + v][a] != tris[9 * i + 3 * v + a]) return -(long)(2 + i);
    if (nh != (size_t)n) return -(long)(2 + std::min<size_t>(nh, (size_t)n));
    return (long)n;
}

}
