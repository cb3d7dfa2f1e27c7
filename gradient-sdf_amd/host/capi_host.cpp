/* C wrappers around the host-side exports so that non-C++ hosts (the Python runner of the
 * frame-sharded config C4) can write the same artefacts as Scan3D from an existing gsdf_ctx:
 * mesh (MapGradPixelSdf.cpp:124-175), point cloud (:177-220), sdf text files (:222-296). */
#include <string>
#include <vector>

#include "../../include/gsdf.h"
#include "MarchingCubes.h"

extern "C" {

/* returns the number of faces written, or -1 */
long gsdf_host_extract_mesh(gsdf_ctx* ctx, float voxel_size, const char* path) {
    int64_t n = 0;
    if (gsdf_count(ctx, &n) != GSDF_OK || n <= 0) return -1;
    std::vector<int32_t> k((size_t)n * 3);
    std::vector<float> p((size_t)n * 5);
    if (gsdf_export(ctx, k.data(), p.data(), n, &n, 1, 0) != GSDF_OK) return -1;
    MarchingCubes mc(voxel_size);
    if (!mc.computeIsoSurface(k, p, 0.f) || !mc.savePly(path)) return -1;
    return (long)mc.faces().size();
}

}
