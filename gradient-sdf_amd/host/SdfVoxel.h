/* SdfVoxel -- the per-voxel payload of the reference (cpp/include/sdf_voxel/SdfVoxel.h:45-57). */
#ifndef GSDF_HOST_SDF_VOXEL_H_
#define GSDF_HOST_SDF_VOXEL_H_

#include <unordered_map>
#include "mat.h"

struct SdfVoxel {
    float dist = 0.f;
    Vec3f grad;
    float weight = 0.f;
};

/* what get_tsdf() returns (MapGradPixelSdf.h:133-138); the live map stays in HBM */
using SdfLrMap = std::unordered_map<Vec3i, SdfVoxel, Vec3iHash>;

#endif
