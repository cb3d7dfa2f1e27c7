/*
 * exports.h -- the file exports of MapGradPixelSdf (cpp/include/sdf_tracker/MapGradPixelSdf.cpp:124-296) over a gsdf_ctx:
 * mesh PLY (extract_mesh -> LayeredMarchingCubesNoColor), point-cloud PLY (extract_pc), sparse sdf text files (save_sdf).
 */
#ifndef GSDF_HOST_EXPORTS_H_
#define GSDF_HOST_EXPORTS_H_

#include <string>

#include "../../include/gsdf.h"

namespace gsdf_exports {
/* extract_pc -- MapGradPixelSdf.cpp:177-220; *n_rows (nullable) = points written */
bool write_cloud_ply(gsdf_ctx* ctx, float voxel_size, const std::string& filename, long* n_rows);
/* save_sdf -- MapGradPixelSdf.cpp:222-296 */
bool write_sdf_txt(gsdf_ctx* ctx, float voxel_size, const std::string& filename);
/* extract_mesh -- MapGradPixelSdf.cpp:124-175, marching cubes on the device; *n_faces (nullable) = faces written */
bool write_mesh_ply(gsdf_ctx* ctx, float voxel_size, const std::string& filename, long* n_faces);
}

#endif
