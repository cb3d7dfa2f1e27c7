/*
 * MarchingCubes -- host-side iso-surface extraction over the exported voxel map; the facade's
 * stand-in for LayeredMarchingCubesNoColor (cpp/include/mesh/LayeredMarchingCubesNoColor.cpp:354-757).
 *
 * Same semantics: z-y-x sweep over the bounding box, corner numbering bit1=(x+1,y+1,z), bit2=(x+1,y,z),
 * bit4=(x,y,z), bit8=(x,y+1,z), bits 16..128 the same at z+1 (:593-639); a cube is skipped when any
 * corner has weight 0; vertices by linear interpolation with the 1e-7 guards and clamped double mu
 * (:642-662); no vertex de-duplication, degenerate triangles dropped (:686-712); ASCII PLY (:721-757).
 *
 * The case tables are the classic edgeTable / triTable the reference embeds (:67-352), kept as constant data in
 * include/gsdf_mc_tables.h: the triangle list equals the reference's, triangle for triangle, in its order.
 */
#ifndef GSDF_HOST_MARCHING_CUBES_H_
#define GSDF_HOST_MARCHING_CUBES_H_

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "mat.h"

class MarchingCubes {
public:
    explicit MarchingCubes(float voxel_size) : vs_(voxel_size) {}
    /* keys int32[n][3], payload float[n][5] = dist,gx,gy,gz,weight (any order) */
    bool computeIsoSurface(const std::vector<int32_t>& keys, const std::vector<float>& payload, float isoValue = 0.f);
    /* the same mesh from triangles computed elsewhere (gsdf_extract_mesh: 9 floats per triangle, sweep order) */
    void setTriangles(const float* tris, size_t n_tris);
    /* the classic triTable in the layout gsdf_extract_mesh takes: 256 x 16 edge ids, -1 terminated */
    static void fill_table(int8_t out[256 * 16]);
    bool savePly(const std::string& filename) const;
    const std::vector<Vec3f>& vertices() const { return vertices_; }
    const std::vector<std::array<int, 3>>& faces() const { return faces_; }
    /* the case tables (exposed for tests) */
    static const std::vector<int>& triangles(int cube_index);   /* edge ids, 3 per triangle */
    static int edge_mask(int cube_index);

private:
    float vs_;
    std::vector<Vec3f> vertices_;
    std::vector<std::array<int, 3>> faces_;
};

#endif
