/*
 * MapGradPixelSdf -- the Gradient-SDF voxel hash map of the reference
 * (cpp/include/sdf_tracker/MapGradPixelSdf.h:51-151) as a facade over the C-ABI: the map lives in
 * HBM inside a gsdf_ctx, every method is one include/gsdf.h call.
 */
#ifndef GSDF_HOST_MAP_GRAD_PIXEL_SDF_H_
#define GSDF_HOST_MAP_GRAD_PIXEL_SDF_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gsdf.h"
#include "Sdf.h"
#include "SdfVoxel.h"

class RigidPointOptimizer;

class MapGradPixelSdf : public Sdf {
    friend class RigidPointOptimizer;            /* like Sdf.h:58 */
    gsdf_ctx* ctx_ = nullptr;
    float voxel_size_, T_;
    int frame_w_ = 0, frame_h_ = 0;
    void ensure_frame(const DepthImage& depth, const Mat3f& K, NormalEstimator* NEst);
    void check(int rc, const char* what) const;

public:
    /* MapGradPixelSdf(voxel_size, T) -- MapGradPixelSdf.h:99-103; capacity/device are new */
    /* capacity_log2: 2^c voxel records to start with; the table doubles by itself as the map grows, up to 2^max_capacity_log2
     * (8 GiB of records at 28; pass max == capacity for a fixed table, which then reports GSDF_ERR_TABLE_FULL when it overflows) */
    MapGradPixelSdf(float voxel_size, float T, int capacity_log2 = 22, int device = 0, int max_capacity_log2 = 28);
    ~MapGradPixelSdf() override;
    MapGradPixelSdf(const MapGradPixelSdf&) = delete;
    MapGradPixelSdf& operator=(const MapGradPixelSdf&) = delete;

    float tsdf(Vec3f point, Vec3f* grad_ptr) const override;          /* MapGradPixelSdf.h:109-115 */
    float weights(Vec3f point) const override;                         /* MapGradPixelSdf.h:117-125 */
    SdfVoxel getSdf(Vec3i idx) const;                                  /* MapGradPixelSdf.h:127-129 */
    void update(const ColorImage& color, const DepthImage& depth, const Mat3f K, const SE3& pose,
                NormalEstimator* NEst) override;                       /* MapGradPixelSdf.cpp:43-122 */
    void set_zmin(float z_min) override { zmin_ = z_min; gsdf_set_zrange(ctx_, zmin_, zmax_); }
    void set_zmax(float z_max) override { zmax_ = z_max; gsdf_set_zrange(ctx_, zmin_, zmax_); }

    SdfLrMap get_tsdf() const;                                         /* MapGradPixelSdf.h:133-138 (by value) */
    /* vis_ (MapGradPixelSdf.h:70,140-142): opt-in here, call before the first update() */
    void enable_vis(int max_frames) { check(gsdf_enable_vis(ctx_, max_frames), "gsdf_enable_vis"); }
    /* sorted (z,y,x) arrays: keys int32[n][3], payload float[n][5] = dist,gx,gy,gz,weight */
    void export_arrays(std::vector<int32_t>& keys, std::vector<float>& payload) const;
    int64_t size() const;
    int64_t frame_counter() const;

    /* voxel-hash raycaster (BASELINE.json north_star; the reference has none): depth (camera z, 0 = no hit)
     * and camera-frame normals (3 planes, may be null) of the map seen from `pose` through K */
    void raycast(const Mat3f& K, const SE3& pose, int W, int H, float* depth_out, float* normals_out) const;

    bool extract_pc(std::string filename) override;                    /* MapGradPixelSdf.cpp:177-220 */
    bool save_sdf(std::string filename) override;                      /* MapGradPixelSdf.cpp:222-296 */
    bool extract_mesh(std::string filename) override;                  /* MapGradPixelSdf.cpp:124-175 */

    /* fix the frame size / intrinsics / normal-estimator window before driving the *_dev entries directly (update() does
     * it on the first frame) */
    void prepare(int W, int H, const Mat3f& K, NormalEstimator* NEst);
    gsdf_ctx* handle() const { return ctx_; }
    float voxel_size() const { return voxel_size_; }

private:
    float zmin_ = 0.5f, zmax_ = 3.5f;                                  /* Sdf.h:67-68 */
};

#endif
