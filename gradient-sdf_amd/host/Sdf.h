/* Sdf -- abstract map interface of the reference (cpp/include/sdf_tracker/Sdf.h:52-149). */
#ifndef GSDF_HOST_SDF_H_
#define GSDF_HOST_SDF_H_

#include <string>
#include "Image.h"

class Sdf {
public:
    virtual ~Sdf() {}
    /* Sdf.h:113-115 */
    virtual float tsdf(Vec3f point, Vec3f* grad_ptr = nullptr) const = 0;
    virtual float weights(Vec3f point) const = 0;
    /* Sdf.h:117 */
    virtual void update(const ColorImage& color, const DepthImage& depth, const Mat3f K, const SE3& pose,
                        NormalEstimator* NEst = nullptr) = 0;
    /* Sdf.h:119-121 */
    virtual void setup(const ColorImage& color, const DepthImage& depth, const Mat3f K, NormalEstimator* NEst = nullptr) {
        update(color, depth, K, SE3(), NEst);
    }
    virtual void set_zmin(float z_min) = 0;      /* Sdf.h:123-129 */
    virtual void set_zmax(float z_max) = 0;
    /* Sdf.h:135-145 */
    virtual bool extract_mesh(std::string) { return false; }
    virtual bool extract_pc(std::string) { return false; }
    virtual bool save_sdf(std::string) { return false; }
};

#endif
