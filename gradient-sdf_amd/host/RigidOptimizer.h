/* RigidOptimizer / RigidPointOptimizer -- the 6-DoF point-to-SDF tracker interface of the
 * reference (cpp/include/sdf_tracker/RigidOptimizer.h:51-112, RigidPointOptimizer.h:46-73). */
#ifndef GSDF_HOST_RIGID_OPTIMIZER_H_
#define GSDF_HOST_RIGID_OPTIMIZER_H_

#include "MapGradPixelSdf.h"

class RigidOptimizer {
protected:
    int num_iterations_;
    float conv_threshold_;
    float damping_;
    Sdf* tSDF_;                                  /* non-owning, RigidOptimizer.h:62 */
    SE3 pose_ = SE3();                           /* RigidOptimizer.h:64 */
public:
    RigidOptimizer(Sdf* tSDF) : num_iterations_(25), conv_threshold_(1e-3f), damping_(1.f), tSDF_(tSDF) {}   /* :70-76 */
    RigidOptimizer(int num_iterations, float conv_threshold, float damping, Sdf* tSDF)
        : num_iterations_(num_iterations), conv_threshold_(conv_threshold), damping_(damping), tSDF_(tSDF) {}
    virtual ~RigidOptimizer() {}
    void set_num_iterations(int n) { num_iterations_ = n; }            /* :85-97 */
    void set_conv_threshold(float c) { conv_threshold_ = c; }
    void set_damping(float d) { damping_ = d; }
    int num_iterations() const { return num_iterations_; }
    float conv_threshold() const { return conv_threshold_; }
    float damping() const { return damping_; }
    void set_pose(SE3 pose) { pose_ = pose; }                          /* :99 */
    SE3 pose() { return pose_; }                                       /* :101-103 */
    virtual bool optimize(const DepthImage& depth, const Mat3f K) = 0; /* :106 */
};

class RigidPointOptimizer : public RigidOptimizer {
    int last_passes_ = 0;
public:
    RigidPointOptimizer(Sdf* tSDF) : RigidOptimizer(tSDF) {}
    RigidPointOptimizer(int num_iterations, float conv_threshold, float damping, Sdf* tSDF)
        : RigidOptimizer(num_iterations, conv_threshold, damping, tSDF) {}
    /* optimize_sampled(depth, K, sampling) -- RigidPointOptimizer.h:65, .cpp:40-99 (sampling = pixel stride, .cpp:62) */
    bool optimize_sampled(const DepthImage& depth, const Mat3f K, size_t sampling);
    /* optimize() -> optimize_sampled(depth, K, 1) -- RigidPointOptimizer.h:69-72 */
    bool optimize(const DepthImage& depth, const Mat3f K) override { return optimize_sampled(depth, K, 1); }
    int last_passes() const { return last_passes_; }
};

#endif
