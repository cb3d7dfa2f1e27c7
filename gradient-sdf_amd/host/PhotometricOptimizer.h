/*
 * PhotometricOptimizer -- coarse photometric bundle adjustment of the reference
 * (cpp/include/ps_optimizer/PhotometricOptimizer.h:68-186) as a facade over the C-ABI (gsdf_ba_*).
 * The voxel sweeps (getEnergy, solvePose's normal equations, solveDist) run on the GPU over the
 * HBM table; the per-keyframe 6x6 LDLT and the pose update are host-side glue inside libgsdf.
 * The map must have been fused with visibility tracking (MapGradPixelSdf::enable_vis()).
 */
#ifndef GSDF_HOST_PHOTOMETRIC_OPTIMIZER_H_
#define GSDF_HOST_PHOTOMETRIC_OPTIMIZER_H_

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "MapGradPixelSdf.h"

enum LossFunction { L2 = 0, CAUCHY = 1, HUBER = 2, TUKEY = 3, TRUNC_L2 = 4 };   /* loss.h:39-46 */

struct OptSettings {                       /* PhotometricOptimizer.h:49-66 */
    int max_it = 25;
    float conv_threshold = 1e-4f;
    float damping = 1.0f;
    float lambda = 0.5f;                   /* lambda of the weight function (not the LM damping) */
    float reg_weight = 10.0f;
    LossFunction loss = CAUCHY;            /* only TRUNC_L2 changes the computation, as in the reference (:364, :542) */
};

/* float BGR image in [0,1], rows x cols x 3 (cv::Mat CV_32FC3 as produced by ImageLoader::load_color) */
struct ColorImageF {
    int rows = 0, cols = 0;
    std::vector<float> bgr;
};

class PhotometricOptimizer {
    MapGradPixelSdf* tSDF_;
    OptSettings settings_;
    std::vector<int> frame_idx_;
    std::vector<std::shared_ptr<ColorImageF>> images_;
    std::vector<Mat4f> poses_;
    bool uploaded_ = false;

    void check(int rc, const char* what) const {
        if (rc != GSDF_OK) throw std::runtime_error(std::string(what) + ": " + gsdf_last_error());
    }
    void upload() {
        if (uploaded_) return;
        const size_t n = images_.size();
        if (!n || poses_.size() != n || frame_idx_.size() != n) throw std::runtime_error("PhotoBA: images / poses / keyframes differ in length");
        std::vector<float> img, P(16 * n);
        for (size_t i = 0; i < n; ++i) {
            img.insert(img.end(), images_[i]->bgr.begin(), images_[i]->bgr.end());
            for (int k = 0; k < 16; ++k) P[16 * i + k] = poses_[i].m[k];
        }
        check(gsdf_ba_setup(tSDF_->handle(), (int)n, img.data(), P.data(), frame_idx_.data(), settings_.reg_weight), "gsdf_ba_setup");
        check(gsdf_ba_set_loss(tSDF_->handle(), (int)settings_.loss, settings_.lambda), "gsdf_ba_set_loss");
        uploaded_ = true;
    }
    void download() {
        std::vector<float> P(16 * poses_.size());
        check(gsdf_ba_get_poses(tSDF_->handle(), P.data()), "gsdf_ba_get_poses");
        for (size_t i = 0; i < poses_.size(); ++i) for (int k = 0; k < 16; ++k) poses_[i].m[k] = P[16 * i + k];
    }

public:
    /* PhotometricOptimizer(tSDF, voxel_size, K, save_path, settings) -- .cpp:145-157; voxel size and K are the map's */
    explicit PhotometricOptimizer(MapGradPixelSdf* tSDF, OptSettings settings = OptSettings()) : tSDF_(tSDF), settings_(settings) {}

    void setImages(std::vector<std::shared_ptr<ColorImageF>> images) { images_ = std::move(images); uploaded_ = false; }   /* .h:136 */
    void setPoses(const std::vector<Mat4f>& poses) { poses_ = poses; uploaded_ = false; }                                  /* .h:141 */
    void setKeyframes(const std::vector<int>& keyframes) { frame_idx_ = keyframes; uploaded_ = false; }                    /* .h:146 */
    const std::vector<Mat4f>& poses() const { return poses_; }

    float getEnergy() { upload(); float E = 0.f; check(gsdf_ba_energy(tSDF_->handle(), &E), "gsdf_ba_energy"); return E; }           /* .cpp:273 */
    void solveDist(float damping = 1.0f) { upload(); check(gsdf_ba_solve_dist(tSDF_->handle(), damping), "gsdf_ba_solve_dist"); }     /* .cpp:326 */
    void solvePose(float damping = 1.0f) { upload(); check(gsdf_ba_solve_pose(tSDF_->handle(), damping), "gsdf_ba_solve_pose"); download(); }   /* .cpp:499 */
    /* optimize() -- .cpp:611-662; returns the reference's bool (true = converged) */
    bool optimize(std::vector<float>* energies = nullptr) {
        upload();
        std::vector<float> e(2 * (size_t)settings_.max_it + 1);
        int ne = 0, conv = 0;
        check(gsdf_ba_optimize(tSDF_->handle(), settings_.max_it, e.data(), &ne, &conv), "gsdf_ba_optimize");
        download();
        if (energies) energies->assign(e.begin(), e.begin() + ne);
        return conv != 0;
    }
};

#endif
