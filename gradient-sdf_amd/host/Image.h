/* Minimal stand-ins for cv::Mat (CV_32FC1, continuous, metres) and cv::NormalEstimator<float>:
 * the facade only forwards raw pointers and parameters across the C-ABI. */
#ifndef GSDF_HOST_IMAGE_H_
#define GSDF_HOST_IMAGE_H_

#include <vector>
#include "mat.h"

struct DepthImage {                       /* cv::Mat CV_32FC1 -- ImageLoader.h:159-175 */
    int rows = 0, cols = 0;
    std::vector<float> buf;
    const float* data() const { return buf.data(); }
    bool empty() const { return buf.empty(); }
};
struct ColorImage {                       /* accepted and ignored by update() like in the reference */
    int rows = 0, cols = 0;
};

/* cv::NormalEstimator<float>(W, H, K, Size(win, win)) -- normals/NormalEstimator.h:158-165.
 * cache()/compute() run on the GPU inside the engine (gsdf_normals_init / k_normals). */
struct NormalEstimator {
    int width, height, window;
    Mat3f K;
    NormalEstimator(int w, int h, const Mat3f& K_, int win) : width(w), height(h), window(win), K(K_) {}
};

#endif
