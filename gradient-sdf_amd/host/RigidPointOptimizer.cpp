/* RigidPointOptimizer::optimize_sampled -- one gsdf_track_sampled call; pose_ persists across frames
 * (constant-position motion model, RigidOptimizer.h:64). */
#include "RigidOptimizer.h"

#include <algorithm>
#include <iostream>
#include <stdexcept>

bool RigidPointOptimizer::optimize_sampled(const DepthImage& depth, const Mat3f K, size_t sampling) {
    MapGradPixelSdf* map = dynamic_cast<MapGradPixelSdf*>(tSDF_);
    if (!map) throw std::runtime_error("RigidPointOptimizer needs a MapGradPixelSdf");
    int conv = 0, passes = 0;
    const int rc = gsdf_track_sampled(map->ctx_, depth.data(), K.data(), pose_.pose7(), num_iterations_, conv_threshold_,
                                      damping_, (int)std::min<size_t>(sampling, 1u << 20), &conv, &passes);
    if (rc != GSDF_OK) throw std::runtime_error(std::string("gsdf_track_sampled: ") + gsdf_last_error());
    last_passes_ = passes;
    if (conv) std::cout << "... Convergence after " << passes - 1 << " iterations!" << std::endl;   /* .cpp:89 */
    return conv != 0;
}
