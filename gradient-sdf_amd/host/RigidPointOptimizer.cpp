/* RigidPointOptimizer::optimize -- one gsdf_track call; pose_ persists across frames
 * (constant-position motion model, RigidOptimizer.h:64). */
#include "RigidOptimizer.h"

#include <iostream>
#include <stdexcept>

bool RigidPointOptimizer::optimize(const DepthImage& depth, const Mat3f K) {
    MapGradPixelSdf* map = dynamic_cast<MapGradPixelSdf*>(tSDF_);
    if (!map) throw std::runtime_error("RigidPointOptimizer needs a MapGradPixelSdf");
    int conv = 0, passes = 0;
    const int rc = gsdf_track(map->ctx_, depth.data(), K.data(), pose_.pose7(), num_iterations_, conv_threshold_,
                              damping_, &conv, &passes);
    if (rc != GSDF_OK) throw std::runtime_error(std::string("gsdf_track: ") + gsdf_last_error());
    last_passes_ = passes;
    if (conv) std::cout << "... Convergence after " << passes - 1 << " iterations!" << std::endl;   /* .cpp:89 */
    return conv != 0;
}
