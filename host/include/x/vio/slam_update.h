// x/vio/slam_update.h -- the host-side half of x::SlamUpdate that the update path still needs on the CPU
// (src/x/vio/slam_update.cpp:216-242): inverse-depth coordinates of the standard SLAM features about to be initialised.
// (The SLAM rows themselves -- SlamUpdate::processOneTrack, :49-214 -- are built on the GPU by xk_slam_rows.)
#pragma once
#include "x/common/types.h"
#include "x/vision/types.h"

namespace x {
class SlamUpdate {
 public:
  // alpha, beta = the track's LAST observation, rho = rho_0 (slam_update.cpp:216-242)
  static void computeInverseDepthsNew(const TrackList &new_trks, double rho_0, Matrix &ivds) {
    const size_t n = new_trks.size();
    ivds = Matrix::Zero((int)n * 3, 1);
    for (size_t j = 0; j < n; ++j) computeOneInverseDepthNew(new_trks[j].back(), rho_0, (unsigned int)j, ivds);
  }
  static void computeOneInverseDepthNew(const Feature &feature, double rho_0, unsigned int idx, Matrix &ivds) {
    ivds(3 * idx) = feature.getX();
    ivds(3 * idx + 1) = feature.getY();
    ivds(3 * idx + 2) = rho_0;
  }
};
}  // namespace x
