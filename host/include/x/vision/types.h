// x/vision/types.h -- the input types the update path reads (include/x/vision/types.h:177-198,
// feature.h:29, track.h:32 of the reference): only Feature::getX()/getY() and the window lists.
#pragma once
#include <vector>

namespace x {
struct Attitude { double ax = 0, ay = 0, az = 0, aw = 1; };     // (x,y,z,w)
struct Translation { double tx = 0, ty = 0, tz = 0; };
using AttitudeList = std::vector<Attitude>;
using TranslationList = std::vector<Translation>;

class Feature {
 public:
  Feature() = default;
  Feature(double x, double y) : x_(x), y_(y) {}
  double getX() const { return x_; }   // normalised, undistorted image coordinates
  double getY() const { return y_; }
 private:
  double x_ = 0, y_ = 0;
};
using Track = std::vector<Feature>;
using TrackList = std::vector<Track>;
}  // namespace x
