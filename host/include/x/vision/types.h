// x/vision/types.h -- the input types the update path reads (include/x/vision/types.h:83-198, feature.h:29,
// track.h:32 of the reference): Feature::getX()/getY(), Track (a feature list with a unique id), the window lists,
// and the match records the multi-agent updates consume (MsckfMatch / SlamMatch, types.h:83-116).
#pragma once
#include <memory>
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

typedef unsigned long long uniqueId;                  // types.h:81
// Track (track.h:32): a std::vector<Feature> with an id unique across the run (the tracker assigns it).
class Track : public std::vector<Feature> {
 public:
  using std::vector<Feature>::vector;
  Track() = default;
  uniqueId getId() const { return id_; }
  void setId(uniqueId id) { id_ = id; }
 private:
  uniqueId id_ = 0;
};
using TrackList = std::vector<Track>;
using TrackPtr = std::shared_ptr<Track>;

class SimpleState;
// One of this agent's MSCKF tracks also seen by agent `uav_id` (types.h:83-103).
struct MsckfMatch {
  std::shared_ptr<SimpleState> state;
  int uav_id = -1;
  TrackPtr received_track_ptr;
  uniqueId id_current_track = (uniqueId)-1;
  uniqueId id_received_track = (uniqueId)-1;
  MsckfMatch(int uav_id_, uniqueId id_current_track_, uniqueId id_received_track_, TrackPtr received_track,
             std::shared_ptr<SimpleState> state_)
      : state(std::move(state_)), uav_id(uav_id_), received_track_ptr(std::move(received_track)),
        id_current_track(id_current_track_), id_received_track(id_received_track_) {}
};
using MsckfMatches = std::vector<MsckfMatch>;

// One of this agent's persistent features matched to a persistent feature of agent `uav_id` (types.h:105-116).
struct SlamMatch {
  std::shared_ptr<SimpleState> state;
  int uav_id = -1;
  int current_feature_id = -1;     // slot in this agent's feature state
  int received_feature_id = -1;    // slot in the sender's
  SlamMatch(int uav_id_, int current_feature_id_, int received_feature_id_, std::shared_ptr<SimpleState> state_)
      : state(std::move(state_)), uav_id(uav_id_), current_feature_id(current_feature_id_),
        received_feature_id(received_feature_id_) {}
  SlamMatch() = delete;
};
using SlamMatches = std::vector<SlamMatch>;
}  // namespace x
