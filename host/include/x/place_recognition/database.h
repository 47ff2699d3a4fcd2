// x/place_recognition/database.h -- mirror of x::VLAD, x::Keyframe and x::Database
// (include/x/place_recognition/{vlad,keyframe,database}.h, src/x/place_recognition/{vlad,keyframe,database}.cpp)
// and of the matching front half of PlaceRecognition::findCorrespondences (place_recognition.cpp:137-390).
//
// The reference keeps descriptors and VLADs in cv::Mat and the vocabulary in a DBoW3::Vocabulary; here a
// descriptor set is a row-major byte matrix and the vocabulary is the plain tree DBoW3 stores (PRVocabulary).
// Every bit operation -- the vocabulary descent, the XOR / OR aggregation, the Hamming norms, the 2-NN search --
// runs on the GPU behind the xk_pr_* entry points (include/xk.h); keyframes (SimpleState payload, tracks,
// descriptors, VLAD) stay in device memory.  There is no CPU fallback.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "xk.h"

namespace x {

struct Descriptors {               // cv::Mat of CV_8UC1 rows
  int rows = 0, cols = 32;
  std::vector<unsigned char> data;
  const unsigned char *ptr() const { return data.empty() ? nullptr : data.data(); }
};
using VLADVec = std::vector<unsigned char>;   // clusters x descriptor bytes (types.h:36: cv::Mat)

struct PRVocabulary {              // what DBoW3::Vocabulary holds (types.h:33), as Vocabulary::fromStream reads it
  int k = 0, L = 0, kmax = 0, desc_bytes = 32;
  std::vector<unsigned char> node_desc;       // [n_nodes][desc_bytes]
  std::vector<int> children;                  // [n_nodes][kmax], -1 padded, file order
  std::vector<int> word_of_node, node_of_word;
  int nNodes() const { return (int)word_of_node.size(); }
};

class Keyframe {                   // keyframe.h / keyframe.cpp: the store keeps its payload on the device
 public:
  Keyframe(Descriptors descriptors, const double *d_payload, const double *d_tracks, long tag)
      : descriptors_(std::move(descriptors)), d_payload_(d_payload), d_tracks_(d_tracks), tag_(tag) {}
  const Descriptors &getDescriptors() const { return descriptors_; }   // MSCKF, SLAM, OPP order (keyframe.cpp:40-52)
  const double *payload() const { return d_payload_; }                 // DEVICE, xk_pack_payload layout (SimpleState)
  const double *tracks() const { return d_tracks_; }                   // DEVICE, packed track observations
  long tag() const { return tag_; }
 private:
  Descriptors descriptors_;
  const double *d_payload_, *d_tracks_;
  long tag_;
};
using KeyframePtr = std::shared_ptr<Keyframe>;

struct Candidate {                 // what VIO::processOtherRequests sends back (vio.cpp:489-495), device pointers
  int index = -1;                  // position in the database, oldest first; -1: no candidate
  double score = 0.0;
  long tag = -1;
  const double *d_payload = nullptr, *d_tracks = nullptr;
  int n_descriptors = 0;
};

class Database {                   // database.h / database.cpp
 public:
  // payload_doubles / tracks_doubles: sizes of the per-keyframe device buffers the store copies
  Database(xk_handle *xk, const PRVocabulary &vocabulary, double pr_score_thr, long payload_doubles, long tracks_doubles,
           int max_descriptors = 1024);
  ~Database();
  Database(const Database &) = delete;
  Database &operator=(const Database &) = delete;

  void addKeyframe(const KeyframePtr &keyframe);                         // database.cpp:51-61
  void findCandidate(int uav_id, const VLADVec &query_vlad, Candidate &best_candidate);   // :30-49
  VLADVec computeVLAD(const Descriptors &descriptors);                   // :26-28 -> VLAD::computeVLAD (vlad.cpp:40-66)
  Descriptors keyframeDescriptors(int index);
  int size() const;

  // matcher_->knnMatch(received, current, matches_, 2) (place_recognition.cpp:249): idx / dist [rows][2]
  void knnMatch(const Descriptors &received, const Descriptors &current, std::vector<int> &idx, std::vector<int> &dist);

 private:
  xk_handle *xk_;
  xk_pr *pr_ = nullptr;
  double pr_score_thr_;
  int desc_bytes_;
};

// The rest of findCorrespondences' non-GT branch on the 2-NN result: distance + ratio test (:252-263), optional
// RANSAC inlier mask (:268-281; the essential-matrix estimate itself is the caller's), duplicate removal (:283-301).
struct GoodMatch { int queryIdx, trainIdx; };
std::vector<GoodMatch> goodMatches(const std::vector<int> &idx, const std::vector<int> &dist, double pr_min_distance,
                                   double pr_ratio_thr, const std::vector<unsigned char> *inlier_mask = nullptr);

// Classification of place_recognition.cpp:311-388.
enum class MatchKind { MSCKF_OPP, SLAM_SLAM, SLAM_OPP, OPP_OPP };
struct ClassifiedMatch { MatchKind kind; int current, received; };
std::vector<ClassifiedMatch> classifyMatches(const std::vector<GoodMatch> &good, int n_cur_msckf, int n_cur_slam,
                                             int n_rec_msckf, int n_rec_slam);
}  // namespace x
