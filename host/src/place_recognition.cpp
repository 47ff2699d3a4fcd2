// Mirror of src/x/place_recognition/{vlad,database,keyframe}.cpp and the matching front half of
// place_recognition.cpp; the bit work runs in libxk.so.
#include "x/place_recognition/database.h"

#include <stdexcept>

using namespace x;

static void check(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}

Database::Database(xk_handle *xk, const PRVocabulary &v, double pr_score_thr, long payload_doubles, long tracks_doubles,
                   int max_descriptors)
    : xk_(xk), pr_score_thr_(pr_score_thr), desc_bytes_(v.desc_bytes) {
  check(xk_, xk_pr_create(xk_, v.k, v.L, v.nNodes(), v.kmax, v.desc_bytes, v.node_desc.data(), v.children.data(),
                          v.word_of_node.data(), v.node_of_word.data(), (int)v.node_of_word.size(), payload_doubles,
                          tracks_doubles, max_descriptors, &pr_),
        "xk_pr_create");
}

Database::~Database() { xk_pr_destroy(pr_); }

int Database::size() const { return xk_pr_size(pr_); }

VLADVec Database::computeVLAD(const Descriptors &d) {
  VLADVec v((size_t)xk_pr_vlad_bytes(pr_));
  check(xk_, xk_pr_compute_vlad(pr_, d.ptr(), d.rows, v.data()), "xk_pr_compute_vlad");
  return v;
}

void Database::addKeyframe(const KeyframePtr &kf) {
  const Descriptors &d = kf->getDescriptors();             // cv::Mat desc = keyframe->getDescriptors();
  check(xk_, xk_pr_add_keyframe(pr_, d.ptr(), d.rows, kf->payload(), kf->tracks(), kf->tag()), "xk_pr_add_keyframe");
}

void Database::findCandidate(int uav_id, const VLADVec &query_vlad, Candidate &best) {
  best = Candidate();
  if ((int)query_vlad.size() != xk_pr_vlad_bytes(pr_)) throw std::runtime_error("findCandidate: VLAD of the wrong size");
  check(xk_, xk_pr_find_candidate(pr_, uav_id, query_vlad.data(), pr_score_thr_, &best.index, &best.score, &best.tag),
        "xk_pr_find_candidate");
  if (best.index >= 0)
    check(xk_, xk_pr_keyframe(pr_, best.index, &best.d_payload, &best.d_tracks, &best.n_descriptors, nullptr, nullptr),
          "xk_pr_keyframe");
}

Descriptors Database::keyframeDescriptors(int index) {
  Descriptors d;
  d.cols = desc_bytes_;
  check(xk_, xk_pr_keyframe(pr_, index, nullptr, nullptr, &d.rows, nullptr, nullptr), "xk_pr_keyframe");
  d.data.resize((size_t)d.rows * d.cols);
  if (d.rows) check(xk_, xk_pr_keyframe(pr_, index, nullptr, nullptr, nullptr, nullptr, d.data.data()), "xk_pr_keyframe");
  return d;
}

void Database::knnMatch(const Descriptors &rec, const Descriptors &cur, std::vector<int> &idx, std::vector<int> &dist) {
  idx.assign(2 * (size_t)rec.rows, -1);
  dist.assign(2 * (size_t)rec.rows, 0);
  check(xk_, xk_pr_knn_match(pr_, rec.ptr(), rec.rows, cur.ptr(), cur.rows, idx.data(), dist.data()), "xk_pr_knn_match");
}

std::vector<GoodMatch> x::goodMatches(const std::vector<int> &idx, const std::vector<int> &dist, double pr_min_distance,
                                      double pr_ratio_thr, const std::vector<unsigned char> *mask) {
  std::vector<GoodMatch> good;
  for (size_t q = 0; 2 * q + 1 < idx.size(); ++q) {
    if (idx[2 * q + 1] < 0) continue;                        // fewer than two neighbours
    const float d0 = (float)dist[2 * q], d1 = (float)dist[2 * q + 1];   // cv::DMatch::distance is a float
    if (d0 < pr_min_distance && d0 < d1 * pr_ratio_thr) good.push_back({(int)q, idx[2 * q]});
  }
  if (good.empty()) return good;
  if (mask) {                                                // :275-281
    int corr_id = 0;
    for (size_t i = 0; i < mask->size(); i++)
      if (!(*mask)[i]) { good.erase(good.begin() + (long)i - corr_id); corr_id++; }
  }
  std::vector<int> remove_ids;                               // :283-296
  for (size_t i = 0; i < good.size(); i++)
    for (size_t j = i; j < good.size(); j++)
      if (i != j && (good[i].queryIdx == good[j].queryIdx || good[i].trainIdx == good[j].trainIdx)) {
        remove_ids.push_back((int)j);
        break;
      }
  int corr_id = 0;                                           // :297-301
  for (const int remove_id : remove_ids) {
    const long pos = (long)remove_id - corr_id;
    if (pos >= 0 && pos < (long)good.size()) good.erase(good.begin() + pos);
    corr_id++;
  }
  return good;
}

std::vector<ClassifiedMatch> x::classifyMatches(const std::vector<GoodMatch> &good, int n_cur_msckf, int n_cur_slam,
                                                int n_rec_msckf, int n_rec_slam) {
  const int MAX_CURR_MSCKF_ID = n_cur_msckf, MAX_CURR_SLAM_ID = n_cur_slam + MAX_CURR_MSCKF_ID;
  const int MAX_REC_MSCKF_ID = n_rec_msckf, MAX_REC_SLAM_ID = n_rec_slam + MAX_REC_MSCKF_ID;
  std::vector<ClassifiedMatch> out;
  for (const GoodMatch &m : good) {
    if (m.queryIdx < MAX_REC_MSCKF_ID && m.trainIdx >= MAX_CURR_SLAM_ID)
      out.push_back({MatchKind::MSCKF_OPP, m.trainIdx - MAX_CURR_SLAM_ID, m.queryIdx});
    if (m.queryIdx >= MAX_REC_MSCKF_ID && m.queryIdx < MAX_REC_SLAM_ID) {
      if (m.trainIdx >= MAX_CURR_MSCKF_ID && m.trainIdx < MAX_CURR_SLAM_ID)
        out.push_back({MatchKind::SLAM_SLAM, m.trainIdx - MAX_CURR_MSCKF_ID, m.queryIdx - MAX_REC_MSCKF_ID});
      if (m.trainIdx >= MAX_CURR_SLAM_ID)
        out.push_back({MatchKind::SLAM_OPP, m.trainIdx - MAX_CURR_SLAM_ID, m.queryIdx - MAX_REC_MSCKF_ID});
    }
    if (m.queryIdx >= MAX_REC_SLAM_ID && m.trainIdx >= MAX_CURR_SLAM_ID)
      out.push_back({MatchKind::OPP_OPP, m.trainIdx - MAX_CURR_SLAM_ID, m.queryIdx - MAX_REC_SLAM_ID});
  }
  return out;
}
