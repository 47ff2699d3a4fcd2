// Drives the place-recognition mirror from a text case file (written by tests/test_gpu_place_host.py) and prints
// what x::Database / goodMatches / classifyMatches return, one line per operation.
//   header : k L n_nodes kmax desc_bytes n_words thr min_distance ratio
//   arrays : node_desc, children, word_of_node, node_of_word
//   ops    : A tag n <n*desc_bytes bytes>            add keyframe
//            F uav n <bytes>                         findCandidate with the VLAD of these descriptors
//            M nq nt ncm ncs nrm nrs <bytes q> <bytes t>   knnMatch + goodMatches + classifyMatches
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "x/place_recognition/database.h"

using namespace x;

static Descriptors readDesc(std::istream &in, int rows, int cols) {
  Descriptors d;
  d.rows = rows; d.cols = cols;
  d.data.resize((size_t)rows * cols);
  for (auto &b : d.data) { int v; in >> v; b = (unsigned char)v; }
  return d;
}

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s case.txt\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  PRVocabulary v;
  int nn, nw;
  double thr, min_distance, ratio;
  in >> v.k >> v.L >> nn >> v.kmax >> v.desc_bytes >> nw >> thr >> min_distance >> ratio;
  v.node_desc.resize((size_t)nn * v.desc_bytes);
  for (auto &b : v.node_desc) { int t; in >> t; b = (unsigned char)t; }
  v.children.resize((size_t)nn * v.kmax);
  for (auto &c : v.children) in >> c;
  v.word_of_node.resize(nn);
  for (auto &c : v.word_of_node) in >> c;
  v.node_of_word.resize(nw);
  for (auto &c : v.node_of_word) in >> c;
  xk_handle *xk = nullptr;
  if (xk_create(0, 4, 0, 4, &xk) != XK_OK) { std::fprintf(stderr, "xk_create failed\n"); return 1; }
  try {
    Database db(xk, v, thr, 0, 0, 2048);
    std::string op;
    while (in >> op) {
      if (op == "A") {
        long tag; int n;
        in >> tag >> n;
        db.addKeyframe(std::make_shared<Keyframe>(readDesc(in, n, v.desc_bytes), nullptr, nullptr, tag));
        std::printf("A %d\n", db.size());
      } else if (op == "F") {
        int uav, n;
        in >> uav >> n;
        Candidate c;
        db.findCandidate(uav, db.computeVLAD(readDesc(in, n, v.desc_bytes)), c);
        unsigned long long bits;
        std::memcpy(&bits, &c.score, 8);
        std::printf("F %d %llx %ld %d\n", c.index, bits, c.tag, c.n_descriptors);
      } else if (op == "M") {
        int nq, nt, ncm, ncs, nrm, nrs;
        in >> nq >> nt >> ncm >> ncs >> nrm >> nrs;
        Descriptors q = readDesc(in, nq, v.desc_bytes), t = readDesc(in, nt, v.desc_bytes);
        std::vector<int> idx, dist;
        db.knnMatch(q, t, idx, dist);
        std::vector<GoodMatch> good = goodMatches(idx, dist, min_distance, ratio);
        std::printf("M %zu", good.size());
        for (const GoodMatch &g : good) std::printf(" %d:%d", g.queryIdx, g.trainIdx);
        std::printf(" |");
        for (const ClassifiedMatch &c : classifyMatches(good, ncm, ncs, nrm, nrs)) std::printf(" %d:%d:%d", (int)c.kind, c.current, c.received);
        std::printf("\n");
      }
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    xk_destroy(xk);
    return 1;
  }
  xk_destroy(xk);
  return 0;
}
