// xk_place.hip.h -- binary-VLAD request filter and descriptor matching for the multi-agent exchange (gfx950).
//
//   xk_vlad_build     VLAD::computeVLAD          (src/x/place_recognition/vlad.cpp:40-66) with the vocabulary
//                     descent of DBoW3 (third_party/DBow3/src/Vocabulary.cpp:880-914, DescManip.h:72-97)
//   xk_vlad_hamming   VLAD::computeScore's Hamming norm against every stored keyframe
//                     (vlad.cpp:68-75, called from Database::findCandidate, database.cpp:30-49)
//   xk_desc_knn2      matcher_->knnMatch(received, current, matches, 2)   (place_recognition.cpp:249)
//
// Byte / bit work on a few KB: one launch each, everything integer, results bit-exact by construction.
// Descriptors are handled as little-endian 32-bit words (desc_bytes % 4 == 0, <= 64 bytes).
#pragma once
#include <hip/hip_runtime.h>

#define XK_PR_MAXW 16   // 32-bit words per descriptor (ORB: 8)

struct XkVladArgs {
  const unsigned int *desc;       // [n][W] query descriptors
  int n, W;
  const unsigned int *node_desc;  // [n_nodes][W]
  const int *children;            // [n_nodes][kmax], -1 padded, file order
  int kmax;
  const int *word_of_node;        // [n_nodes]
  const int *node_of_word;        // [n_words]
  unsigned int *vlad;             // [clusters][W], zeroed before the launch
};

// One thread per descriptor: greedy descent (first minimum wins: strict '<' over the children in file order),
// then  vlad[word] |= desc ^ centroid(word).
__global__ __launch_bounds__(256) void xk_vlad_build(XkVladArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n) return;
  unsigned int d[XK_PR_MAXW];
#pragma unroll
  for (int w = 0; w < XK_PR_MAXW; ++w) d[w] = (w < a.W) ? a.desc[(size_t)t * a.W + w] : 0u;
  int node = 0;
  while (true) {
    const int *ch = a.children + (size_t)node * a.kmax;
    if (ch[0] < 0) break;                                   // leaf
    unsigned int best = 0xffffffffu;
    int nxt = ch[0];
    for (int c = 0; c < a.kmax; ++c) {
      const int id = ch[c];
      if (id < 0) break;
      const unsigned int *nd = a.node_desc + (size_t)id * a.W;
      unsigned int dist = 0;
#pragma unroll
      for (int w = 0; w < XK_PR_MAXW; ++w)
        if (w < a.W) dist += __popc(d[w] ^ nd[w]);
      if (dist < best) { best = dist; nxt = id; }
    }
    node = nxt;
  }
  const int word = a.word_of_node[node];
  const unsigned int *cen = a.node_desc + (size_t)a.node_of_word[word] * a.W;   // getWord(id): the word's own node
#pragma unroll
  for (int w = 0; w < XK_PR_MAXW; ++w)
    if (w < a.W) {
      const unsigned int x = d[w] ^ cen[w];
      if (x) atomicOr(&a.vlad[(size_t)word * a.W + w], x);
    }
}

// ham[s] = popcount(query ^ vlad_s) for every stored keyframe s: one workgroup per keyframe.
struct XkVladHamArgs {
  const unsigned int *query;   // [VW]
  const unsigned int *store;   // [slots][VW]
  int VW;
  int *ham;                    // [slots]
};
__global__ __launch_bounds__(256) void xk_vlad_hamming(XkVladHamArgs a) {
  __shared__ int part[4];
  const unsigned int *v = a.store + (size_t)blockIdx.x * a.VW;
  int s = 0;
  for (int i = threadIdx.x; i < a.VW; i += blockDim.x) s += __popc(a.query[i] ^ v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.ham[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// Two nearest train descriptors of every query in ascending (distance, train index) order; -1 / INT_MAX-ish
// where the train set has fewer than two rows.  One thread per query, the train set walked through LDS tiles.
struct XkKnnArgs {
  const unsigned int *query, *train;   // [nq][W], [nt][W]
  int nq, nt, W;
  int *idx, *dist;                     // [nq][2]
};
#define XK_KNN_TILE 256
__global__ __launch_bounds__(256) void xk_desc_knn2(XkKnnArgs a) {
  __shared__ unsigned int tile[XK_KNN_TILE * XK_PR_MAXW];
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int d[XK_PR_MAXW];
#pragma unroll
  for (int w = 0; w < XK_PR_MAXW; ++w) d[w] = (q < a.nq && w < a.W) ? a.query[(size_t)q * a.W + w] : 0u;
  int i0 = -1, i1 = -1, d0 = 1 << 30, d1 = 1 << 30;
  for (int base = 0; base < a.nt; base += XK_KNN_TILE) {
    const int cnt = min(XK_KNN_TILE, a.nt - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * a.W; i += blockDim.x) tile[i] = a.train[(size_t)base * a.W + i];
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      int dist = 0;
#pragma unroll
      for (int w = 0; w < XK_PR_MAXW; ++w)
        if (w < a.W) dist += __popc(d[w] ^ tile[t * a.W + w]);
      // train indices arrive in ascending order, so strict '<' keeps the earlier index on ties
      if (dist < d0) { d1 = d0; i1 = i0; d0 = dist; i0 = base + t; }
      else if (dist < d1) { d1 = dist; i1 = base + t; }
    }
  }
  if (q < a.nq) {
    a.idx[2 * q] = i0; a.idx[2 * q + 1] = i1;
    a.dist[2 * q] = d0; a.dist[2 * q + 1] = d1;
  }
}
