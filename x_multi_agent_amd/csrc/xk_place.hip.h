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
// where the train set has fewer than two rows.  A workgroup takes 32 queries; its 256 threads are 32 queries x 8
// segments of every 256-row train tile (streamed through LDS), and the eight partial top-2 lists of a query are
// merged lexicographically on (distance, index) -- so the result does not depend on how the scan was split.
struct XkKnnArgs {
  const unsigned int *query, *train;   // [nq][W], [nt][W]
  int nq, nt, W;
  int *idx, *dist;                     // [nq][2]
};
#define XK_KNN_TILE 256
#define XK_KNN_Q 32
__device__ __forceinline__ void xk_knn_push(int d, int i, int &d0, int &i0, int &d1, int &i1) {
  if (d < d0 || (d == d0 && i < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
  else if (d < d1 || (d == d1 && i < i1)) { d1 = d; i1 = i; }
}
__global__ __launch_bounds__(256) void xk_desc_knn2(XkKnnArgs a) {
  __shared__ unsigned int tile[XK_KNN_TILE * XK_PR_MAXW];
  __shared__ int part[8][XK_KNN_Q][4];
  const int ql = threadIdx.x & (XK_KNN_Q - 1), seg = threadIdx.x >> 5;
  const int q = blockIdx.x * XK_KNN_Q + ql;
  unsigned int d[XK_PR_MAXW];
#pragma unroll
  for (int w = 0; w < XK_PR_MAXW; ++w) d[w] = (q < a.nq && w < a.W) ? a.query[(size_t)q * a.W + w] : 0u;
  int i0 = -1, i1 = -1, d0 = 1 << 30, d1 = 1 << 30;
  for (int base = 0; base < a.nt; base += XK_KNN_TILE) {
    const int cnt = min(XK_KNN_TILE, a.nt - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * a.W; i += blockDim.x) tile[i] = a.train[(size_t)base * a.W + i];
    __syncthreads();
    const int t1 = min(cnt, 32 * seg + 32);
    for (int t = 32 * seg; t < t1; ++t) {
      int dist = 0;
#pragma unroll
      for (int w = 0; w < XK_PR_MAXW; ++w)
        if (w < a.W) dist += __popc(d[w] ^ tile[t * a.W + w]);
      // (within a thread the train indices arrive in ascending order: strict '<' keeps the earlier one on ties)
      if (dist < d0) { d1 = d0; i1 = i0; d0 = dist; i0 = base + t; }
      else if (dist < d1) { d1 = dist; i1 = base + t; }
    }
  }
  part[seg][ql][0] = d0; part[seg][ql][1] = i0; part[seg][ql][2] = d1; part[seg][ql][3] = i1;
  __syncthreads();
  if (seg == 0 && q < a.nq) {
    int e0 = 1 << 30, j0 = -1, e1 = 1 << 30, j1 = -1;
    for (int s = 0; s < 8; ++s) {
      if (part[s][ql][1] >= 0) xk_knn_push(part[s][ql][0], part[s][ql][1], e0, j0, e1, j1);
      if (part[s][ql][3] >= 0) xk_knn_push(part[s][ql][2], part[s][ql][3], e0, j0, e1, j1);
    }
    a.idx[2 * q] = j0; a.idx[2 * q + 1] = j1;
    a.dist[2 * q] = e0; a.dist[2 * q + 1] = e1;
  }
}
