// One request/response tick of the multi-agent protocol, in C++ over the C ABIs (include/xk.h, include/xk_fleet.h) -- what
// VIO::processOtherRequests and VIO::processOtherMeasurements compose in the reference (vio.cpp:462-570):
//
//   every agent   stores a keyframe: its SimpleState snapshot packed ON THE DEVICE (xk_pack_payload) + the observations of
//                 the tracks it shares + the descriptors of what it sees                       (x::Database::addKeyframe)
//   requester     sends the binary VLAD of its current descriptors to a peer                   (computeVLAD, xk_fleet_send_recv)
//   responder     scores the request against its keyframe store and answers with the best keyframe's snapshot + tracks,
//                 straight from HBM into the RCCL send buffer                                   (findCandidate, device copies)
//   requester     fuses the shared tracks against the snapshot where RCCL left it              (xk_ci_round_device:
//                 MsckfUpdate CI block + Updater::applyCI per entry), then runs its regular visual update on the covariance
//                 the CI entries left                                                           (xk_visual_update_staged)
//
// Two ways to run it:
//   xk_fleet_example case.bin out.bin                      LOOP-BACK, one process, one GPU: this process plays both agents (two
//                                                          handles, two keyframe stores); every message still goes through
//                                                          xk_fleet_send_recv -- a communicator of one rank, sent to itself --
//                                                          and the receiving agent reads it where it landed.  This is what the
//                                                          GPU test runs (a one-GPU box cannot host two RCCL ranks).
//   xk_fleet_example case.bin out.bin 2 <rank> <idfile>    TWO RANKS over RCCL (one GPU each; rank 0 writes the 128-byte unique
//                                                          id to <idfile>, rank 1 waits for it): both agents request, answer
//                                                          and fuse symmetrically; rank r writes out.bin.<r>.
//   xk_fleet_example case.bin out.bin dry <N> <rank> <idfile>   DRY RUN of rank <rank> in a ring of N agents (first-contact insurance:
//                                                          no multi-GPU node is available to the builder).  One process walks
//                                                          the multi-rank branch -- unique-id file written AND read back,
//                                                          device = rank mod visible devices, rank-indexed scenario, ring partners
//                                                          (rank asks rank+1, answers rank-1), message sizes -- with every
//                                                          send/recv on a ONE-rank communicator and the partners' messages produced
//                                                          by an in-process agent whose buffers are already in HBM.
//   in : N K n_shared sigma_img ci_msckf_w pr_score_thr | vocabulary: k L n_nodes kmax desc_bytes n_words, node_desc, children,
//        word_of_node, node_of_word | per agent (2): q[4N] p[3N] P[n*n] L_k[K] obs[2 sum L] n_kf_desc kf_desc[..] n_q_desc q_desc[..]
//   out: found tag n_fused | correction[n] | P_post[n*n]        (of the requester: agent 0 in loop-back mode)
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "x/place_recognition/database.h"
#include "xk.h"
#include "xk_fleet.h"

using namespace x;

#define HIPOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
static void xkok(xk_handle *h, int rc, const char *what) {
  if (rc != XK_OK) throw std::runtime_error(std::string(what) + ": " + xk_strerror(rc) + " (" + (h ? xk_last_error(h) : "") + ")");
}

static std::vector<double> slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<double> v(sz / sizeof(double));
  if (fread(v.data(), sizeof(double), v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}

struct Scenario {                        // one agent's problem + what it sees
  std::vector<double> q, p, P, obs;
  std::vector<int> off;
  Descriptors kf_desc, query_desc;
};

struct Agent {
  int id = 0, N = 0, K = 0, n = 0, n_shared = 0;
  long pay_n = 0, trk_n = 0, vlad_n = 0;  // doubles
  xk_handle *h = nullptr;
  xk_fleet *f = nullptr;
  std::unique_ptr<Database> db;
  const Scenario *sc = nullptr;
  // device buffers: the keyframe being stored, the request / response this agent SENDS, and where messages LAND
  double *d_pay = nullptr, *d_trk = nullptr, *d_vlad_out = nullptr, *d_vlad_in = nullptr, *d_resp_out = nullptr, *d_resp_in = nullptr;
  double *d_ci_pay = nullptr, *d_ci_trk = nullptr;   // [2][pay_n], [2][trk_n]: what xk_ci_round_device reads
  hipStream_t stream = nullptr;

  void create(int id_, int N_, int K_, int n_shared_, const Scenario &s, const PRVocabulary &voc, double thr, int world, int rank,
              const unsigned char *uid) {
    id = id_; N = N_; K = K_; n = 15 + 6 * N; n_shared = n_shared_; sc = &s;
    xkok(nullptr, xk_create(0, N, 0, K, &h), "xk_create");
    stream = (hipStream_t)xk_stream(h);
    pay_n = xk_payload_doubles(N, 0);
    trk_n = (long)n_shared * (1 + 2 * N);
    db = std::make_unique<Database>(h, voc, thr, pay_n, trk_n, 1024);
    vlad_n = 1;
    for (int l = 0; l < voc.L; ++l) vlad_n *= voc.k;
    vlad_n = vlad_n * voc.desc_bytes / 8;
    xkok(h, xk_fleet_create(h, uid, world, rank, &f), "xk_fleet_create");
    for (double **b : {&d_pay, &d_resp_out, &d_resp_in}) HIPOK(hipMalloc((void **)b, sizeof(double) * (2 + pay_n + trk_n)));
    HIPOK(hipMalloc((void **)&d_trk, sizeof(double) * trk_n));
    HIPOK(hipMalloc((void **)&d_vlad_out, sizeof(double) * vlad_n));
    HIPOK(hipMalloc((void **)&d_vlad_in, sizeof(double) * vlad_n));
    HIPOK(hipMalloc((void **)&d_ci_pay, sizeof(double) * 2 * pay_n));
    HIPOK(hipMalloc((void **)&d_ci_trk, sizeof(double) * 2 * trk_n));
    // the agent's filter state: window, tracks, prior -- resident from here on
    xkok(h, xk_stage_window(h, s.q.data(), s.p.data(), N), "xk_stage_window");
    xkok(h, xk_stage_tracks(h, s.off.data(), s.obs.data(), K), "xk_stage_tracks");
    xkok(h, xk_stage_slam(h, nullptr, nullptr, nullptr, nullptr, 0), "xk_stage_slam");
    xkok(h, xk_upload_P(h, s.P.data(), n, n), "xk_upload_P");
    // the observations of the shared tracks travel next to the snapshot: [n_shared][1 + 2N], length first
    std::vector<double> pk(trk_n, 0.0);
    for (int j = 0; j < n_shared; ++j) {
      const int L = s.off[j + 1] - s.off[j];
      pk[(size_t)j * (1 + 2 * N)] = L;
      memcpy(&pk[(size_t)j * (1 + 2 * N) + 1], &s.obs[2 * (size_t)s.off[j]], sizeof(double) * 2 * L);
    }
    HIPOK(hipMemcpy(d_trk, pk.data(), sizeof(double) * trk_n, hipMemcpyHostToDevice));
  }

  // VIO keyframe insertion (vio.cpp:447-450 assembles the SimpleState; Database::addKeyframe): snapshot packed on the device
  void storeKeyframe(long tag) {
    double dyn[16] = {0};
    dyn[9] = 1.0;
    xkok(h, xk_pack_payload(h, (double)id, (double)tag, dyn, d_pay, nullptr), "xk_pack_payload");
    HIPOK(hipStreamSynchronize(stream));
    db->addKeyframe(std::make_shared<Keyframe>(sc->kf_desc, d_pay, d_trk, tag));
  }
  // VIO::getDescriptors -> Database::computeVLAD: the request, left in this agent's send buffer
  void prepareRequest() {
    const VLADVec v = db->computeVLAD(sc->query_desc);
    if ((long)v.size() != vlad_n * 8) throw std::runtime_error("VLAD size");
    HIPOK(hipMemcpyAsync(d_vlad_out, v.data(), v.size(), hipMemcpyHostToDevice, stream));
  }
  // VIO::processOtherRequests (vio.cpp:462-496): score the request that landed at `d_request`, answer into d_resp_out
  void answer(int requester, const double *d_request) {
    VLADVec v((size_t)vlad_n * 8);
    HIPOK(hipMemcpyAsync(v.data(), d_request, v.size(), hipMemcpyDeviceToHost, stream));
    HIPOK(hipStreamSynchronize(stream));
    Candidate c;
    db->findCandidate(requester, v, c);
    const double hdr[2] = {c.index >= 0 ? 1.0 : 0.0, (double)c.tag};
    HIPOK(hipMemcpyAsync(d_resp_out, hdr, sizeof(hdr), hipMemcpyHostToDevice, stream));
    if (c.index >= 0) {   // the stored keyframe goes from HBM to the send buffer; it never visits the host
      HIPOK(hipMemcpyAsync(d_resp_out + 2, c.d_payload, sizeof(double) * pay_n, hipMemcpyDeviceToDevice, stream));
      HIPOK(hipMemcpyAsync(d_resp_out + 2 + pay_n, c.d_tracks, sizeof(double) * trk_n, hipMemcpyDeviceToDevice, stream));
    }
    HIPOK(hipStreamSynchronize(stream));
  }
  // VIO::processOtherMeasurements (vio.cpp:498-570) on the response that landed at `d_response`, then the regular update
  void fuseAndUpdate(const double *d_response, double sigma_img, double ci_w, std::vector<double> &out) {
    double hdr[2];
    HIPOK(hipMemcpyAsync(hdr, d_response, sizeof(hdr), hipMemcpyDeviceToHost, stream));
    HIPOK(hipStreamSynchronize(stream));
    int n_fused = 0;
    if (hdr[0] != 0.0) {
      // slot 0 = this agent (its snapshot slot is not read), slot 1 = the sender, exactly where RCCL left its snapshot
      HIPOK(hipMemcpyAsync(d_ci_pay + pay_n, d_response + 2, sizeof(double) * pay_n, hipMemcpyDeviceToDevice, stream));
      HIPOK(hipMemcpyAsync(d_ci_trk, d_trk, sizeof(double) * trk_n, hipMemcpyDeviceToDevice, stream));
      HIPOK(hipMemcpyAsync(d_ci_trk + trk_n, d_response + 2 + pay_n, sizeof(double) * trk_n, hipMemcpyDeviceToDevice, stream));
      // only the lengths of the shared tracks and the window sizes come to the host
      std::vector<int> tl(2 * n_shared), nv(2), self(n_shared);
      std::vector<double> len(n_shared), hdr8(8);
      for (int j = 0; j < n_shared; ++j) {
        HIPOK(hipMemcpyAsync(&len[j], d_response + 2 + pay_n + (size_t)j * (1 + 2 * N), sizeof(double), hipMemcpyDeviceToHost, stream));
        tl[j] = sc->off[j + 1] - sc->off[j];
        self[j] = j;
      }
      HIPOK(hipMemcpyAsync(hdr8.data(), d_response + 2, sizeof(double) * 8, hipMemcpyDeviceToHost, stream));
      HIPOK(hipStreamSynchronize(stream));
      for (int j = 0; j < n_shared; ++j) tl[n_shared + j] = (int)len[j];
      nv[0] = N; nv[1] = (int)hdr8[5];
      xkok(h, xk_ci_round_device(h, d_ci_pay, pay_n, 2, 0, d_ci_trk, n_shared, tl.data(), nv.data(), self.data(), sigma_img, ci_w,
                                 &n_fused, nullptr), "xk_ci_round_device");
    }
    std::vector<double> corr(n);
    std::vector<int> inl(K);
    xkok(h, xk_visual_update_staged(h, sigma_img, corr.data(), inl.data(), nullptr, nullptr, nullptr), "xk_visual_update_staged");
    std::vector<double> P((size_t)n * n);
    xkok(h, xk_download_P(h, P.data(), n, n), "xk_download_P");
    out = {hdr[0], hdr[1], (double)n_fused};
    out.insert(out.end(), corr.begin(), corr.end());
    out.insert(out.end(), P.begin(), P.end());
  }
  void destroy() {
    db.reset();
    if (f) xk_fleet_destroy(f);
    for (double *b : {d_pay, d_trk, d_vlad_out, d_vlad_in, d_resp_out, d_resp_in, d_ci_pay, d_ci_trk})
      if (b) hipFree(b);
    if (h) xk_destroy(h);
  }
};

static void writeOut(const std::string &path, const std::vector<double> &v) {
  FILE *f = fopen(path.c_str(), "wb");
  fwrite(v.data(), sizeof(double), v.size(), f);
  fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s case.bin out.bin [2 rank idfile | dry N rank idfile]\n", argv[0]); return 2; }
  const int world = argc > 3 ? (std::string(argv[3]) == "dry" ? -1 : atoi(argv[3])) : 1, rank = argc > 4 ? atoi(argv[4]) : 0;
  const std::vector<double> in = slurp(argv[1]);
  size_t at = 0;
  const int N = (int)in[at++], K = (int)in[at++], n_shared = (int)in[at++];
  const double sigma_img = in[at++], ci_w = in[at++], thr = in[at++];
  PRVocabulary voc;
  voc.k = (int)in[at++]; voc.L = (int)in[at++];
  const int nn = (int)in[at++];
  voc.kmax = (int)in[at++]; voc.desc_bytes = (int)in[at++];
  const int nw = (int)in[at++];
  voc.node_desc.resize((size_t)nn * voc.desc_bytes);
  for (auto &b : voc.node_desc) b = (unsigned char)in[at++];
  voc.children.resize((size_t)nn * voc.kmax);
  for (auto &c : voc.children) c = (int)in[at++];
  voc.word_of_node.resize(nn);
  for (auto &c : voc.word_of_node) c = (int)in[at++];
  voc.node_of_word.resize(nw);
  for (auto &c : voc.node_of_word) c = (int)in[at++];
  const int n = 15 + 6 * N;
  Scenario sc[2];
  for (Scenario &s : sc) {
    s.q.assign(&in[at], &in[at] + 4 * N); at += 4 * N;
    s.p.assign(&in[at], &in[at] + 3 * N); at += 3 * N;
    s.P.assign(&in[at], &in[at] + (size_t)n * n); at += (size_t)n * n;
    s.off.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) s.off[k + 1] = s.off[k] + (int)in[at++];
    s.obs.assign(&in[at], &in[at] + 2 * (size_t)s.off[K]); at += 2 * (size_t)s.off[K];
    for (Descriptors *d : {&s.kf_desc, &s.query_desc}) {
      d->rows = (int)in[at++]; d->cols = voc.desc_bytes;
      d->data.resize((size_t)d->rows * d->cols);
      for (auto &b : d->data) b = (unsigned char)in[at++];
    }
  }
  try {
    unsigned char uid[XK_FLEET_ID_BYTES];
    if (world == 1) {
      // ---- loop-back: both agents in this process; a message = xk_fleet_send_recv to oneself, read by the other agent
      Agent a, b;
      xkok(nullptr, xk_fleet_unique_id(uid), "xk_fleet_unique_id");
      a.create(0, N, K, n_shared, sc[0], voc, thr, 1, 0, uid);
      xkok(nullptr, xk_fleet_unique_id(uid), "xk_fleet_unique_id");
      b.create(1, N, K, n_shared, sc[1], voc, thr, 1, 0, uid);
      a.storeKeyframe(100);
      b.storeKeyframe(200);
      a.prepareRequest();                                                       // A asks B
      xkok(a.h, xk_fleet_send_recv(a.f, a.d_vlad_out, a.vlad_n, 0, a.d_vlad_in, a.vlad_n, 0), "request");
      xkok(a.h, xk_fleet_wait(a.f), "wait");
      b.answer(a.id, a.d_vlad_in);                                              // ... B finds the request where it landed
      xkok(b.h, xk_fleet_send_recv(b.f, b.d_resp_out, 2 + b.pay_n + b.trk_n, 0, b.d_resp_in, 2 + b.pay_n + b.trk_n, 0), "response");
      xkok(b.h, xk_fleet_wait(b.f), "wait");
      std::vector<double> out;
      a.fuseAndUpdate(b.d_resp_in, sigma_img, ci_w, out);                       // ... and A the response
      writeOut(argv[2], out);
      printf("ok loop-back: keyframe %s (tag %ld), %d shared tracks fused, n=%d K=%d\n", out[0] != 0.0 ? "received" : "refused", (long)out[1],
             (int)out[2], n, K);
      a.destroy();
      b.destroy();
    } else {
      // ---- several ranks over RCCL (two real ones, or a dry run of one rank of a ring): each agent requests from its right-hand
      //      neighbour, answers its left-hand neighbour and fuses against the keyframe that comes back
      const bool dry = std::string(argv[3]) == "dry";
      const int fleet = dry ? atoi(argv[4]) : world, me_rank = dry ? atoi(argv[5]) : rank;
      if ((!dry && (world != 2 || argc < 6)) || (dry && (argc < 7 || fleet < 2 || me_rank < 0 || me_rank >= fleet)))
        throw std::runtime_error("multi-rank mode: case.bin out.bin 2 <rank> <idfile> | case.bin out.bin dry <N> <rank> <idfile>");
      const std::string idfile = dry ? argv[6] : argv[5];
      const int ask = (me_rank + 1) % fleet, asked_by = (me_rank + fleet - 1) % fleet;       // the ring
      if (me_rank == 0 || dry) {
        xkok(nullptr, xk_fleet_unique_id(uid), "xk_fleet_unique_id");
        FILE *f = fopen((idfile + ".tmp").c_str(), "wb");
        if (!f || fwrite(uid, 1, sizeof(uid), f) != sizeof(uid)) throw std::runtime_error("cannot write the unique id file");
        fclose(f);
        rename((idfile + ".tmp").c_str(), idfile.c_str());
      }
      if (me_rank != 0 || dry) {
        unsigned char got[XK_FLEET_ID_BYTES];
        FILE *f = nullptr;
        for (int i = 0; i < 6000 && !(f = fopen(idfile.c_str(), "rb")); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || fread(got, 1, sizeof(got), f) != sizeof(got)) throw std::runtime_error("no unique id from rank 0");
        fclose(f);
        if (dry && memcmp(got, uid, sizeof(uid)) != 0) throw std::runtime_error("unique id file does not read back");
        memcpy(uid, got, sizeof(uid));
      }
      int ndev = 0;
      HIPOK(hipGetDeviceCount(&ndev));
      if (!dry && me_rank >= ndev) throw std::runtime_error("fewer devices than ranks");
      HIPOK(hipSetDevice(dry ? me_rank % ndev : me_rank));
      Agent me, partner;                    // (partner: dry run only -- the neighbours' messages, produced in this process)
      const int comm_world = dry ? 1 : fleet, comm_rank = dry ? 0 : me_rank;
      me.create(me_rank, N, K, n_shared, sc[me_rank % 2], voc, thr, comm_world, comm_rank, uid);
      const long rn = 2 + me.pay_n + me.trk_n;
      me.storeKeyframe(100 * (me_rank + 1));
      me.prepareRequest();
      if (dry) {
        unsigned char uid2[XK_FLEET_ID_BYTES];
        xkok(nullptr, xk_fleet_unique_id(uid2), "xk_fleet_unique_id");
        partner.create(ask, N, K, n_shared, sc[ask % 2], voc, thr, 1, 0, uid2);
        if (partner.pay_n != me.pay_n || partner.trk_n != me.trk_n || partner.vlad_n != me.vlad_n) throw std::runtime_error("message sizes differ between ranks");
        partner.storeKeyframe(100 * (ask + 1));
        partner.prepareRequest();
        HIPOK(hipStreamSynchronize(partner.stream));
        // my request leaves through RCCL (to myself: the one real rank) ...; the left-hand neighbour's request lands in MY receive buffer
        xkok(me.h, xk_fleet_send_recv(me.f, me.d_vlad_out, me.vlad_n, 0, partner.d_vlad_in, me.vlad_n, 0), "request");
        xkok(me.h, xk_fleet_wait(me.f), "wait");
        HIPOK(hipMemcpy(me.d_vlad_in, partner.d_vlad_out, sizeof(double) * me.vlad_n, hipMemcpyDeviceToDevice));
        me.answer(asked_by, me.d_vlad_in);                                        // my responder half (answer discarded: the asker is virtual)
        partner.answer(me_rank, partner.d_vlad_in);                                // the right-hand neighbour answers MY request
        xkok(me.h, xk_fleet_send_recv(me.f, partner.d_resp_out, rn, 0, me.d_resp_in, rn, 0), "response");
        xkok(me.h, xk_fleet_wait(me.f), "wait");
      } else {
        xkok(me.h, xk_fleet_send_recv(me.f, me.d_vlad_out, me.vlad_n, ask, me.d_vlad_in, me.vlad_n, asked_by), "request");
        xkok(me.h, xk_fleet_wait(me.f), "wait");
        me.answer(asked_by, me.d_vlad_in);
        xkok(me.h, xk_fleet_send_recv(me.f, me.d_resp_out, rn, asked_by, me.d_resp_in, rn, ask), "response");
        xkok(me.h, xk_fleet_wait(me.f), "wait");
      }
      std::vector<double> out;
      me.fuseAndUpdate(me.d_resp_in, sigma_img, ci_w, out);
      writeOut(std::string(argv[2]) + "." + std::to_string(me_rank), out);
      printf("ok %srank %d of %d (asks %d, answers %d; device %d of %d; %ld-byte responses): keyframe %s (tag %ld), %d shared tracks fused\n",
             dry ? "dry-run " : "", me_rank, fleet, ask, asked_by, dry ? me_rank % ndev : me_rank, ndev, 8 * rn,
             out[0] != 0.0 ? "received" : "refused", (long)out[1], (int)out[2]);
      me.destroy();
      if (dry) partner.destroy();
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
