// xk_fleet.cpp -- include/xk_fleet.h on RCCL.  Host code only (no kernels): communicator set-up and the two exchange
// patterns of the CI step, queued on the update engine's stream.
#include "../../include/xk_fleet.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

struct xk_fleet {
  xk_handle *h;
  ncclComm_t comm;
  int world, rank, device;
  hipStream_t stream;
  char err[256];
};

static_assert(sizeof(ncclUniqueId) == XK_FLEET_ID_BYTES, "ncclUniqueId is 128 bytes");

static int ffail(xk_fleet *f, int code, const char *what, ncclResult_t r) {
  if (f) snprintf(f->err, sizeof(f->err), "%s: %s", what, ncclGetErrorString(r));
  return code;
}
#define NCHK(f, call)                                                   \
  do {                                                                  \
    ncclResult_t r_ = (call);                                           \
    if (r_ != ncclSuccess) return ffail((f), XK_EDEVICE, #call, r_);    \
  } while (0)

extern "C" int xk_fleet_unique_id(unsigned char id[XK_FLEET_ID_BYTES]) {
  if (!id) return XK_EINVAL;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return XK_EDEVICE;
  memcpy(id, &u, XK_FLEET_ID_BYTES);
  return XK_OK;
}

extern "C" int xk_fleet_create(xk_handle *h, const unsigned char id[XK_FLEET_ID_BYTES], int world, int rank, xk_fleet **out) {
  if (!h || !id || !out || world < 1 || rank < 0 || rank >= world) return XK_EINVAL;
  xk_fleet *f = (xk_fleet *)calloc(1, sizeof(xk_fleet));
  if (!f) return XK_ENOMEM;
  f->h = h; f->world = world; f->rank = rank;
  f->stream = (hipStream_t)xk_stream(h);
  if (hipStreamGetDevice(f->stream, &f->device) != hipSuccess || hipSetDevice(f->device) != hipSuccess) { free(f); return XK_EDEVICE; }
  ncclUniqueId u;
  memcpy(&u, id, XK_FLEET_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&f->comm, world, u, rank);
  if (r != ncclSuccess) { free(f); return XK_EDEVICE; }
  *out = f;
  return XK_OK;
}

extern "C" int xk_fleet_destroy(xk_fleet *f) {
  if (!f) return XK_OK;
  (void)hipSetDevice(f->device);
  (void)hipStreamSynchronize(f->stream);
  ncclCommDestroy(f->comm);
  free(f);
  return XK_OK;
}

extern "C" int xk_fleet_world(const xk_fleet *f) { return f ? f->world : 0; }
extern "C" int xk_fleet_rank(const xk_fleet *f) { return f ? f->rank : -1; }
extern "C" const char *xk_fleet_last_error(const xk_fleet *f) { return f ? f->err : "null fleet"; }

extern "C" int xk_fleet_all_gather(xk_fleet *f, const double *d_send, double *d_recv, long count) {
  if (!f || !d_send || !d_recv || count <= 0) return XK_EINVAL;
  if (hipSetDevice(f->device) != hipSuccess) return XK_EDEVICE;
  NCHK(f, ncclAllGather(d_send, d_recv, (size_t)count, ncclDouble, f->comm, f->stream));
  return XK_OK;
}

extern "C" int xk_fleet_send_recv(xk_fleet *f, const double *d_send, long send_count, int send_peer, double *d_recv,
                                  long recv_count, int recv_peer) {
  if (!f || send_peer >= f->world || recv_peer >= f->world) return XK_EINVAL;
  if ((send_peer >= 0 && (!d_send || send_count <= 0)) || (recv_peer >= 0 && (!d_recv || recv_count <= 0))) return XK_EINVAL;
  if (hipSetDevice(f->device) != hipSuccess) return XK_EDEVICE;
  NCHK(f, ncclGroupStart());
  // (a group that was opened is always closed, also when a call inside it fails: a communicator left inside a group would
  //  swallow every later call)
  ncclResult_t rs = ncclSuccess, rr = ncclSuccess;
  if (send_peer >= 0) rs = ncclSend(d_send, (size_t)send_count, ncclDouble, send_peer, f->comm, f->stream);
  if (rs == ncclSuccess && recv_peer >= 0) rr = ncclRecv(d_recv, (size_t)recv_count, ncclDouble, recv_peer, f->comm, f->stream);
  const ncclResult_t re = ncclGroupEnd();
  NCHK(f, rs);
  NCHK(f, rr);
  NCHK(f, re);
  return XK_OK;
}

extern "C" int xk_fleet_wait(xk_fleet *f) {
  if (!f) return XK_EINVAL;
  if (hipSetDevice(f->device) != hipSuccess || hipStreamSynchronize(f->stream) != hipSuccess) return XK_EDEVICE;
  return XK_OK;
}
