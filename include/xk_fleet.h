/* xk_fleet.h -- the inter-agent exchange of the CI step on RCCL, as a C ABI next to xk.h.
 *
 * The reference's agents exchange SimpleState snapshots (+ the shared tracks' observations) through the host
 * application's transport (ROS topics in the reference's wrapper; VIO::processOtherMeasurements, vio.cpp:498-570, is the
 * receiving end).  One agent per GPU on an 8 x MI355X node, the same messages travel over RCCL / xGMI between DEVICE
 * buffers: xk_pack_payload writes the snapshot straight into the send buffer, xk_ci_round_device reads the gathered
 * snapshots where RCCL left them.  Broadcast mode (BASELINE config 4): one ncclAllGather.  Request/response mode
 * (config 5, vio.cpp:455-496): one grouped ncclSend / ncclRecv pair per agent.  Everything is queued on the handle's
 * stream (xk_stream), so an exchange orders itself behind the update that produced the snapshot.
 *
 * Lives in libxk_fleet.so (links libxk.so and librccl.so) so that libxk.so itself has no communication dependency.
 * The 128-byte unique id is produced by ONE rank and has to reach the others through whatever the host already uses
 * (MPI, a socket, a file): that bootstrap is the application's, as it is with NCCL. */
#ifndef XK_FLEET_H_
#define XK_FLEET_H_
#include "xk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct xk_fleet xk_fleet;
#define XK_FLEET_ID_BYTES 128

/* ncclGetUniqueId: call on one rank, distribute the 128 bytes to all. */
int xk_fleet_unique_id(unsigned char id[XK_FLEET_ID_BYTES]);
/* ncclCommInitRank on the handle's device; collective over all `world` ranks. */
int xk_fleet_create(xk_handle *h, const unsigned char id[XK_FLEET_ID_BYTES], int world, int rank, xk_fleet **out);
int xk_fleet_destroy(xk_fleet *f);
int xk_fleet_world(const xk_fleet *f);
int xk_fleet_rank(const xk_fleet *f);
const char *xk_fleet_last_error(const xk_fleet *f);

/* Broadcast mode: every agent's `count` doubles at d_send -> d_recv[world][count] on every agent (ncclAllGather). */
int xk_fleet_all_gather(xk_fleet *f, const double *d_send, double *d_recv, long count);
/* Request/response mode: send `send_count` doubles to send_peer and receive `recv_count` from recv_peer in one group
 * (a peer < 0 skips that half).  Used for the VLAD request (as doubles or bytes reinterpreted) and the keyframe reply. */
int xk_fleet_send_recv(xk_fleet *f, const double *d_send, long send_count, int send_peer, double *d_recv, long recv_count,
                       int recv_peer);
/* Waits for the exchanges queued so far (hipStreamSynchronize on the handle's stream). */
int xk_fleet_wait(xk_fleet *f);

#ifdef __cplusplus
}
#endif
#endif /* XK_FLEET_H_ */
