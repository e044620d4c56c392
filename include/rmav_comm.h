/* rmav_comm.h - multi-GPU: the path's one collective, replacing the MPI rank probe / data-parallel workers of
 * gym_reinmav/run.py:18-21,177-182.
 * Part of the C ABI of librmav.so; included by rmav.h (conventions, rmav_handle, status codes: there). */
#ifndef RMAV_COMM_H
#define RMAV_COMM_H

#include "rmav.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- multi-GPU: the path's one collective ------------------------------------------ */
/* Envs shard over ranks by contiguous ranges of GLOBAL env id: rank r of W owns base + (r < rem) envs
 * starting at r*base + min(r, rem), base = n_total / W, rem = n_total % W (create each rank's handle with
 * env_id_base = that start).  The data path needs no communication; the only exchange is the all-gather of
 * per-env episode statistics once per rollout.  The communicator wraps an RCCL communicator (librccl.so.1 is
 * loaded on first use; one process per GPU).  Rank 0 calls rmav_comm_unique_id and hands the 128 bytes to the
 * other ranks out of band (file, MPI, a torch store ...); every rank then calls rmav_comm_create. */
typedef struct rmav_comm_s *rmav_comm;
#define RMAV_COMM_ID_BYTES 128
/* Optional, before any other rmav_comm_* call of the process: resolve the five collective entry points (ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString) from THIS shared object instead of librccl.so.1 - a
 * site's own RCCL build, or the test suite's stand-in that lets two rank processes share ONE GPU (tests/stub_rccl). */
int rmav_comm_use_library(const char *path);
int rmav_comm_unique_id(void *id_out /* RMAV_COMM_ID_BYTES bytes, host */);
int rmav_comm_create(rmav_comm *out, const void *id, int rank, int world, int device);
int rmav_comm_destroy(rmav_comm c);
/* What the communicator is: rank / world as passed to rmav_comm_create, and what the collective library itself reports for its
 * communicator (ncclCommUserRank / ncclCommCount; -1 when the library does not export them).  Any pointer may be NULL.  Lets a
 * launcher assert that RCCL really connected `world` ranks (the role of the MPI rank probe of gym_reinmav/run.py:18-21,177-182). */
int rmav_comm_info(rmav_comm c, int *rank_out, int *world_out, int *lib_rank_out, int *lib_world_out);
/* One tiny all-gather on the communicator's own stream, awaited on the HOST for at most timeout_s seconds (< 0: no limit):
 * RMAV_OK, or RMAV_ERR_TIMEOUT.  RCCL connects its transports inside the FIRST collective's enqueue - a host-side exchange
 * with the peers that blocks when one of them is gone - so a caller that wants a bounded set-up runs rmav_comm_create +
 * rmav_comm_warmup on a helper thread and joins it with a deadline (gym_reinmav_amd.distributed.NativeStatsExchange does);
 * no handle and no handle's stream is involved. */
int rmav_comm_warmup(rmav_comm c, double timeout_s);
/* returns_out f32 [n_total], lengths_out i32 [n_total] (DEVICE pointers) <- return / length of every env's most
 * recently finished episode, in global env order, on every rank.  Enqueued on the handle's stream (pack ->
 * ncclAllGather over xGMI -> unpack); does not synchronise.  Needs RMAV_F_TRACK_EPISODES. */
int rmav_allgather_stats(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out);
/* The same exchange in two halves, so that it overlaps the next rollout: _post packs this rank's payload on the
 * handle's stream (a stream-ordered snapshot) and runs the ncclAllGather on the communicator's OWN (high-priority)
 * stream behind a signal word; up to eight exchanges may be in flight, and a ninth post blocks the HOST until the
 * oldest one has finished (back pressure - nothing is ever inserted into the handle's stream).  _result makes the
 * handle's stream wait for the most recently posted gather and unpacks it.  rmav_allgather_stats = _post followed
 * by _result.  ~15 us of host time per post (two small launches, one event record, one RCCL enqueue). */
int rmav_allgather_stats_post(rmav_handle h, rmav_comm c, int64_t n_total);
/* Optional, BEFORE the rollout whose statistics the next _post will exchange: the next call of `h` that is ONE fused launch over
 * all of its envs (rmav_rollout fused, rmav_rollout_policy) then writes the snapshot itself and publishes per-wavefront arrival
 * words the communicator's stream polls, so that _post puts nothing into the handle's stream.  Same snapshot, same result; if no
 * such launch happens before _post (single-step launches, a sliced launch) or another stepping launch follows it, _post packs as
 * usual.  The communicator stream's wait is bounded: 2 s from the moment the armed launch begins on the device (10 min overall);
 * past that this rank's payload is poisoned (return NaN, length -1), the collective is issued all the same - no peer hangs - and
 * _wait / _result return RMAV_ERR_TIMEOUT for THAT post only.  One armed exchange per handle and communicator at a time. */
int rmav_allgather_stats_arm(rmav_handle h, rmav_comm c, int64_t n_total);
int rmav_allgather_stats_result(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out);
/* HOST-side bounded wait for the most recently posted exchange (polls its completion event; touches no stream):
 * RMAV_OK once the gather has finished, RMAV_ERR_TIMEOUT after timeout_s seconds (< 0: no limit).  Lets a caller probe a
 * freshly created communicator - post, wait with a deadline, only then _result - without ever parking a handle's stream
 * behind a collective that may never complete. */
int rmav_allgather_stats_wait(rmav_comm c, double timeout_s);
/* The send side of that exchange alone, for callers that own the collective (torch.distributed over RCCL):
 * send_out i32 [2][cmax] (DEVICE) <- bit patterns of the per-env last returns, then the last lengths, zero padded
 * from num_envs to cmax (the largest shard).  A stream-ordered snapshot in one small launch, so the next
 * rollout may overwrite the per-env arrays while the collective is still in flight. */
int rmav_pack_stats(rmav_handle h, int64_t cmax, int32_t *send_out);

#ifdef __cplusplus
}
#endif
#endif
