// capi_comm.hip -- C ABI, part 6: the one-process-per-GPU exchange UNDER the boundary ("bp_ctx owns ... the RCCL comm";
// 8e: MSM by point range + one all-gather of the partial sums, NTT by independent columns + one all-gather of finished columns).
// A Rust host that runs one process per GPU binds these few calls and needs no RCCL binding of its own: the library is linked
// against librccl and enqueues the collective on the context's stream, between its own kernels.
//   rank 0:     bp_comm_unique_id(id)            -> the host carries the 128 bytes to the other ranks (file, socket, MPI: its business)
//   every rank: bp_comm_init_rank(ctx, id, rank, world)
//               bp_msm_g1_allgather(ctx, shard, ...)     = sum over ALL ranks' (point, scalar) pairs, the same 96 bytes on every rank
//               bp_ntt_columns_allgather(ctx, ...)       = every rank ends with every finished column
//               bp_comm_destroy(ctx)
// Elliptic-curve addition is not an RCCL reduction operator, so the MSM's "reduce" is ONE ncclAllGather of BP_MSM_BLOB_BYTES per rank
// followed by the device-side slot-wise sum of the gathered records (msm_blob_sum) and one 22-KB device-to-host copy.
// What has run where: RCCL worlds of ONE rank on the build pool's single-GPU boxes (tests/test_gpu_dist.py); more ranks only through the
// same calls over gloo-rehearsed control flow.  No SCALE run exists yet -- the numbers of a real world > 1 are not claimed anywhere.
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ctx.hpp"

#include "capi_common.hpp"

using namespace bp;

static_assert(BP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id the host carries between ranks is RCCL's ncclUniqueId");

static int comm_fail(bp_ctx* ctx, const char* what, ncclResult_t r, int line) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, ncclGetErrorString(r));
  return fail(ctx, BP_ERR_COMM, buf, hipSuccess, __FILE__, line);
}
#define BP_NCCL(ctx, call)                                                   \
  do {                                                                       \
    ncclResult_t r__ = (call);                                               \
    if (r__ != ncclSuccess) return comm_fail(ctx, #call, r__, __LINE__);     \
  } while (0)

int bp_comm_unique_id(uint8_t id[BP_COMM_ID_BYTES]) {
  if (!id) return BP_ERR_INVALID_ARG;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return BP_ERR_COMM;
  memcpy(id, u.internal, BP_COMM_ID_BYTES);
  return BP_OK;
}

int bp_comm_init_rank(bp_ctx* ctx, const uint8_t id[BP_COMM_ID_BYTES], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return BP_ERR_INVALID_ARG;
  if (is_group(ctx)) return fail(ctx, BP_ERR_INVALID_ARG, "a bp_init_multi context combines its shards itself; communicators belong to plain (one GPU) contexts", hipSuccess, __FILE__, __LINE__);
  if (ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "this context already has a communicator (bp_comm_destroy first)", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  ncclUniqueId u;
  memcpy(u.internal, id, BP_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  BP_NCCL(ctx, ncclCommInitRank(&c, world, u, rank));
  ctx->comm = c;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  for (auto& e : ctx->comm_ev)
    if (!e) BP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDefault));
  return BP_OK;
}

int bp_comm_info(bp_ctx* ctx, int* rank, int* world) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
  if (world) *world = ctx->comm ? ctx->comm_world : 0;
  return BP_OK;
}

namespace bp {
void comm_release(bp_ctx* ctx) {           // also from bp_destroy
  if (ctx->comm) {
    (void)ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->comm_world = 0;
  for (auto& e : ctx->comm_ev)
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
}
}  // namespace bp

int bp_comm_destroy(bp_ctx* ctx) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  if (ctx->stream) BP_HIP(ctx, stream_wait(ctx->stream));
  comm_release(ctx);
  return BP_OK;
}

// sum over ALL ranks of sum_i s_i P_{first + i} over the rank's own shard: record -> ONE ncclAllGather -> device pre-sum -> one D2H of
// one record -> host Horner + normalisation.  Everything up to the copy is enqueued on the context's stream without a host wait.
int bp_msm_g1_allgather(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                        int scalars_on_device, uint8_t out96[96]) {
  if (!ctx || !out96 || !fmt_ok(scalar_fmt) || (n_scalars && !scalars)) return BP_ERR_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "no communicator on this context (bp_comm_init_rank)", hipSuccess, __FILE__, __LINE__);
  const size_t world = (size_t)ctx->comm_world;
  DeviceGuard guard(ctx->device);
  uint8_t *mine, *gathered, *summed;
  BP_TRY(ws_get(ctx, "comm.mine", BP_MSM_BLOB_BYTES, (void**)&mine));
  BP_TRY(ws_get(ctx, "comm.gathered", world * BP_MSM_BLOB_BYTES, (void**)&gathered));
  BP_TRY(ws_get(ctx, "comm.summed", BP_MSM_BLOB_BYTES, (void**)&summed));
  if (ctx->comm_host_cap < world * BP_MSM_BLOB_BYTES) {          // pinned landing area of the path's single device-to-host copy
    if (ctx->comm_host) BP_HIP(ctx, hipHostFree(ctx->comm_host));
    ctx->comm_host = nullptr;
    ctx->comm_host_cap = 0;
    BP_HIP(ctx, hipHostMalloc(&ctx->comm_host, world * BP_MSM_BLOB_BYTES, hipHostMallocDefault));
    ctx->comm_host_cap = world * BP_MSM_BLOB_BYTES;
  }
  BP_TRY(bp_msm_g1_blob_device_async(ctx, srs_handle, first, scalars, n_scalars, scalar_fmt, scalars_on_device, mine));
  hipStream_t st = ctx->stream;
  BP_HIP(ctx, hipEventRecord(ctx->comm_ev[0], st));
  BP_NCCL(ctx, ncclAllGather(mine, gathered, BP_MSM_BLOB_BYTES, ncclUint8, (ncclComm_t)ctx->comm, st));       // the path's single collective
  BP_TRY(msm_blobs_sum_device_run(ctx, gathered, world, summed, false));
  BP_HIP(ctx, hipMemcpyAsync(ctx->comm_host, summed, BP_MSM_BLOB_BYTES, hipMemcpyDeviceToHost, st));          // the path's single D2H
  BP_HIP(ctx, hipEventRecord(ctx->comm_ev[1], st));
  BP_HIP(ctx, stream_wait(st));                                                                                 // the only host wait
  float ms = 0;
  if (hipEventElapsedTime(&ms, ctx->comm_ev[0], ctx->comm_ev[1]) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
  ctx->comm_exchange_ms = ms;
  g1_proj r;
  uint32_t magic;
  memcpy(&magic, ctx->comm_host, 4);
  int rc;
  if (magic != 0) {
    rc = msm_blobs_combine((const uint8_t*)ctx->comm_host, 1, &r);
  } else {               // the ranks' window layouts differ (unequal shard lengths across a width threshold): every record to the host
    BP_HIP(ctx, hipMemcpyAsync(ctx->comm_host, gathered, world * BP_MSM_BLOB_BYTES, hipMemcpyDeviceToHost, st));
    BP_HIP(ctx, stream_wait(st));
    rc = msm_blobs_combine((const uint8_t*)ctx->comm_host, world, &r);
  }
  if (rc == BP_ERR_BAD_SCALAR) return fail(ctx, rc, "scalar >= q in a canonical-bytes input (on some rank)", hipSuccess, __FILE__, __LINE__);
  if (rc != BP_OK) return fail(ctx, rc, "gathered MSM records", hipSuccess, __FILE__, __LINE__);
  host_encode96(out96, r);
  return BP_OK;
}

int bp_comm_last_exchange_ms(bp_ctx* ctx, float* ms) {
  if (!ctx || !ms) return BP_ERR_INVALID_ARG;
  *ms = ctx->comm_exchange_ms;
  return BP_OK;
}

// Finished NTT columns of all ranks, in place: d_columns holds world x columns_per_rank columns of 2^log_n Montgomery elements, rank r's
// block at column r * columns_per_rank; this rank's block is filled (its transforms ran on this context's stream), the others are
// overwritten.  ONE ncclAllGather on the context's stream, waited for.
int bp_ntt_columns_allgather(bp_ctx* ctx, void* d_columns, uint32_t log_n, size_t columns_per_rank) {
  if (!ctx || !d_columns || log_n > 32) return BP_ERR_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "no communicator on this context (bp_comm_init_rank)", hipSuccess, __FILE__, __LINE__);
  const size_t block = columns_per_rank * ((size_t)1 << log_n) * sizeof(fr_t);
  if (block == 0) return BP_OK;
  DeviceGuard guard(ctx->device);
  uint8_t* base = (uint8_t*)d_columns;
  BP_NCCL(ctx, ncclAllGather(base + (size_t)ctx->comm_rank * block, base, block, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));     // in place
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
