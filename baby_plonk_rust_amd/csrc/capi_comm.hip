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
//
// FAILURE SEMANTICS (round 6).  The reference panics where something is wrong (setup.rs:34, utils.rs:65,108); it never blocks.  A
// collective turns one rank's early return into every other rank's endless wait, so:
//   (1) a rank NEVER skips a collective it was asked to join.  A local failure in front of bp_msm_g1_allgather's ncclAllGather (unknown
//       handle, range error, out of memory, a launch error) writes a POISONED record -- header only, err = the code, err_rank = the
//       rank (MsmBlobHeader) -- and still enqueues the all-gather; every rank then returns that code.  bp_ntt_columns_allgather has
//       no record to carry a status, so a 32-byte AGREEMENT all-gather goes first: (local status, log_n, columns per rank) of every
//       rank; the columns travel only when all ranks report success and the same shape, otherwise every rank returns the same error.
//   (2) every buffer a collective needs is allocated by bp_comm_init_rank, not on the way to the collective.
//   (3) every host wait behind a collective is BOUNDED (bp_comm_set_timeout_ms, default 120 s): the wait polls the stream and
//       ncclCommGetAsyncError; on an asynchronous error or when the bound expires the communicator is aborted (ncclCommAbort), the
//       context loses it (bp_comm_info reports world 0) and the call returns BP_ERR_COMM.  ncclCommInitRank itself runs on a helper
//       thread under the same bound: when it expires (a rank never arrived) the call returns BP_ERR_COMM, the helper stays parked
//       inside RCCL until the process ends and the context refuses further communicators.
// What has run where: RCCL worlds of ONE rank on the build pool's single-GPU boxes (tests/test_gpu_dist.py: the happy path, an injected
// local failure that still advances the collective counter, a 1-ms bound and an init whose second rank never comes); more ranks only
// through the same calls over gloo-rehearsed control flow.  No SCALE run exists yet -- no number of a real world > 1 is claimed anywhere.
#include <rccl/rccl.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
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

// the agreement word of one rank (comm_agree): 32 bytes
struct AgreeWord {
  int32_t rc;
  uint32_t tag;
  uint64_t a, b, c;
};
static_assert(sizeof(AgreeWord) == 32, "agreement word");
enum : uint32_t { AGREE_INIT = 0x494e4954u, AGREE_COLUMNS = 0x434f4c53u };

// device layout of ctx->comm_dev (world = ranks of the communicator)
static inline uint8_t* dev_mine(bp_ctx* ctx) { return ctx->comm_dev; }
static inline uint8_t* dev_summed(bp_ctx* ctx) { return ctx->comm_dev + BP_MSM_BLOB_BYTES; }
static inline uint8_t* dev_gathered(bp_ctx* ctx) { return ctx->comm_dev + 2 * (size_t)BP_MSM_BLOB_BYTES; }
static inline uint8_t* dev_agree(bp_ctx* ctx) { return dev_gathered(ctx) + (size_t)ctx->comm_world * BP_MSM_BLOB_BYTES; }     // mine | all[world]
static inline size_t dev_bytes(size_t world) { return (2 + world) * (size_t)BP_MSM_BLOB_BYTES + (1 + world) * sizeof(AgreeWord); }
// pinned mirror: world records, then mine | all[world] agreement words
static inline uint8_t* host_agree(bp_ctx* ctx) { return (uint8_t*)ctx->comm_host + (size_t)ctx->comm_world * BP_MSM_BLOB_BYTES; }
static inline size_t host_bytes(size_t world) { return world * (size_t)BP_MSM_BLOB_BYTES + (1 + world) * sizeof(AgreeWord); }

namespace bp {
// Gives the communicator up: in-flight collectives are aborted (ncclCommAbort is the one RCCL call that is safe while peers are
// missing), the context keeps its buffers for a later bp_comm_init_rank.
static void comm_abort(bp_ctx* ctx) {
  if (ctx->comm) {
    (void)ncclCommAbort((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->comm_world = 0;
  ctx->comm_rank = 0;
}

void comm_release(bp_ctx* ctx) {           // also from bp_destroy
  if (ctx->comm) {
    ncclResult_t ar = ncclSuccess;
    const bool healthy = ncclCommGetAsyncError((ncclComm_t)ctx->comm, &ar) == ncclSuccess && ar == ncclSuccess;
    if (healthy) (void)ncclCommDestroy((ncclComm_t)ctx->comm);
    else (void)ncclCommAbort((ncclComm_t)ctx->comm);          // a destroy would wait for peers that may be gone
    ctx->comm = nullptr;
  }
  ctx->comm_world = 0;
  ctx->comm_rank = 0;
  for (auto& e : ctx->comm_ev)
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  if (ctx->comm_dev) {
    (void)hipFree(ctx->comm_dev);
    ctx->comm_dev = nullptr;
  }
  if (ctx->comm_host) {
    (void)hipHostFree(ctx->comm_host);
    ctx->comm_host = nullptr;
    ctx->comm_host_cap = 0;
  }
}
}  // namespace bp

// Wait for the context's stream behind a collective, under the communicator's bound.  The stream is polled (as stream_wait does), the
// communicator's asynchronous error is read every few polls, and when the bound expires the communicator is aborted: a peer that
// never joined must cost this rank an error, not its life.
static int comm_stream_wait(bp_ctx* ctx, const char* what, int line) {
  const auto t0 = std::chrono::steady_clock::now();
  const uint32_t bound_ms = ctx->comm_timeout_ms;
  auto next_health = t0 + std::chrono::microseconds(200);
  for (;;) {
    const hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess) return BP_OK;
    if (e != hipErrorNotReady) return fail(ctx, BP_ERR_HIP, what, e, __FILE__, line);
    const auto now = std::chrono::steady_clock::now();
    if (bound_ms && now - t0 > std::chrono::milliseconds(bound_ms)) {
      comm_abort(ctx);
      char buf[256];
      snprintf(buf, sizeof buf, "%s: the collective did not finish within %u ms (bp_comm_set_timeout_ms) -- a rank is missing or stuck; "
               "communicator aborted", what, bound_ms);
      return fail(ctx, BP_ERR_COMM, buf, hipSuccess, __FILE__, line);
    }
    if (ctx->comm && now >= next_health) {            // the communicator's own view, a few thousand times per second at most
      next_health = now + std::chrono::microseconds(200);
      ncclResult_t ar = ncclSuccess;
      const ncclResult_t q = ncclCommGetAsyncError((ncclComm_t)ctx->comm, &ar);
      if (q != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress)) {
        comm_abort(ctx);
        char buf[256];
        snprintf(buf, sizeof buf, "%s: the communicator reports an asynchronous error (%s); communicator aborted", what,
                 ncclGetErrorString(q != ncclSuccess ? q : ar));
        return fail(ctx, BP_ERR_COMM, buf, hipSuccess, __FILE__, line);
      }
    }
    if (now - t0 > std::chrono::milliseconds(6)) usleep(50);      // a long wait does not spin a core
    else __builtin_ia32_pause();
  }
}

// an RCCL call of a collective: its failure costs the communicator (peers can no longer be matched call for call)
#define BP_NCCL_OR_ABORT(ctx, call)                                          \
  do {                                                                       \
    ncclResult_t r__ = (call);                                               \
    if (r__ != ncclSuccess) {                                                \
      comm_abort(ctx);                                                       \
      return comm_fail(ctx, #call, r__, __LINE__);                           \
    }                                                                        \
  } while (0)

// Every rank contributes (rc, tag, a, b, c); every rank returns the same verdict: BP_OK when all ranks report rc == BP_OK and the
// same (tag, a, b, c), the lowest failing rank's code when some rank reports a failure, BP_ERR_INVALID_ARG when the ranks disagree.
// One 32-byte all-gather on the context's stream + two small copies, waited for under the bound.
static int comm_agree(bp_ctx* ctx, uint32_t tag, int local_rc, uint64_t a, uint64_t b, uint64_t c, const char* what) {
  const size_t world = (size_t)ctx->comm_world;
  AgreeWord* h = reinterpret_cast<AgreeWord*>(host_agree(ctx));
  h[0] = AgreeWord{local_rc, tag, a, b, c};
  AgreeWord* d = reinterpret_cast<AgreeWord*>(dev_agree(ctx));
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemcpyAsync(d, h, sizeof(AgreeWord), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) {              // even now the collective is not skipped: the device word may be stale, the peers at least are not left waiting
    (void)hipGetLastError();
  }
  ctx->comm_collectives++;
  BP_NCCL_OR_ABORT(ctx, ncclAllGather(d, d + 1, sizeof(AgreeWord), ncclUint8, (ncclComm_t)ctx->comm, st));
  BP_HIP(ctx, hipMemcpyAsync(h + 1, d + 1, world * sizeof(AgreeWord), hipMemcpyDeviceToHost, st));
  BP_TRY(comm_stream_wait(ctx, what, __LINE__));
  if (e != hipSuccess) return fail(ctx, BP_ERR_HIP, what, e, __FILE__, __LINE__);
  for (size_t r = 0; r < world; r++)
    if (h[1 + r].rc != BP_OK) {
      if ((int)r == ctx->comm_rank) return h[1 + r].rc;          // this rank's own failure: its own text is already in last_error
      char buf[200];
      snprintf(buf, sizeof buf, "%s: rank %zu failed before the collective (code %d); every rank returns it", what, r, (int)h[1 + r].rc);
      const int code = h[1 + r].rc < 0 && h[1 + r].rc >= BP_ERR_COMM ? h[1 + r].rc : BP_ERR_INVALID_ARG;
      return fail(ctx, code, buf, hipSuccess, __FILE__, __LINE__);
    }
  for (size_t r = 0; r < world; r++)
    if (h[1 + r].tag != tag || h[1 + r].a != a || h[1 + r].b != b || h[1 + r].c != c) {
      char buf[200];
      snprintf(buf, sizeof buf, "%s: rank %zu joined with other arguments (%llu, %llu against %llu, %llu here)", what, r,
               (unsigned long long)h[1 + r].a, (unsigned long long)h[1 + r].b, (unsigned long long)a, (unsigned long long)b);
      return fail(ctx, BP_ERR_INVALID_ARG, buf, hipSuccess, __FILE__, __LINE__);
    }
  return BP_OK;
}

int bp_comm_unique_id(uint8_t id[BP_COMM_ID_BYTES]) {
  if (!id) return BP_ERR_INVALID_ARG;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return BP_ERR_COMM;
  memcpy(id, u.internal, BP_COMM_ID_BYTES);
  return BP_OK;
}

int bp_comm_set_timeout_ms(bp_ctx* ctx, uint32_t ms) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  ctx->comm_timeout_ms = ms;
  return BP_OK;
}

int bp_comm_stats(bp_ctx* ctx, uint64_t* collectives, uint32_t* timeout_ms) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (collectives) *collectives = ctx->comm_collectives;
  if (timeout_ms) *timeout_ms = ctx->comm_timeout_ms;
  return BP_OK;
}

namespace {
// ncclCommInitRank blocks until every rank of the world has called it.  It runs on a helper thread so that the caller can stop
// waiting: the job is shared, so a helper that outlives its caller (the bound expired) touches nothing that has been freed.
struct InitJob {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  ncclResult_t result = ncclSuccess;
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 0;
  ncclUniqueId id;
};
}  // namespace

int bp_comm_init_rank(bp_ctx* ctx, const uint8_t id[BP_COMM_ID_BYTES], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return BP_ERR_INVALID_ARG;
  if (is_group(ctx)) return fail(ctx, BP_ERR_INVALID_ARG, "a bp_init_multi context combines its shards itself; communicators belong to plain (one GPU) contexts", hipSuccess, __FILE__, __LINE__);
  if (ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "this context already has a communicator (bp_comm_destroy first)", hipSuccess, __FILE__, __LINE__);
  if (ctx->comm_init_stuck) return fail(ctx, BP_ERR_COMM, "an earlier bp_comm_init_rank of this context ran into its bound and is still inside RCCL; use a fresh context (or process)", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  // everything a collective will need, BEFORE the communicator exists: a failure here is a plain local error, no peer is involved yet
  comm_release(ctx);
  for (auto& e : ctx->comm_ev) BP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDefault));
  BP_HIP(ctx, hipMalloc((void**)&ctx->comm_dev, dev_bytes((size_t)world)));
  BP_HIP(ctx, hipHostMalloc(&ctx->comm_host, host_bytes((size_t)world), hipHostMallocDefault));
  ctx->comm_host_cap = host_bytes((size_t)world);
  auto job = std::make_shared<InitJob>();
  job->device = ctx->device;
  job->rank = rank;
  job->world = world;
  memcpy(job->id.internal, id, BP_COMM_ID_BYTES);
  std::thread helper([job] {
    ncclComm_t c = nullptr;
    ncclResult_t r = ncclSystemError;
    if (hipSetDevice(job->device) == hipSuccess) r = ncclCommInitRank(&c, job->world, job->id, job->rank);
    std::lock_guard<std::mutex> lk(job->mu);
    job->comm = c;
    job->result = r;
    job->done = true;
    job->cv.notify_all();
  });
  {
    std::unique_lock<std::mutex> lk(job->mu);
    const uint32_t bound_ms = ctx->comm_timeout_ms;
    const bool in_time = bound_ms ? job->cv.wait_for(lk, std::chrono::milliseconds(bound_ms), [&] { return job->done; })
                                  : (job->cv.wait(lk, [&] { return job->done; }), true);
    if (!in_time) {
      lk.unlock();
      helper.detach();                       // parked inside RCCL's bootstrap until the process ends; it owns nothing of the context
      ctx->comm_init_stuck = true;
      comm_release(ctx);
      char buf[256];
      snprintf(buf, sizeof buf, "ncclCommInitRank(rank %d of %d) did not return within %u ms (bp_comm_set_timeout_ms): a rank never called "
               "bp_comm_init_rank with this id", rank, world, bound_ms);
      return fail(ctx, BP_ERR_COMM, buf, hipSuccess, __FILE__, __LINE__);
    }
  }
  helper.join();
  if (job->result != ncclSuccess) {
    comm_release(ctx);
    return comm_fail(ctx, "ncclCommInitRank", job->result, __LINE__);
  }
  ctx->comm = job->comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  // First collective of the communicator, still inside the init call where every rank is known to be present: RCCL connects its
  // channels on first use, and the ranks agree on what they are about to exchange (world, record size) -- a peer built from another
  // version of this library is an error here, not a misread record later.
  const int rc = comm_agree(ctx, AGREE_INIT, BP_OK, (uint64_t)world, (uint64_t)BP_MSM_BLOB_BYTES, sizeof(AgreeWord), "bp_comm_init_rank");
  if (rc != BP_OK) {
    comm_abort(ctx);
    comm_release(ctx);
    return rc;
  }
  return BP_OK;
}

int bp_comm_info(bp_ctx* ctx, int* rank, int* world) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
  if (world) *world = ctx->comm ? ctx->comm_world : 0;
  return BP_OK;
}

int bp_comm_destroy(bp_ctx* ctx) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  int rc = BP_OK;
  if (ctx->stream) rc = ctx->comm ? comm_stream_wait(ctx, "bp_comm_destroy", __LINE__) : (stream_wait(ctx->stream) == hipSuccess ? BP_OK : BP_ERR_HIP);
  comm_release(ctx);
  return rc;
}

// sum over ALL ranks of sum_i s_i P_{first + i} over the rank's own shard: record -> ONE ncclAllGather -> device pre-sum -> one D2H of
// one record -> host Horner + normalisation.  Everything up to the copy is enqueued on the context's stream without a host wait.
int bp_msm_g1_allgather(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                        int scalars_on_device, uint8_t out96[96]) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "no communicator on this context (bp_comm_init_rank)", hipSuccess, __FILE__, __LINE__);
  const size_t world = (size_t)ctx->comm_world;
  DeviceGuard guard(ctx->device);
  uint8_t *mine = dev_mine(ctx), *gathered = dev_gathered(ctx), *summed = dev_summed(ctx);
  hipStream_t st = ctx->stream;
  // This rank's record.  From here to the all-gather NOTHING returns: whatever goes wrong locally becomes a poisoned record.
  int local = (!out96 || !fmt_ok(scalar_fmt) || (n_scalars && !scalars)) ? BP_ERR_INVALID_ARG : BP_OK;
  if (local != BP_OK) (void)fail(ctx, local, "bp_msm_g1_allgather: null pointer or bad scalar format", hipSuccess, __FILE__, __LINE__);
  if (local == BP_OK) local = bp_msm_g1_blob_device_async(ctx, srs_handle, first, scalars, n_scalars, scalar_fmt, scalars_on_device, mine);
  std::string local_text;
  if (local != BP_OK) {
    local_text = ctx->last_error;
    if (msm_blob_poison_run(ctx, mine, local, (uint32_t)ctx->comm_rank) != BP_OK) (void)hipGetLastError();      // the all-gather is enqueued regardless
  }
  (void)hipEventRecord(ctx->comm_ev[0], st);
  ctx->comm_collectives++;
  BP_NCCL_OR_ABORT(ctx, ncclAllGather(mine, gathered, BP_MSM_BLOB_BYTES, ncclUint8, (ncclComm_t)ctx->comm, st));       // the path's single collective
  BP_TRY(msm_blobs_sum_device_run(ctx, gathered, world, summed, false));
  BP_HIP(ctx, hipMemcpyAsync(ctx->comm_host, summed, BP_MSM_BLOB_BYTES, hipMemcpyDeviceToHost, st));          // the path's single D2H
  BP_HIP(ctx, hipEventRecord(ctx->comm_ev[1], st));
  BP_TRY(comm_stream_wait(ctx, "bp_msm_g1_allgather", __LINE__));                                              // the only host wait
  float ms = 0;
  if (hipEventElapsedTime(&ms, ctx->comm_ev[0], ctx->comm_ev[1]) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
  ctx->comm_exchange_ms = ms;
  if (local != BP_OK) {                  // this rank's own failure, with its own text; the peers read it from the record
    ctx->last_error = local_text;
    return local;
  }
  uint32_t bad_rank = 0;
  int rc = msm_blob_poisoned((const uint8_t*)ctx->comm_host, 1, &bad_rank);
  if (rc != 0) {
    char buf[200];
    snprintf(buf, sizeof buf, "bp_msm_g1_allgather: rank %u failed before the collective (code %d); every rank returns it", bad_rank, rc);
    return fail(ctx, rc, buf, hipSuccess, __FILE__, __LINE__);
  }
  g1_proj r;
  uint32_t magic;
  memcpy(&magic, ctx->comm_host, 4);
  if (magic != 0) {
    rc = msm_blobs_combine((const uint8_t*)ctx->comm_host, 1, &r);
  } else {               // the ranks' window layouts differ (unequal shard lengths across a width threshold): every record to the host
    BP_HIP(ctx, hipMemcpyAsync(ctx->comm_host, gathered, world * BP_MSM_BLOB_BYTES, hipMemcpyDeviceToHost, st));
    BP_TRY(comm_stream_wait(ctx, "bp_msm_g1_allgather", __LINE__));
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
// overwritten.  An agreement all-gather of 32 bytes (status and shape of every rank), then ONE ncclAllGather of the columns on the
// context's stream, waited for under the bound.
int bp_ntt_columns_allgather(bp_ctx* ctx, void* d_columns, uint32_t log_n, size_t columns_per_rank) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, BP_ERR_INVALID_ARG, "no communicator on this context (bp_comm_init_rank)", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  int local = BP_OK;
  size_t block = 0, all = 0;
  if (!d_columns && columns_per_rank) local = fail(ctx, BP_ERR_INVALID_ARG, "bp_ntt_columns_allgather: null column buffer", hipSuccess, __FILE__, __LINE__);
  else if (log_n > 28) local = fail(ctx, BP_ERR_TOO_LARGE, "bp_ntt_columns_allgather: columns longer than 2^28 elements", hipSuccess, __FILE__, __LINE__);
  else if (__builtin_mul_overflow(columns_per_rank, ((size_t)1 << log_n) * sizeof(fr_t), &block) ||
           __builtin_mul_overflow(block, (size_t)ctx->comm_world, &all))
    local = fail(ctx, BP_ERR_TOO_LARGE, "bp_ntt_columns_allgather: columns_per_rank * world * 2^log_n * 32 overflows", hipSuccess, __FILE__, __LINE__);
  // the ranks' statuses and shapes first: a rank that cannot take part says so INSIDE a collective instead of staying away from one
  BP_TRY(comm_agree(ctx, AGREE_COLUMNS, local, log_n, columns_per_rank, 0, "bp_ntt_columns_allgather"));
  if (block == 0) return BP_OK;
  uint8_t* base = (uint8_t*)d_columns;
  ctx->comm_collectives++;
  BP_NCCL_OR_ABORT(ctx, ncclAllGather(base + (size_t)ctx->comm_rank * block, base, block, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));     // in place
  BP_TRY(comm_stream_wait(ctx, "bp_ntt_columns_allgather", __LINE__));
  return BP_OK;
}
