// capi_ntt.hip -- C ABI, part 3: ntt_381 / i_ntt_381 (utils.rs:63-129), host and device forms, columns and one large transform over
// the members of a group context, roots of unity, scalar format conversion.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"

#include "capi_common.hpp"

using namespace bp;
// ---------------------------------------------------------------------------------------------- DFT
int bp_ntt_fr_device(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride) {
  if (!ctx || (!d_data && batch)) return BP_ERR_INVALID_ARG;
  if (batch == 0) {                   // nothing was enqueued: no events to read
    ctx->ntt_ms = 0;
    return BP_OK;
  }
  DeviceGuard guard(ctx->device);
  BP_TRY(ntt_run(ctx, (fr_t*)d_data, log_n, inverse, batch, stride));
  BP_HIP(ctx, stream_wait(ctx->stream));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->ntt_ms, ctx->ev[0], ctx->ev[1]));
  ctx->ntt_async_pending = false;
  return BP_OK;
}

// Enqueue only: the transform runs on the context's stream behind whatever was enqueued before; bp_synchronize (or any blocking
// entry point on this context) waits for it.  Back-to-back transforms on HBM-resident data then cost their kernels, not a host
// round trip each (16 us of a 0.17 ms call at 2^20).
int bp_ntt_fr_device_async(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride) {
  if (!ctx || (!d_data && batch)) return BP_ERR_INVALID_ARG;
  if (batch == 0) return BP_OK;
  DeviceGuard guard(ctx->device);
  BP_TRY(ntt_run(ctx, (fr_t*)d_data, log_n, inverse, batch, stride));
  ctx->ntt_async_pending = true;
  ctx->ntt_members = 1;
  return BP_OK;
}

// Group context, batch > 1: the columns are independent transforms (NTT option i), column j goes to member
// j mod R; every member uploads, transforms and downloads its columns on its own stream, nothing is exchanged.
static int ntt_columns_over_members(bp_ctx* ctx, uint8_t* data, uint32_t log_n, int inverse, int scalar_fmt, size_t batch, size_t stride) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const size_t N = (size_t)1 << log_n, R = sh.size();
  std::vector<fr_t*> dbuf(R, nullptr);
  std::vector<size_t> cnt(R, 0);
  // one host thread per member: the column copies from and to pageable memory are staged by the thread that issues them
  std::vector<int> rcs(R, BP_OK);
  for (size_t r = 0; r < R; r++) cnt[r] = batch / R + (r < batch % R ? 1 : 0);
  over_members(ctx, R, [&](size_t r) { return cnt[r] != 0; }, [&](size_t r) {
    bp_ctx* m = sh[r];
    DeviceGuard guard(m->device);
    int rc = ws_get(m, "io.ntt", cnt[r] * N * sizeof(fr_t), (void**)&dbuf[r]);
    for (size_t j = 0; j < cnt[r] && rc == BP_OK; j++) {
      hipError_t e = hipMemcpyAsync(dbuf[r] + j * N, data + (r + j * R) * stride * sizeof(fr_t), N * sizeof(fr_t), hipMemcpyHostToDevice, m->stream);
      if (e != hipSuccess) rc = fail(m, BP_ERR_HIP, "NTT column upload", e, __FILE__, __LINE__);
    }
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(m, dbuf[r], cnt[r] * N, 0);
    if (rc == BP_OK) rc = ntt_run(m, dbuf[r], log_n, inverse, cnt[r], N);
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(m, dbuf[r], cnt[r] * N, 1);
    for (size_t j = 0; j < cnt[r] && rc == BP_OK; j++) {
      hipError_t e = hipMemcpyAsync(data + (r + j * R) * stride * sizeof(fr_t), dbuf[r] + j * N, N * sizeof(fr_t), hipMemcpyDeviceToHost, m->stream);
      if (e != hipSuccess) rc = fail(m, BP_ERR_HIP, "NTT column download", e, __FILE__, __LINE__);
    }
    rcs[r] = rc;
  });
  int rc = BP_OK;
  for (size_t r = 0; r < R; r++)
    if (rc == BP_OK) rc = lift(ctx, sh[r], rcs[r]);
  float ms = 0;
  for (size_t r = 0; r < R; r++) {                       // wait for every member, also after a failure elsewhere
    if (cnt[r] == 0) continue;
    DeviceGuard guard(sh[r]->device);
    hipError_t e = stream_wait(sh[r]->stream);
    if (e != hipSuccess && rc == BP_OK) rc = fail(ctx, BP_ERR_HIP, "NTT columns", e, __FILE__, __LINE__);
    float t = 0;
    if (rc == BP_OK && hipEventElapsedTime(&t, sh[r]->ev[0], sh[r]->ev[1]) == hipSuccess) ms = std::max(ms, t);
  }
  ctx->ntt_ms = ms;
  ctx->ntt_passes = sh[0]->ntt_passes;
  ctx->ntt_members = (uint32_t)std::min(R, batch);
  return rc;
}

// Group context, ONE large transform from host memory (NTT option ii): N = 2^(l_1 + s).  Member g uploads the
// columns r in its slice of [0, 2^s) -- every member over its own PCIe link -- and runs pass 1 on them; the members then swap
// blocks of the intermediate buffer (member g' collects the rows e_1 of its slice: R - 1 peer copies of N / R^2 elements each,
// over xGMI), run the remaining passes on their e_1 and download their outputs (index = e_1 mod 2^(l_1): runs of 2^(l_1) / R
// elements).  Buffers keep the full N-element layout on every member, so the kernels address exactly as on one GPU.
// Returns 1 (not an error code of the ABI) when the shape does not split; only then does the caller run the transform on the
// leader.  Any other failure is returned as it is: the download phase writes `data` from every member at once, so after a
// failed copy the buffer may be part input, part output, and must not be transformed again.
static int ntt_one_over_members(bp_ctx* ctx, uint8_t* data, uint32_t log_n, int inverse, int scalar_fmt) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const uint32_t R = (uint32_t)sh.size();
  if (!ntt_split_ok(log_n, R)) return 1;
  uint32_t l1 = 0;
  ntt_split_shape(log_n, &l1);
  const size_t N = (size_t)1 << log_n, S = (size_t)1 << (log_n - l1), L1 = (size_t)1 << l1;     // rows of pass 1 x row length
  const size_t cols = S / R, rows = L1 / R, esz = sizeof(fr_t);
  std::vector<fr_t*> dbuf(R, nullptr), tbuf(R, nullptr);
  std::vector<int> rcs(R, BP_OK);
  for (uint32_t g = 0; g < R; g++) {
    DeviceGuard guard(sh[g]->device);
    BP_TRY(lift(ctx, sh[g], ws_get(sh[g], "io.ntt", N * esz, (void**)&dbuf[g])));
    BP_TRY(lift(ctx, sh[g], ntt_tmp_buffer(sh[g], log_n, &tbuf[g])));
  }
  // copies from and to pageable host memory are staged by the calling thread: one host thread per member keeps every PCIe link busy
  auto on_members = [&](const std::function<int(uint32_t)>& work) {
    over_members(ctx, R, [](size_t) { return true; }, [&](size_t g) { rcs[g] = work((uint32_t)g); });
    for (uint32_t g = 0; g < R; g++)
      if (rcs[g] != BP_OK) return lift(ctx, sh[g], rcs[g]);
    return (int)BP_OK;
  };
  int rc = on_members([&](uint32_t g) -> int {                 // column slice up, pass 1
    bp_ctx* m = sh[g];
    DeviceGuard guard(m->device);
    hipError_t e = hipMemcpy2DAsync(dbuf[g] + g * cols, S * esz, data + g * cols * esz, S * esz, cols * esz, L1, hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) return fail(m, BP_ERR_HIP, "NTT column-slice upload", e, __FILE__, __LINE__);
    if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(m, dbuf[g], N, 0));      // other columns: unused memory, converted and ignored
    BP_TRY(ntt_run_part(m, dbuf[g], log_n, inverse, 1, N, 0, g, R));
    BP_HIP(m, hipEventRecord(m->ev[4], m->stream));               // pass 1 of this member is behind this event: the exchange waits for it
    BP_HIP(m, stream_wait(m->stream));
    return BP_OK;
  });
  if (rc != BP_OK) return rc;
  for (uint32_t to = 0; to < R && rc == BP_OK; to++) {         // member `to` collects rows [to * rows, (to + 1) * rows) of everybody's columns,
    bp_ctx* m = sh[to];                                        // then runs the remaining passes on its e_1
    DeviceGuard guard(m->device);
    for (uint32_t from = 0; from < R && rc == BP_OK; from++) {
      if (from == to) continue;
      const size_t off = (size_t)to * rows * S + (size_t)from * cols;
      // stream order, not the host wait above, is what ties the copy to `from`'s pass 1 (the event was recorded on from's stream)
      hipError_t e = hipStreamWaitEvent(m->stream, sh[from]->ev[4], 0);
      if (e == hipSuccess) e = hipMemcpy2DAsync(tbuf[to] + off, S * esz, tbuf[from] + off, S * esz, cols * esz, rows, hipMemcpyDefault, m->stream);   // the runtime finds the two devices
      if (e != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "NTT block exchange", e, __FILE__, __LINE__);
    }
    if (rc == BP_OK) rc = lift(ctx, m, ntt_run_part(m, dbuf[to], log_n, inverse, 1, N, 1, to, R));
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = lift(ctx, m, fr_convert_run(m, dbuf[to], N, 1));
  }
  for (uint32_t g = 0; g < R; g++) {                           // every member, also after a failure elsewhere
    DeviceGuard guard(sh[g]->device);
    hipError_t e = stream_wait(sh[g]->stream);
    if (e != hipSuccess && rc == BP_OK) rc = fail(ctx, BP_ERR_HIP, "NTT over the members", e, __FILE__, __LINE__);
  }
  if (rc != BP_OK) return rc;
  rc = on_members([&](uint32_t g) -> int {                     // outputs e_1 + 2^(l_1) m, e_1 in the member's slice
    bp_ctx* m = sh[g];
    DeviceGuard guard(m->device);
    hipError_t e = hipMemcpy2DAsync(data + g * rows * esz, L1 * esz, dbuf[g] + g * rows, L1 * esz, rows * esz, S, hipMemcpyDeviceToHost, m->stream);
    if (e != hipSuccess) return fail(m, BP_ERR_HIP, "NTT output download", e, __FILE__, __LINE__);
    BP_HIP(m, stream_wait(m->stream));
    return BP_OK;
  });
  if (rc != BP_OK) return rc;
  float ms0 = 0, ms1 = 0;
  for (uint32_t g = 0; g < R; g++) {
    DeviceGuard guard(sh[g]->device);
    float a = 0, b2 = 0;
    if (hipEventElapsedTime(&a, sh[g]->ev[0], sh[g]->ev[1]) == hipSuccess) ms0 = std::max(ms0, a);
    if (hipEventElapsedTime(&b2, sh[g]->ev[2], sh[g]->ev[3]) == hipSuccess) ms1 = std::max(ms1, b2);
  }
  ctx->ntt_ms = ms0 + ms1;                                      // kernels only (slowest member of each phase); the exchange is not in it
  ctx->ntt_passes = sh[0]->ntt_passes;
  ctx->ntt_members = R;
  return BP_OK;
}

int bp_ntt_fr(bp_ctx* ctx, void* data, uint32_t log_n, int inverse, int scalar_fmt, size_t batch, size_t stride) {
  if (!ctx || !fmt_ok(scalar_fmt) || (!data && batch)) return BP_ERR_INVALID_ARG;
  if (log_n > 28) return fail(ctx, BP_ERR_TOO_LARGE, "NTT length > 2^28", hipSuccess, __FILE__, __LINE__);
  if (batch == 0) return BP_OK;
  const size_t N = (size_t)1 << log_n;
  if (batch > 1 && stride < N) return fail(ctx, BP_ERR_INVALID_ARG, "NTT stride < N", hipSuccess, __FILE__, __LINE__);
  if (batch > 65535) return fail(ctx, BP_ERR_TOO_LARGE, "NTT batch > 65535", hipSuccess, __FILE__, __LINE__);
  if (is_group(ctx) && batch > 1) return ntt_columns_over_members(ctx, (uint8_t*)data, log_n, inverse, scalar_fmt, batch, stride);
  if (is_group(ctx) && log_n >= knob_u32("BP_NTT_GROUP_SPLIT_FROM", 22, 11, 29)) {     // one large transform: every member's PCIe link and a share of the work
    const int rc = ntt_one_over_members(ctx, (uint8_t*)data, log_n, inverse, scalar_fmt);
    if (rc != 1) return rc;                 // done, or a real failure (reported, never papered over: host data may be partly written);
  }                                         // 1 = the shape does not split over this many members: on the leader
  DeviceGuard guard(ctx->device);
  ctx->ntt_members = 1;
  const size_t span = (batch - 1) * stride + N;
  fr_t* d;
  BP_TRY(upload_fr(ctx, "io.ntt", data, span, span, scalar_fmt, &d));
  BP_TRY(ntt_run(ctx, d, log_n, inverse, batch, stride));
  BP_TRY(download_fr(ctx, d, data, span, scalar_fmt));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->ntt_ms, ctx->ev[0], ctx->ev[1]));
  return BP_OK;
}

int bp_ntt_last_stats(bp_ctx* ctx, float* device_ms, uint32_t* passes) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (ctx->ntt_async_pending) {         // the last transform was only enqueued: its events are read now (0 while it is still running)
    DeviceGuard guard(ctx->device);
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) {
      ctx->ntt_ms = ms;
      ctx->ntt_async_pending = false;
    } else {
      (void)hipGetLastError();
      ctx->ntt_ms = 0;
    }
  }
  if (device_ms) *device_ms = ctx->ntt_ms;
  if (passes) *passes = ctx->ntt_passes;
  return BP_OK;
}
int bp_ntt_last_members(bp_ctx* ctx) { return ctx ? (int)ctx->ntt_members : BP_ERR_INVALID_ARG; }

// utils.rs:39-43: ROOT_OF_UNITY.pow([2^32 / group_order, 0, 0, 0]) -- integer division, as written
bool host_root_of_unity(fr_t& out, uint64_t group_order) {
  if (group_order == 0) return false;                       // division by zero panics in the reference
  const uint64_t e = ((uint64_t)1 << 32) / group_order;
  uint32_t e32[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
  Fr::pow(out, fr_root_of_unity(false), e32, 2);
  return true;
}
int bp_root_of_unity(uint64_t group_order, int scalar_fmt, uint8_t out32[32]) {
  if (!out32 || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return BP_ERR_INVALID_ARG;
  fr_mont_to_bytes(out32, w, scalar_fmt);
  return BP_OK;
}
int bp_roots_of_unity(bp_ctx* ctx, uint64_t group_order, int scalar_fmt, void* out) {
  if (!ctx || !out || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return fail(ctx, BP_ERR_INVALID_ARG, "group_order == 0", hipSuccess, __FILE__, __LINE__);
  if (group_order > ((uint64_t)1 << 28)) return fail(ctx, BP_ERR_TOO_LARGE, "group_order > 2^28", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.roots", group_order * sizeof(fr_t), (void**)&d));
  BP_TRY(roots_run(ctx, w, group_order, d));
  return download_fr(ctx, d, out, group_order, scalar_fmt);
}

int bp_fr_convert(const void* in, size_t n, int from_fmt, int to_fmt, void* out) {
  if (!fmt_ok(from_fmt) || !fmt_ok(to_fmt) || (n && (!in || !out))) return BP_ERR_INVALID_ARG;
  for (size_t i = 0; i < n; i++) {
    fr_t v;
    if (!fr_bytes_to_mont(v, (const uint8_t*)in + 32 * i, from_fmt)) return BP_ERR_BAD_SCALAR;
    fr_mont_to_bytes((uint8_t*)out + 32 * i, v, to_fmt);
  }
  return BP_OK;
}

int bp_fr_synthetic_device(bp_ctx* ctx, void* d_out, size_t n, uint64_t seed) {
  if (!ctx || (n && !d_out)) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_synthetic_run(ctx, (fr_t*)d_out, n, seed));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
