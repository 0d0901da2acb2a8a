// capi_prove.hip -- C ABI, part 5: the native prover (circuit preprocessing, bp_prove, transcript test vector).
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
// ---------------------------------------------------------------------------------------------- prover
int bp_circuit_load(bp_ctx* ctx, uint32_t log_n, const void* const columns[8], int scalar_fmt, int columns_on_device, uint64_t* handle) {
  if (!ctx || !columns || !handle || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  if (log_n < 3 || log_n > 24) return fail(ctx, BP_ERR_INVALID_ARG, "circuit: log_n must be in 3..24", hipSuccess, __FILE__, __LINE__);
  if (columns_on_device && scalar_fmt != BP_FR_MONT) return fail(ctx, BP_ERR_INVALID_ARG, "device columns must be Montgomery", hipSuccess, __FILE__, __LINE__);
  for (int k = 0; k < 8; k++)
    if (!columns[k]) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  const size_t n = (size_t)1 << log_n;
  fr_t* lag = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&lag, 8 * n * sizeof(fr_t)));
  for (int k = 0; k < 8; k++) {
    hipError_t e = hipMemcpyAsync(lag + (size_t)k * n, columns[k], n * sizeof(fr_t), columns_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(lag);
      return fail(ctx, BP_ERR_HIP, "circuit upload", e, __FILE__, __LINE__);
    }
  }
  int rc = BP_OK;
  if (scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(ctx, lag, 8 * n, 0);
  CircuitEntry e;
  if (rc == BP_OK) rc = circuit_build(ctx, log_n, lag, &e);
  if (rc != BP_OK) {
    (void)hipFree(lag);
    return rc;
  }
  if (is_group(ctx) || knob_u32("BP_PROVE_COSET_ONE", 0, 0, 1) == 1) {       // round 3 by coset over the members: their shares of the coset tables (experiment: one device, all four)
    rc = circuit_split_build(ctx, e);
    if (rc != BP_OK) {
      circuit_release(e);
      return rc;
    }
  }
  *handle = ctx->next_handle++;
  ctx->circuits[*handle] = e;
  return BP_OK;
}
int bp_circuit_free(bp_ctx* ctx, uint64_t handle) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  auto it = ctx->circuits.find(handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  BP_HIP(ctx, stream_wait(ctx->stream));
  circuit_release(it->second);
  ctx->circuits.erase(it);
  return BP_OK;
}
int bp_make_s_polynomials(uint32_t log_n, const uint32_t* wire_ids, void* s1, void* s2, void* s3) {
  if (!wire_ids || !s1 || !s2 || !s3 || log_n > 26) return BP_ERR_INVALID_ARG;
  const size_t n = (size_t)1 << log_n, cells = 3 * n;
  fr_t w;
  host_root_of_unity(w, n);
  std::vector<fr_t> root(n);                                  // roots_of_unity(group_order), utils.rs:45-52
  root[0] = Fr::one();
  for (size_t i = 1; i < n; i++) Fr::mul(root[i], root[i - 1], w);
  fr_t col[3], two, three;
  col[0] = Fr::one();
  Fr::add(two, col[0], col[0]);
  Fr::add(three, two, col[0]);
  col[1] = two;
  col[2] = three;
  auto label = [&](size_t cell) { fr_t r; Fr::mul(r, root[cell / 3], col[cell % 3]); return r; };      // Cell::label, utils.rs:28-37
  // cells grouped by variable, row-major inside a group (program.rs:80-99)
  std::vector<uint64_t> order(cells);
  for (size_t k = 0; k < cells; k++) order[k] = ((uint64_t)wire_ids[k] << 32) | k;
  std::sort(order.begin(), order.end());
  fr_t* out[3] = {(fr_t*)s1, (fr_t*)s2, (fr_t*)s3};
  for (size_t a = 0; a < cells;) {
    size_t b = a;
    while (b < cells && (order[b] >> 32) == (order[a] >> 32)) b++;
    for (size_t j = a; j < b; j++) {                          // uses[j] -> uses[j + 1 cyclically] receives label(uses[j]), :126-137
      const size_t cell = (size_t)(order[j] & 0xffffffffu), next = (size_t)(order[j + 1 < b ? j + 1 : a] & 0xffffffffu);
      out[next % 3][next / 3] = label(cell);
    }
    a = b;
  }
  return BP_OK;
}
int bp_circuit_commitments(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, uint8_t out768[768]) {
  if (!ctx || !out768) return BP_ERR_INVALID_ARG;
  auto it = ctx->circuits.find(circuit_handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  const size_t n = (size_t)1 << it->second.log_n;
  const fr_t* polys[8];
  size_t lens[8];
  g1_proj cm[8];
  for (int k = 0; k < 8; k++) { polys[k] = it->second.coef + (size_t)k * n; lens[k] = n; }
  BP_TRY(commit_many(ctx, srs_handle, polys, lens, 8, cm));
  for (int k = 0; k < 8; k++) host_encode96(out768 + 96 * k, cm[k]);
  return BP_OK;
}
int bp_prove(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, const void* a, const void* b, const void* c, const void* public_input,
             int scalar_fmt, int witness_on_device, const uint8_t blinders[352], uint8_t proof[624]) {
  if (!ctx || !a || !b || !c || !blinders || !proof || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  if (witness_on_device && scalar_fmt != BP_FR_MONT) return fail(ctx, BP_ERR_INVALID_ARG, "device witness must be Montgomery", hipSuccess, __FILE__, __LINE__);
  auto it = ctx->circuits.find(circuit_handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  SrsEntry* srs;
  BP_TRY(srs_find(ctx, srs_handle, &srs));
  const size_t n = (size_t)1 << it->second.log_n;
  (void)srs;      // an SRS shorter than group_order + 6 powers truncates the commitments exactly as Setup::commit's zip does (msm.rs:29)
  fr_t blind[11];
  for (int j = 0; j < 11; j++)
    if (!fr_bytes_to_mont(blind[j], blinders + 32 * j, BP_FR_BYTES_LE)) return fail(ctx, BP_ERR_BAD_SCALAR, "blinder >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* wit;
  BP_TRY(ws_get(ctx, "prove.witness", 4 * n * sizeof(fr_t), (void**)&wit));
  const void* cols[4] = {a, b, c, public_input};
  // host witness of 2^18 gates and more on one device: round 1 stages the columns itself, uploads beside the commitments (BP_PROVE_STAGED=0: off)
  bool staged = !witness_on_device && !is_group(ctx) && it->second.log_n >= 18;
  {
    const char* v = knob("BP_PROVE_STAGED");
    if (v && *v == '0') staged = false;
    if (v && *v == '1') staged = !witness_on_device && !is_group(ctx);
  }
  if (staged) {
    const ProveStaged hw = {{a, b, c, public_input}, scalar_fmt};
    return prove_run(ctx, srs_handle, it->second, wit, blind, proof, public_input == nullptr, &hw);
  }
  const hipMemcpyKind kind = witness_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  for (int k = 0; k < 4; k++) {
    if (cols[k]) BP_HIP(ctx, hipMemcpyAsync(wit + (size_t)k * n, cols[k], n * sizeof(fr_t), kind, ctx->stream));
    else BP_HIP(ctx, hipMemsetAsync(wit + (size_t)k * n, 0, n * sizeof(fr_t), ctx->stream));
  }
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, wit, 4 * n, 0));
  return prove_run(ctx, srs_handle, it->second, wit, blind, proof, public_input == nullptr);
}
int bp_prove_last_stats(bp_ctx* ctx, float round_ms[5], float* total_ms) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (round_ms) memcpy(round_ms, ctx->prove_ms, 5 * sizeof(float));
  if (total_ms) *total_ms = ctx->prove_ms[5];
  return BP_OK;
}
int bp_transcript_test_vector(uint8_t out32[32]) {
  if (!out32) return BP_ERR_INVALID_ARG;
  transcript_test_vector(out32);
  return BP_OK;
}
