// srs.hip -- host drivers of the SRS kernels (srs_kernels.hpp).
#include <stdlib.h>

#include "ctx.hpp"
#include "srs_kernels.hpp"

namespace bp {

int srs_decode_run(bp_ctx* ctx, const uint8_t* d_bytes, size_t n, g1_affine* d_out) {
  if (n == 0) return BP_OK;
  uint32_t* status;
  BP_TRY(ws_get(ctx, "srs.status", 4, (void**)&status));
  BP_HIP(ctx, hipMemsetAsync(status, 0, 4, ctx->stream));
  hipLaunchKernelGGL(srs_decode96, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_bytes, n, d_out, status);
  BP_HIP(ctx, hipGetLastError());
  uint32_t h = 0;
  BP_HIP(ctx, hipMemcpyAsync(&h, status, 4, hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  if (h) return fail(ctx, BP_ERR_BAD_POINT, h & 1 ? "non-canonical point encoding" : "point not on the curve", hipSuccess, __FILE__, __LINE__);
  return BP_OK;
}
int srs_encode_run(bp_ctx* ctx, const g1_affine* d_in, size_t n, uint8_t* d_bytes) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(srs_encode96, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, d_bytes);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
int srs_from_projective_run(bp_ctx* ctx, const g1_proj* d_in, size_t n, g1_affine* d_out) {
  if (n == 0) return BP_OK;
  // points per inversion: 2^16 lanes (one wave per SIMD), 8..64 points each.  Measured at 2^20 points (load of the literal seam):
  // 4.55 / 3.90 / 4.00 / 4.25 ms at 8 / 16 / 32 / 64 -- fewer inversions against a longer dependent chain through memory
  uint32_t group = 8;
  while (group < 64 && (n / group) > 65536) group <<= 1;
  {
    const char* v = knob("BP_SRS_PROJ_GROUP");        // experiment knob
    if (v) {
      const long g = strtol(v, nullptr, 10);
      if (g >= 1 && g <= 4096) group = (uint32_t)g;
    }
  }
  const size_t lanes = (n + group - 1) / group;
  hipLaunchKernelGGL(srs_from_projective, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, group, d_out);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
int srs_generate_run(bp_ctx* ctx, const fr_t& a, const fr_t& d, int mode, size_t first, size_t n, g1_affine* d_out) {
  if (n == 0) return BP_OK;
  // the table of the generator's multiples (1 MiB) lives in a workspace of the context: built by the first call, in stream order
  // (ctx->gen_table_ready: set only after the table kernel was accepted by the runtime -- the workspace map gains its entry before
  // the allocation, so the map alone would call a table "there" after a failed first call)
  g1_affine28* table;
  BP_TRY(ws_get(ctx, "srs.gen_table", (size_t)GEN_WINDOWS * GEN_DIGITS * sizeof(g1_affine28), (void**)&table));
  if (!ctx->gen_table_ready || ctx->gen_table_ptr != (const void*)table) {
    hipLaunchKernelGGL(srs_gen_table, dim3((GEN_WINDOWS * GEN_DIGITS + 255) / 256), dim3(256), 0, ctx->stream, table);
    BP_HIP(ctx, hipGetLastError());
    ctx->gen_table_ready = true;
    ctx->gen_table_ptr = table;
  }
  const size_t lanes = (n + SRS_GEN_GROUP - 1) / SRS_GEN_GROUP;
  hipLaunchKernelGGL(srs_generate_fb, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, a, d, mode, first, n, table, d_out);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

}  // namespace bp
