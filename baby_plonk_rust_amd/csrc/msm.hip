// msm.hip -- host driver of the G1 MSM pipeline (kernels in msm_kernels.cuh).
#include <stdlib.h>
#include <string.h>

#include "ctx.hpp"
#include "msm_kernels.cuh"

namespace bp {

static uint32_t ilog2_floor(size_t n) {
  uint32_t l = 0;
  while ((n >> (l + 1)) != 0) l++;
  return l;
}
static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = getenv(name);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// Window width: minimise W * (n + 2 * 2^(c-1)) (bucket adds + reduction adds), bounded by the int16 digit
// array and by the 128 KiB LDS histogram (c <= 16).
static void make_plan(MsmPlan& plan, size_t n) {
  memset(&plan, 0, sizeof plan);
  plan.n = (uint32_t)n;
  uint32_t c = n < 32 ? 4 : ilog2_floor(n) - 3;
  if (c < 4) c = 4;
  if (c > MSM_MAX_C) c = MSM_MAX_C;
  c = env_u32("BP_MSM_C", c);
  if (c < 2) c = 2;
  if (c > MSM_MAX_C) c = MSM_MAX_C;
  // digits d_w = ((k + bias) >> c*w & mask) - 2^(c-1) with bias = sum_w 2^(c-1) 2^(cw); needs k + bias < 2^(cW)
  // for every k < q.  Take W = ceil(256 / c) and check the bound with the real q; add a window if it fails.
  static const uint32_t q_minus_1[8] = {0x00000000u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  uint32_t W = (256 + c - 1) / c;
  for (;;) {
    uint32_t bias[10] = {0};
    for (uint32_t w = 0; w < W; w++) {
      uint32_t bit = c * w + c - 1;
      if (bit < 320) bias[bit >> 5] |= 1u << (bit & 31);
    }
    uint64_t carry = 0;
    uint32_t sum[10];
    for (int j = 0; j < 10; j++) {
      carry += (uint64_t)(j < 8 ? q_minus_1[j] : 0u) + bias[j];
      sum[j] = (uint32_t)carry;
      carry >>= 32;
    }
    // highest set bit of sum must be below c*W
    int top = -1;
    for (int b = 319; b >= 0; b--)
      if ((sum[b >> 5] >> (b & 31)) & 1) { top = b; break; }
    if (top < (int)(c * W) && c * W <= 288) {
      memcpy(plan.bias, bias, sizeof plan.bias);
      break;
    }
    W++;
  }
  plan.c = c;
  plan.W = W;
  plan.B = 1u << (c - 1);
  const uint64_t entries = (uint64_t)W * n;
  uint32_t chunk = 4;
  while (chunk < 64 && entries / chunk > 262144) chunk <<= 1;
  plan.chunk = env_u32("BP_MSM_CHUNK", chunk);
  uint32_t slices = 1024 / W;
  if (slices < 1) slices = 1;
  while (slices > 1 && n / slices < 1024) slices >>= 1;
  plan.slices = slices;
  uint32_t seg = 1;
  while (seg < 32 && (uint64_t)W * plan.B / seg > 65536) seg <<= 1;
  plan.seg = env_u32("BP_MSM_SEG", seg);
}

int msm_run(bp_ctx* ctx, const g1_affine* d_points, size_t n, const fr_t* d_scalars, int fmt, g1_proj* host_out) {
  if (n == 0) {
    *host_out = g1_identity();
    ctx->msm_accumulate_ms = ctx->msm_total_ms = 0;
    ctx->msm_adds = 0;
    return BP_OK;
  }
  if (n >= (1ull << 31)) return fail(ctx, BP_ERR_TOO_LARGE, "MSM length >= 2^31", hipSuccess, __FILE__, __LINE__);
  MsmPlan plan;
  make_plan(plan, n);
  const uint32_t W = plan.W, B = plan.B, total = W * B;
  const uint64_t max_entries = (uint64_t)W * n;
  const uint64_t n_chunks = (max_entries + plan.chunk - 1) / plan.chunk;
  const uint32_t lanes_per_window = (B + plan.seg - 1) / plan.seg;
  const uint32_t blocks_per_window = (lanes_per_window + 255) / 256;

  int16_t* digits;
  uint32_t *counts, *offsets, *cursors, *sorted;
  g1_proj *bucket_sum, *partial, *block_out, *window_sum;
  BP_TRY(ws_get(ctx, "msm.digits", max_entries * sizeof(int16_t), (void**)&digits));
  BP_TRY(ws_get(ctx, "msm.counts", (size_t)total * 4, (void**)&counts));
  BP_TRY(ws_get(ctx, "msm.offsets", ((size_t)total + 1) * 4, (void**)&offsets));
  BP_TRY(ws_get(ctx, "msm.cursors", (size_t)total * 4, (void**)&cursors));
  BP_TRY(ws_get(ctx, "msm.sorted", max_entries * 4, (void**)&sorted));
  BP_TRY(ws_get(ctx, "msm.bucket_sum", (size_t)total * sizeof(g1_proj), (void**)&bucket_sum));
  BP_TRY(ws_get(ctx, "msm.partial", 2 * n_chunks * sizeof(g1_proj), (void**)&partial));
  BP_TRY(ws_get(ctx, "msm.block_out", (size_t)W * blocks_per_window * sizeof(g1_proj), (void**)&block_out));
  BP_TRY(ws_get(ctx, "msm.window_sum", (size_t)W * sizeof(g1_proj), (void**)&window_sum));
  g1_proj* h_windows;
  BP_TRY(pinned_get(ctx, (size_t)W * sizeof(g1_proj), (void**)&h_windows));

  static bool lds_attr_set = false;
  if (!lds_attr_set) {            // a full-window histogram at c = 16 needs 128 KiB of dynamic LDS
    BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_count, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_attr_set = true;
  }
  hipStream_t st = ctx->stream;
  BP_HIP(ctx, hipEventRecord(ctx->ev[0], st));
  BP_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)total * 4, st));
  hipLaunchKernelGGL(msm_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, digits);
  const size_t hist_bytes = (size_t)B * 4;
  hipLaunchKernelGGL(msm_count, dim3(plan.slices, W), dim3(256), hist_bytes, st, digits, plan, counts);
  hipLaunchKernelGGL(scan_u32, dim3(1), dim3(1024), 0, st, counts, total, offsets, cursors);
  hipLaunchKernelGGL(msm_scatter, dim3(plan.slices, W), dim3(256), hist_bytes, st, digits, plan, cursors, sorted);
  BP_HIP(ctx, hipEventRecord(ctx->ev[1], st));
  hipLaunchKernelGGL(msm_accumulate, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, d_points, sorted, offsets, plan,
                     bucket_sum, partial);
  BP_HIP(ctx, hipEventRecord(ctx->ev[2], st));
  hipLaunchKernelGGL(msm_fixup, dim3((total + 255) / 256), dim3(256), 0, st, offsets, plan, bucket_sum, partial);
  hipLaunchKernelGGL(msm_reduce, dim3(blocks_per_window, W), dim3(256), 256 * sizeof(g1_proj), st, offsets, plan, bucket_sum,
                     block_out);
  hipLaunchKernelGGL(msm_window_finish, dim3((W + 63) / 64), dim3(64), 0, st, block_out, blocks_per_window, W, window_sum);
  BP_HIP(ctx, hipGetLastError());
  BP_HIP(ctx, hipMemcpyAsync(h_windows, window_sum, (size_t)W * sizeof(g1_proj), hipMemcpyDeviceToHost, st));
  BP_HIP(ctx, hipEventRecord(ctx->ev[3], st));
  BP_HIP(ctx, hipStreamSynchronize(st));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->msm_accumulate_ms, ctx->ev[1], ctx->ev[2]));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->msm_total_ms, ctx->ev[0], ctx->ev[3]));
  ctx->msm_c = plan.c;
  ctx->msm_adds = max_entries;      // upper bound: zero digits are skipped (about n*W/2^c of them)
  host_horner(*host_out, h_windows, W, plan.c);
  return BP_OK;
}

}  // namespace bp
