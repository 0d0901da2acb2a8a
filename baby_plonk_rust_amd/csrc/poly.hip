// poly.hip -- host drivers of the polynomial kernels (poly_kernels.hpp).
#include <string.h>

#include <algorithm>

#include "ctx.hpp"
#include "poly_kernels.hpp"

namespace bp {

// coeffs_evaluate (polynomial.rs:34-45): sum_i c_i x^i.  The reference spends one 256-bit pow per term;
// the sum is the same field element however it is associated.
int poly_eval_run(bp_ctx* ctx, const fr_t* d_coeffs, size_t n, const fr_t& x, fr_t* host_out) {
  if (n == 0) {
    *host_out = Fr::zero();
    return BP_OK;
  }
  uint32_t K = 16;
  while ((n + K - 1) / K > 256 * 1024 && K < (1u << 20)) K <<= 1;     // at most 1024 workgroups
  const size_t lanes = (n + K - 1) / K;
  const unsigned blocks = (unsigned)((lanes + 255) / 256);
  fr_t *partial, *result;
  BP_TRY(ws_get(ctx, "poly.partial", (size_t)blocks * sizeof(fr_t), (void**)&partial));
  BP_TRY(ws_get(ctx, "poly.result", 16 * sizeof(fr_t), (void**)&result));
  hipLaunchKernelGGL(poly_eval_partial, dim3(blocks), dim3(256), 256 * sizeof(fr_t), ctx->stream, d_coeffs, n, x, K, partial);
  hipLaunchKernelGGL(fr_sum_small, dim3(1), dim3(256), 256 * sizeof(fr_t), ctx->stream, partial, blocks, result);
  BP_HIP(ctx, hipGetLastError());
  BP_HIP(ctx, hipMemcpyAsync(host_out, result, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}

// k evaluations enqueued together, one copy of the k results, one wait (the six of prover round 4, prover.rs:502-541: six
// sequential calls were six stream waits of ~55 us each)
int poly_eval_many_run(bp_ctx* ctx, int k, const fr_t* const* d_coeffs, const size_t* n, const fr_t* x, fr_t* host_out) {
  if (k <= 0) return BP_OK;
  if (k > 8) return fail(ctx, BP_ERR_INVALID_ARG, "poly_eval_many: more than 8 evaluations", hipSuccess, __FILE__, __LINE__);
  PolyEvalMany a;
  memset(&a, 0, sizeof a);
  size_t total = 0;
  unsigned max_blocks = 1;
  for (int j = 0; j < k; j++) {
    uint32_t K = 16;
    while ((n[j] + K - 1) / K > 256 * 1024 && K < (1u << 20)) K <<= 1;
    const size_t lanes = (n[j] + K - 1) / K;
    a.c[j] = d_coeffs[j];
    a.n[j] = n[j];
    a.x[j] = x[j];
    a.K[j] = K;
    a.blocks[j] = (uint32_t)((lanes + 255) / 256);       // 0 for an empty polynomial: its sum over no partials is zero
    a.off[j] = (uint32_t)total;
    total += a.blocks[j];
    max_blocks = std::max(max_blocks, (unsigned)a.blocks[j]);
  }
  fr_t *partial, *result;
  BP_TRY(ws_get(ctx, "poly.partial", (total ? total : 1) * sizeof(fr_t), (void**)&partial));
  BP_TRY(ws_get(ctx, "poly.result", 16 * sizeof(fr_t), (void**)&result));
  hipLaunchKernelGGL(poly_eval_partial_many, dim3(max_blocks, (unsigned)k), dim3(256), 256 * sizeof(fr_t), ctx->stream, a, partial);
  hipLaunchKernelGGL(fr_sum_small_many, dim3((unsigned)k), dim3(256), 256 * sizeof(fr_t), ctx->stream, a, partial, result);
  BP_HIP(ctx, hipGetLastError());
  BP_HIP(ctx, hipMemcpyAsync(host_out, result, (size_t)k * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}

// True polynomial quotient of a (na coeffs, a[na-1] != 0) by b (nb coeffs, b[nb-1] != 0), na >= nb.
// d_a is clobbered by the general path.  host_b = host copy of b (to pick the fast path).
int poly_div_run(bp_ctx* ctx, fr_t* d_a, size_t na, const fr_t* d_b, size_t nb, const fr_t& b0, const fr_t& b_lead, bool binomial,
                 fr_t* d_q, size_t nq) {
  fr_t lead_inv;
  fr_invert(lead_inv, b_lead);
  if (binomial) {
    const size_t m = nb - 1;
    fr_t f;
    Fr::mul(f, b0, lead_inv);
    Fr::neg(f, f);                                   // f = -b0 / bm
    const size_t max_len = (nq + m - 1) / m;
    const uint32_t K = 32;
    const size_t chunks = (max_len + K - 1) / K;
    fr_t *head, *carry;
    BP_TRY(ws_get(ctx, "poly.div_head", chunks * m * sizeof(fr_t), (void**)&head));
    BP_TRY(ws_get(ctx, "poly.div_carry", chunks * m * sizeof(fr_t), (void**)&carry));
    const size_t lanes = m * chunks;
    hipLaunchKernelGGL(poly_div_binomial_local, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, d_a, nq, m, f,
                       lead_inv, K, chunks, d_q, head);
    if (chunks > 1) {
      if (chunks > 4096 && m <= 16) {        // very long chains (x - zeta at 2^20): G segments per chain, a workgroup each
        uint32_t G = (uint32_t)((chunks + 1023) / 1024);
        if (G > 64) G = 64;
        fr_t *seg_map, *seg_carry;
        BP_TRY(ws_get(ctx, "poly.div_seg_map", 2 * m * G * sizeof(fr_t), (void**)&seg_map));
        BP_TRY(ws_get(ctx, "poly.div_seg_carry", m * G * sizeof(fr_t), (void**)&seg_carry));
        hipLaunchKernelGGL(poly_div_binomial_carry_seg<0>, dim3((unsigned)m, G), dim3(1024), 2048 * sizeof(fr_t), ctx->stream, nq, m, f, K, chunks, G,
                           head, seg_map, seg_carry, carry);
        hipLaunchKernelGGL(poly_div_seg_scan, dim3((unsigned)m), dim3(64), 0, ctx->stream, G, seg_map, seg_carry);
        hipLaunchKernelGGL(poly_div_binomial_carry_seg<1>, dim3((unsigned)m, G), dim3(1024), 2048 * sizeof(fr_t), ctx->stream, nq, m, f, K, chunks, G,
                           head, seg_map, seg_carry, carry);
      } else if (chunks > 64 && m <= 4096)   // long chains: a workgroup per chain
        hipLaunchKernelGGL(poly_div_binomial_carry_wg, dim3((unsigned)m), dim3(1024), 2048 * sizeof(fr_t), ctx->stream, nq, m, f, K, chunks,
                           head, carry);
      else
        hipLaunchKernelGGL(poly_div_binomial_carry, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, nq, m, f, K, chunks,
                           head, carry);
      hipLaunchKernelGGL(poly_div_binomial_apply, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, nq, m, f, K, chunks,
                         carry, d_q);
    }
  } else {
    hipLaunchKernelGGL(poly_div_general, dim3(1), dim3(1024), 0, ctx->stream, d_a, na, d_b, nb, lead_inv, d_q);
  }
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

// effective length (trailing zeros trimmed) and the number of non-zero entries in [lo, hi)
int fr_nonzero_stats_run(bp_ctx* ctx, const fr_t* d_a, size_t n, size_t lo, size_t hi, size_t* eff_len, size_t* nonzero_in_range) {
  unsigned long long* d_out;
  BP_TRY(ws_get(ctx, "poly.nzstats", 16, (void**)&d_out));
  BP_HIP(ctx, hipMemsetAsync(d_out, 0, 16, ctx->stream));
  if (n) hipLaunchKernelGGL(fr_nonzero_stats, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, ctx->stream, d_a, n, lo, hi, d_out);
  unsigned long long h[2];
  BP_HIP(ctx, hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  *eff_len = (size_t)h[0];
  if (nonzero_in_range) *nonzero_in_range = (size_t)h[1];
  return BP_OK;
}
// d_q[0..*n) -> its non-zero elements in order, in place (through a workspace); *n = how many remain.  All on the device.
int fr_compact_nonzero_run(bp_ctx* ctx, fr_t* d_q, size_t* n) {
  if (*n == 0) return BP_OK;
  if (*n >= ((size_t)1 << 32)) return fail(ctx, BP_ERR_TOO_LARGE, "compaction longer than 2^32", hipSuccess, __FILE__, __LINE__);
  const uint32_t n_tiles = (uint32_t)((*n + COMPACT_TILE - 1) / COMPACT_TILE);
  uint32_t* tiles;
  fr_t* tmp;
  unsigned long long* d_total;
  BP_TRY(ws_get(ctx, "poly.compact_tiles", (size_t)n_tiles * 4, (void**)&tiles));
  BP_TRY(ws_get(ctx, "poly.compact_tmp", *n * sizeof(fr_t), (void**)&tmp));
  BP_TRY(ws_get(ctx, "poly.nzstats", 16, (void**)&d_total));
  hipLaunchKernelGGL(fr_compact_count, dim3(n_tiles), dim3(256), 0, ctx->stream, d_q, *n, tiles);
  hipLaunchKernelGGL(fr_compact_scan, dim3(1), dim3(256), 0, ctx->stream, tiles, n_tiles, d_total);
  hipLaunchKernelGGL(fr_compact_scatter, dim3(n_tiles), dim3(256), 0, ctx->stream, d_q, *n, tiles, tmp);
  BP_HIP(ctx, hipGetLastError());
  unsigned long long m = 0;
  BP_HIP(ctx, hipMemcpyAsync(&m, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  if (m) BP_HIP(ctx, hipMemcpyAsync(d_q, tmp, (size_t)m * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  *n = (size_t)m;
  return BP_OK;
}
int fr_scale_powers_run(bp_ctx* ctx, const fr_t* d_a, size_t n, const fr_t& w, fr_t* d_out) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fr_scale_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_a, n, w, d_out);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

// out = scan of `in` by products; *d_total (device, 1 element) receives the product of all n elements
int fr_scan_mul_run(bp_ctx* ctx, const fr_t* d_in, size_t n, int reverse, int inclusive, fr_t* d_out, fr_t* d_total) {
  const uint32_t n_tiles = (uint32_t)((n + SCANM_TILE - 1) / SCANM_TILE);
  if (n_tiles > 256 * 64) return fail(ctx, BP_ERR_TOO_LARGE, "scan longer than 2^25", hipSuccess, __FILE__, __LINE__);
  fr_t* tiles;
  BP_TRY(ws_get(ctx, reverse ? "poly.scan_tiles_r" : "poly.scan_tiles_f", (size_t)(n_tiles ? n_tiles : 1) * sizeof(fr_t), (void**)&tiles));
  const size_t lds = 256 * sizeof(fr_t);
  if (n_tiles) hipLaunchKernelGGL(fr_scan_tile_products, dim3(n_tiles), dim3(256), lds, ctx->stream, d_in, n, reverse, tiles);
  hipLaunchKernelGGL(fr_scan_tiles, dim3(1), dim3(256), lds, ctx->stream, tiles, n_tiles, d_total);
  if (n_tiles) hipLaunchKernelGGL(fr_scan_apply, dim3(n_tiles), dim3(256), lds, ctx->stream, d_in, n, reverse, inclusive, tiles, d_out);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

// prover.rs:279-319 on device-resident Montgomery columns; d_z receives n values (z_0 .. z_{n-1})
int grand_product_run(bp_ctx* ctx, const fr_t* a, const fr_t* b, const fr_t* c, const fr_t* s1, const fr_t* s2, const fr_t* s3, size_t n,
                      const fr_t& beta, const fr_t& gamma, const fr_t& k1, const fr_t& k2, const fr_t& root, fr_t* d_z, const fr_t* d_roots) {
  if (n == 0) return BP_OK;
  fr_t *roots, *num, *den, *pn, *sd, *totals;
  if (d_roots) roots = const_cast<fr_t*>(d_roots);           // the caller keeps roots_of_unity(n) resident (read-only here)
  else BP_TRY(ws_get(ctx, "gp.roots", n * sizeof(fr_t), (void**)&roots));
  BP_TRY(ws_get(ctx, "gp.num", n * sizeof(fr_t), (void**)&num));
  BP_TRY(ws_get(ctx, "gp.den", n * sizeof(fr_t), (void**)&den));
  BP_TRY(ws_get(ctx, "gp.pn", n * sizeof(fr_t), (void**)&pn));
  BP_TRY(ws_get(ctx, "gp.sd", n * sizeof(fr_t), (void**)&sd));
  BP_TRY(ws_get(ctx, "gp.totals", 2 * sizeof(fr_t), (void**)&totals));
  if (!d_roots) BP_TRY(roots_run(ctx, root, n, roots));                        // roots_of_unity(group_order), prover.rs:282
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(grand_product_terms, dim3(blocks), dim3(256), 0, ctx->stream, a, b, c, s1, s2, s3, roots, n, beta, gamma, k1, k2, num, den);
  BP_TRY(fr_scan_mul_run(ctx, num, n, 0, 0, pn, totals));                      // exclusive prefix products of the numerators
  BP_TRY(fr_scan_mul_run(ctx, den, n, 1, 1, sd, totals + 1));                  // inclusive suffix products of the denominators
  fr_t h_tot[2];
  BP_HIP(ctx, hipMemcpyAsync(h_tot, totals, 2 * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  if (big_is_zero(h_tot[1])) return fail(ctx, BP_ERR_DIV_ZERO, "round 2: a permutation denominator is zero (invert().unwrap())", hipSuccess, __FILE__, __LINE__);
  if (!big_eq(h_tot[0], h_tot[1])) return fail(ctx, BP_ERR_ASSERT, "round 2: z_n != 1 (prover.rs:319)", hipSuccess, __FILE__, __LINE__);
  fr_t td_inv;
  fr_invert(td_inv, h_tot[1]);
  hipLaunchKernelGGL(grand_product_combine, dim3(blocks), dim3(256), 0, ctx->stream, pn, sd, td_inv, n, d_z);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

}  // namespace bp
