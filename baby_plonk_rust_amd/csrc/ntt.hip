// ntt.hip -- host driver of the Fr NTT (kernels in ntt_kernels.hpp) and of the element-wise Fr kernels.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctx.hpp"
#include "ntt_kernels.hpp"

namespace bp {

static fr_t host_root_pow2(bool inverse, uint32_t log_order) {
  // w_{2^log_order} = ROOT_OF_UNITY^(2^(32 - log_order))   (utils.rs:39-43)
  fr_t w = fr_root_of_unity(inverse);
  for (uint32_t i = log_order; i < 32; i++) Fr::sqr(w, w);
  return w;
}

static int make_table(bp_ctx* ctx, const fr_t& base, uint32_t count, uint32_t shift, const fr_t* d_scale, fr_t* d_out_fr,
                      tw29_t* d_out_tw) {
  hipLaunchKernelGGL(ntt_make_table, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, base, count, shift, d_scale, d_out_fr,
                     d_out_tw);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

int ntt_init_tables(bp_ctx* ctx) {
  for (int inv = 0; inv < 2; inv++) {
    BP_HIP(ctx, hipMalloc((void**)&ctx->small_tw[inv], 512 * sizeof(tw29_t)));
    BP_TRY(make_table(ctx, host_root_pow2(inv != 0, NTT_SMALL_MAX_LOG), 512, 0, nullptr, nullptr, ctx->small_tw[inv]));
  }
  // per function and per device: set for every context (this runs in bp_init, after hipSetDevice)
  BP_HIP(ctx, hipFuncSetAttribute((const void*)ntt_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_strided, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_last, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_strided_swz, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_last_swz, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  return BP_OK;
}

// one radix-4 group (four elements, two stages in registers) per lane: 2^(l-2) * columns lanes, capped at 512 (the kernels'
// launch bound: a group needs ~140 registers).  BP_NTT_LANES_SHIFT = 1 doubles the lanes (they only help the load/store phases).
static unsigned pass_threads(uint32_t l, uint32_t cl) {
  unsigned t = (((1u << l) << cl) >> 2) << knob_u32("BP_NTT_LANES_SHIFT", 0, 0, 1);
  return t > 512 ? 512 : (t < 64 ? 64 : t);
}

static void make_ntt_plan(NttPlan& plan, uint32_t k) {
  memset(&plan, 0, sizeof plan);
  plan.k = k;
  plan.P = k <= NTT_SMALL_MAX_LOG ? 1 : (k <= 2 * NTT_MAX_PASS_LOG ? 2 : 3);
  // 2^20 as 7 + 7 + 6: small tiles keep three workgroups per CU busy (0.152 ms against 0.159 for 10 + 10, whose 74-KiB tiles
  // leave one 512-lane workgroup per CU); 2^19 stays 10 + 9 (0.080 against 0.084).  profiles/r02_ntt_radix4_ab.txt
  if (plan.P == 2 && k >= knob_u32("BP_NTT_THREE_PASS_FROM", 20, 11, 29)) plan.P = 3;
  uint32_t rem = k;
  for (uint32_t i = 0; i < plan.P; i++) {          // balanced widths, larger ones first
    uint32_t left = plan.P - i;
    plan.l[i] = (rem + left - 1) / left;
    rem -= plan.l[i];
    plan.cl[i] = ntt_tile_cols_log(plan.l[i]);
    if (plan.l[i] <= 7) plan.cl[i] = knob_u32("BP_NTT_COLS_LOG_L7", plan.cl[i], 0, 3);
    if (plan.l[i] == 8) plan.cl[i] = knob_u32("BP_NTT_COLS_LOG_L8", plan.cl[i], 0, 3);
    if (plan.l[i] == 9) plan.cl[i] = knob_u32("BP_NTT_COLS_LOG_L9", plan.cl[i], 0, 3);
    if (plan.l[i] == 10) plan.cl[i] = knob_u32("BP_NTT_COLS_LOG_L10", plan.cl[i], 0, 3);
    if (plan.cl[i] > 3) plan.cl[i] = 3;
    // an override must still fit the 160 KiB of LDS (tile rows x (columns + pad) + stage twiddles, 36 B each)
    if ((((size_t)1 << plan.l[i]) * ((1u << plan.cl[i]) + 1) + ntt_tw_slots(plan.l[i]) + 2) * N29 * 4 + 16 > 160 * 1024) plan.cl[i] = ntt_tile_cols_log(plan.l[i]);
  }
  // three passes of widths (a, a, a - 1) with 8-column tiles everywhere (a <= 7: 2^20 = 7 + 7 + 6): the short digit goes in the MIDDLE --
  // the last pass pays no inter-pass twiddle product, so it is the cheapest place for a stage: 0.148 against 0.152 ms at 2^20, three alternating
  // rounds (profiles/r05_ntt_order_ab.txt); 2^22 (8, 7, 7) and 2^23 (8, 8, 7) show no difference between the orders and keep theirs
  if (plan.P == 3 && plan.l[0] <= 7 && plan.l[0] == plan.l[1] && plan.l[2] + 1 == plan.l[1]) {
    const uint32_t t = plan.l[1];
    plan.l[1] = plan.l[2];
    plan.l[2] = t;
    for (uint32_t i = 1; i < 3; i++) plan.cl[i] = ntt_tile_cols_log(plan.l[i]);
  }
  // experiment knob: BP_NTT_SPLIT="k:l1,l2,l3" replaces the digit widths of one size (widths must add up to k, each <= 10)
  if (const char* e = knob("BP_NTT_SPLIT")) {
    unsigned kk = 0, a = 0, b = 0, c = 0;
    const int got = sscanf(e, "%u:%u,%u,%u", &kk, &a, &b, &c);
    if (got >= 3 && kk == k && a + b + c == k && a <= (unsigned)NTT_MAX_PASS_LOG && b <= (unsigned)NTT_MAX_PASS_LOG && c <= (unsigned)NTT_MAX_PASS_LOG &&
        a >= 2 && b >= 1 && k > (uint32_t)NTT_SMALL_MAX_LOG) {
      plan.P = c ? 3 : 2;
      plan.l[0] = a; plan.l[1] = b; plan.l[2] = c;
      for (uint32_t i = 0; i < plan.P; i++) plan.cl[i] = ntt_tile_cols_log(plan.l[i]);
    }
  }
  plan.h = (k + 1) / 2;
}

static int get_tables(bp_ctx* ctx, uint32_t k, int inverse, NttTables** out) {
  const uint32_t key = k * 2 + (inverse ? 1 : 0);
  auto it = ctx->ntt_tables.find(key);
  if (it != ctx->ntt_tables.end()) {
    *out = &it->second;
    return BP_OK;
  }
  NttTables t;
  t.h = (k + 1) / 2;
  const uint32_t nlo = 1u << t.h, nhi = 1u << (k - t.h);
  const fr_t w = host_root_pow2(inverse != 0, k);
  BP_HIP(ctx, hipMalloc((void**)&t.lo, (size_t)nlo * sizeof(tw29_t)));
  BP_HIP(ctx, hipMalloc((void**)&t.hi, (size_t)nhi * sizeof(tw29_t)));
  BP_TRY(make_table(ctx, w, nlo, 0, nullptr, nullptr, t.lo));
  BP_TRY(make_table(ctx, w, nhi, t.h, nullptr, nullptr, t.hi));
  if (inverse) {
    // N^-1 = (2^k)^-1 in Montgomery form (utils.rs:126: Scalar::from(n).invert())
    fr_t two = Fr::one(), n_m = Fr::one(), n_inv;
    Fr::dbl(two, two);
    for (uint32_t i = 0; i < k; i++) Fr::mul(n_m, n_m, two);
    fr_invert(n_inv, n_m);
    BP_HIP(ctx, hipMalloc((void**)&t.n_inv, sizeof(fr_t)));
    BP_HIP(ctx, hipMemcpyAsync(t.n_inv, &n_inv, sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
    BP_HIP(ctx, stream_wait(ctx->stream));    // n_inv lives on this stack frame
    BP_HIP(ctx, hipMalloc((void**)&t.hi_scaled, (size_t)nhi * sizeof(tw29_t)));
    BP_TRY(make_table(ctx, w, nhi, t.h, t.n_inv, nullptr, t.hi_scaled));
    BP_HIP(ctx, hipMalloc((void**)&t.n_inv_tw, sizeof(tw29_t)));
    BP_TRY(make_table(ctx, w, 1, 0, t.n_inv, nullptr, t.n_inv_tw));       // w^0 * N^-1
  }
  ctx->ntt_tables[key] = t;
  *out = &ctx->ntt_tables[key];
  return BP_OK;
}

// One transform cut in two phases for a group context (NTT option ii; capi_ntt.hip ntt_one_over_members): with the
// digits N = 2^(l_1 + s), phase 0 is pass 1 on the tiles of a COLUMN slice (all d_1, r in the part's range), phase 1 the passes
// 2 .. P on the slice of e_1 (each e_1 owns 2^s contiguous elements of the intermediate buffer).  Between the phases the members
// exchange blocks of the intermediate buffer.  Every member keeps buffers in the full N-element layout, so addresses are the
// single-GPU ones and a part is just a range of tiles: [part, part + 1) * tiles / parts in every pass (e_1 is the most
// significant part of every later pass's tile index).
bool ntt_split_ok(uint32_t k, uint32_t parts) {
  if (k <= NTT_SMALL_MAX_LOG || parts < 2 || (parts & (parts - 1)) || parts > 8) return false;
  NttPlan plan;
  make_ntt_plan(plan, k);
  uint32_t lg = 0;
  while ((1u << lg) < parts) lg++;
  const uint32_t s0 = k - plan.l[0];
  if (s0 < plan.cl[0] + lg) return false;                                    // pass 1: whole tiles of columns per part
  if (plan.l[0] < plan.cl[plan.P - 1] + lg) return false;                    // last pass: whole tiles of e_1 per part
  return true;
}
void ntt_split_shape(uint32_t k, uint32_t* l1) {
  NttPlan plan;
  make_ntt_plan(plan, k);
  *l1 = plan.l[0];
}
int ntt_tmp_buffer(bp_ctx* ctx, uint32_t k, fr_t** out) { return ws_get(ctx, "ntt.tmp", ((size_t)1 << k) * sizeof(fr_t), (void**)out); }

// phase -1: the whole transform; 0 / 1: see above (batch must be 1)
int ntt_run_part(bp_ctx* ctx, fr_t* d_data, uint32_t k, int inverse, size_t batch, size_t stride, int phase, uint32_t part, uint32_t parts) {
  if (k > 28) return fail(ctx, BP_ERR_TOO_LARGE, "NTT length > 2^28", hipSuccess, __FILE__, __LINE__);
  if (batch == 0) return BP_OK;
  if (batch > 65535) return fail(ctx, BP_ERR_TOO_LARGE, "NTT batch > 65535", hipSuccess, __FILE__, __LINE__);
  const size_t N = (size_t)1 << k;
  if (batch > 1 && stride < N) return fail(ctx, BP_ERR_INVALID_ARG, "NTT stride < N", hipSuccess, __FILE__, __LINE__);
  if (phase >= 0 && (batch != 1 || !ntt_split_ok(k, parts) || part >= parts))
    return fail(ctx, BP_ERR_INVALID_ARG, "NTT phase", hipSuccess, __FILE__, __LINE__);
  NttPlan plan;
  make_ntt_plan(plan, k);
  NttTables* tab;
  BP_TRY(get_tables(ctx, k, inverse, &tab));
  hipStream_t st = ctx->stream;
  const tw29_t* small = ctx->small_tw[inverse ? 1 : 0];
  BP_HIP(ctx, hipEventRecord(ctx->ev[phase == 1 ? 2 : 0], st));
  if (plan.P == 1) {
    const size_t lds = (((N + 1) & ~(size_t)1) + ntt_tw_slots(k)) * N29 * 4 + 16;
    hipLaunchKernelGGL(ntt_small, dim3((unsigned)batch), dim3(256), lds, st, d_data, stride, k, small,
                       inverse ? tab->n_inv_tw : (const tw29_t*)nullptr);
  } else {
    // ping-pong: pass 1 data -> tmp, middle passes in tmp, last pass tmp -> data
    fr_t* tmp;
    BP_TRY(ws_get(ctx, "ntt.tmp", batch * N * sizeof(fr_t), (void**)&tmp));
    // 2^l x 8 tiles with l <= 7 run unpadded with swizzled rows (39 KiB at l = 7, bank-conflict-free drain); BP_NTT_SWIZZLE=0: padded
    const bool swz_on = knob_u32("BP_NTT_SWIZZLE", 1, 0, 1) != 0;
    auto swizzled = [&](uint32_t l, uint32_t cl) { return swz_on && cl == 3 && l <= 7 && l >= 2; };
    auto tile_lds = [&](uint32_t l, uint32_t cl) {
      const uint32_t C = 1u << cl, tstride = ((1u << l) * (swizzled(l, cl) || C == 1 ? C : C + 1) + 1) & ~1u;
      return ((size_t)tstride + ntt_tw_slots(l)) * N29 * 4 + 16;
    };
    auto tile_range = [&](unsigned tiles, unsigned* first, unsigned* count) {           // the part's share of a pass's tiles
      *first = phase < 0 ? 0u : tiles / parts * part;
      *count = phase < 0 ? tiles : tiles / parts;
    };
    uint32_t s = k;
    for (uint32_t i = 0; i + 1 < plan.P; i++) {
      const uint32_t l = plan.l[i], cl = plan.cl[i];
      s -= l;
      if ((phase == 0 && i > 0) || (phase == 1 && i == 0)) continue;
      const size_t lds = tile_lds(l, cl);
      const tw29_t* hi = (inverse && i == 0) ? tab->hi_scaled : tab->hi;   // N^-1 rides on the first twiddle
      if (tab->full[i] && tab->full_ls[i] != ((l << 8) | s)) {
        BP_HIP(ctx, stream_wait(st));
        BP_HIP(ctx, hipFree(tab->full[i]));
        tab->full[i] = nullptr;
      }
      if (!tab->full[i] && k <= knob_u32("BP_NTT_FULL_TWIDDLES_MAX_LOG", 24, 0, 28)) {      // built once per (N, direction, pass)
        const size_t M = (size_t)1 << (l + s);
        tab->full_ls[i] = (l << 8) | s;
        BP_HIP(ctx, hipMalloc((void**)&tab->full[i], M * sizeof(tw29_t)));
        hipLaunchKernelGGL(ntt_make_pass_table, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, tab->lo, hi, tab->h, l, s, k - (l + s),
                           tab->full[i]);
        BP_HIP(ctx, hipGetLastError());
      }
      unsigned first, count;
      tile_range((unsigned)(N >> (l + cl)), &first, &count);
      NttStridedArgs sa;
      sa.src = i == 0 ? (const fr_t*)d_data : (const fr_t*)tmp;
      sa.src_stride = i == 0 ? stride : N;
      sa.k = k; sa.l = l; sa.s = s; sa.cl = cl;
      sa.small_tw = small;
      sa.tile0 = first;
      sa.h = tab->h;
      sa.dst = (decltype(sa.dst))tmp;
      sa.dst_stride = N;
      sa.tw_lo = (decltype(sa.tw_lo))tab->lo; sa.tw_hi = (decltype(sa.tw_hi))hi; sa.tw_full = (decltype(sa.tw_full))tab->full[i];
      hipLaunchKernelGGL(swizzled(l, cl) ? ntt_pass_strided_swz : ntt_pass_strided, dim3(count, (unsigned)batch), dim3(pass_threads(l, cl)), lds, st, sa);
    }
    if (phase != 0) {
      const uint32_t l = plan.l[plan.P - 1], cl = plan.cl[plan.P - 1];
      const size_t lds = tile_lds(l, cl);
      unsigned first, count;
      tile_range((unsigned)(N >> (l + cl)), &first, &count);
      NttLastArgs la;
      la.src = tmp;
      la.src_stride = N;
      la.small_tw = small;
      la.tile0 = first;
      la.plan = plan;
      la.dst = (decltype(la.dst))d_data;
      la.dst_stride = stride;
      hipLaunchKernelGGL(swizzled(l, cl) ? ntt_pass_last_swz : ntt_pass_last, dim3(count, (unsigned)batch), dim3(pass_threads(l, cl)), lds, st, la);
    }
  }
  BP_HIP(ctx, hipGetLastError());
  BP_HIP(ctx, hipEventRecord(ctx->ev[phase == 1 ? 3 : 1], st));
  ctx->ntt_passes = plan.P;
  return BP_OK;
}
int ntt_run(bp_ctx* ctx, fr_t* d_data, uint32_t k, int inverse, size_t batch, size_t stride) {
  return ntt_run_part(ctx, d_data, k, inverse, batch, stride, -1, 0, 1);
}

int fr_convert_run(bp_ctx* ctx, fr_t* d, size_t n, int dir) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fr_convert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, n, dir);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
int fr_binary_run(bp_ctx* ctx, const fr_t* a, size_t na, const fr_t* b, size_t nb, fr_t* out, size_t n, int op) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fr_binary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, na, b, nb, out, n, op);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
int fr_scalar_run(bp_ctx* ctx, const fr_t* a, const fr_t& s, fr_t* out, size_t n, int op) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fr_scalar_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, s, out, n, op);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
int fr_synthetic_run(bp_ctx* ctx, fr_t* d_out, size_t n, uint64_t seed) {
  if (n == 0) return BP_OK;
  hipLaunchKernelGGL(fr_synthetic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n, seed);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}
// out[j] = w^j, j < n   (roots_of_unity, utils.rs:45-52)
int roots_run(bp_ctx* ctx, const fr_t& w, size_t n, fr_t* d_out) {
  if (n == 0) return BP_OK;
  return make_table(ctx, w, (uint32_t)n, 0, nullptr, d_out, nullptr);
}

}  // namespace bp
