// msm.hip -- host driver of the G1 MSM pipeline (kernels in msm_kernels.hpp).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "ctx.hpp"
#include "msm_kernels.hpp"

namespace bp {

static uint32_t ilog2_floor(size_t n) {
  uint32_t l = 0;
  while ((n >> (l + 1)) != 0) l++;
  return l;
}
// radix for table width c with W windows (R = 0: power-of-two windows); cached per width
static const MsmRadix& msm_radix_for(uint32_t c, uint32_t W) {
  static MsmRadix cache[MSM_MAX_TABLE_C + 1];
  static bool done[MSM_MAX_TABLE_C + 1];
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!done[c]) {
    done[c] = true;
    // Widths from 21 bits only (2^20 allocated buckets and more).  Fewer live buckets save tree time only where a tree level runs for
    // several wave rounds: at c = 22 (2^21 buckets, 2^24 points) the radix takes 0.3 ms off the 5.3-ms tail and 0.65 ms off the MSM; at
    // c = 20 every wide level of the 2^19-bucket tree is ONE round of two waves per SIMD, a fifth of the workgroups leaving early frees
    // no SIMD earlier, and the division costs msm_part_count 9 us: 2.60-2.64 against 2.63 ms, no gain (profiles/r06_tail_ab.txt).
    // Experiment build: BP_MSM_RADIX=0 keeps power-of-two windows everywhere, BP_MSM_RADIX_FROM moves the threshold.
    if (knob_u32("BP_MSM_RADIX", 1, 0, 1) != 0 && c >= knob_u32("BP_MSM_RADIX_FROM", 21, 8, 30)) cache[c] = msm_radix_compute(c, W);
  }
  return cache[c];
}
// live buckets of table width c (the prefix of the 2^(c-1) allocated ones that digits can reach)
uint32_t msm_table_radix(uint32_t c, uint32_t W) { return msm_radix_for(c, W).R; }

// Window width: minimise W * (n + 2 * 2^(c-1)) (bucket adds + reduction adds).  Per-window bucket sets (any point set) are
// bounded by the 128 KiB LDS histogram of one window (c <= 16); with fixed-base tables wider windows go through the
// partitioned sort (plan.parts > 1), up to MSM_MAX_TABLE_C.
// table_c != 0: the SRS carries fixed-base window tables built for that width (row length table_stride)
static void make_plan(MsmPlan& plan, size_t n, uint32_t table_c, size_t table_stride, uint32_t J = 1) {
  memset(&plan, 0, sizeof plan);
  plan.n = (uint32_t)n;
  plan.J = J;
  plan.nj[0] = (uint32_t)n;
  if (table_c & MSM_NAF_FLAG) {         // every-position tables: odd NAF digits of width w, 2^(w-2) buckets, one bucket set
    const uint32_t w = table_c & 0xffu;
    plan.naf = w;
    plan.c = w - 1;
    plan.W = 255 / w + 1;               // digit slots per scalar (most scalars fill 256 / (w + 1) of them)
    plan.B = 1u << (w - 2);
    plan.total = J * plan.B;
    plan.wbuckets = 0;
    plan.wpoints = (uint32_t)table_stride;
    plan.parts = plan.B <= (1u << MSM_HIST_LOG) ? 1 : plan.B >> MSM_HIST_LOG;
    // lanes: one wave round (131 072) for the worst case of every slot filled; the kernels cut the sorted list by the actual
    // entry count (a uniform scalar fills 256 / (w + 1) + ~0.45 of its slots: 15.49 / 13.91 / 12.71 at w = 16 / 18 / 20)
    uint32_t chunk = (uint32_t)(((uint64_t)plan.W * n * J + 131071) / 131072);
    if (chunk < 4) chunk = 4;
    if (chunk > 1024) chunk = 1024;
    plan.chunk = knob_u32("BP_MSM_CHUNK", chunk, 1, 1024);
    plan.lanes = knob_u32("BP_MSM_CHUNK", 0, 1, 1024) ? 0u : (uint32_t)(((uint64_t)plan.W * n * J + plan.chunk - 1) / plan.chunk);
    plan.slices = 1;
    plan.seg = 1;
    return;
  }
  uint32_t c = n < 32 ? 4 : ilog2_floor(n) - 3;
  if (c < 4) c = 4;
  if (c > MSM_MAX_C) c = MSM_MAX_C;
  c = knob_u32("BP_MSM_C", c, 2, MSM_MAX_C);
  if (c > (uint32_t)MSM_MAX_C) c = MSM_MAX_C;          // per-window bucket sets: one LDS histogram per window
  if (table_c) c = table_c;                          // tables: up to MSM_MAX_TABLE_C through the partitioned sort
  if (c < 2) c = 2;
  // digits d_w = ((k + bias) >> c*w & mask) - 2^(c-1) with bias = sum_w 2^(c-1) 2^(cw); needs k + bias < 2^(cW)
  // for every k < q.  Take W = ceil(256 / c) and check the bound with the real q; add a window if it fails.
  static const uint32_t q_minus_1[8] = {0x00000000u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  // Windows 0..W-2 are signed: d_w = ((k + bias) >> cw & mask) - 2^(c-1), bias = sum_{w<W-1} 2^(c-1) 2^(cw).
  // The top window is unsigned (digit = remaining high bits, carry included) and must fit a bucket index,
  // i.e. be <= 2^(c-1) for every k < q.  Smallest such W:
  uint32_t W = (255 + c - 1) / c;
  if (W < 2) W = 2;
  for (;;) {
    uint32_t bias[10] = {0};
    for (uint32_t w = 0; w + 1 < W; w++) {
      uint32_t bit = c * w + c - 1;
      if (bit < 320) bias[bit >> 5] |= 1u << (bit & 31);
    }
    uint64_t carry = 0;
    uint32_t sum[11] = {0};
    for (int j = 0; j < 10; j++) {
      carry += (uint64_t)(j < 8 ? q_minus_1[j] : 0u) + bias[j];
      sum[j] = (uint32_t)carry;
      carry >>= 32;
    }
    // top digit of the largest scalar: (q - 1 + bias) >> c (W - 1), as an exact 320-bit shift
    const uint32_t o = c * (W - 1);
    bool ok = o < 288;
    uint64_t top = 0;
    for (int bit = 319; bit >= (int)o && ok; bit--) {
      if ((sum[bit >> 5] >> (bit & 31)) & 1) {
        if (bit - (int)o >= 32) ok = false; else top |= 1ull << (bit - o);
      }
    }
    if (ok && top <= (1ull << (c - 1))) {
      memcpy(plan.bias, bias, sizeof plan.bias);
      break;
    }
    W++;
  }
  if (table_c) {                                     // fixed-base tables: the radix need not be a power of two
    const MsmRadix& rx = msm_radix_for(c, W);
    if (rx.R) {
      plan.radix = rx.R;
      memcpy(plan.radix_m, rx.m, sizeof plan.radix_m);
      memcpy(plan.bias, rx.bias, sizeof plan.bias);
    }
  }
  plan.c = c;
  plan.W = W;
  plan.B = 1u << (c - 1);
  plan.total = table_c ? J * plan.B : W * plan.B;
  plan.wbuckets = table_c ? 0 : plan.B;
  plan.wpoints = table_c ? (uint32_t)table_stride : 0;
  plan.parts = plan.B <= (1u << MSM_HIST_LOG) ? 1 : plan.B >> MSM_HIST_LOG;
  const uint64_t entries = (uint64_t)W * n * J;
  // entries per lane: the kernel runs 2 waves per SIMD = 131072 lanes at a time and every lane does the same work, so the
  // lane count should land just under a whole number of such rounds.  (A power-of-two chunk wasted up to a third of the last
  // round whenever windows * n was not a power of two: 13 or 15 windows.)  Measured with tables (r02_chunk_rounds_ab.txt):
  //   up to 2^20 entries (2^16 points): half a round -- chains of 16 leave two partials per bucket instead of eight (0.531 -> 0.482 ms);
  //   up to 2*10^7 entries (2^20 points and the prover's 2^20 + 6): ONE round (2^20: 3.13 -> 3.03 ms, 2^18 1.004 -> 0.979, 2^17 0.723 -> 0.675);
  //   beyond: two rounds (2^21 .. 2^24 are equal or 1-3 % better with two: the second round evens out the lanes' finish times).
  // Per-window buckets (no tables) are short: cap the chunk at 64.
  const uint32_t chunk_cap = table_c ? 1024u : 64u;
  uint32_t chunk;
  if (table_c && entries <= (1u << 20)) chunk = (uint32_t)(entries >> 16);
  else if (table_c && entries <= 20000000u) chunk = (uint32_t)((entries + 131071) / 131072);      // 2^24 and a little more: the prover's SRS has n + 6 points
  else chunk = (uint32_t)((entries + 262143) / 262144);
  if (chunk < 4) chunk = 4;
  if (chunk > chunk_cap) chunk = chunk_cap;
  plan.chunk = knob_u32("BP_MSM_CHUNK", chunk, 1, 1024);
  plan.lanes = knob_u32("BP_MSM_CHUNK", 0, 1, 1024) ? 0u : (uint32_t)((entries + plan.chunk - 1) / plan.chunk);      // = the host's n_chunks
  // count/scatter workgroups per window: each flushes its whole LDS histogram with global atomics, so fewer, fatter
  // slices are cheaper (~32 Ki points each) as long as >= 256 workgroups remain to fill the CUs (measured: 2^16, 2^20, 2^24)
  uint32_t slices = (uint32_t)(n >> 15), lo = 256 / W, hi = 1024 / W;
  if (slices < lo) slices = lo;
  if (slices > hi) slices = hi;
  if (slices < 1) slices = 1;
  while (slices > 1 && n / slices < 1024) slices >>= 1;
  plan.slices = knob_u32("BP_MSM_SLICES", slices, 1, 4096);
  uint32_t seg = 1;
  while (seg < 32 && plan.total / seg > 65536) seg <<= 1;
  plan.seg = knob_u32("BP_MSM_SEG", seg, 1, 1024);
}

// unsaturated copy of an SRS (128-byte slots) (allocated here, owned by the SRS entry)
int srs_to28_run(bp_ctx* ctx, const g1_affine* d_in, size_t n, g1_affine28** d_out) {
  g1_affine28* d = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&d, (n ? n : 1) * sizeof(g1_affine28)));
  if (n) hipLaunchKernelGGL(srs_to28, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, d);
  BP_HIP(ctx, hipGetLastError());
  *d_out = d;
  return BP_OK;
}

int srs_to28_into(bp_ctx* ctx, const g1_affine* d_in, size_t n, g1_affine28* d_out) {
  if (n) hipLaunchKernelGGL(srs_to28, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, d_out);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

uint32_t msm_table_windows(uint32_t c) {
  if (c & MSM_NAF_FLAG) return MSM_NAF_ROWS;
  MsmPlan plan;
  make_plan(plan, 1, c, 1);
  return plan.W;
}

// table[w * n + i] = 2^(c w) P_i; row 0 is the unsaturated SRS copy itself
uint32_t srs_table_rows(uint32_t c) { return msm_table_windows(c); }

int srs_tables_run(bp_ctx* ctx, const g1_affine* d_points, const g1_affine28* d_points28, size_t n, uint32_t c, g1_affine28** d_table,
                   uint32_t* windows) {
  const uint32_t W = msm_table_windows(c);
  const uint32_t radix = (c & MSM_NAF_FLAG) ? 0u : msm_table_radix(c, W);       // rows R^w P_i where the MSM cuts radix-R digits (MsmPlan::radix)
  if ((uint64_t)W * n >= (1ull << 31))
    return fail(ctx, BP_ERR_TOO_LARGE, "fixed-base tables: windows * points >= 2^31", hipSuccess, __FILE__, __LINE__);
  g1_affine28* t = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&t, (size_t)W * (n ? n : 1) * sizeof(g1_affine28)));
  if (n) {
    BP_HIP(ctx, hipMemcpyAsync(t, d_points28, n * sizeof(g1_affine28), hipMemcpyDeviceToDevice, ctx->stream));
    if (radix) hipLaunchKernelGGL(srs_window_tables<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_points28, n, c, W, radix, t);
    else hipLaunchKernelGGL(srs_window_tables<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_points28, n, (c & MSM_NAF_FLAG) ? 1u : c, W, 0u, t);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = stream_wait(ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(t);
    return fail(ctx, BP_ERR_HIP, "srs_window_tables", e, __FILE__, __LINE__);
  }
  *d_table = t;
  *windows = W;
  return BP_OK;
}

// accumulator slot (lazy 28-bit limbs, as the kernels leave it) -> the reference's Montgomery limbs
static g1_proj slot_to_proj(const proj28_slot* slot) {
  g1_proj28 p;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(slot);
  for (int j = 0; j < N28; j++) { p.x.l[j] = src[j]; p.y.l[j] = src[N28 + j]; p.z.l[j] = src[2 * N28 + j]; }
  return g1_proj_from_28(p);
}

// dynamic-LDS limits are per function AND per device: set them for every context at creation (bp_init), after hipSetDevice
int msm_init_device(bp_ctx* ctx) {
  // a full-window histogram at c = 16 needs 128 KiB of dynamic LDS
#ifdef BP_EXPERIMENT
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_count, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
  // these two also hold a few KiB of static LDS: the dynamic limit must leave room for it
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_final<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_final<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_count<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_count<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_scatter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_scatter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_part_scatter<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_part_scatter<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_part_scatter<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_part_scatter<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  return BP_OK;
}

// d_points28: the unsaturated SRS copy, or (table_c != 0) row 0 of its fixed-base tables with rows table_stride apart.
// Enqueues the whole pipeline and the device-to-host copy of the window sums on ctx->stream; waits for nothing.
int msm_launch(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
               int slot, void* d_blob, MsmPending* out) {
  return msm_launch_many(ctx, d_points28, 1, &d_scalars, &n, fmt, table_c, table_stride, slot, d_blob, out);
}

// J scalar vectors against the same points in one pipeline (J > 1: fixed-base tables only, results in the pinned slot only);
// vector j has n_each[j] scalars.  msm_finish then delivers J results.
int msm_launch_many(bp_ctx* ctx, const g1_affine28* d_points28, uint32_t J, const fr_t* const* d_scalars_each, const size_t* n_each, int fmt,
                    uint32_t table_c, size_t table_stride, int slot, void* d_blob, MsmPending* out) {
  *out = MsmPending();
  out->blob = d_blob != nullptr;
  if (J < 1 || J > MSM_MAX_BATCH || (J > 1 && (!table_c || d_blob)))
    return fail(ctx, BP_ERR_INVALID_ARG, "MSM batch", hipSuccess, __FILE__, __LINE__);
  size_t n = 0;
  for (uint32_t j = 0; j < J; j++) n = n_each[j] > n ? n_each[j] : n;
#ifdef BP_EXPERIMENT
  const fr_t* d_scalars = d_scalars_each[0];           // the records-first passes (experiment builds) take a single vector
#endif
  out->J = J;
  if (n == 0) {
    if (d_blob) {
      MsmBlobHeader hdr;
      memset(&hdr, 0, sizeof hdr);
      hdr.magic = MSM_BLOB_MAGIC;
      hipLaunchKernelGGL(msm_write_blob, dim3(1), dim3(64), 0, ctx->stream, (const proj28_slot*)nullptr, hdr, (uint8_t*)d_blob);
      BP_HIP(ctx, hipGetLastError());
    }
    return BP_OK;
  }
  if (slot < 0 || slot >= MSM_SLOTS) return fail(ctx, BP_ERR_INVALID_ARG, "MSM result slot", hipSuccess, __FILE__, __LINE__);
  if (n >= (1ull << 31)) return fail(ctx, BP_ERR_TOO_LARGE, "MSM length >= 2^31", hipSuccess, __FILE__, __LINE__);
  MsmPlan plan;
  make_plan(plan, n, table_c, table_stride, J);
  MsmScalars scalars_all;
  for (uint32_t j = 0; j < MSM_MAX_BATCH; j++) {
    plan.nj[j] = j < J ? (uint32_t)n_each[j] : 0u;
    scalars_all.p[j] = j < J ? d_scalars_each[j] : nullptr;
  }
  if ((uint64_t)plan.W * n * J >= (1ull << 32))        // positions in the bucket-sorted list are 32-bit
    return fail(ctx, BP_ERR_TOO_LARGE, "MSM length * windows >= 2^32", hipSuccess, __FILE__, __LINE__);
  const uint32_t W = plan.W, B = plan.B, total = plan.total, Wr = table_c ? J : W;   // Wr: bucket sets left after accumulation
  const uint64_t max_entries = (uint64_t)W * n * J;
  const uint64_t n_chunks = (max_entries + plan.chunk - 1) / plan.chunk;
  // bucket reduction: the bit-plane tree over one bucket set (tables) or over the forest of W bucket sets (table-free); the running sums per
  // window of rounds 1-4 (msm_reduce: blocks_per_window, plan.seg) exist in the experiment build only (BP_MSM_REDUCE=1)
  const uint32_t blocks_per_window = (!EXPERIMENT_BUILD || table_c) ? 0 : ((B + plan.seg - 1) / plan.seg + 255) / 256;
  // Bucket reduction without tables (W bucket sets of 2^(c-1) buckets): the same bit-plane tree as with tables, over a forest of W trees
  // (round 5; rounds 1-4 ran segmented running sums per window, msm_reduce: a dependent chain of ~46 additions per lane, 256 VGPRs + spills,
  // 0.57 ms at 2^20 points -- kept in the experiment build as BP_MSM_REDUCE=1 for the A/B), then two levels of the Horner form over each
  // tree's c values (msm_planes_window_quads): `quads` values per window go to the host.
  const bool reduce_running = EXPERIMENT_BUILD && !table_c && knob_u32("BP_MSM_REDUCE", 0, 0, 1) == 1;
  const uint32_t quads = (table_c || reduce_running) ? 1u : (plan.c + 2) / 4;          // ceil((c - 1) / 4); c = 2: one quad (A + T_0)
  const uint32_t per_window = table_c ? plan.c : quads;     // slots per window that go to the host

  uint32_t *counts, *offsets, *cursors, *sorted;
  proj28_slot *bucket_sum, *partial, *block_out, *window_sum;
  // a bucket is "long" when it spans >= FIXUP_LONG chunks, so at most n_chunks / FIXUP_LONG + 1 buckets can be long
  const uint32_t long_cap = (uint32_t)(n_chunks / FIXUP_LONG + 1);
  // control words, zeroed by ONE memset per MSM: [0] long-bucket counter, [1] scalar status, [2] long-run counter of the sort,
  // [4] ticket + [5 ..] partition sizes of msm_part_count, then one ticket per queued long bucket (msm_fixup_long)
  // ... then one ticket per long RUN of the sort (msm_radix_long_count; at most 2^16 final runs).  The memset covers the words in use.
  uint32_t* ctl;
  const size_t ctl_fixed = 8 + PART_MAX + long_cap;
  BP_TRY(ws_get(ctx, "msm.ctl", (ctl_fixed + 65536) * 4, (void**)&ctl));
  uint32_t* run_ticket = ctl + ctl_fixed;
  BP_TRY(ws_get(ctx, "msm.counts", (size_t)total * 4, (void**)&counts));        // bucket sizes (histogram sort; long runs of the other two)
  BP_TRY(ws_get(ctx, "msm.offsets", ((size_t)total + 1) * 4, (void**)&offsets));
  BP_TRY(ws_get(ctx, "msm.cursors", (size_t)total * 4, (void**)&cursors));
  uint32_t *long_count = ctl, *long_list, *long_ticket = ctl + 8 + PART_MAX;
  BP_TRY(ws_get(ctx, "msm.long_list", (size_t)long_cap * 4, (void**)&long_list));
  // slice sums of the buckets that take several workgroups: those have > FIXUP_LONG_SPLIT_FROM partials, so there are few of them, but the
  // slot is addressed by the bucket's place in the list
  proj28_slot* long_scratch;
  BP_TRY(ws_get(ctx, "msm.long_scratch", (size_t)long_cap * FIXUP_LONG_SLICES * sizeof(proj28_slot), (void**)&long_scratch));
  uint32_t* tile_sums;
  BP_TRY(ws_get(ctx, "msm.tile_sums", 4096 * 4, (void**)&tile_sums));
  BP_TRY(ws_get(ctx, "msm.sorted", max_entries * 4, (void**)&sorted));
  BP_TRY(ws_get(ctx, "msm.bucket_sum", (size_t)total * sizeof(proj28_slot), (void**)&bucket_sum));
  BP_TRY(ws_get(ctx, "msm.partial", 2 * n_chunks * sizeof(proj28_slot), (void**)&partial));
  const uint32_t n_planes = Wr * per_window;        // tables: A and the c - 1 bit planes; else one sum per window
  if (n_planes > (uint32_t)MSM_MAX_WINDOWS) return fail(ctx, BP_ERR_TOO_LARGE, "MSM windows", hipSuccess, __FILE__, __LINE__);
  block_out = nullptr;
  if (reduce_running) BP_TRY(ws_get(ctx, "msm.block_out", (size_t)Wr * blocks_per_window * sizeof(proj28_slot), (void**)&block_out));
  proj28_slot* roots = nullptr;                     // table-free: the W roots (A, T_0 .. T_{c-2}) in front of the per-window Horner pass
  if (!table_c && !reduce_running) BP_TRY(ws_get(ctx, "msm.roots", (size_t)W * plan.c * sizeof(proj28_slot), (void**)&roots));
  BP_TRY(ws_get(ctx, "msm.window_sum", (size_t)n_planes * sizeof(proj28_slot) + 16, (void**)&window_sum));     // + the status word
  // pinned staging: MSM_SLOTS result areas of the largest possible size, so earlier pending results stay where they are
  constexpr size_t slot_bytes = (size_t)MSM_MAX_WINDOWS * sizeof(proj28_slot) + 16;
  uint8_t* h_base;
  BP_TRY(pinned_get(ctx, MSM_SLOTS * slot_bytes, (void**)&h_base));
  proj28_slot* h_windows = reinterpret_cast<proj28_slot*>(h_base + (size_t)slot * slot_bytes);
  hipStream_t st = ctx->stream;
  BP_HIP(ctx, hipEventRecord(ctx->ev[0], st));
  // Bucket sort, three builds: the partition sort (default wherever its 2^pb <= 2^11 partitions leave final runs
  // that one workgroup sorts: up to ~5 * 10^7 entries), the two-level radix sort (beyond), the one-histogram counting sort of
  // round 1 (c <= 16 only; kept selectable for A/B: BP_MSM_SORT=0 histogram, 1 two-level, 2 partition).
  uint32_t kb = 0;
  while ((1ull << kb) < total) kb++;
  uint32_t pb = 0;
  // final runs of ~12-24 Ki entries; per-window bucket sets (no tables) take runs of half that length: 2^9 buckets of 16 entries per run sort
  // faster than 2^9 buckets of 32 (tail 0.84 -> 0.80 ms at 2^20, 0.83 -> 0.80 at 2^18, 0.63 -> 0.60 at 2^16: tools/part_bits_probe.sh, round 5)
  const uint64_t run_min = table_c ? 12288 : 6144;
  while (pb < kb && (max_entries >> (pb + 1)) >= run_min) pb++;
  pb = knob_u32("BP_MSM_RADIX_BITS", pb, 0, 16);
  if (pb + MSM_HIST_LOG < kb) pb = kb - MSM_HIST_LOG;                   // a final run's buckets must fit one LDS histogram
  if (pb > kb) pb = kb;
  if (pb > 16) pb = 16;
  if (pb > PART_MAX_BITS && (max_entries >> PART_MAX_BITS) <= 32768) pb = PART_MAX_BITS;     // final runs of up to 32 Ki entries are fine for one workgroup
  if (pb + MSM_HIST_LOG < kb) pb = kb - MSM_HIST_LOG;
  // record form of the partition sort: the packed word holds the bucket's low kb - pb bits, the sign and the entry (point or
  // table index, vb - 1 bits).  Where a few more partitions make the record fit one word (c = 20 at 2^20: 2^12 instead of 2^10)
  // they are taken: half the bytes through the scatter and the final sort (BP_MSM_PACKED=2 keeps the run length instead).
  const uint64_t idx_max = (uint64_t)(n - 1) + (uint64_t)(plan.naf ? 255u : W - 1) * plan.wpoints;
  uint32_t vb = 1;
  while ((idx_max >> (vb - 1)) != 0) vb++;                              // vb - 1 = bits of the largest entry, + 1 for the sign
  const uint32_t packed_env = knob_u32("BP_MSM_PACKED", 1, 0, 2);
  if (packed_env == 1 && kb + vb > 32 + pb && kb + vb - 32 <= PART_MAX_BITS && knob_u32("BP_MSM_RADIX_BITS", 99, 0, 16) == 99) pb = kb + vb - 32;
  const uint32_t sort_env = knob_u32("BP_MSM_SORT", 2, 0, 3);      // 3: the two-level sort (first level from the scalars) at sizes where the partition sort is the default
  const bool hist_ok = plan.parts == 1 && !plan.naf && J == 1;
  const int sort_mode = (EXPERIMENT_BUILD && sort_env == 0 && hist_ok) ? 0 : ((((sort_env == 1 || sort_env == 3) && J == 1) || pb > PART_MAX_BITS) ? 1 : 2);
  if (sort_mode != 2 && J > 1) return fail(ctx, BP_ERR_TOO_LARGE, "MSM batch too long for the partition sort", hipSuccess, __FILE__, __LINE__);
  const uint32_t rbits = kb - pb, n_final = 1u << pb;
  const size_t rhist = ((size_t)1 << rbits) * 4;
  BP_HIP(ctx, hipMemsetAsync(ctl, 0, (ctl_fixed + n_final) * 4, st));
  uint32_t* rlong_n = ctl + 2;
  if (sort_mode == 2) {
    const bool packed = rbits + vb <= 32 && packed_env != 0;
    const uint32_t rec_bytes = packed ? 4 : 8;
    // scalars per workgroup: the staging area (slice * W records) within 64 KiB, and >= 512 workgroups where n allows
    uint32_t slice = 64;
    while (slice < 1024 && (uint64_t)2 * slice * W * rec_bytes <= 65536 && (uint64_t)slice * 512 < (uint64_t)n * J) slice <<= 1;
    slice = knob_u32("BP_MSM_PART_SLICE", slice, 64, 4096);
    if (slice < 64 || slice > 4096 || (uint64_t)slice * W * rec_bytes > 65536) slice = 64;
    // Two-word records (2^21 .. 2^23 points at c = 20: bucket-low bits + sign + a 25..27-bit table index exceed one word): the 64-KiB rule leaves
    // slices of 512 scalars, and a workgroup's fixed cost for its 2^12-entry tables (~12 us) then outweighs its records (~7 us).  A slice of
    // 1 024 scalars stages 104 KiB; that fits beside the 48 KiB of tables when the per-record partition array of the flat write-out is dropped,
    // i.e. with the partition-major write-out (BP_MSM_PART_WIDE8=0 keeps the 512-scalar slices).
    const bool wide8 = !packed && slice == 512 && (uint64_t)1024 * 512 < (uint64_t)n * J && (size_t)3 * n_final * 4 + (size_t)1024 * W * 8 <= 156 * 1024 &&
                       knob_u32("BP_MSM_PART_WIDE8", 1, 0, 1) != 0;
    if (wide8) slice = 1024;
    const uint32_t cap = slice * W, n_slices = (uint32_t)(((uint64_t)n * J + slice - 1) / slice);
    const unsigned threads = slice >= 1024 ? 1024u : (slice <= 256 ? 256u : slice);
    uint32_t *recs = nullptr, *rvals = nullptr, *roff, *cur, *rlong_list;
    BP_TRY(ws_get(ctx, "msm.rkeys0", max_entries * 4, (void**)&recs));
    if (!packed) BP_TRY(ws_get(ctx, "msm.rvals0", max_entries * 4, (void**)&rvals));
    BP_TRY(ws_get(ctx, "msm.run_off", ((size_t)3 * n_final + 16) * 4, (void**)&roff));
    BP_TRY(ws_get(ctx, "msm.run_cur", (size_t)n_final * 4, (void**)&cur));
    BP_TRY(ws_get(ctx, "msm.run_long", ((size_t)n_final + 2) * 4, (void**)&rlong_list));
    // the count pass keeps nothing per slice, so its slices are its own: fatter ones (~256 workgroups) flush their 2^pb partition
    // sizes with a quarter of the global atomics (BP_MSM_COUNT_SLICE forces a size)
    uint32_t slice1 = slice;
    while (slice1 < 8192 && (uint64_t)slice1 * 256 < (uint64_t)n * J) slice1 <<= 1;      // measured at 2^20: 48 / 41 / 34 / 48 us at 1 024 / 2 048 / 4 096 / 8 192
    slice1 = knob_u32("BP_MSM_COUNT_SLICE", slice1, 64, 65536);
    const uint32_t n_slices1 = (uint32_t)(((uint64_t)n * J + slice1 - 1) / slice1);
    hipLaunchKernelGGL(msm_part_count, dim3(n_slices1), dim3(slice1 >= 1024 ? 1024u : threads), 0, st, scalars_all, fmt, plan, slice1, pb, rbits, ctl + 4, roff, cur, long_count + 1);
    // a slice's share of a partition: long -> partition-major write-out, short -> one lane per record (BP_MSM_PART_FLAT = 0 / 1 forces)
    const uint32_t flat_env = knob_u32("BP_MSM_PART_FLAT", 2, 0, 2);
    const bool flat = wide8 ? false : (flat_env == 2 ? (cap >> pb) < 8 : flat_env == 1);
    const size_t part_lds = (size_t)3 * n_final * 4 + (size_t)cap * rec_bytes + (flat ? (size_t)cap * 2 : 0);
    const dim3 lgrid(64, n_final < 4 ? n_final : 4);
    // short final runs (2^12 runs of ~3 Ki records at c = 20): 512-lane workgroups (measured 70 / 60 / 72 us at 256 / 512 / 1024 lanes)
    const unsigned final_threads = knob_u32("BP_MSM_FINAL_THREADS", (max_entries >> pb) <= 8192 ? 512 : 1024, 256, 1024) & ~63u;
    if (packed) {
      if (flat) hipLaunchKernelGGL((msm_part_scatter<true, true>), dim3(n_slices), dim3(threads), part_lds, st, scalars_all, fmt, plan, slice, pb, rbits, vb, cap, cur, recs, rvals);
      else hipLaunchKernelGGL((msm_part_scatter<true, false>), dim3(n_slices), dim3(threads), part_lds, st, scalars_all, fmt, plan, slice, pb, rbits, vb, cap, cur, recs, rvals);
      const RunRecords<true> rr{recs, nullptr, (1u << rbits) - 1u, vb};
      hipLaunchKernelGGL(msm_radix_final<true>, dim3(n_final < 4096 ? n_final : 4096), dim3(final_threads), rhist, st, rr, roff, n_final, rbits, total, offsets,
                         sorted, rlong_n, rlong_list, counts);
      if (n_final > 1) {
        hipLaunchKernelGGL(msm_radix_long_count<true>, lgrid, dim3(1024), rhist, st, rr, roff, n_final, rbits, total, rlong_n, rlong_list, counts, offsets,
                           cursors, run_ticket);
        hipLaunchKernelGGL(msm_radix_long_scatter<true>, lgrid, dim3(1024), rhist, st, rr, roff, rbits, rlong_n, rlong_list, cursors, sorted);
      }
    } else {
      if (flat) hipLaunchKernelGGL((msm_part_scatter<false, true>), dim3(n_slices), dim3(threads), part_lds, st, scalars_all, fmt, plan, slice, pb, rbits, vb, cap, cur, recs, rvals);
      else hipLaunchKernelGGL((msm_part_scatter<false, false>), dim3(n_slices), dim3(threads), part_lds, st, scalars_all, fmt, plan, slice, pb, rbits, vb, cap, cur, recs, rvals);
      const RunRecords<false> rr{recs, rvals, (1u << rbits) - 1u, 0u};
      hipLaunchKernelGGL(msm_radix_final<false>, dim3(n_final < 4096 ? n_final : 4096), dim3(final_threads), rhist, st, rr, roff, n_final, rbits, total, offsets,
                         sorted, rlong_n, rlong_list, counts);
      if (n_final > 1) {
        hipLaunchKernelGGL(msm_radix_long_count<false>, lgrid, dim3(1024), rhist, st, rr, roff, n_final, rbits, total, rlong_n, rlong_list, counts, offsets,
                           cursors, run_ticket);
        hipLaunchKernelGGL(msm_radix_long_scatter<false>, lgrid, dim3(1024), rhist, st, rr, roff, rbits, rlong_n, rlong_list, cursors, sorted);
      }
    }
  } else if (sort_mode == 1) {
    const uint32_t lv[2] = {pb <= 8 ? pb : pb - pb / 2, pb <= 8 ? 0 : pb / 2};
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *run_off[3], *cnt, *cur;
    BP_TRY(ws_get(ctx, "msm.rkeys0", max_entries * 4, (void**)&keys[0]));
    BP_TRY(ws_get(ctx, "msm.rvals0", max_entries * 4, (void**)&vals[0]));
    if (pb) {
      BP_TRY(ws_get(ctx, "msm.rkeys1", max_entries * 4, (void**)&keys[1]));
      BP_TRY(ws_get(ctx, "msm.rvals1", max_entries * 4, (void**)&vals[1]));
    }
    uint32_t* roff;
    BP_TRY(ws_get(ctx, "msm.run_off", ((size_t)3 * n_final + 16) * 4, (void**)&roff));
    run_off[0] = roff;                                   // {0, W n}
    run_off[1] = roff + 4;                               // after level 1: 2^lv[0] + 1 entries
    run_off[2] = roff + 8 + n_final;                     // after level 2: n_final + 1 entries
    BP_TRY(ws_get(ctx, "msm.run_cnt", (size_t)n_final * 4, (void**)&cnt));
    BP_TRY(ws_get(ctx, "msm.run_cur", (size_t)n_final * 4, (void**)&cur));
    const uint32_t whole[2] = {0u, (uint32_t)max_entries};
    BP_HIP(ctx, hipMemcpyAsync(run_off[0], whole, sizeof whole, hipMemcpyHostToDevice, st));
    uint32_t* rlong_list;
    BP_TRY(ws_get(ctx, "msm.run_long", ((size_t)n_final + 2) * 4, (void**)&rlong_list));
    // Level 1 of the two-level sort, two builds.  Straight from the scalars, as the partition sort does it (msm_part_count + msm_part_scatter
    // with two-word records into 2^lv[0] runs: no record pass, one array's traffic less -- the default since round 4: 2^24 points, tail
    // 5.87 -> 5.32 ms, profiles/r04_sort24_hybrid_ab.txt), or records first (msm_digit_records writes one (bucket, entry) pair per scalar and
    // window, then msm_radix_count / _scatter partition them: a single level, NAF digits, or BP_MSM_SORT=1 in the experiment build).
    uint32_t runs = 1, shift = kb, side = 0;
    int first_level = 0;
    const bool hybrid = sort_env != 1 && lv[0] && lv[1] && !plan.naf;
    if (hybrid) {
      const uint32_t pb1 = lv[0], rb1 = kb - pb1;
      uint32_t slice = 64;
      while (slice < 1024 && (uint64_t)2 * slice * W * 8 <= 65536 && (uint64_t)slice * 512 < (uint64_t)n) slice <<= 1;
      const uint32_t cap = slice * W, n_slices = (uint32_t)(((uint64_t)n + slice - 1) / slice);
      const unsigned threads = slice >= 1024 ? 1024u : (slice <= 256 ? 256u : slice);
      uint32_t slice1 = slice;
      while (slice1 < 8192 && (uint64_t)slice1 * 256 < (uint64_t)n) slice1 <<= 1;
      const uint32_t n_slices1 = (uint32_t)(((uint64_t)n + slice1 - 1) / slice1);
      hipLaunchKernelGGL(msm_part_count, dim3(n_slices1), dim3(slice1 >= 1024 ? 1024u : threads), 0, st, scalars_all, fmt, plan, slice1, pb1, rb1, ctl + 4, run_off[1], cur,
                         long_count + 1);
      const size_t part_lds = (size_t)3 * (1u << pb1) * 4 + (size_t)cap * 8;
      hipLaunchKernelGGL((msm_part_scatter<false, false>), dim3(n_slices), dim3(threads), part_lds, st, scalars_all, fmt, plan, slice, pb1, rb1, 0u, cap, cur, keys[1],
                         vals[1]);
      runs = 1u << pb1;
      shift = rb1;
      side = 1;
      first_level = 1;
    } else {
#ifdef BP_EXPERIMENT
    if (plan.naf)
      hipLaunchKernelGGL(msm_naf_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, keys[0], vals[0], long_count + 1);
    else
      hipLaunchKernelGGL(msm_digit_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, keys[0], vals[0], long_count + 1);
#else
    return fail(ctx, BP_ERR_INVALID_ARG, "records-first sort is an experiment build", hipSuccess, __FILE__, __LINE__);      // unreachable: two levels always take level 1 from the scalars
#endif
    }
    for (int level = first_level; level < 2 && lv[level]; level++) {
      const uint32_t bits = lv[level], nd = 1u << bits, n_sub = runs * nd;
      shift -= bits;
      uint64_t per_run = max_entries / runs;
      uint32_t gx = (uint32_t)((per_run + RADIX_SLICE - 1) / RADIX_SLICE) + (runs > 1 ? 1 : 0);
      if (gx > 4096) gx = 4096;
      const uint32_t t = (n_sub + SCAN_TILE - 1) / SCAN_TILE;
      BP_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)n_sub * 4, st));
      hipLaunchKernelGGL(msm_radix_count, dim3(gx, runs), dim3(1024), 0, st, keys[side], run_off[level], shift, bits, cnt);
      hipLaunchKernelGGL(scan_tile_sums, dim3(t), dim3(256), 0, st, cnt, n_sub, tile_sums);
      hipLaunchKernelGGL(scan_block_sums, dim3(1), dim3(256), 0, st, tile_sums, t, run_off[level + 1] + n_sub);
      hipLaunchKernelGGL(scan_apply, dim3(t), dim3(256), 0, st, cnt, n_sub, tile_sums, run_off[level + 1], cur);
      hipLaunchKernelGGL(msm_radix_scatter, dim3(gx, runs), dim3(1024), RADIX_SLICE * 9, st, keys[side], vals[side], run_off[level], shift, bits,
                         cur, keys[side ^ 1], vals[side ^ 1]);
      side ^= 1;
      runs = n_sub;
    }
    const uint32_t level_count = (lv[0] ? 1 : 0) + (lv[1] ? 1 : 0);
    const RunRecords<false> rr{keys[side], vals[side], (1u << rbits) - 1u, 0u};
    hipLaunchKernelGGL(msm_radix_final<false>, dim3(runs < 4096 ? runs : 4096), dim3(1024), rhist, st, rr, run_off[level_count], runs, rbits, total, offsets,
                       sorted, rlong_n, rlong_list, counts);
    if (runs > 1) {                 // long runs (none for uniformly random scalars beyond the top window's): slice-parallel
      const dim3 lgrid(256, runs < 16 ? runs : 16);
      hipLaunchKernelGGL(msm_radix_long_count<false>, lgrid, dim3(1024), rhist, st, rr, run_off[level_count], runs, rbits, total, rlong_n, rlong_list, counts,
                         offsets, cursors, run_ticket);
      hipLaunchKernelGGL(msm_radix_long_scatter<false>, lgrid, dim3(1024), rhist, st, rr, run_off[level_count], rbits, rlong_n, rlong_list, cursors, sorted);
    }
  } else {
#ifdef BP_EXPERIMENT
    int16_t* digits = nullptr;
  const size_t hist_bytes = (size_t)B * 4;
  const unsigned hist_threads = B >= 4096 ? 1024 : 256;   // a big histogram owns the CU's LDS: fill the CU with one workgroup
  const uint32_t n_tiles = (total + SCAN_TILE - 1) / SCAN_TILE;      // <= 4096 (total <= 2^24)
    BP_TRY(ws_get(ctx, "msm.digits", max_entries * sizeof(int16_t), (void**)&digits));
    BP_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)total * 4, st));
    hipLaunchKernelGGL(msm_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, digits, long_count + 1);
    hipLaunchKernelGGL(msm_count, dim3(plan.slices, W), dim3(hist_threads), hist_bytes, st, digits, plan, counts);
    hipLaunchKernelGGL(scan_tile_sums, dim3(n_tiles), dim3(256), 0, st, counts, total, tile_sums);
    hipLaunchKernelGGL(scan_block_sums, dim3(1), dim3(256), 0, st, tile_sums, n_tiles, offsets + total);
    hipLaunchKernelGGL(scan_apply, dim3(n_tiles), dim3(256), 0, st, counts, total, tile_sums, offsets, cursors);
    hipLaunchKernelGGL(msm_scatter, dim3(plan.slices, W), dim3(hist_threads), hist_bytes, st, digits, plan, cursors, sorted);
#else
    return fail(ctx, BP_ERR_INVALID_ARG, "counting sort is an experiment build", hipSuccess, __FILE__, __LINE__);      // unreachable: sort_mode 0 needs a knob
#endif
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[1], st));
  const dim3 acc_grid((unsigned)((n_chunks + 255) / 256));
  hipStream_t ast = st;
  if (ctx->acc_stream) {                  // experiment (BP_ACC_LOW_PRIORITY): the accumulation on its low-priority stream, chained by events
    ast = ctx->acc_stream;
    BP_HIP(ctx, hipEventRecord(ctx->acc_ev[0], st));
    BP_HIP(ctx, hipStreamWaitEvent(ast, ctx->acc_ev[0], 0));
  }
  switch (knob_u32("BP_MSM_ACC_WAVES", 2, 2, 4)) {
#ifdef BP_EXPERIMENT          // three / four waves per SIMD: 168 / 128 VGPRs, both spill (tools/kernel_resources.py) and measured slower
    case 3: hipLaunchKernelGGL(msm_accumulate<3>, acc_grid, dim3(256), 0, ast, d_points28, sorted, offsets, plan, bucket_sum, partial); break;
    case 4: hipLaunchKernelGGL(msm_accumulate<4>, acc_grid, dim3(256), 0, ast, d_points28, sorted, offsets, plan, bucket_sum, partial); break;
#endif
    default: hipLaunchKernelGGL(msm_accumulate<2>, acc_grid, dim3(256), 0, ast, d_points28, sorted, offsets, plan, bucket_sum, partial);
  }
  if (ctx->acc_stream) {
    BP_HIP(ctx, hipEventRecord(ctx->acc_ev[1], ast));
    BP_HIP(ctx, hipStreamWaitEvent(st, ctx->acc_ev[1], 0));
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[2], st));
  // fix-up of the buckets cut by chunk edges.  Long buckets (tables at c <= 17: every one of the 2^15 buckets spans several chunks):
  // two lanes per bucket = one wave per SIMD, half the chain.  Short buckets (wide windows, per-window bucket sets): most buckets
  // sit inside one chunk -- one lane per chunk edge.  BP_MSM_FIXUP=1 / 2 forces the per-bucket / per-edge form.
  const uint32_t fixup_env = knob_u32("BP_MSM_FIXUP", 0, 0, 2);
  const bool by_edges = fixup_env ? fixup_env == 2 : max_entries / total < 2ull * plan.chunk;
  if (by_edges)
    hipLaunchKernelGGL(msm_fixup_edges, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st, offsets, plan, bucket_sum, partial, long_count,
                       long_list, long_cap);
  else if (table_c)
    hipLaunchKernelGGL(msm_fixup<2>, dim3((unsigned)(((uint64_t)total * 2 + 255) / 256)), dim3(256), 0, st, offsets, plan, bucket_sum, partial,
                       long_count, long_list, long_cap);
  else
    hipLaunchKernelGGL(msm_fixup<1>, dim3((total + 255) / 256), dim3(256), 0, st, offsets, plan, bucket_sum, partial, long_count, long_list,
                       long_cap);
  hipLaunchKernelGGL(msm_fixup_long, dim3(512), dim3(256), 256 * sizeof(proj28_slot), st, offsets, plan, bucket_sum, partial,
                     long_count, long_list, long_cap, long_scratch, long_ticket);
  if (!reduce_running) {
    // The tree over the `total` leaves (tables: B = 2^(c-1) buckets, J B for a batch; table-free: the forest of W trees, W B leaves),
    // level by level.  A level with at least PLANES_WIDE_MIN additions is throughput bound: one addition per lane through HBM
    // (msm_planes_level01 / msm_planes_level: six such levels at 2^19 leaves, seven for the table-free forest of 16 x 2^15).  The rest
    // is latency bound: up to PLANES_STEP_LOG levels per launch inside workgroups, cooperative additions (msm_planes_step).
    // A node of 2^k buckets carries k + 1 values; the two ping-pong buffers hold at most B values (level 1).
    const uint32_t levels = plan.c - 1;               // >= 1 (make_plan keeps c >= 2)
    const uint64_t wide_min = knob_u32("BP_MSM_PLANES_WIDE_MIN", 24000, 256, 1u << 30);
    uint32_t k = 0, nodes = total, n_wide = 0;            // total = J B leaves: a forest of J trees (J > 1: the vectors of a batch)
    while (n_wide < levels && (uint64_t)(total >> (n_wide + 1)) * (n_wide + 1) >= wide_min) n_wide++;
    if (n_wide == levels) n_wide = levels - 1;            // the last level is always a step: it writes the result where the epilogue reads it
    // levels 0 and 1 have the same number of additions, so a wide level 0 comes with a wide level 1 and the two run fused (msm_planes_level01);
    // a lone wide leaf level (two-level trees, c = 3: knobs only) takes the step path in the shipped library -- its single-level LEAF kernel
    // is the one tree kernel that spills (tools/kernel_resources.py) and lives in the experiment build only (BP_MSM_PLANES_FUSE01=0)
    if (!EXPERIMENT_BUILD && n_wide < 2) n_wide = 0;
    proj28_slot* tmp[2] = {nullptr, nullptr};
    {
      size_t need[2] = {0, 0};
      uint32_t kk = 0, flip = 0;
      while (kk < levels) {
        const uint32_t m = kk < n_wide ? 1 : (levels - kk < PLANES_STEP_LOG ? levels - kk : PLANES_STEP_LOG);
        kk += m;
        if (kk < levels) {
          const size_t slots = (size_t)(total >> kk) * (kk + 1);
          if (slots > need[flip]) need[flip] = slots;
        }
        flip ^= 1;
      }
      if (need[0]) BP_TRY(ws_get(ctx, "msm.planes_tmp0", need[0] * sizeof(proj28_slot), (void**)&tmp[0]));
      if (need[1]) BP_TRY(ws_get(ctx, "msm.planes_tmp1", need[1] * sizeof(proj28_slot), (void**)&tmp[1]));
    }
    const proj28_slot* in = bucket_sum;
    int flip = 0;
    const bool fuse01 = n_wide >= 2 && knob_u32("BP_MSM_PLANES_FUSE01", 1, 0, 1) != 0;
    while (k < levels) {
      const bool leaf = k == 0;
      if (leaf && fuse01) {                 // levels 0 and 1 in one launch, four buckets per lane
        nodes >>= 2;
        flip ^= 1;                          // the output takes the buffer level 1 would have taken
        hipLaunchKernelGGL(msm_planes_level01, dim3((nodes + 255) / 256), dim3(256), 0, st, offsets, in, nodes, tmp[flip]);
        in = tmp[flip];
        k = 2;
      } else if (k < n_wide) {
        nodes >>= 1;
        const uint64_t items = (uint64_t)nodes * (k + 1);
        const dim3 grid((unsigned)((items + 255) / 256));
#ifdef BP_EXPERIMENT
        if (leaf) hipLaunchKernelGGL(msm_planes_level<true>, grid, dim3(256), 0, st, offsets, in, k, nodes, tmp[flip]);
        else
#endif
        hipLaunchKernelGGL(msm_planes_level<false>, grid, dim3(256), 0, st, offsets, in, k, nodes, tmp[flip]);
        in = tmp[flip];
        k += 1;
      } else {
        const uint32_t m = levels - k < PLANES_STEP_LOG ? levels - k : PLANES_STEP_LOG;
        nodes >>= m;
        const bool last = k + m == levels;
        proj28_slot* out_nodes = last ? (table_c ? window_sum : roots) : tmp[flip];
        uint32_t* status_out = last && table_c ? reinterpret_cast<uint32_t*>(window_sum + n_planes) : (uint32_t*)nullptr;
        const dim3 grid(nodes, k + 1);
        const size_t lds = ((size_t)2 << m) * sizeof(proj28_slot);
        if (leaf) hipLaunchKernelGGL(msm_planes_step<true>, grid, dim3(256), lds, st, offsets, in, k, m, out_nodes, long_count + 1, offsets + total, status_out);
        else hipLaunchKernelGGL(msm_planes_step<false>, grid, dim3(256), lds, st, offsets, in, k, m, out_nodes, long_count + 1, offsets + total, status_out);
        in = out_nodes;
        k += m;
      }
      flip ^= 1;
    }
    if (!table_c)           // W roots of c values each -> W x quads partial Horner values (the host finishes what msm.rs:42-46, 107-115 compute)
      hipLaunchKernelGGL(msm_planes_window_quads, dim3(W), dim3(64), 0, st, roots, plan.c, quads, window_sum, long_count + 1, offsets + total,
                         reinterpret_cast<uint32_t*>(window_sum + n_planes));
  } else {
#ifdef BP_EXPERIMENT
    hipLaunchKernelGGL(msm_reduce, dim3(blocks_per_window, Wr), dim3(256), 256 * sizeof(proj28_slot), st, offsets, plan, bucket_sum,
                       block_out);
    hipLaunchKernelGGL(msm_window_finish, dim3(Wr), dim3(256), 256 * sizeof(proj28_slot), st, block_out, blocks_per_window, window_sum,
                       long_count + 1, offsets + total, reinterpret_cast<uint32_t*>(window_sum + n_planes));
#endif
  }
  BP_HIP(ctx, hipGetLastError());
  if (d_blob) {
    MsmBlobHeader hdr;
    memset(&hdr, 0, sizeof hdr);
    hdr.magic = MSM_BLOB_MAGIC;
    hdr.c = plan.c;
    hdr.Wr = table_c ? 1 : Wr;
    hdr.n_planes = n_planes;
    hdr.tables = plan.naf ? 2u : (table_c != 0 ? 1u : 0u);
    hdr.quads = table_c ? 0u : quads;
    hipLaunchKernelGGL(msm_write_blob, dim3(1), dim3(256), 0, st, window_sum, hdr, (uint8_t*)d_blob);
    BP_HIP(ctx, hipGetLastError());
  } else {
    BP_HIP(ctx, hipMemcpyAsync(h_windows, window_sum, (size_t)n_planes * sizeof(proj28_slot) + 8, hipMemcpyDeviceToHost, st));    // sums + status + entry count
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[3], st));
  out->empty = false;
  out->tables = plan.naf ? 2u : (table_c != 0 ? 1u : 0u);
  out->c = plan.c;
  out->Wr = table_c ? 1 : Wr;
  out->n_planes = n_planes;
  out->quads = quads;
  out->adds = max_entries;      // upper bound (blob mode keeps it); msm_finish replaces it by the exact count of non-zero digits
  out->h_windows = h_windows;
  return BP_OK;
}

// waits for the stream, checks the scalar status and runs the host epilogue: window sums / planes back to the reference's
// Montgomery limbs, then Horner (msm.rs:107-115).  The timing stats describe the last launched MSM of this ctx.
int msm_finish(bp_ctx* ctx, const MsmPending& pend, g1_proj* host_out) {
  ctx->msm_async_pending = false;
  if (pend.empty) {
    for (uint32_t j = 0; host_out && j < pend.J; j++) host_out[j] = g1_identity();
    ctx->msm_accumulate_ms = ctx->msm_total_ms = 0;
    ctx->msm_adds = 0;
    if (pend.blob) BP_HIP(ctx, stream_wait(ctx->stream));
    return BP_OK;
  }
  BP_HIP(ctx, stream_wait(ctx->stream));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->msm_accumulate_ms, ctx->ev[1], ctx->ev[2]));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->msm_total_ms, ctx->ev[0], ctx->ev[3]));
  ctx->msm_c = pend.tables == 2 ? (MSM_NAF_FLAG | (pend.c + 1)) : pend.c;       // as given to bp_srs_precompute
  ctx->msm_tables = pend.tables != 0;
  ctx->msm_adds = pend.adds;
  if (pend.blob) {                      // the result stayed in HBM (bp_msm_g1_blob_device); its status word travels in the record
    if (host_out) *host_out = g1_identity();
    return BP_OK;
  }
  const proj28_slot* h_windows = static_cast<const proj28_slot*>(pend.h_windows);
  const uint32_t n_planes = pend.n_planes;
  const uint32_t* tail = reinterpret_cast<const uint32_t*>(h_windows + n_planes);
  if (tail[0]) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q in a canonical-bytes input", hipSuccess, __FILE__, __LINE__);
  ctx->msm_adds = tail[1];             // entries of the bucket-sorted list = non-zero digits = bucket additions performed
  std::vector<g1_proj> windows(n_planes);
  for (uint32_t w = 0; w < n_planes; w++) windows[w] = slot_to_proj(&h_windows[w]);
  if (pend.tables) {
    for (uint32_t j = 0; j < pend.J; j++)              // a batch: c values per vector, one Horner pass each
      host_plane_horner(host_out[j], windows.data() + (size_t)j * pend.c, pend.Wr, pend.c, pend.tables == 2);
  } else {
    host_quad_horner(*host_out, windows.data(), pend.Wr, pend.c, pend.quads);
  }
  return BP_OK;
}

// Host side of the one-process-per-GPU exchange: n_blobs records (BP_MSM_BLOB_BYTES apart, host memory).  Records that
// share the window layout (the normal case: equal shard sizes) are added slot by slot first, so the Horner pass over the
// c bit positions (msm.rs:107-115) runs once instead of once per rank.
int msm_blobs_combine(const uint8_t* blobs, size_t n_blobs, g1_proj* out) {
  g1_proj total = g1_identity();
  std::vector<bool> done(n_blobs, false);
  for (size_t i = 0; i < n_blobs; i++) {
    MsmBlobHeader h;
    memcpy(&h, blobs + i * BP_MSM_BLOB_BYTES, sizeof h);
    if (h.magic != MSM_BLOB_MAGIC || h.n_planes > (uint32_t)MSM_MAX_WINDOWS) return BP_ERR_INVALID_ARG;
    if (h.err != 0) return h.err < 0 && h.err >= BP_ERR_COMM ? h.err : BP_ERR_INVALID_ARG;       // a rank's poisoned record: its code for every rank (msm_blob_poisoned names the rank)
    // the Horner passes below index the slots by (Wr, c): a record whose header does not describe its own slot count (another
    // library version on a peer rank, a truncated gather) is rejected here instead of being read past its end
    if (h.n_planes != 0) {
      const bool width_ok = h.tables == 2 ? (h.c >= 5 && h.c <= 21) : (h.tables <= 1 && h.c >= 2 && h.c <= (uint32_t)MSM_MAX_TABLE_C);
      const uint32_t per = h.tables ? h.c : (h.quads ? h.quads : 1u);
      if (!width_ok || h.Wr < 1 || per > 8 * 4 || h.n_planes != h.Wr * per) return BP_ERR_INVALID_ARG;
    }
    if (h.status) return BP_ERR_BAD_SCALAR;
  }
  for (size_t i = 0; i < n_blobs; i++) {
    if (done[i]) continue;
    MsmBlobHeader h;
    memcpy(&h, blobs + i * BP_MSM_BLOB_BYTES, sizeof h);
    done[i] = true;
    if (h.n_planes == 0) continue;
    std::vector<g1_proj> sum(h.n_planes);
    const proj28_slot* slots = reinterpret_cast<const proj28_slot*>(blobs + i * BP_MSM_BLOB_BYTES + sizeof(MsmBlobHeader));
    for (uint32_t w = 0; w < h.n_planes; w++) sum[w] = slot_to_proj(&slots[w]);
    for (size_t k = i + 1; k < n_blobs; k++) {
      MsmBlobHeader g;
      memcpy(&g, blobs + k * BP_MSM_BLOB_BYTES, sizeof g);
      if (done[k] || g.c != h.c || g.Wr != h.Wr || g.n_planes != h.n_planes || g.tables != h.tables || g.quads != h.quads) continue;
      done[k] = true;
      const proj28_slot* sk = reinterpret_cast<const proj28_slot*>(blobs + k * BP_MSM_BLOB_BYTES + sizeof(MsmBlobHeader));
      for (uint32_t w = 0; w < h.n_planes; w++) g1_add(sum[w], sum[w], slot_to_proj(&sk[w]));
    }
    g1_proj part;
    if (h.tables) host_plane_horner(part, sum.data(), h.Wr, h.c, h.tables == 2);
    else host_quad_horner(part, sum.data(), h.Wr, h.c, h.quads ? h.quads : 1u);
    g1_add(total, total, part);
  }
  *out = total;
  return BP_OK;
}

// first poisoned record among n_blobs host records: its error code (0: none) and the rank that wrote it
int msm_blob_poisoned(const uint8_t* blobs, size_t n_blobs, uint32_t* rank) {
  for (size_t i = 0; i < n_blobs; i++) {
    MsmBlobHeader h;
    memcpy(&h, blobs + i * BP_MSM_BLOB_BYTES, sizeof h);
    if (h.magic == MSM_BLOB_MAGIC && h.err != 0) {
      if (rank) *rank = h.err_rank;
      return h.err < 0 && h.err >= BP_ERR_COMM ? h.err : BP_ERR_INVALID_ARG;
    }
  }
  return 0;
}

// this rank's record when its MSM failed before the collective: header only, on the context's stream
int msm_blob_poison_run(bp_ctx* ctx, void* d_blob, int err, uint32_t rank) {
  MsmBlobHeader hdr;
  memset(&hdr, 0, sizeof hdr);
  hdr.magic = MSM_BLOB_MAGIC;
  hdr.err = err;
  hdr.err_rank = rank;
  hipLaunchKernelGGL(msm_write_blob, dim3(1), dim3(64), 0, ctx->stream, (const proj28_slot*)nullptr, hdr, (uint8_t*)d_blob);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

int msm_blobs_sum_device_run(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out, bool wait) {
  hipLaunchKernelGGL(msm_blob_sum, dim3((MSM_MAX_WINDOWS + 7) / 8), dim3(64), 0, ctx->stream, (const uint8_t*)d_blobs, (uint32_t)n_blobs,
                     (uint32_t)BP_MSM_BLOB_BYTES, (uint8_t*)d_out);
  BP_HIP(ctx, hipGetLastError());
  if (wait) BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}

int msm_run(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
            g1_proj* host_out) {
  MsmPending pend;
  BP_TRY(msm_launch(ctx, d_points28, n, d_scalars, fmt, table_c, table_stride, 0, nullptr, &pend));
  return msm_finish(ctx, pend, host_out);
}

}  // namespace bp
