// msm.hip -- host driver of the G1 MSM pipeline (kernels in msm_kernels.hpp).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ctx.hpp"
#include "msm_kernels.hpp"

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

// Window width: minimise W * (n + 2 * 2^(c-1)) (bucket adds + reduction adds).  Per-window bucket sets (any point set) are
// bounded by the 128 KiB LDS histogram of one window (c <= 16); with fixed-base tables wider windows go through the
// partitioned sort (plan.parts > 1), up to MSM_MAX_TABLE_C.
// table_c != 0: the SRS carries fixed-base window tables built for that width (row length table_stride)
static void make_plan(MsmPlan& plan, size_t n, uint32_t table_c, size_t table_stride) {
  memset(&plan, 0, sizeof plan);
  plan.n = (uint32_t)n;
  if (table_c & MSM_NAF_FLAG) {         // every-position tables: odd NAF digits of width w, 2^(w-2) buckets, one bucket set
    const uint32_t w = table_c & 0xffu;
    plan.naf = w;
    plan.c = w - 1;
    plan.W = 255 / w + 1;               // digit slots per scalar (most scalars fill 256 / (w + 1) of them)
    plan.B = 1u << (w - 2);
    plan.total = plan.B;
    plan.wbuckets = 0;
    plan.wpoints = (uint32_t)table_stride;
    plan.parts = plan.B <= (1u << MSM_HIST_LOG) ? 1 : plan.B >> MSM_HIST_LOG;
    // lanes: one wave round (131 072) for the worst case of every slot filled; the kernels cut the sorted list by the actual
    // entry count (a uniform scalar fills 256 / (w + 1) + ~0.45 of its slots: 15.49 / 13.91 / 12.71 at w = 16 / 18 / 20)
    uint32_t chunk = (uint32_t)(((uint64_t)plan.W * n + 131071) / 131072);
    if (chunk < 4) chunk = 4;
    if (chunk > 1024) chunk = 1024;
    plan.chunk = env_u32("BP_MSM_CHUNK", chunk);
    plan.lanes = getenv("BP_MSM_CHUNK") ? 0u : (uint32_t)(((uint64_t)plan.W * n + plan.chunk - 1) / plan.chunk);
    plan.slices = 1;
    plan.seg = 1;
    return;
  }
  uint32_t c = n < 32 ? 4 : ilog2_floor(n) - 3;
  if (c < 4) c = 4;
  if (c > MSM_MAX_C) c = MSM_MAX_C;
  c = env_u32("BP_MSM_C", c);
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
  plan.c = c;
  plan.W = W;
  plan.B = 1u << (c - 1);
  plan.total = table_c ? plan.B : W * plan.B;
  plan.wbuckets = table_c ? 0 : plan.B;
  plan.wpoints = table_c ? (uint32_t)table_stride : 0;
  plan.parts = plan.B <= (1u << MSM_HIST_LOG) ? 1 : plan.B >> MSM_HIST_LOG;
  const uint64_t entries = (uint64_t)W * n;
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
  plan.chunk = env_u32("BP_MSM_CHUNK", chunk);
  plan.lanes = getenv("BP_MSM_CHUNK") ? 0u : (uint32_t)((entries + plan.chunk - 1) / plan.chunk);      // = the host's n_chunks
  // count/scatter workgroups per window: each flushes its whole LDS histogram with global atomics, so fewer, fatter
  // slices are cheaper (~32 Ki points each) as long as >= 256 workgroups remain to fill the CUs (measured: 2^16, 2^20, 2^24)
  uint32_t slices = (uint32_t)(n >> 15), lo = 256 / W, hi = 1024 / W;
  if (slices < lo) slices = lo;
  if (slices > hi) slices = hi;
  if (slices < 1) slices = 1;
  while (slices > 1 && n / slices < 1024) slices >>= 1;
  plan.slices = env_u32("BP_MSM_SLICES", slices);
  uint32_t seg = 1;
  while (seg < 32 && plan.total / seg > 65536) seg <<= 1;
  plan.seg = env_u32("BP_MSM_SEG", seg);
}

// 112-byte unsaturated copy of an SRS (allocated here, owned by the SRS entry)
int srs_to28_run(bp_ctx* ctx, const g1_affine* d_in, size_t n, g1_affine28** d_out) {
  g1_affine28* d = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&d, (n ? n : 1) * sizeof(g1_affine28)));
  if (n) hipLaunchKernelGGL(srs_to28, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, n, d);
  BP_HIP(ctx, hipGetLastError());
  *d_out = d;
  return BP_OK;
}

uint32_t msm_table_windows(uint32_t c) {
  if (c & MSM_NAF_FLAG) return MSM_NAF_ROWS;
  MsmPlan plan;
  make_plan(plan, 1, c, 1);
  return plan.W;
}

// table[w * n + i] = 2^(c w) P_i; row 0 is the unsaturated SRS copy itself
int srs_tables_run(bp_ctx* ctx, const g1_affine* d_points, const g1_affine28* d_points28, size_t n, uint32_t c, g1_affine28** d_table,
                   uint32_t* windows) {
  const uint32_t W = msm_table_windows(c);
  if ((uint64_t)W * n >= (1ull << 31))
    return fail(ctx, BP_ERR_TOO_LARGE, "fixed-base tables: windows * points >= 2^31", hipSuccess, __FILE__, __LINE__);
  g1_affine28* t = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&t, (size_t)W * (n ? n : 1) * sizeof(g1_affine28)));
  if (n) {
    BP_HIP(ctx, hipMemcpyAsync(t, d_points28, n * sizeof(g1_affine28), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(srs_window_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_points, n, (c & MSM_NAF_FLAG) ? 1u : c, W, t);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
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
  // a full-window histogram at c = 16 needs 128 KiB of dynamic LDS, the plane tree 88 KiB
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_planes_block<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_planes_block<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_count, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // these two also hold a few KiB of static LDS: the dynamic limit must leave room for it
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_final, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_count, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  BP_HIP(ctx, hipFuncSetAttribute((const void*)msm_radix_long_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  return BP_OK;
}

// d_points28: the unsaturated SRS copy, or (table_c != 0) row 0 of its fixed-base tables with rows table_stride apart.
// Enqueues the whole pipeline and the device-to-host copy of the window sums on ctx->stream; waits for nothing.
int msm_launch(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
               int slot, void* d_blob, MsmPending* out) {
  *out = MsmPending();
  out->blob = d_blob != nullptr;
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
  make_plan(plan, n, table_c, table_stride);
  if ((uint64_t)plan.W * n >= (1ull << 32))        // positions in the bucket-sorted list are 32-bit
    return fail(ctx, BP_ERR_TOO_LARGE, "MSM length * windows >= 2^32", hipSuccess, __FILE__, __LINE__);
  const uint32_t W = plan.W, B = plan.B, total = plan.total, Wr = table_c ? 1 : W;   // Wr: windows left after accumulation
  const uint64_t max_entries = (uint64_t)W * n;
  const uint64_t n_chunks = (max_entries + plan.chunk - 1) / plan.chunk;
  // bucket reduction: running sums per window, or (tables: one bucket set) the bit-plane tree in two stages of l1 + l2 levels
  const uint32_t planes_log = env_u32("BP_MSM_PLANES_LOG", plan.c > 16 ? 7 : PLANES_BLOCK_LOG) == 7 ? 7 : PLANES_BLOCK_LOG;
  const uint32_t l1 = plan.c - 1 < planes_log ? plan.c - 1 : planes_log, l2 = plan.c - 1 - l1;
  const uint32_t blocks_per_window = table_c ? 1u << l2 : ((B + plan.seg - 1) / plan.seg + 255) / 256;
  const uint32_t per_block = table_c ? l1 + 1 : 1, per_window = table_c ? plan.c : 1;     // slots

  int16_t* digits = nullptr;
  uint32_t *counts, *offsets, *cursors, *sorted;
  proj28_slot *bucket_sum, *partial, *block_out, *window_sum;
  if (plan.parts == 1 && !plan.naf) BP_TRY(ws_get(ctx, "msm.digits", max_entries * sizeof(int16_t), (void**)&digits));
  BP_TRY(ws_get(ctx, "msm.counts", (size_t)total * 4 + 8, (void**)&counts));       // + [0] long-bucket counter, [1] scalar status: one memset
  BP_TRY(ws_get(ctx, "msm.offsets", ((size_t)total + 1) * 4, (void**)&offsets));
  BP_TRY(ws_get(ctx, "msm.cursors", (size_t)total * 4, (void**)&cursors));
  // a bucket is "long" when it spans >= FIXUP_LONG chunks, so at most n_chunks / FIXUP_LONG + 1 buckets can be long
  const uint32_t long_cap = (uint32_t)(n_chunks / FIXUP_LONG + 1);
  uint32_t *long_count, *long_list;
  long_count = counts + total;
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
  BP_TRY(ws_get(ctx, "msm.block_out", (size_t)Wr * blocks_per_window * per_block * sizeof(proj28_slot), (void**)&block_out));
  BP_TRY(ws_get(ctx, "msm.window_sum", (size_t)n_planes * sizeof(proj28_slot) + 16, (void**)&window_sum));     // + the status word
  // pinned staging: MSM_SLOTS result areas of the largest possible size, so earlier pending results stay where they are
  constexpr size_t slot_bytes = (size_t)MSM_MAX_WINDOWS * sizeof(proj28_slot) + 16;
  uint8_t* h_base;
  BP_TRY(pinned_get(ctx, MSM_SLOTS * slot_bytes, (void**)&h_base));
  proj28_slot* h_windows = reinterpret_cast<proj28_slot*>(h_base + (size_t)slot * slot_bytes);
  hipStream_t st = ctx->stream;
  BP_HIP(ctx, hipEventRecord(ctx->ev[0], st));
  const size_t hist_bytes = (size_t)B * 4;
  const unsigned hist_threads = B >= 4096 ? 1024 : 256;   // a big histogram owns the CU's LDS: fill the CU with one workgroup
  const uint32_t n_tiles = (total + SCAN_TILE - 1) / SCAN_TILE;      // <= 4096 (total <= 2^24)
  // Bucket sort.  c <= 16: the one-histogram counting sort (msm_count / msm_scatter).  Wider windows: the partitioned (radix)
  // sort -- every store coalesced or L2-merged.  (At c = 16 the two cost the same, 0.34 vs 0.35 ms at 2^20: the radix sort
  // moves 8-byte records three times.)  BP_MSM_SORT=1 forces the radix sort everywhere (tests, A/B).
  const bool radix = plan.parts > 1 || plan.naf || env_u32("BP_MSM_SORT", 0) == 1;
  if (radix) {
    uint32_t kb = 0;
    while ((1ull << kb) < total) kb++;
    uint32_t pb = 0;
    while (pb < kb && (max_entries >> (pb + 1)) >= 12288) pb++;             // final runs of ~12-24 Ki entries
    pb = env_u32("BP_MSM_RADIX_BITS", pb);
    if (pb + MSM_HIST_LOG < kb) pb = kb - MSM_HIST_LOG;                   // a final run's buckets must fit one LDS histogram
    if (pb > kb) pb = kb;
    if (pb > 16) pb = 16;
    const uint32_t lv[2] = {pb <= 8 ? pb : pb - pb / 2, pb <= 8 ? 0 : pb / 2}, rbits = kb - pb, n_final = 1u << pb;
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
    uint32_t* rlong;                                     // [0] number of long final runs, [1..] their list
    BP_TRY(ws_get(ctx, "msm.run_long", ((size_t)n_final + 2) * 4, (void**)&rlong));
    BP_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)total * 4 + 8, st));     // bucket sizes of long runs + long-bucket counter + scalar status
    BP_HIP(ctx, hipMemsetAsync(rlong, 0, 4, st));
    if (plan.naf)
      hipLaunchKernelGGL(msm_naf_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, keys[0], vals[0], long_count + 1);
    else
      hipLaunchKernelGGL(msm_digit_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, keys[0], vals[0], long_count + 1);
    uint32_t runs = 1, shift = kb, side = 0;
    for (int level = 0; level < 2 && lv[level]; level++) {
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
    const size_t rhist = ((size_t)1 << rbits) * 4;
    hipLaunchKernelGGL(msm_radix_final, dim3(runs < 4096 ? runs : 4096), dim3(1024), rhist, st, keys[side], vals[side], run_off[level_count], runs,
                       rbits, total, offsets, sorted, rlong, rlong + 1);
    if (runs > 1) {                 // long runs (none for uniformly random scalars beyond the top window's): slice-parallel
      const dim3 lgrid(256, runs < 16 ? runs : 16);
      hipLaunchKernelGGL(msm_radix_long_count, lgrid, dim3(1024), rhist, st, keys[side], run_off[level_count], rbits, rlong, rlong + 1, counts);
      hipLaunchKernelGGL(msm_radix_long_prefix, dim3(runs < 64 ? runs : 64), dim3(1024), 0, st, run_off[level_count], runs, rbits, total, rlong,
                         rlong + 1, counts, offsets, cursors);
      hipLaunchKernelGGL(msm_radix_long_scatter, lgrid, dim3(1024), rhist, st, keys[side], vals[side], run_off[level_count], rbits, rlong, rlong + 1,
                         cursors, sorted);
    }
  } else {
    BP_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)total * 4 + 8, st));
    hipLaunchKernelGGL(msm_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, fmt, plan, digits, long_count + 1);
    hipLaunchKernelGGL(msm_count, dim3(plan.slices, W), dim3(hist_threads), hist_bytes, st, digits, plan, counts);
    hipLaunchKernelGGL(scan_tile_sums, dim3(n_tiles), dim3(256), 0, st, counts, total, tile_sums);
    hipLaunchKernelGGL(scan_block_sums, dim3(1), dim3(256), 0, st, tile_sums, n_tiles, offsets + total);
    hipLaunchKernelGGL(scan_apply, dim3(n_tiles), dim3(256), 0, st, counts, total, tile_sums, offsets, cursors);
    hipLaunchKernelGGL(msm_scatter, dim3(plan.slices, W), dim3(hist_threads), hist_bytes, st, digits, plan, cursors, sorted);
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[1], st));
  const dim3 acc_grid((unsigned)((n_chunks + 255) / 256));
  switch (env_u32("BP_MSM_ACC_WAVES", 2)) {
    case 3: hipLaunchKernelGGL(msm_accumulate<3>, acc_grid, dim3(256), 0, st, d_points28, sorted, offsets, plan, bucket_sum, partial); break;
    case 4: hipLaunchKernelGGL(msm_accumulate<4>, acc_grid, dim3(256), 0, st, d_points28, sorted, offsets, plan, bucket_sum, partial); break;
    default: hipLaunchKernelGGL(msm_accumulate<2>, acc_grid, dim3(256), 0, st, d_points28, sorted, offsets, plan, bucket_sum, partial);
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[2], st));
  if (table_c)      // 2^15 buckets of ~8 partials each: two lanes per bucket = one wave per SIMD, half the chain
    hipLaunchKernelGGL(msm_fixup<2>, dim3((unsigned)(((uint64_t)total * 2 + 255) / 256)), dim3(256), 0, st, offsets, plan, bucket_sum, partial,
                       long_count, long_list, long_cap);
  else
    hipLaunchKernelGGL(msm_fixup<1>, dim3((total + 255) / 256), dim3(256), 0, st, offsets, plan, bucket_sum, partial, long_count, long_list,
                       long_cap);
  hipLaunchKernelGGL(msm_fixup_long, dim3(512, FIXUP_LONG_SLICES), dim3(256), 256 * sizeof(proj28_slot), st, offsets, plan, bucket_sum, partial,
                     long_count, long_list, long_cap, long_scratch);
  hipLaunchKernelGGL(msm_fixup_long_merge, dim3(512), dim3(256), 256 * sizeof(proj28_slot), st, offsets, plan, bucket_sum, long_count, long_list,
                     long_cap, long_scratch);
  if (table_c) {
    if (planes_log == 7)
      hipLaunchKernelGGL((msm_planes_block<128, 2>), dim3(blocks_per_window, Wr), dim3(128), 256 * sizeof(proj28_slot), st, offsets, plan,
                         bucket_sum, l1, block_out);
    else
      hipLaunchKernelGGL((msm_planes_block<256, 1>), dim3(blocks_per_window, Wr), dim3(256), 512 * sizeof(proj28_slot), st, offsets, plan,
                         bucket_sum, l1, block_out);
    // merge steps of at most 7 levels each until one node (A and the c - 1 planes) is left; c <= 16 needs one
    uint32_t k = l1, r = l2;
    const proj28_slot* in = block_out;
    proj28_slot* tmp[2] = {nullptr, nullptr};
    if (r > 7) {
      BP_TRY(ws_get(ctx, "msm.planes_tmp0", ((size_t)1 << (r - 7)) * (k + 8) * sizeof(proj28_slot), (void**)&tmp[0]));
      if (r > 14) BP_TRY(ws_get(ctx, "msm.planes_tmp1", ((size_t)1 << (r - 14)) * (k + 15) * sizeof(proj28_slot), (void**)&tmp[1]));
    }
    int flip = 0;
    do {
      const uint32_t m = r < 7 ? r : 7, nodes = 1u << (r - m);
      const bool last = r == m;
      proj28_slot* out_nodes = last ? window_sum : tmp[flip];
      hipLaunchKernelGGL(msm_planes_window, dim3(k + 1, nodes), dim3(512), 256 * sizeof(proj28_slot), st, in, k, m, out_nodes, long_count + 1,
                         offsets + total, last ? reinterpret_cast<uint32_t*>(window_sum + n_planes) : (uint32_t*)nullptr);
      in = out_nodes;
      flip ^= 1;
      k += m;
      r -= m;
    } while (r > 0);
  } else {
    hipLaunchKernelGGL(msm_reduce, dim3(blocks_per_window, Wr), dim3(256), 256 * sizeof(proj28_slot), st, offsets, plan, bucket_sum,
                       block_out);
    hipLaunchKernelGGL(msm_window_finish, dim3(Wr), dim3(256), 256 * sizeof(proj28_slot), st, block_out, blocks_per_window, window_sum,
                       long_count + 1, offsets + total, reinterpret_cast<uint32_t*>(window_sum + n_planes));
  }
  BP_HIP(ctx, hipGetLastError());
  if (d_blob) {
    MsmBlobHeader hdr;
    memset(&hdr, 0, sizeof hdr);
    hdr.magic = MSM_BLOB_MAGIC;
    hdr.c = plan.c;
    hdr.Wr = Wr;
    hdr.n_planes = n_planes;
    hdr.tables = plan.naf ? 2u : (table_c != 0 ? 1u : 0u);
    hipLaunchKernelGGL(msm_write_blob, dim3(1), dim3(256), 0, st, window_sum, hdr, (uint8_t*)d_blob);
    BP_HIP(ctx, hipGetLastError());
  } else {
    BP_HIP(ctx, hipMemcpyAsync(h_windows, window_sum, (size_t)n_planes * sizeof(proj28_slot) + 8, hipMemcpyDeviceToHost, st));    // sums + status + entry count
  }
  BP_HIP(ctx, hipEventRecord(ctx->ev[3], st));
  out->empty = false;
  out->tables = plan.naf ? 2u : (table_c != 0 ? 1u : 0u);
  out->c = plan.c;
  out->Wr = Wr;
  out->n_planes = n_planes;
  out->adds = max_entries;      // upper bound (blob mode keeps it); msm_finish replaces it by the exact count of non-zero digits
  out->h_windows = h_windows;
  return BP_OK;
}

// waits for the stream, checks the scalar status and runs the host epilogue: window sums / planes back to the reference's
// Montgomery limbs, then Horner (msm.rs:107-115).  The timing stats describe the last launched MSM of this ctx.
int msm_finish(bp_ctx* ctx, const MsmPending& pend, g1_proj* host_out) {
  if (pend.empty) {
    if (host_out) *host_out = g1_identity();
    ctx->msm_accumulate_ms = ctx->msm_total_ms = 0;
    ctx->msm_adds = 0;
    if (pend.blob) BP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BP_OK;
  }
  BP_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
  if (pend.tables) host_plane_horner(*host_out, windows.data(), pend.Wr, pend.c, pend.tables == 2);
  else host_horner(*host_out, windows.data(), pend.Wr, pend.c);
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
      if (done[k] || g.c != h.c || g.Wr != h.Wr || g.n_planes != h.n_planes || g.tables != h.tables) continue;
      done[k] = true;
      const proj28_slot* sk = reinterpret_cast<const proj28_slot*>(blobs + k * BP_MSM_BLOB_BYTES + sizeof(MsmBlobHeader));
      for (uint32_t w = 0; w < h.n_planes; w++) g1_add(sum[w], sum[w], slot_to_proj(&sk[w]));
    }
    g1_proj part;
    if (h.tables) host_plane_horner(part, sum.data(), h.Wr, h.c, h.tables == 2);
    else host_horner(part, sum.data(), h.Wr, h.c);
    g1_add(total, total, part);
  }
  *out = total;
  return BP_OK;
}

int msm_blobs_sum_device_run(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out) {
  hipLaunchKernelGGL(msm_blob_sum, dim3((MSM_MAX_WINDOWS + 7) / 8), dim3(64), 0, ctx->stream, (const uint8_t*)d_blobs, (uint32_t)n_blobs,
                     (uint32_t)BP_MSM_BLOB_BYTES, (uint8_t*)d_out);
  BP_HIP(ctx, hipGetLastError());
  BP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return BP_OK;
}

int msm_run(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
            g1_proj* host_out) {
  MsmPending pend;
  BP_TRY(msm_launch(ctx, d_points28, n, d_scalars, fmt, table_c, table_stride, 0, nullptr, &pend));
  return msm_finish(ctx, pend, host_out);
}

}  // namespace bp
