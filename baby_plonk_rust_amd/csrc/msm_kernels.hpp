// msm_kernels.hpp -- G1 multi-scalar multiplication on gfx950: sum_i s_i * P_i.
// Replaces BucketMSM::bucket_msm (src/msm.rs:76-118) behind Setup::commit (src/setup.rs:32-37).
//
// The reference walks 64 four-bit windows with 15 buckets each and ~60 projective adds per point.
// Any bucket method yields the same group element, and parity is defined on its affine encoding
// (SURVEY.md section 8a), so the device algorithm is chosen for the hardware:
//
//   1. msm_digits       scalars (Montgomery Fr, coalesced 32-B loads) -> canonical integer ->
//                       W signed c-bit digits (bias trick, no carry chain), written window-major.
//   2. msm_count        per (window, slice) workgroup: LDS histogram of the 2^(c-1) buckets of one
//                       window (c = 16 -> 128 KiB, which is why LDS size picks c), flushed to HBM.
//   3. scan_*           exclusive scan of all W * 2^(c-1) bucket sizes -> bucket offsets (3 launches).
//   4. msm_scatter      same LDS histogram; one global atomic per (workgroup, bucket) reserves a range,
//                       LDS atomics rank inside it; point indices land bucket-sorted (counting sort).
//   5. msm_accumulate   the hot loop.  The bucket-sorted index list is cut into equal chunks, one per
//                       lane, regardless of bucket boundaries: every lane performs the same number of
//                       complete mixed additions (gathered 112-B points of the unsaturated SRS copy, 14 x 28-bit
//                       lazy limbs of fp28.hpp, accumulator in VGPRs),
//                       so wave utilisation does not depend on the scalar distribution.  Runs that
//                       cover a whole bucket are stored as the bucket sum; runs cut by a chunk edge go
//                       to a per-lane partial slot.
//   6. msm_fixup        buckets that straddle chunks: add their partials.
//   7. msm_reduce       sum_b b * S_b per window by segmented running sums + LDS tree; with fixed-base tables
//      msm_planes_*     (one bucket set for all windows) as bit planes T_j = sum of the buckets with bit j set:
//                       a binary tree in LDS, two additions per bucket, dependent chain of log2(B) additions.
//   host epilogue       Horner over the W window sums / the bit planes (c doublings per window, msm.rs:107-115)
//                       + one affine normalisation.
//
// Group law: complete RCB formulas (g1.hpp) -- branch-free, so P+P / P+(-P) / identity need no
// divergent special cases.
#pragma once
#include "g1.hpp"
#include "g1_28.hpp"
#include "msm_digits.hpp"

namespace bp {

constexpr int MSM_MAX_C = 16;            // per-window bucket sets (any point set): one LDS histogram per window
constexpr int MSM_MAX_TABLE_C = 24;      // fixed-base tables: partitioned counting sort above 16 bits
constexpr uint32_t MSM_HIST_LOG = 15;    // buckets per LDS histogram (128 KiB of the 160)
constexpr int MSM_MAX_WINDOWS = 128;

struct MsmPlan {
  uint32_t n;          // number of (point, scalar) pairs
  uint32_t c;          // window bits
  uint32_t W;          // windows
  uint32_t B;          // buckets per window = 2^(c-1)
  uint32_t chunk;      // entries per lane in msm_accumulate when every digit is non-zero (sizes the lane count and the partial slots)
  uint32_t lanes;      // 0: the chunk is fixed (BP_MSM_CHUNK).  Else the lanes of msm_accumulate = ceil(W n / chunk): the kernels cut
                       // the sorted list into ceil(M / lanes) entries per lane from the ACTUAL entry count M (zero digits are
                       // not entries: small or sparse scalars keep every lane busy with short chains instead of a few lanes with long ones)
  uint32_t slices;     // workgroups per window in count/scatter
  uint32_t seg;        // buckets per lane in msm_reduce
  uint32_t bias[9];    // sum_{w < W-1} 2^(c-1) * 2^(c*w)   (the top window is unsigned)
  // Two bucket layouts share every kernel.  Per-window buckets (any point set): window w owns buckets
  // [w B, (w+1) B) and an entry is the point index i.  Fixed-base tables (bp_srs_precompute): the SRS carries
  // T[w][i] = 2^(cw) P_i, every window feeds the SAME B buckets and an entry is the table index w * stride + i,
  // so the reduction and the Horner epilogue run over one window instead of W.
  uint32_t total;      // number of buckets: W * B, or B with tables
  uint32_t wbuckets;   // bucket-index stride per window: B, or 0 with tables
  uint32_t wpoints;    // point-index stride per window: 0, or the table row length with tables
  // Windows wider than 16 bits (fixed-base tables only): one window's 2^(c-1) buckets no longer fit one LDS histogram
  // (parts > 1); those MSMs take the partitioned (radix) bucket sort, section 4c.
  uint32_t parts;      // 1, or B >> 15
  // Fixed-base tables of EVERY bit position (T[p][i] = 2^p P_i, 256 rows: the 288 GB of HBM pay for it up to ~2^21 points) carry
  // the scalar's width-`naf` non-adjacent form: odd digits |d| < 2^(naf-1) at arbitrary positions, at most one per `naf`
  // positions -- 256 / (naf + 1) bucket additions per scalar on average instead of 256 / c, with only 2^(naf-2) buckets
  // (bucket (|d| - 1) / 2).  Then c = naf - 1 (so that B = 2^(c-1) as everywhere), W = digit slots per scalar.
  uint32_t naf;        // 0, or the NAF width
  // Several scalar vectors against ONE table set in one pipeline (the commitments of a prover round, prover.rs:249-251,
  // 483-485, 640-641): vector j owns the buckets [j B, (j + 1) B), so the sort, the accumulation, the fix-up and the tree run
  // once over J bucket sets and deliver J results.  n is then the longest vector; scalar (j, i) is element j * n + i of the
  // pipeline's index space, elements with i >= nj[j] do not exist.  Partition sort only (msm_part_*); J = 1 everywhere else.
  uint32_t J;          // 1 .. MSM_MAX_BATCH
  uint32_t nj[4];      // length of vector j
  // Fixed-base tables free the digit radix from being a power of two (round 6): T[w][i] = R^w P_i for ANY R.  W windows of c bits cover
  // c W >= 256 bits, but a scalar has 255: at c = 20 (13 windows) five bits of every bucket index are bought and never used.  With
  // R = the smallest (low Hamming weight) even number whose W-th power exceeds 2^256 the signed digits |d| <= R / 2 fill R / 2 buckets
  // instead of 2^(c-1): 425 984 instead of 524 288 at 13 windows, 1 327 104 instead of 2 097 152 at 12 -- a fifth to a third of the
  // bucket tree's leaves (and of its additions) gone; the live buckets are a PREFIX of the 2^(c-1) the tree is sized for, the rest
  // stay empty and whole waves of the tree skip them.  radix = 0: power-of-two windows (every table-free MSM; widths that waste < 10 %).
  // Digits: k' = k + bias (bias = sum_{w < W-1} (R / 2) R^w), f = k' / R^W as a 288-bit fraction = the top limbs of k' * radix_m
  // (radix_m = ceil(2^512 / R^W)) + 2 ulp, then W times "multiply by R, take the integer part" from the top digit down; signed digit
  // = that - R / 2 below the top window.  Exact: the fraction's error is in [0, 2^-257 + 2^-287) and 1 / R^W > 2^-256.93 (make_plan).
  uint32_t radix;
  uint32_t radix_m[8];
};
constexpr uint32_t MSM_MAX_BATCH = 4;
struct MsmScalars { const fr_t* p[MSM_MAX_BATCH]; };

// ---------------------------------------------------------------- 1. digits
#ifdef BP_EXPERIMENT      // one-histogram counting sort (BP_MSM_SORT=0): digit array pass
// fmt 0: 32-byte little-endian canonical (Scalar::to_bytes), 1: Montgomery limbs (Scalar::to_array)
__global__ void __launch_bounds__(256) msm_digits(const fr_t* __restrict__ scalars, int fmt, MsmPlan plan,
                                                   int16_t* __restrict__ digits, uint32_t* __restrict__ status) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plan.n) return;
  fr_t k = scalars[i];
  if (fmt == 1) {
    Fr::from_mont(k, k);                           // msm.rs:126: scalar.to_bytes() = canonical integer
  } else {
    fr_t t;                                        // Scalar::from_bytes rejects values >= q (scalar.rs:264-288)
    if (!big_sub(t, k, Fr::modulus())) atomicOr(status, 1u);
  }
  uint32_t kp[10];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    carry += (uint64_t)(j < 8 ? k.l[j] : 0u) + plan.bias[j];
    kp[j] = (uint32_t)carry;
    carry >>= 32;
  }
  kp[9] = 0;
  const uint32_t c = plan.c, mask = (1u << c) - 1u, half = 1u << (c - 1);
  for (uint32_t w = 0; w < plan.W; w++) {
    uint32_t o = c * w, word = o >> 5, sh = o & 31;
    uint64_t two = (uint64_t)kp[word] | ((uint64_t)kp[word + 1] << 32);
    // windows below the top one are signed (bias already added); the top window keeps its small unsigned value
    int32_t d = (int32_t)((uint32_t)(two >> sh) & mask) - (w + 1 < plan.W ? (int32_t)half : 0);
    digits[(size_t)w * plan.n + i] = (int16_t)d;
  }
}

#endif  // BP_EXPERIMENT
// bucket index inside a window for a non-zero digit: |d| - 1 in [0, 2^(c-1))
__device__ __forceinline__ uint32_t digit_bucket(int32_t d) { return (uint32_t)(d < 0 ? -d : d) - 1u; }

// ---------------------------------------------------------------- 2. count
extern __shared__ uint32_t msm_lds_hist[];

__device__ __forceinline__ void slice_range(const MsmPlan& plan, uint32_t slice, uint32_t& lo, uint32_t& hi) {
  uint32_t per = (plan.n + plan.slices - 1) / plan.slices;
  lo = slice * per;
  hi = lo + per < plan.n ? lo + per : plan.n;
  if (lo > plan.n) lo = plan.n;
}

#ifdef BP_EXPERIMENT      // one-histogram counting sort: count pass
__global__ void __launch_bounds__(1024) msm_count(const int16_t* __restrict__ digits, MsmPlan plan,
                                                  uint32_t* __restrict__ counts) {
  const uint32_t w = blockIdx.y, B = plan.B;
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) msm_lds_hist[b] = 0;
  __syncthreads();
  uint32_t lo, hi;
  slice_range(plan, blockIdx.x, lo, hi);
  const int16_t* dw = digits + (size_t)w * plan.n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int32_t d = dw[i];
    if (d != 0) atomicAdd(&msm_lds_hist[digit_bucket(d)], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
    uint32_t v = msm_lds_hist[b];
    if (v) atomicAdd(&counts[(size_t)w * plan.wbuckets + b], v);
  }
}

#endif  // BP_EXPERIMENT
// ---------------------------------------------------------------- 3. scan (three small launches)
// offsets[0..total] = exclusive prefix sums of counts[0..total); cursors = copy of offsets[0..total).
// Tile = 4096 counts per workgroup (256 lanes x 16).  scan_tile_sums -> scan_block_sums -> scan_apply.
constexpr uint32_t SCAN_TILE = 4096, SCAN_PER_LANE = 16;

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* lds, uint32_t& block_total) {
  // wave-level inclusive scan by shuffles, then a 4-entry LDS hop across the waves of the workgroup
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t n = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += n;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    uint32_t x = lds[w];
    if ((uint32_t)w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  block_total = tot;
  return base + incl - v;
}
__global__ void __launch_bounds__(256) scan_tile_sums(const uint32_t* __restrict__ counts, uint32_t total, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_LANE;
  uint32_t s = 0;
#pragma unroll
  for (uint32_t j = 0; j < SCAN_PER_LANE; j++) s += base + j < total ? counts[base + j] : 0;
  uint32_t tot;
  (void)block_exclusive_scan_256(s, lds, tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
// in-place exclusive scan of up to 256 * 16 tile sums by one workgroup; writes the grand total to *total_out
__global__ void __launch_bounds__(256) scan_block_sums(uint32_t* __restrict__ tile_sums, uint32_t n_tiles, uint32_t* __restrict__ total_out) {
  __shared__ uint32_t lds[4];
  const uint32_t base = threadIdx.x * SCAN_PER_LANE;
  uint32_t v[SCAN_PER_LANE], s = 0;
#pragma unroll
  for (uint32_t j = 0; j < SCAN_PER_LANE; j++) {
    v[j] = base + j < n_tiles ? tile_sums[base + j] : 0;
    s += v[j];
  }
  uint32_t tot, run = block_exclusive_scan_256(s, lds, tot);
#pragma unroll
  for (uint32_t j = 0; j < SCAN_PER_LANE; j++) {
    if (base + j < n_tiles) tile_sums[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == 0) *total_out = tot;
}
__global__ void __launch_bounds__(256) scan_apply(const uint32_t* __restrict__ counts, uint32_t total, const uint32_t* __restrict__ tile_sums,
                                                   uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_LANE;
  uint32_t v[SCAN_PER_LANE], s = 0;
#pragma unroll
  for (uint32_t j = 0; j < SCAN_PER_LANE; j++) {
    v[j] = base + j < total ? counts[base + j] : 0;
    s += v[j];
  }
  uint32_t tot, run = tile_sums[blockIdx.x] + block_exclusive_scan_256(s, lds, tot);
#pragma unroll
  for (uint32_t j = 0; j < SCAN_PER_LANE; j++) {
    if (base + j < total) {
      offsets[base + j] = run;
      cursors[base + j] = run;
    }
    run += v[j];
  }
}

#ifdef BP_EXPERIMENT      // one-histogram counting sort: scatter pass
// ---------------------------------------------------------------- 4. scatter (counting sort)
__global__ void __launch_bounds__(1024) msm_scatter(const int16_t* __restrict__ digits, MsmPlan plan,
                                                    uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
  const uint32_t w = blockIdx.y, B = plan.B;
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) msm_lds_hist[b] = 0;
  __syncthreads();
  uint32_t lo, hi;
  slice_range(plan, blockIdx.x, lo, hi);
  const int16_t* dw = digits + (size_t)w * plan.n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int32_t d = dw[i];
    if (d != 0) atomicAdd(&msm_lds_hist[digit_bucket(d)], 1u);
  }
  __syncthreads();
  // reserve this workgroup's range in every bucket it touches; LDS now holds the range start
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
    uint32_t v = msm_lds_hist[b];
    if (v) msm_lds_hist[b] = atomicAdd(&cursors[(size_t)w * plan.wbuckets + b], v);
  }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int32_t d = dw[i];
    if (d != 0) {
      uint32_t pos = atomicAdd(&msm_lds_hist[digit_bucket(d)], 1u);
      sorted[pos] = (i + w * plan.wpoints) | (d < 0 ? 0x80000000u : 0u);
    }
  }
}

#endif  // BP_EXPERIMENT
// exclusive scan of one value per lane over a 1024-lane workgroup (16 waves); lds16: 16 words
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* lds16) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += t;
  }
  if (lane == 63) lds16[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; w++) base += lds16[w];
  __syncthreads();
  return base + incl - v;
}
// ---------------------------------------------------------------- 4c. partitioned bucket sort (MSD radix, coalesced)
// msm_scatter pays for every entry with an isolated 4-byte store: a workgroup's ~32 Ki entries land in ~32 Ki different bucket
// regions, the lines leave L2 partially written, and HBM sees ~10x the payload (PMC WRITE_SIZE, round 1).  Here the entries are
// first PARTITIONED by the high bits of their bucket id into runs small enough (~12 Ki entries, ~48 KiB of output) that the
// final counting sort of a run -- one workgroup, all of the run's buckets -- scatters inside a region L2 holds and merges:
//   msm_digit_records   scalar -> W (key, value) records: key = bucket id (0xffffffff for a zero digit), value = entry | sign
//   msm_radix_count     per run and per slice of 8 192 records: LDS histogram of the next `bits` key bits -> sub-run sizes
//   (scan_*)            exclusive scan of the sub-run sizes = sub-run offsets (sub-runs of run j are consecutive inside run j)
//   msm_radix_scatter   same slices: rank in LDS, stage the slice sorted by digit, write every sub-run's share lane-adjacent
//   msm_radix_final     one workgroup per final run: histogram of its 2^rbits buckets, prefix = the buckets' offsets (written
//                       to `offsets` directly: no global count / scan over buckets), then the entries in bucket order
// One or two partition levels of <= 8 bits each; the record arrays ping-pong.
constexpr uint32_t RADIX_SLICE = 8192, RADIX_PER_LANE = RADIX_SLICE / 1024, RADIX_MAX_DIGITS = 256, RADIX_EMPTY = 0xffffffffu;

#ifdef BP_EXPERIMENT      // records-first level 1 of the two-level sort (BP_MSM_SORT=1; NAF digits): the shipped build takes level 1 from the scalars
__global__ void __launch_bounds__(256) msm_digit_records(const fr_t* __restrict__ scalars, int fmt, MsmPlan plan, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals, uint32_t* __restrict__ status) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plan.n) return;
  fr_t k = scalars[i];
  if (fmt == 1) {
    Fr::from_mont(k, k);
  } else {
    fr_t t;
    if (!big_sub(t, k, Fr::modulus())) atomicOr(status, 1u);
  }
  uint32_t kp[10];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    carry += (uint64_t)(j < 8 ? k.l[j] : 0u) + plan.bias[j];
    kp[j] = (uint32_t)carry;
    carry >>= 32;
  }
  kp[9] = 0;
  const uint32_t c = plan.c, mask = (1u << c) - 1u, half = 1u << (c - 1);
  for (uint32_t w = 0; w < plan.W; w++) {
    uint32_t o = c * w, word = o >> 5, sh = o & 31;
    uint64_t two = (uint64_t)kp[word] | ((uint64_t)kp[word + 1] << 32);
    int32_t d = (int32_t)((uint32_t)(two >> sh) & mask) - (w + 1 < plan.W ? (int32_t)half : 0);
    const size_t at = (size_t)w * plan.n + i;
    keys[at] = d == 0 ? RADIX_EMPTY : w * plan.wbuckets + digit_bucket(d);
    vals[at] = (i + w * plan.wpoints) | (d < 0 ? 0x80000000u : 0u);
  }
}

#endif  // BP_EXPERIMENT
#ifdef BP_EXPERIMENT      // every-position tables with NAF digits (bp_srs_precompute(h, 256 + w))
// Width-w NAF records of every scalar (plan.naf = w): slot s of scalar i at [s * n + i].  With E = (k >> pos) + carry:
// E even -> next position; E odd -> e = E mod 2^w, digit d = e (carry 0) or e - 2^w (carry 1) when e >= 2^(w-1), pos += w.
// k < 2^255, so every digit sits at a position <= 255 and there are at most 255 / w + 1 of them (= plan.W slots).
__global__ void __launch_bounds__(256) msm_naf_records(const fr_t* __restrict__ scalars, int fmt, MsmPlan plan, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, uint32_t* __restrict__ status) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plan.n) return;
  fr_t k = scalars[i];
  if (fmt == 1) {
    Fr::from_mont(k, k);
  } else {
    fr_t t;
    if (!big_sub(t, k, Fr::modulus())) atomicOr(status, 1u);
  }
  uint32_t kw[10];
#pragma unroll
  for (int j = 0; j < 8; j++) kw[j] = k.l[j];
  kw[8] = 0;
  kw[9] = 0;
  const uint32_t w = plan.naf, mask = (1u << w) - 1u, half = 1u << (w - 1);
  uint32_t pos = 0, carry = 0, slot = 0;
  while (pos < 256 && slot < plan.W) {
    // next position >= pos whose bit differs from the carry (there E is odd); none below 256: the rest of E is zero
    const uint32_t flip = carry ? 0xffffffffu : 0u;
    uint32_t word = pos >> 5, x = ((kw[word] ^ flip) >> (pos & 31));
    uint32_t p = pos;
    if (x) {
      p += __ffs(x) - 1;
    } else {
      p = (word + 1) << 5;
      for (word++; word < 8 && (kw[word] ^ flip) == 0; word++) p += 32;
      if (word >= 8) {
        if (!carry) break;                    // carry == 0: k has no bits left.  carry == 1 reaches here only past bit 255 (bit 255 of k is 0)
        p = 256;
      } else {
        p += __ffs(kw[word] ^ flip) - 1;
      }
    }
    if (p >= 256) break;
    const uint32_t wd = p >> 5, sh = p & 31;
    const uint64_t two = (uint64_t)kw[wd] | ((uint64_t)kw[wd + 1] << 32);
    const uint32_t e = ((uint32_t)(two >> sh) & mask) + carry;        // odd, < 2^w
    const bool neg = e >= half;
    const uint32_t mag = neg ? (1u << w) - e : e;                     // |d|, odd, < 2^(w-1)
    carry = neg ? 1u : 0u;
    const size_t at = (size_t)slot * plan.n + i;
    keys[at] = (mag - 1) >> 1;
    vals[at] = (i + p * plan.wpoints) | (neg ? 0x80000000u : 0u);
    slot++;
    pos = p + w;
  }
  for (; slot < plan.W; slot++) keys[(size_t)slot * plan.n + i] = RADIX_EMPTY;
}

#endif  // BP_EXPERIMENT
// run r = records [run_off[r], run_off[r + 1]); the workgroups blockIdx.x, blockIdx.x + gridDim.x, ... take its slices
__global__ void __launch_bounds__(1024) msm_radix_count(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ run_off, uint32_t shift,
                                                        uint32_t bits, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t h[RADIX_MAX_DIGITS];
  const uint32_t run = blockIdx.y, nd = 1u << bits, lo = run_off[run], hi = run_off[run + 1];
  if (lo + (uint64_t)blockIdx.x * RADIX_SLICE >= hi) return;
  if (threadIdx.x < nd) h[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t base = lo + (uint64_t)blockIdx.x * RADIX_SLICE; base < hi; base += (uint64_t)gridDim.x * RADIX_SLICE) {
    const uint32_t end = base + RADIX_SLICE < hi ? (uint32_t)base + RADIX_SLICE : hi;
    for (uint32_t j = (uint32_t)base + threadIdx.x; j < end; j += blockDim.x) {
      const uint32_t key = keys[j];
      if (key != RADIX_EMPTY) atomicAdd(&h[(key >> shift) & (nd - 1)], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < nd && h[threadIdx.x]) atomicAdd(&cnt[run * nd + threadIdx.x], h[threadIdx.x]);
}

extern __shared__ uint32_t msm_radix_lds[];       // st_key[RADIX_SLICE] | st_val[RADIX_SLICE] | st_d[RADIX_SLICE bytes]
__global__ void __launch_bounds__(1024) msm_radix_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                          const uint32_t* __restrict__ run_off, uint32_t shift, uint32_t bits,
                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t h[RADIX_MAX_DIGITS], lbase[RADIX_MAX_DIGITS], gbase[RADIX_MAX_DIGITS], scan16[16];
  uint32_t* st_key = msm_radix_lds;
  uint32_t* st_val = st_key + RADIX_SLICE;
  uint8_t* st_d = reinterpret_cast<uint8_t*>(st_val + RADIX_SLICE);
  const uint32_t run = blockIdx.y, nd = 1u << bits, lo = run_off[run], hi = run_off[run + 1];
  for (uint64_t base = lo + (uint64_t)blockIdx.x * RADIX_SLICE; base < hi; base += (uint64_t)gridDim.x * RADIX_SLICE) {   // uniform over the workgroup
    if (threadIdx.x < nd) h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t key[RADIX_PER_LANE], val[RADIX_PER_LANE], rank[RADIX_PER_LANE];
#pragma unroll
    for (uint32_t j = 0; j < RADIX_PER_LANE; j++) {
      const uint64_t at = base + j * 1024 + threadIdx.x;
      key[j] = at < hi ? keys[at] : RADIX_EMPTY;
      if (key[j] != RADIX_EMPTY) {
        val[j] = vals[at];
        rank[j] = atomicAdd(&h[(key[j] >> shift) & (nd - 1)], 1u);
      }
    }
    __syncthreads();
    const uint32_t mine = threadIdx.x < nd ? h[threadIdx.x] : 0u;
    const uint32_t ex = block_exclusive_scan_1024(mine, scan16);
    if (threadIdx.x < nd) {
      lbase[threadIdx.x] = ex;
      gbase[threadIdx.x] = mine ? atomicAdd(&cursor[run * nd + threadIdx.x], mine) : 0u;
    }
    __syncthreads();
    const uint32_t live = lbase[nd - 1] + h[nd - 1];
#pragma unroll
    for (uint32_t j = 0; j < RADIX_PER_LANE; j++) {
      if (key[j] != RADIX_EMPTY) {
        const uint32_t d = (key[j] >> shift) & (nd - 1), pos = lbase[d] + rank[j];
        st_key[pos] = key[j];
        st_val[pos] = val[j];
        st_d[pos] = (uint8_t)d;
      }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < live; j += blockDim.x) {
      const uint32_t d = st_d[j], g = gbase[d] + (j - lbase[d]);
      keys_out[g] = st_key[j];
      vals_out[g] = st_val[j];
    }
    __syncthreads();
  }
}

// Records of the final runs come in two forms.  Two arrays (keys = bucket id, vals = entry | sign << 31): the two-level sort above
// and every case the packed form cannot hold.  PACKED: one word per record, (bucket's low rbits bits) << vb | sign << (vb - 1) | entry
// -- the high bits of the bucket id are the run's number, so they need not travel (msm_part_scatter; half the bytes).
template <bool PACKED>
struct RunRecords {
  const uint32_t* __restrict__ keys;     // PACKED: the records
  const uint32_t* __restrict__ vals;     // PACKED: unused
  uint32_t mask, vb;                     // mask = 2^rbits - 1
  __device__ __forceinline__ uint32_t load(uint32_t j) const { return keys[j]; }
  __device__ __forceinline__ uint32_t low(uint32_t rec) const { return PACKED ? rec >> vb : rec & mask; }     // bucket inside the run
  __device__ __forceinline__ uint32_t val(uint32_t j, uint32_t rec) const {
    if (!PACKED) return vals[j];
    return (rec & ((1u << (vb - 1)) - 1u)) | (((rec >> (vb - 1)) & 1u) << 31);
  }
};

// One workgroup per final run (grid-stride).  The run's keys share their high bits; its 2^rbits buckets are
// [run << rbits, (run + 1) << rbits).  Writes offsets[bucket] for those (only below `total`) and the run's entries in bucket
// order; the scattered 4-byte stores stay inside the run's own ~48 KiB of `sorted`, which L2 merges into whole lines.
// keys may still hold RADIX_EMPTY records when no partition level ran (single run, two-array form): they are skipped.
// Runs longer than RADIX_LONG_RUN records (skewed scalars; the narrow top window, whose few buckets collect n entries) are
// not sorted by one workgroup: they are queued in long_list and handled slice-parallel by the msm_radix_long_* kernels
// (zero_counts != null: the queueing workgroup clears the run's bucket counters for them).
constexpr uint32_t RADIX_LONG_RUN = 1u << 16;
template <bool PACKED>
__global__ void __launch_bounds__(1024) msm_radix_final(RunRecords<PACKED> recs, const uint32_t* __restrict__ run_off, uint32_t n_runs, uint32_t rbits,
                                                        uint32_t total, uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted,
                                                        uint32_t* __restrict__ long_n, uint32_t* __restrict__ long_list,
                                                        uint32_t* __restrict__ zero_counts) {
  __shared__ uint32_t scan16[16];
  __shared__ uint32_t carry;
  const uint32_t nb = 1u << rbits;
  for (uint32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
    const uint32_t lo = run_off[run], hi = run_off[run + 1];
    if (hi - lo > RADIX_LONG_RUN && n_runs > 1) {          // uniform over the workgroup
      if (threadIdx.x == 0) long_list[atomicAdd(long_n, 1u)] = run;
      if (zero_counts)
        for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x)
          if ((((uint64_t)run << rbits) + b) < total) zero_counts[((size_t)run << rbits) + b] = 0;
      continue;
    }
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) msm_lds_hist[b] = 0;
    __syncthreads();
    for (uint32_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
      const uint32_t key = recs.load(j);
      if (PACKED || key != RADIX_EMPTY) atomicAdd(&msm_lds_hist[recs.low(key)], 1u);
    }
    __syncthreads();
    // exclusive prefix over the nb bucket sizes, 1024 at a time; the prefix replaces the size in LDS (it becomes the cursor).
    // Without a partition level the run still holds its zero-digit records: output positions start at `lo` and close up.
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += blockDim.x) {
      const uint32_t b = b0 + threadIdx.x, v = b < nb ? msm_lds_hist[b] : 0u;
      const uint32_t ex = block_exclusive_scan_1024(v, scan16) + carry;
      if (b < nb) {
        msm_lds_hist[b] = ex;
        const uint64_t bucket = ((uint64_t)run << rbits) + b;
        if (bucket < total) offsets[bucket] = lo + ex;
      }
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) carry = ex + v;
      __syncthreads();
    }
    if (run == n_runs - 1 && threadIdx.x == 0) offsets[total] = lo + carry;
    for (uint32_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
      const uint32_t key = recs.load(j);
      if (PACKED || key != RADIX_EMPTY) sorted[lo + atomicAdd(&msm_lds_hist[recs.low(key)], 1u)] = recs.val(j, key);
    }
    __syncthreads();
  }
}

// long runs, slice-parallel: count (+ prefix = offsets and cursors of the run's buckets, by the run's last workgroup) -> scatter.
// grid (slices, lanes of the list); workgroup (x, y) takes slices x, x + gridDim.x, ... of the long runs y, y + gridDim.y, ...
// Round 6: two launches instead of three.  Every long run k is visited by exactly gridDim.x workgroups (those with blockIdx.y = k mod
// gridDim.y, with or without slices of their own); each takes a ticket of the run when its slices are counted, and the one that takes
// the last ticket scans the run's bucket sizes (what msm_radix_long_prefix did in a launch of its own).  ticket[k]: zero before the
// launch, zero again after it.  The sizes are only ever touched by device-scope atomics (as in msm_part_count), so the last workgroup
// reads them with an addition of zero and needs no cache maintenance.
template <bool PACKED>
__global__ void __launch_bounds__(1024) msm_radix_long_count(RunRecords<PACKED> recs, const uint32_t* __restrict__ run_off, uint32_t n_runs, uint32_t rbits,
                                                             uint32_t total, const uint32_t* __restrict__ long_n, const uint32_t* __restrict__ long_list,
                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors,
                                                             uint32_t* __restrict__ ticket) {
  __shared__ uint32_t scan16[16];
  __shared__ uint32_t carry, last_s;
  const uint32_t nb = 1u << rbits, n_long = *long_n;
  for (uint32_t k = blockIdx.y; k < n_long; k += gridDim.y) {
    const uint32_t run = long_list[k], lo = run_off[run], hi = run_off[run + 1];
    for (uint64_t base = lo + (uint64_t)blockIdx.x * RADIX_SLICE; base < hi; base += (uint64_t)gridDim.x * RADIX_SLICE) {
      for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) msm_lds_hist[b] = 0;
      __syncthreads();
      const uint32_t end = base + RADIX_SLICE < hi ? (uint32_t)base + RADIX_SLICE : hi;
      for (uint32_t j = (uint32_t)base + threadIdx.x; j < end; j += blockDim.x) atomicAdd(&msm_lds_hist[recs.low(recs.load(j))], 1u);
      __syncthreads();
      for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t v = msm_lds_hist[b];
        if (v) atomicAdd(&counts[((size_t)run << rbits) + b], v);
      }
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this workgroup's additions have been performed before it takes its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
      last_s = atomicAdd(&ticket[k], 1u) == gridDim.x - 1 ? 1u : 0u;
      carry = 0;
    }
    __syncthreads();
    if (!last_s) continue;                                 // uniform over the workgroup
    if (threadIdx.x == 0) ticket[k] = 0;                   // as found, for the next launch
    for (uint32_t b0 = 0; b0 < nb; b0 += blockDim.x) {
      const uint32_t b = b0 + threadIdx.x;
      const uint64_t bucket = ((uint64_t)run << rbits) + b;
      const uint32_t v = (b < nb && bucket < total) ? atomicAdd(&counts[bucket], 0u) : 0u;
      const uint32_t ex = block_exclusive_scan_1024(v, scan16) + carry;
      if (b < nb && bucket < total) {
        offsets[bucket] = lo + ex;
        cursors[bucket] = lo + ex;
      }
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) carry = ex + v;
      __syncthreads();
    }
    if (run == n_runs - 1 && threadIdx.x == 0) offsets[total] = lo + carry;
    __syncthreads();
  }
}
template <bool PACKED>
__global__ void __launch_bounds__(1024) msm_radix_long_scatter(RunRecords<PACKED> recs, const uint32_t* __restrict__ run_off, uint32_t rbits,
                                                               const uint32_t* __restrict__ long_n, const uint32_t* __restrict__ long_list,
                                                               uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
  const uint32_t nb = 1u << rbits, n_long = *long_n;
  for (uint32_t k = blockIdx.y; k < n_long; k += gridDim.y) {
    const uint32_t run = long_list[k], lo = run_off[run], hi = run_off[run + 1];
    for (uint64_t base = lo + (uint64_t)blockIdx.x * RADIX_SLICE; base < hi; base += (uint64_t)gridDim.x * RADIX_SLICE) {
      for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) msm_lds_hist[b] = 0;
      __syncthreads();
      const uint32_t end = base + RADIX_SLICE < hi ? (uint32_t)base + RADIX_SLICE : hi;
      for (uint32_t j = (uint32_t)base + threadIdx.x; j < end; j += blockDim.x) atomicAdd(&msm_lds_hist[recs.low(recs.load(j))], 1u);
      __syncthreads();
      for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
        const uint32_t v = msm_lds_hist[b];
        if (v) msm_lds_hist[b] = atomicAdd(&cursors[((size_t)run << rbits) + b], v);
      }
      __syncthreads();
      for (uint32_t j = (uint32_t)base + threadIdx.x; j < end; j += blockDim.x) {
        const uint32_t rec = recs.load(j);
        sorted[atomicAdd(&msm_lds_hist[recs.low(rec)], 1u)] = recs.val(j, rec);
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------- 4d. partition sort (one level, digits recomputed, no digit or record pass)
// The default bucket sort.  The histogram sort of section 4 pays one global atomic per (workgroup, bucket) it touches -- ~10^7 of
// them at 2^20 points, twice -- and isolated 4-byte stores; the two-level sort of 4c moves 8-byte records three times.  Here:
//   msm_part_count     a workgroup per slice of scalars: Montgomery -> canonical -> digits in registers, LDS histogram of the
//                      entries' PARTITIONS (the high pb bits of the bucket id, <= 2^12), one global atomic per (workgroup,
//                      partition); the workgroup that finishes last (a ticket) scans the 2^pb sizes = the final runs' offsets
//   msm_part_scatter   the same slices, digits recomputed (32 B of scalar re-read instead of 128 B of records written and read):
//                      rank in LDS, reserve the slice's range in every partition, stage sorted by partition, write each
//                      partition's share lane-adjacent
//   msm_radix_final    one workgroup per final run, as in 4c (packed records where they fit: RunRecords)
// No digit array, no record arrays before the final runs, no scan launches.
constexpr uint32_t PART_MAX_BITS = 12, PART_MAX = 1u << PART_MAX_BITS;

// every (bucket id, entry | sign << 31) of scalar i, in window (or NAF slot) order.  status != null: canonical-bytes inputs are
// range-checked (Scalar::from_bytes rejects values >= q, scalar.rs:264-288)
// the scalar as the canonical integer the digits are cut from
__device__ __forceinline__ void msm_scalar_canon(fr_t& k, int fmt, uint32_t* __restrict__ status) {
  if (fmt == 1) {
    Fr::from_mont(k, k);                           // msm.rs:126: scalar.to_bytes() = canonical integer
  } else if (status) {
    fr_t t;
    if (!big_sub(t, k, Fr::modulus())) atomicOr(status, 1u);
  }
}
template <class F>
__device__ __forceinline__ void msm_canon_entries(const fr_t& k, uint32_t i, uint32_t bucket_base, const MsmPlan& plan, F&& emit) {
  if (plan.naf) {                                  // as msm_naf_records
    uint32_t kw[10];
#pragma unroll
    for (int j = 0; j < 8; j++) kw[j] = k.l[j];
    kw[8] = 0;
    kw[9] = 0;
    const uint32_t w = plan.naf, mask = (1u << w) - 1u, half = 1u << (w - 1);
    uint32_t pos = 0, carry = 0, slot = 0;
    while (pos < 256 && slot < plan.W) {
      const uint32_t flip = carry ? 0xffffffffu : 0u;
      uint32_t word = pos >> 5, x = ((kw[word] ^ flip) >> (pos & 31));
      uint32_t p = pos;
      if (x) {
        p += __ffs(x) - 1;
      } else {
        p = (word + 1) << 5;
        for (word++; word < 8 && (kw[word] ^ flip) == 0; word++) p += 32;
        if (word >= 8) {
          if (!carry) break;
          p = 256;
        } else {
          p += __ffs(kw[word] ^ flip) - 1;
        }
      }
      if (p >= 256) break;
      const uint32_t wd = p >> 5, sh = p & 31;
      const uint64_t two = (uint64_t)kw[wd] | ((uint64_t)kw[wd + 1] << 32);
      const uint32_t e = ((uint32_t)(two >> sh) & mask) + carry;
      const bool neg = e >= half;
      const uint32_t mag = neg ? (1u << w) - e : e;
      carry = neg ? 1u : 0u;
      emit(bucket_base + ((mag - 1) >> 1), (i + p * plan.wpoints) | (neg ? 0x80000000u : 0u));
      slot++;
      pos = p + w;
    }
    return;
  }
  if (plan.radix) {                                // digits in radix R (MsmPlan::radix, msm_digits.hpp), top window first
    msm_radix_digits(k.l, plan.radix, plan.radix_m, plan.bias, plan.W, [&](uint32_t w, int32_t d) {
      if (d != 0) emit(bucket_base + digit_bucket(d), (i + w * plan.wpoints) | (d < 0 ? 0x80000000u : 0u));
    });
    return;
  }
  uint32_t kp[10];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    carry += (uint64_t)(j < 8 ? k.l[j] : 0u) + plan.bias[j];
    kp[j] = (uint32_t)carry;
    carry >>= 32;
  }
  kp[9] = 0;
  const uint32_t c = plan.c, mask = (1u << c) - 1u, half = 1u << (c - 1);
  for (uint32_t w = 0; w < plan.W; w++) {
    const uint32_t o = c * w, word = o >> 5, sh = o & 31;
    const uint64_t two = (uint64_t)kp[word] | ((uint64_t)kp[word + 1] << 32);
    const int32_t d = (int32_t)((uint32_t)(two >> sh) & mask) - (w + 1 < plan.W ? (int32_t)half : 0);
    if (d != 0) emit(bucket_base + w * plan.wbuckets + digit_bucket(d), (i + w * plan.wpoints) | (d < 0 ? 0x80000000u : 0u));
  }
}
template <class F>
__device__ __forceinline__ void msm_scalar_entries(fr_t k, uint32_t i, uint32_t bucket_base, int fmt, const MsmPlan& plan,
                                                   uint32_t* __restrict__ status, F&& emit) {
  msm_scalar_canon(k, fmt, status);
  msm_canon_entries(k, i, bucket_base, plan, emit);
}

// element x of the pipeline's index space (vector x / n, position x % n): its scalar, position and bucket base; false: no such element
__device__ __forceinline__ bool msm_slice_scalar(const MsmScalars& sc, uint64_t x, const MsmPlan& plan, fr_t& k, uint32_t& i, uint32_t& base) {
  uint32_t j = 0;
  i = (uint32_t)x;
  if (plan.J > 1) {
    j = (uint32_t)(x / plan.n);
    i = (uint32_t)(x - (uint64_t)j * plan.n);
  }
  if (i >= plan.nj[j]) return false;
  const fr_t* __restrict__ src = sc.p[0];
  if (j == 1) src = sc.p[1];
  if (j == 2) src = sc.p[2];
  if (j == 3) src = sc.p[3];
  k = src[i];
  base = j * plan.B;
  return true;
}
// the entries of the scalars [lo, hi) of the pipeline's index space
template <class F>
__device__ __forceinline__ void msm_slice_entries(const MsmScalars& sc, uint64_t lo, uint64_t hi, int fmt, const MsmPlan& plan,
                                                  uint32_t* __restrict__ status, F&& emit) {
  for (uint64_t x = lo + threadIdx.x; x < hi; x += blockDim.x) {
    fr_t k;
    uint32_t i, base;
    if (!msm_slice_scalar(sc, x, plan, k, i, base)) continue;
    msm_scalar_entries(k, i, base, fmt, plan, status, emit);
  }
}

// ctl: [0] ticket of finished workgroups, [1 .. 2^pb] partition sizes (zero before the launch).  slice: scalars per workgroup.
// run_off[0 .. 2^pb] and cursor[0 .. 2^pb) are written by the workgroup that takes the last ticket.
__global__ void __launch_bounds__(1024) msm_part_count(MsmScalars scalars, int fmt, MsmPlan plan, uint32_t slice, uint32_t pb,
                                                       uint32_t rbits, uint32_t* __restrict__ ctl, uint32_t* __restrict__ run_off,
                                                       uint32_t* __restrict__ cursor, uint32_t* __restrict__ status) {
  __shared__ uint32_t h[PART_MAX], scan16[16], carry_s, last_s;
  const uint32_t P = 1u << pb;
  uint32_t* cnt = ctl + 1;
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) h[p] = 0;
  __syncthreads();
  const uint64_t all = (uint64_t)plan.J * plan.n, lo = (uint64_t)blockIdx.x * slice, hi = lo + slice < all ? lo + slice : all;
  msm_slice_entries(scalars, lo, hi, fmt, plan, status, [&](uint32_t bucket, uint32_t) { atomicAdd(&h[bucket >> rbits], 1u); });
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x)
    if (h[p]) atomicAdd(&cnt[p], h[p]);
  // Every wave's additions have been performed (vmcnt = 0) before the workgroup takes its ticket.  The sizes are only ever
  // touched by device-scope atomics -- here and, below, read by one (an addition of zero) -- so they need no cache maintenance:
  // a release fence per lane (buffer_wbl2) cost this kernel 270 of its 300 us.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    last_s = atomicAdd(&ctl[0], 1u) == gridDim.x - 1 ? 1u : 0u;
    carry_s = 0;
  }
  __syncthreads();
  if (!last_s) return;
  for (uint32_t p0 = 0; p0 < P; p0 += blockDim.x) {
    const uint32_t p = p0 + threadIdx.x;
    const uint32_t v = p < P ? atomicAdd(&cnt[p], 0u) : 0u;
    const uint32_t ex = block_exclusive_scan_1024(v, scan16) + carry_s;
    if (p < P) {
      run_off[p] = ex;
      cursor[p] = ex;
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_s = ex + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) run_off[P] = carry_s;
}

// LDS (dynamic): h[P] | lbase[P] | gbase[P] | records[cap] (PACKED) or keys[cap] | vals[cap] | FLAT: partition of every staged
// record, u16[cap]   (cap = slice * W, P = 2^pb)
// FLAT picks the write-out.  false: partition-major, 16 lanes per partition -- for shares of >= 8 records per partition (c <= 16:
// 2^10 partitions).  true: one lane per staged record, consecutive lanes = consecutive records of a partition -- for short
// shares (c = 20 at 2^20: 2^12 partitions, ~3 records per slice each), where the partition-major loop leaves 13 lanes in 16 idle.
extern __shared__ uint32_t msm_part_lds[];
template <bool PACKED, bool FLAT>
__global__ void __launch_bounds__(1024) msm_part_scatter(MsmScalars scalars, int fmt, MsmPlan plan, uint32_t slice, uint32_t pb,
                                                         uint32_t rbits, uint32_t vb, uint32_t cap, uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t scan16[16], carry_s;
  const uint32_t P = 1u << pb, rmask = (1u << rbits) - 1u;
  uint32_t* h = msm_part_lds;
  uint32_t* lbase = h + P;
  uint32_t* gbase = lbase + P;
  uint32_t* st_key = gbase + P;
  uint32_t* st_val = st_key + cap;
  uint16_t* st_part = reinterpret_cast<uint16_t*>(PACKED ? st_val : st_val + cap);
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) h[p] = 0;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const uint64_t all = (uint64_t)plan.J * plan.n, lo = (uint64_t)blockIdx.x * slice, hi = lo + slice < all ? lo + slice : all;
  // One scalar per lane (the usual shape: slices of <= 1 024 scalars under 1 024 lanes): the canonical integer stays in registers between
  // the counting pass and the staging pass -- the second Montgomery conversion of round 5's scatter is gone (uniform over the workgroup)
  const bool one = slice <= blockDim.x;
  fr_t kc;
  uint32_t ic = 0, bc = 0;
  bool live = false;
  if (one && lo + threadIdx.x < hi) {
    live = msm_slice_scalar(scalars, lo + threadIdx.x, plan, kc, ic, bc);
    if (live) msm_scalar_canon(kc, fmt, nullptr);
  }
  if (one) {
    if (live) msm_canon_entries(kc, ic, bc, plan, [&](uint32_t bucket, uint32_t) { atomicAdd(&h[bucket >> rbits], 1u); });
  } else {
    msm_slice_entries(scalars, lo, hi, fmt, plan, nullptr, [&](uint32_t bucket, uint32_t) { atomicAdd(&h[bucket >> rbits], 1u); });
  }
  __syncthreads();
  // this slice's share of every partition: place in the staging area (lbase) and in the partition's final run (gbase).  The
  // reservations (one returning global atomic per partition the slice touches) are all issued before the scan's barriers, so
  // their latencies overlap each other and the scan instead of adding up over the 2^pb / lanes rounds
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
    const uint32_t v = h[p];
    gbase[p] = v ? atomicAdd(&cursor[p], v) : 0u;
  }
  for (uint32_t p0 = 0; p0 < P; p0 += blockDim.x) {
    const uint32_t p = p0 + threadIdx.x;
    const uint32_t v = p < P ? h[p] : 0u;
    const uint32_t ex = block_exclusive_scan_1024(v, scan16) + carry_s;
    if (p < P) {
      lbase[p] = ex;
      h[p] = 0;
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_s = ex + v;
    __syncthreads();
  }
  auto stage = [&](uint32_t bucket, uint32_t val) {
    const uint32_t part = bucket >> rbits, pos = lbase[part] + atomicAdd(&h[part], 1u);
    if (PACKED) {
      st_key[pos] = ((bucket & rmask) << vb) | ((val >> 31) << (vb - 1)) | (val & 0x7fffffffu);
    } else {
      st_key[pos] = bucket;
      st_val[pos] = val;
    }
    if (FLAT) st_part[pos] = (uint16_t)part;
  };
  if (one) {
    if (live) msm_canon_entries(kc, ic, bc, plan, stage);
  } else {
    msm_slice_entries(scalars, lo, hi, fmt, plan, nullptr, stage);
  }
  __syncthreads();
  if (FLAT) {
    const uint32_t live = carry_s;
    for (uint32_t j = threadIdx.x; j < live; j += blockDim.x) {
      const uint32_t p = st_part[j], g = gbase[p] + (j - lbase[p]);
      keys_out[g] = st_key[j];
      if (!PACKED) vals_out[g] = st_val[j];
    }
  } else {      // 16 lanes per partition, four partitions per wave at a time
    const uint32_t sub = threadIdx.x >> 4, l16 = threadIdx.x & 15, n_sub = blockDim.x >> 4;
    for (uint32_t p = sub; p < P; p += n_sub) {
      const uint32_t cnt = h[p], src = lbase[p], dst = gbase[p];
      for (uint32_t e = l16; e < cnt; e += 16) {
        keys_out[dst + e] = st_key[src + e];
        if (!PACKED) vals_out[dst + e] = st_val[src + e];
      }
    }
  }
}

// ---------------------------------------------------------------- 5. accumulate
__device__ __forceinline__ g1_affine load_affine(const g1_affine* __restrict__ p) {
  g1_affine r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v[6];
#pragma unroll
  for (int j = 0; j < 6; j++) v[j] = q[j];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    r.x.l[4 * j] = v[j].x; r.x.l[4 * j + 1] = v[j].y; r.x.l[4 * j + 2] = v[j].z; r.x.l[4 * j + 3] = v[j].w;
    r.y.l[4 * j] = v[j + 3].x; r.y.l[4 * j + 1] = v[j + 3].y; r.y.l[4 * j + 2] = v[j + 3].z; r.y.l[4 * j + 3] = v[j + 3].w;
  }
  return r;
}
// point of the unsaturated SRS copy: 112 B = 7 x dwordx4 at the head of its 128-byte slot (one cache line; at a 112-byte pitch
// three gathers in four straddled two lines: accumulate 2.10 -> 1.94 ms at 2^20, 36.9 -> 32.2 ms at 2^24)
__device__ __forceinline__ g1_affine28 load_affine28(const g1_affine28* __restrict__ p) {
  g1_affine28 r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v[7];
#pragma unroll
  for (int j = 0; j < 7; j++) v[j] = q[j];
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < 7; j++) { w[4 * j] = v[j].x; w[4 * j + 1] = v[j].y; w[4 * j + 2] = v[j].z; w[4 * j + 3] = v[j].w; }
#pragma unroll
  for (int j = 0; j < N28; j++) { r.x.l[j] = w[j]; r.y.l[j] = w[N28 + j]; }
  return r;
}
// Projective accumulators travel between the MSM kernels as 44 words (x | y | z, 14 limbs each, + 2 pad)
// = 11 x dwordx4, lazy limbs untouched: no conversion or reduction on the flush path.
struct proj28_slot { uint4 q[11]; };
__device__ __forceinline__ void store_proj28(proj28_slot* __restrict__ dst, const g1_proj28& p) {
  uint32_t w[PROJ28_WORDS];
#pragma unroll
  for (int j = 0; j < N28; j++) { w[j] = p.x.l[j]; w[N28 + j] = p.y.l[j]; w[2 * N28 + j] = p.z.l[j]; }
  w[42] = 0; w[43] = 0;
#pragma unroll
  for (int j = 0; j < 11; j++) dst->q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
__device__ __forceinline__ g1_proj28 load_proj28(const proj28_slot* __restrict__ src) {
  uint32_t w[PROJ28_WORDS];
#pragma unroll
  for (int j = 0; j < 11; j++) { uint4 v = src->q[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
  g1_proj28 p;
#pragma unroll
  for (int j = 0; j < N28; j++) { p.x.l[j] = w[j]; p.y.l[j] = w[N28 + j]; p.z.l[j] = w[2 * N28 + j]; }
  return p;
}
// z limbs all zero: the identity exactly as g1_identity28() writes it (an empty bucket, or a sum of such).  A tree over mostly
// empty buckets (8-bit or sparse scalars on 2^19 buckets) skips the additions of its empty regions on this test: waves whose
// lanes all see two such operands do not execute the addition at all.  (A sum that happens to be the identity with non-zero lazy
// limbs is simply added like any other point.)
__device__ __forceinline__ bool is_blank28(const g1_proj28& p) {
  uint32_t z = 0;
#pragma unroll
  for (int j = 0; j < N28; j++) z |= p.z.l[j];
  return z == 0;
}

// Cooperative complete addition (g1_28.hpp): the COOP consecutive lanes of a group pass the same two points; the six
// products and the three output coordinates travel between the lanes by shuffles; every lane returns the full sum.
// ~1 600 instructions deep instead of ~6 600, at twice the work -- it pays only where fewer additions than lanes / 8 are
// pending (the upper levels of block_tree_sum28), not in the bulk loop.
constexpr uint32_t COOP = 8;
__device__ __forceinline__ g1_proj28 g1_add28_coop(const g1_proj28& a, const g1_proj28& b) {
  const uint32_t role = threadIdx.x & (COOP - 1);
  const CoopProd mine = g1_add28_coop_a(role, a, b);
  CoopProd p[6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
#pragma unroll
    for (int j = 0; j < N28; j++) p[r].l[j] = (uint32_t)__shfl((int)mine.l[j], r, COOP);
  }
  const C28 coord = g1_add28_coop_b(role, p);
  g1_proj28 out;
#pragma unroll
  for (int j = 0; j < N28; j++) {
    out.x.l[j] = (uint32_t)__shfl((int)coord.l[j], 0, COOP);
    out.y.l[j] = (uint32_t)__shfl((int)coord.l[j], 1, COOP);
    out.z.l[j] = (uint32_t)__shfl((int)coord.l[j], 2, COOP);
  }
  return out;
}

// SRS: reference Montgomery limbs (96 B) -> unsaturated copy (112 B in a 128-B slot), once per SRS
__global__ void __launch_bounds__(256) srs_to28(const g1_affine* __restrict__ in, size_t n, g1_affine28* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_affine28 r = g1_affine_to_28(load_affine(&in[i]));
  uint4* q = reinterpret_cast<uint4*>(&out[i]);
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < N28; j++) { w[j] = r.x.l[j]; w[N28 + j] = r.y.l[j]; }
#pragma unroll
  for (int j = 0; j < 7; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}

// Fixed-base window tables: table[w * n + i] = 2^(c w) * P_i for w < W, affine, unsaturated limbs.  One lane per point
// walks its doubling chain; the affine normalisations of a group of TABLE_GROUP rows share one field inversion
// (Montgomery's trick, as G1Projective::batch_normalize does, g1.rs:806-839).  Built once per SRS.
// Round 4: the whole chain runs on the 14 x 28-bit limbs of the hot copy (g1_double28, mul28, fp28_invert) instead of the saturated
// 12 x 32-bit form -- the kernel is nothing but field products, and a product is ~500 instructions there against ~700
// (profiles/r04_table_build_ab.txt).  Rows are stored canonical (canon28), as srs_to28 stores the points themselves.
constexpr int TABLE_GROUP = 16;
__device__ __forceinline__ M28 one_m28() {
  M28 r;
#pragma unroll
  for (int j = 0; j < N28; j++) r.l[j] = One28::limb(j);
  return r;
}
// radix != 0: row w = radix^w P (MsmPlan::radix) -- one small scalar multiplication per row (20 doublings + 3 additions for R = 0xD0000)
// instead of c doublings.
// (RADIX as a template parameter: with the double-and-add in the same body the power-of-two build went from 208 to 328 registers and 35.6 -> 38.0 ms at 2^20 points)
template <bool RADIX>
__global__ void __launch_bounds__(256) srs_window_tables(const g1_affine28* __restrict__ in28, size_t n, uint32_t c, uint32_t W, uint32_t radix,
                                                         g1_affine28* __restrict__ table) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const g1_affine28 a = load_affine28(&in28[i]);
  uint32_t nz = 0;
#pragma unroll
  for (int j = 0; j < N28; j++) nz |= a.x.l[j] | a.y.l[j];
  const bool inf = nz == 0;                             // stays the identity in every row; a point of prime order never doubles to it
  g1_proj28 p = g1_identity28();
  if (!inf) {
    p.x = widen28<C28>(a.x);
    p.y = widen28<C28>(a.y);
    p.z = widen28<C28>(one_m28());
  }
  g1_proj28 row[TABLE_GROUP];
  M28 prefix[TABLE_GROUP];                              // prefix[j] = z_0 z_1 ... z_j of the group
  const M28 one = one_m28();
  for (uint32_t w0 = 1; w0 < W; w0 += TABLE_GROUP) {
    const uint32_t cnt = W - w0 < (uint32_t)TABLE_GROUP ? W - w0 : (uint32_t)TABLE_GROUP;
    for (uint32_t j = 0; j < cnt; j++) {
      if (RADIX) g1_mul_small28(p, p, radix, 32 - __clz(radix));
      else
        for (uint32_t d = 0; d < c; d++) g1_double28(p, p);
      row[j] = p;
      prefix[j] = mul28(j == 0 ? one : prefix[j - 1], p.z);       // (the first product only brings z to the product's bounds)
    }
    M28 inv = fp28_invert(prefix[cnt - 1]);             // 0 -> 0 for the identity
    for (uint32_t j = cnt; j-- > 0;) {
      const M28 zinv = mul28(inv, j ? prefix[j - 1] : one);
      inv = mul28(inv, row[j].z);
      F28n x = canon28(mul28(row[j].x, zinv)), y = canon28(mul28(row[j].y, zinv));
      uint4* q = reinterpret_cast<uint4*>(&table[(size_t)(w0 + j) * n + i]);
      uint32_t wd[28];
#pragma unroll
      for (int t = 0; t < N28; t++) { wd[t] = inf ? 0u : x.l[t]; wd[N28 + t] = inf ? 0u : y.l[t]; }
#pragma unroll
      for (int t = 0; t < 7; t++) q[t] = make_uint4(wd[4 * t], wd[4 * t + 1], wd[4 * t + 2], wd[4 * t + 3]);
    }
  }
}

// entries per lane for a sorted list of M entries (<= plan.chunk, so the lanes and partial slots sized by the host suffice)
__device__ __forceinline__ uint32_t msm_lane_chunk(const MsmPlan& plan, uint32_t M) {
  if (!plan.lanes) return plan.chunk;
  // not below a quarter of the full chunk (nor below 32): a bucket that holds most of a skewed input (scalars 0 / 1: one bucket of n / 2 entries)
  // is cut into entries / chunk partials, which ONE workgroup of msm_fixup_long adds up -- chains of 4 made that 5 ms at 2^20
  const uint32_t c = (uint32_t)(((uint64_t)M + plan.lanes - 1) / plan.lanes);
  const uint32_t small = plan.chunk < 32 ? plan.chunk : 32u, floor_c = plan.chunk / 4 > small ? plan.chunk / 4 : small;
  return c < floor_c ? floor_c : c;
}
// largest g in [0, total) with offsets[g] <= p   (offsets is non-decreasing, offsets[0] = 0)
__device__ __forceinline__ uint32_t bucket_of(const uint32_t* __restrict__ offsets, uint32_t total, uint32_t p) {
  uint32_t lo = 0, hi = total;                 // invariant: offsets[lo] <= p < offsets[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

template <int WAVES>      // waves per SIMD the register allocator must leave room for
__global__ void __launch_bounds__(256, WAVES)
msm_accumulate(const g1_affine28* __restrict__ points, const uint32_t* __restrict__ sorted,
               const uint32_t* __restrict__ offsets, MsmPlan plan, proj28_slot* __restrict__ bucket_sum,
               proj28_slot* __restrict__ partial) {
  const uint32_t total = plan.total;
  const uint32_t M = offsets[total];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)msm_lane_chunk(plan, M));     // the same in every lane: keep it scalar
  const uint64_t p0_64 = (uint64_t)t * chunk;
  if (p0_64 >= M) return;
  const uint32_t p0 = (uint32_t)p0_64;
  const uint32_t p1 = (uint64_t)p0 + chunk < M ? p0 + chunk : M;

  // bucket g = [g_start, g_end); g_end2 = the end of bucket g + 1, fetched one boundary ahead: leaving a bucket then costs no
  // dependent load (the wave used to stall on offsets[g + 1] at every boundary -- four per lane at c = 20, where some lane of a
  // wave leaves a bucket in most iterations -- and to re-read offsets[g] for the "whole bucket" test)
  uint32_t g = bucket_of(offsets, total, p0);
  // (Round 6 tried to hand this bucket to msm_fixup_edges, which bisects for the same position again: ONE 4-byte store per lane here,
  // before the loop, cost the kernel 5 % -- 1.93 -> 2.03 ms at 2^20, 0.194 -> 0.207 at 2^16, three alternating rounds on one box -- against
  // 3 us saved there.  Also tried: the loop below re-ordered so that the flush holds no load (offsets[g + 3] fetched an iteration ahead)
  // and the compiler's wait behind the flush -- a vmcnt(0), i.e. a wait for the flush's own stores in 93 % of the iterations -- disappears:
  // 196 registers instead of 232 and 1.5-3 % SLOWER at 2^16, 2^20 and 2^24.  Both records: profiles/r06_tail_ab.txt.)
  uint32_t g_start = offsets[g], g_end = offsets[g + 1];
  uint32_t g_end2 = g + 2 <= total ? offsets[g + 2] : M;
  uint32_t run_start = p0;
  g1_proj28 acc = g1_identity28();
  // software pipeline, two deep: while entry p is added, the gather of entry p+1 is in flight AND the index of entry p+2 is
  // being fetched -- the gather's address is then in a register when the next iteration starts (with a one-deep pipeline
  // every iteration stalled on `sorted[p+1]` before it could issue the gather: a full memory latency per addition)
  uint32_t e_cur = sorted[p0];
  uint32_t e_ahead = p0 + 1 < p1 ? sorted[p0 + 1] : 0u;
  g1_affine28 q_next = load_affine28(&points[e_cur & 0x7fffffffu]);
  for (uint32_t p = p0; p < p1; p++) {
    if (p >= g_end) {                            // leave bucket g: flush its run [run_start, p)
      const bool complete = run_start == g_start;       // it ended at g_end by construction
      store_proj28(complete ? &bucket_sum[g] : &partial[2 * (size_t)t + (run_start == p0 ? 0 : 1)], acc);
      acc = g1_identity28();
      run_start = p;
      g_start = g_end;                            // the next bucket, or -- when that one is empty -- a binary search: with
      g++;                                        // skewed scalars (all equal: 13 non-empty buckets of 2^19) a linear walk over
      g_end = g_end2;                             // the empty buckets took 40 000 dependent loads per boundary lane (14 ms)
      if (p >= g_end) {
        g = bucket_of(offsets, total, p);
        g_start = offsets[g];
        g_end = offsets[g + 1];
      }
      g_end2 = g + 2 <= total ? offsets[g + 2] : M;
    }
    const uint32_t e = e_cur;
    const g1_affine28 q = q_next;
    if (p + 1 < p1) q_next = load_affine28(&points[e_ahead & 0x7fffffffu]);
    e_cur = e_ahead;
    if (p + 2 < p1) e_ahead = sorted[p + 2];
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < N28; j++) nz |= q.x.l[j] | q.y.l[j];
    if (nz) g1_add_mixed28(acc, q.x, pt_y_signed(q.y, (e >> 31) != 0));     // identity points (x = y = 0) add nothing
  }
  const bool complete = run_start == g_start && p1 == g_end;
  store_proj28(complete ? &bucket_sum[g] : &partial[2 * (size_t)t + (run_start == p0 ? 0 : 1)], acc);
}

// ---------------------------------------------------------------- blob (one process per GPU)
// bp_msm_g1_blob_device: the window sums / bit planes of one rank's MSM as a self-describing record in HBM, so the ranks'
// records can be all-gathered on the device and combined after a single device-to-host copy (bp_msm_blobs_combine).
// err / err_rank: a rank whose MSM failed BEFORE the collective (unknown handle, out of memory, a launch error) still takes part in
// the all-gather with a POISONED record -- magic set, n_planes = 0, err = its negative BP_ERR_* code, err_rank = its rank -- so that no
// peer is left waiting in ncclAllGather and every rank returns the same error (capi_comm.hip; the reference panics, it never blocks).
struct MsmBlobHeader {
  uint32_t magic, c, Wr, n_planes, tables, status, entries;
  int32_t err;
  uint32_t err_rank;
  uint32_t quads;      // table-free records: values per window (msm_planes_window_quads); 0 or 1: one window sum each
  uint32_t pad[6];
};
static_assert(sizeof(MsmBlobHeader) == 64, "blob header");
constexpr uint32_t MSM_BLOB_MAGIC = 0x424d5042u;      // "BPMB"
__global__ void __launch_bounds__(256) msm_write_blob(const proj28_slot* __restrict__ window_sum, MsmBlobHeader hdr, uint8_t* __restrict__ blob) {
  uint4* dst = reinterpret_cast<uint4*>(blob + sizeof(MsmBlobHeader));
  const uint4* src = reinterpret_cast<const uint4*>(window_sum);
  for (uint32_t i = threadIdx.x; i < hdr.n_planes * 11u; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0) {
    if (hdr.n_planes) {
      const uint32_t* tail = reinterpret_cast<const uint32_t*>(window_sum + hdr.n_planes);
      hdr.status = tail[0];             // scalar >= q seen by msm_digits
      hdr.entries = tail[1];
    }
    *reinterpret_cast<MsmBlobHeader*>(blob) = hdr;
  }
}

// Sum of n records of equal layout, slot by slot, on the device (after the ranks' all-gather): one cooperative group of
// COOP lanes per slot walks the n records.  Records that differ in layout leave magic = 0 in the output: the caller then
// combines the gathered records on the host instead.  grid ceil(n_planes / 8), 64 lanes.
__global__ void __launch_bounds__(64) msm_blob_sum(const uint8_t* __restrict__ blobs, uint32_t n_blobs, uint32_t blob_bytes,
                                                   uint8_t* __restrict__ out) {
  const MsmBlobHeader h0 = *reinterpret_cast<const MsmBlobHeader*>(blobs);
  bool same = h0.magic == MSM_BLOB_MAGIC;
  uint32_t status = 0, entries = 0, err_rank = 0;
  int32_t err = 0;
  for (uint32_t k = 0; k < n_blobs; k++) {
    const MsmBlobHeader h = *reinterpret_cast<const MsmBlobHeader*>(blobs + (size_t)k * blob_bytes);
    same = same && h.magic == h0.magic && h.c == h0.c && h.Wr == h0.Wr && h.n_planes == h0.n_planes && h.tables == h0.tables && h.quads == h0.quads;
    status |= h.status;
    entries += h.entries;
    if (err == 0 && h.magic == MSM_BLOB_MAGIC && h.err != 0) { err = h.err; err_rank = h.err_rank; }      // the lowest poisoned rank speaks for all
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    MsmBlobHeader ho = h0;
    ho.magic = (same || err != 0) ? MSM_BLOB_MAGIC : 0u;     // a poisoned gather is reported from this one record: nothing else needs to travel
    ho.status = status;
    ho.entries = entries;
    ho.err = err;
    ho.err_rank = err_rank;
    if (err != 0) ho.n_planes = 0;
    *reinterpret_cast<MsmBlobHeader*>(out) = ho;
  }
  const uint32_t slot = blockIdx.x * (64 / COOP) + threadIdx.x / COOP;
  if (!same || err != 0 || slot >= h0.n_planes) return;      // uniform over a cooperative group
  const proj28_slot* first = reinterpret_cast<const proj28_slot*>(blobs + sizeof(MsmBlobHeader)) + slot;
  g1_proj28 acc = load_proj28(first);
  for (uint32_t k = 1; k < n_blobs; k++) {
    const g1_proj28 b = load_proj28(reinterpret_cast<const proj28_slot*>(blobs + (size_t)k * blob_bytes + sizeof(MsmBlobHeader)) + slot);
    acc = g1_add28_coop(acc, b);
  }
  if ((threadIdx.x & (COOP - 1)) == 0) store_proj28(reinterpret_cast<proj28_slot*>(out + sizeof(MsmBlobHeader)) + slot, acc);
}

// ---------------------------------------------------------------- 6. fixup
// Buckets that straddle chunk edges: add their partials.  Short chains (the common case: 2-3 partials) are
// summed by one lane; a bucket cut into more than FIXUP_LONG chunks (skewed scalars, the narrow top window)
// is queued for msm_fixup_long, where a whole workgroup strides over its partials and tree-reduces in LDS.
constexpr uint32_t FIXUP_LONG = 16;

__device__ __forceinline__ uint32_t partial_slot(uint32_t bucket_start, uint32_t t, uint32_t chunk) {
  return bucket_start <= t * chunk ? 0u : 1u;     // slot 0 = run that begins at the chunk start, 1 = run that ends at its end
}
// lane-to-lane copy of an accumulator inside a group of G lanes
template <int G>
__device__ __forceinline__ g1_proj28 shfl_xor_proj28(const g1_proj28& p, int off) {
  g1_proj28 r;
#pragma unroll
  for (int j = 0; j < N28; j++) {
    r.x.l[j] = (uint32_t)__shfl_xor((int)p.x.l[j], off, G);
    r.y.l[j] = (uint32_t)__shfl_xor((int)p.y.l[j], off, G);
    r.z.l[j] = (uint32_t)__shfl_xor((int)p.z.l[j], off, G);
  }
  return r;
}
// G lanes per bucket: lane `sub` sums the partials t_lo + sub, t_lo + sub + G, ..., then a butterfly over the group.
// G = 1 when buckets are short (per-window buckets: most sit inside one chunk).  With fixed-base tables every one of
// the 2^15 buckets spans ~8 chunks and a single lane's chain of ~9 additions was the whole cost of this kernel; more
// lanes per bucket shorten the chain but every wave still issues whole additions, so G = 2 (one wave per SIMD, chain
// of 5) is the optimum: measured 154 us (G = 1), 218 us (G = 8).
template <int G>
__global__ void __launch_bounds__(256, 2)
msm_fixup(const uint32_t* __restrict__ offsets, MsmPlan plan, proj28_slot* __restrict__ bucket_sum,
          const proj28_slot* __restrict__ partial, uint32_t* __restrict__ long_count, uint32_t* __restrict__ long_list,
          uint32_t long_cap) {
  const uint32_t total = plan.total;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, g = tid / G, sub = tid % G;
  if (g >= total) return;                         // every exit below is uniform over the G lanes of a bucket
  const uint32_t a = offsets[g], b = offsets[g + 1];
  if (a == b) return;
  const uint32_t chunk = msm_lane_chunk(plan, offsets[total]);
  const uint32_t t_lo = a / chunk, t_hi = (b - 1) / chunk;
  if (t_lo == t_hi) return;                       // the whole bucket sat inside one chunk: already stored
  if (t_hi - t_lo >= FIXUP_LONG * G) {
    uint32_t k = 0;
    if (sub == 0) k = atomicAdd(long_count, 1u);
    k = (uint32_t)__shfl((int)k, 0, G);
    if (k < long_cap) {
      if (sub == 0) long_list[k] = g;
      return;
    }                                             // list full (cannot happen: cap = number of buckets that can be this long)
  }
  g1_proj28 acc = g1_identity28();
  bool first = true;
  for (uint32_t t = t_lo + sub; t <= t_hi; t += G) {
    g1_proj28 q = load_proj28(&partial[2 * (size_t)t + partial_slot(a, t, chunk)]);
    if (first) acc = q; else g1_add28(acc, acc, q);
    first = false;
  }
  for (int off = G / 2; off > 0; off >>= 1) {
    g1_proj28 other = shfl_xor_proj28<G>(acc, off);
    g1_add28(acc, acc, other);
  }
  if (sub == 0) store_proj28(&bucket_sum[g], acc);
}

// The same fix-up, one lane per CHUNK EDGE instead of per bucket, for short buckets (wide windows: 2^19 buckets of ~26 entries
// under chunks of ~100): there most buckets sit inside one chunk and a lane per bucket leaves three lanes in four idle while
// their wave issues whole additions (238 us at c = 20, 2^20 points).  Lane t looks at the edge between chunks t - 1 and t
// (position t * chunk), finds the bucket that holds it by bisection and fixes that bucket up if this edge is the first one
// strictly inside it -- every straddling bucket has exactly one such edge.
__global__ void __launch_bounds__(256, 2)
msm_fixup_edges(const uint32_t* __restrict__ offsets, MsmPlan plan, proj28_slot* __restrict__ bucket_sum,
                const proj28_slot* __restrict__ partial, uint32_t* __restrict__ long_count, uint32_t* __restrict__ long_list,
                uint32_t long_cap) {
  const uint32_t total = plan.total, M = offsets[total];
  const uint32_t chunk = msm_lane_chunk(plan, M);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const uint64_t p64 = (uint64_t)t * chunk;
  if (p64 >= M) return;
  const uint32_t p = (uint32_t)p64;
  const uint32_t g = bucket_of(offsets, total, p);
  const uint32_t a = offsets[g], b = offsets[g + 1];
  if (a == p || a / chunk != t - 1) return;         // the bucket starts at this edge, or an earlier edge already lies inside it
  const uint32_t t_lo = t - 1, t_hi = (b - 1) / chunk;
  if (t_hi - t_lo >= FIXUP_LONG) {
    const uint32_t k = atomicAdd(long_count, 1u);
    if (k < long_cap) {
      long_list[k] = g;
      return;
    }
  }
  g1_proj28 acc = load_proj28(&partial[2 * (size_t)t_lo + partial_slot(a, t_lo, chunk)]);
  for (uint32_t tt = t_lo + 1; tt <= t_hi; tt++) {
    g1_proj28 q = load_proj28(&partial[2 * (size_t)tt + partial_slot(a, tt, chunk)]);
    g1_add28(acc, acc, q);
  }
  store_proj28(&bucket_sum[g], acc);
}

extern __shared__ uint4 msm_lds_tree[];
__device__ __forceinline__ g1_proj28 block_tree_sum28(g1_proj28 v, uint32_t live);

// Queued (long) buckets.  A bucket of up to FIXUP_LONG_SPLIT_FROM partials is summed by one workgroup; a longer one -- a bucket that
// holds a large share of a skewed input (scalars 0 / 1: one bucket of n / 2 entries = 16 Ki partials at 2^20) -- by FIXUP_LONG_SLICES
// workgroups, one slice of its partials each, into `scratch`; the workgroup whose slice is finished last (a ticket per
// queued bucket, zero before the launch and zero again after it) adds the slice sums.
// An empty queue (the normal case: uniformly random scalars on a table) costs one launch of workgroups that read one word and leave.
constexpr uint32_t FIXUP_LONG_SLICES = 8, FIXUP_LONG_SPLIT_FROM = 2048;
__global__ void __launch_bounds__(256, 2)
msm_fixup_long(const uint32_t* __restrict__ offsets, MsmPlan plan, proj28_slot* __restrict__ bucket_sum,
               const proj28_slot* __restrict__ partial, const uint32_t* __restrict__ long_count,
               const uint32_t* __restrict__ long_list, uint32_t long_cap, proj28_slot* __restrict__ scratch, uint32_t* __restrict__ ticket) {
  __shared__ uint32_t arrived;
  const uint32_t n_long = *long_count < long_cap ? *long_count : long_cap;
  if (n_long == 0) return;
  const uint32_t chunk = msm_lane_chunk(plan, offsets[plan.total]);
  // work items dealt round-robin over a one-dimensional grid: 255 long buckets (8-bit scalars) keep 255 workgroups busy, one
  // bucket of 16 Ki partials (scalars 0 / 1) keeps eight
  // first the buckets one workgroup sums alone (item = bucket), then the slices of the split ones (item = bucket, slice)
  for (uint32_t item = blockIdx.x; item < n_long * (1 + FIXUP_LONG_SLICES); item += gridDim.x) {
    const bool whole = item < n_long;
    const uint32_t k = whole ? item : (item - n_long) / FIXUP_LONG_SLICES, y = whole ? 0u : (item - n_long) % FIXUP_LONG_SLICES;
    const uint32_t g = long_list[k];
    const uint32_t a = offsets[g], b = offsets[g + 1];
    const uint32_t t_lo = a / chunk, t_hi = (b - 1) / chunk, P = t_hi - t_lo + 1;
    const bool split = P > FIXUP_LONG_SPLIT_FROM;                  // uniform over the workgroup (and over the bucket's workgroups)
    if (split == whole) continue;
    const uint32_t per = split ? (P + FIXUP_LONG_SLICES - 1) / FIXUP_LONG_SLICES : P;
    const uint32_t s_lo = t_lo + y * per, s_hi = s_lo + per - 1 < t_hi ? s_lo + per - 1 : t_hi;      // inclusive; empty when s_lo > t_hi
    g1_proj28 acc = g1_identity28();
    for (uint64_t t = (uint64_t)s_lo + threadIdx.x; t <= s_hi; t += blockDim.x) {
      g1_proj28 q = load_proj28(&partial[2 * (size_t)t + partial_slot(a, (uint32_t)t, chunk)]);
      g1_add28(acc, acc, q);
    }
    g1_proj28 tot = block_tree_sum28(acc, per < blockDim.x ? per : blockDim.x);          // lanes beyond the slice hold the identity
    if (!split) {
      if (threadIdx.x == 0) store_proj28(&bucket_sum[g], tot);
      __syncthreads();
      continue;
    }
    // publish this slice's sum, take a ticket; the last of the FIXUP_LONG_SLICES workgroups of bucket k merges.  One lane stores,
    // releases at agent scope and signals; the merging workgroup acquires before it reads the other slices (MI355X guide, G16)
    if (threadIdx.x == 0) {
      store_proj28(&scratch[(size_t)k * FIXUP_LONG_SLICES + y], tot);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t t = atomicAdd(&ticket[k], 1u);
      if (t == FIXUP_LONG_SLICES - 1) {
        ticket[k] = 0;                                           // as found, for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      arrived = t;
    }
    __syncthreads();
    if (arrived == FIXUP_LONG_SLICES - 1) {                        // uniform over the workgroup
      g1_proj28 v = g1_identity28();
      if (threadIdx.x < FIXUP_LONG_SLICES) v = load_proj28(&scratch[(size_t)k * FIXUP_LONG_SLICES + threadIdx.x]);
      g1_proj28 sum = block_tree_sum28(v, FIXUP_LONG_SLICES);
      if (threadIdx.x == 0) store_proj28(&bucket_sum[g], sum);
    }
    __syncthreads();
  }
}

// LDS tree sum of one value per lane (used by msm_fixup_long)
__device__ __forceinline__ g1_proj28 block_tree_sum28(g1_proj28 v, uint32_t live) {
  proj28_slot* tree = reinterpret_cast<proj28_slot*>(msm_lds_tree);
  store_proj28(&tree[threadIdx.x], v);
  __syncthreads();
  uint32_t stride = 1;
  while (stride < live) stride <<= 1;
  const uint32_t group = threadIdx.x / COOP, n_groups = blockDim.x / COOP;      // one cooperative addition per group of lanes
  for (stride >>= 1; stride > 0; stride >>= 1) {
    for (uint32_t i = group; i < stride; i += n_groups) {
      if (i + stride < live) {
        g1_proj28 a = load_proj28(&tree[i]), b = load_proj28(&tree[i + stride]);
        a = g1_add28_coop(a, b);
        if ((threadIdx.x & (COOP - 1)) == 0) store_proj28(&tree[i], a);
      }
    }
    __syncthreads();
  }
  return load_proj28(&tree[0]);
}

#ifdef BP_EXPERIMENT     // rounds 1-4: the table-free reduction (BP_MSM_REDUCE=1); the shipped library runs the bit-plane tree over every bucket set
// ---------------------------------------------------------------- 7a. reduce, per-window buckets: T_w = sum_b (b+1) * S_{w,b}
// Segmented running sums: grid (blocks_per_window, W), 256 lanes; a lane handles `seg` consecutive buckets, then
// scales its share by its first bucket index and the block tree-sums in LDS.  With W * B buckets there is enough
// parallel work to hide the ~50-operation chain per lane.  out[w * gridDim.x + blockIdx.x] = this block's share.
__global__ void __launch_bounds__(256, 2)
msm_reduce(const uint32_t* __restrict__ offsets, MsmPlan plan, const proj28_slot* __restrict__ bucket_sum,
           proj28_slot* __restrict__ block_out) {
  const uint32_t w = blockIdx.y, B = plan.B, seg = plan.seg;
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * seg;      // first bucket (0-based) of this lane
  g1_proj28 acc = g1_identity28(), wsum = g1_identity28();
  if (first < B) {
    const uint32_t last = first + seg < B ? first + seg : B;
    for (uint32_t b = last; b-- > first;) {
      const size_t g = (size_t)w * B + b;
      if (offsets[g + 1] != offsets[g]) {
        g1_proj28 s = load_proj28(&bucket_sum[g]);
        g1_add28(acc, acc, s);
      }
      g1_add28(wsum, wsum, acc);                  // after the loop: sum_b (b - first + 1) * S_b
    }
    if (first != 0) {                             // + first * acc
      g1_proj28 scaled;
      g1_mul_small28(scaled, acc, first, 32 - __clz(first));
      g1_add28(wsum, wsum, scaled);
    }
  }
  g1_proj28 tot = block_tree_sum28(wsum, blockDim.x);
  if (threadIdx.x == 0) store_proj28(&block_out[(size_t)w * gridDim.x + blockIdx.x], tot);
}

// window_sum[w] = sum of the block shares of window w: one workgroup per window, LDS tree over <= 256 shares
// (the last kernel of an MSM also moves the scalar-status word behind the sums, so that one copy brings everything to the host)
__global__ void __launch_bounds__(256, 2) msm_window_finish(const proj28_slot* __restrict__ block_out, uint32_t blocks_per_window,
                                                             proj28_slot* __restrict__ window_sum, const uint32_t* __restrict__ status_in,
                                                             const uint32_t* __restrict__ entries_in, uint32_t* __restrict__ status_out) {
  const uint32_t w = blockIdx.x;
  if (w == 0 && threadIdx.x == 0) { status_out[0] = *status_in; status_out[1] = *entries_in; }    // [1]: non-zero digits = bucket additions done
  g1_proj28 v = g1_identity28();
  for (uint32_t j = threadIdx.x; j < blocks_per_window; j += blockDim.x) {       // > 256 shares: fold first
    g1_proj28 q = load_proj28(&block_out[(size_t)w * blocks_per_window + j]);
    g1_add28(v, v, q);
  }
  g1_proj28 tot = block_tree_sum28(v, blocks_per_window < blockDim.x ? blocks_per_window : blockDim.x);
  if (threadIdx.x == 0) store_proj28(&window_sum[w], tot);
}
#endif

// ---------------------------------------------------------------- 7b. reduce, fixed-base tables (one bucket set):  sum_b (b + 1) S_b  by bit planes
//   sum_b (b + 1) S_b = A + sum_j 2^j T_j,    A = sum_b S_b,   T_j = sum of the buckets whose index b has bit j set.
// A binary tree over the bucket index yields A and every T_j with two additions per bucket and a dependent chain
// of only log2(B) additions (the running-sum method needs a chain of ~50 additions/doublings per lane, and at one
// wave per SIMD that chain, not the work, was the cost).  A node covering 2^k buckets carries (A, T_0 .. T_{k-1});
// merging the siblings (l, r):   A = A_l + A_r,   T_j = T_j,l + T_j,r  (j < k),   T_k = A_r.
// Levels with at least a wave round of additions (wide windows only) go through HBM one addition per lane (msm_planes_level);
// the rest run up to six levels per launch inside workgroups with cooperative additions (msm_planes_step);
// the c values (A, T_0 .. T_{c-2}) go to the host, whose Horner pass over bit positions replaces
// the scaling by 2^j (msm.rs:107-115 does the same doublings per window).

// cur: 2^levels leaves (one slot each).  Returns the buffer holding the root: A at [0], T_j at [1 + j].
// planes == false: plain tree sum, only slot [0] of the result is meaningful.
// Every addition is shared by a group of COOP lanes (g1_add28_coop: ~1 600 instructions deep instead of ~6 600); the additions of
// a level come first in the item order, the moves T_k = A_r are done by single lanes afterwards, so a level whose additions fit
// the workgroup's groups costs exactly one addition.
__device__ __forceinline__ proj28_slot* planes_tree(proj28_slot* cur, proj28_slot* nxt, uint32_t levels, bool planes) {
  const uint32_t group = threadIdx.x / COOP, n_groups = blockDim.x / COOP;
  const bool lead = (threadIdx.x & (COOP - 1)) == 0;
  for (uint32_t k = 0; k < levels; k++) {
    const uint32_t merges = 1u << (levels - k - 1);
    const uint32_t in_per = planes ? k + 1 : 1, out_per = planes ? k + 2 : 1, adds = merges * in_per;
    for (uint32_t item = group; item < adds; item += n_groups) {               // uniform over a cooperative group
      const uint32_t m = item / in_per, v = item - m * in_per;
      const proj28_slot* L = cur + (size_t)(2 * m) * in_per;
      g1_proj28 a = load_proj28(&L[v]), b = load_proj28(&L[in_per + v]);
      if (!(is_blank28(a) && is_blank28(b))) a = g1_add28_coop(a, b);            // uniform over the group (all its lanes hold the same points)
      if (lead) store_proj28(&nxt[(size_t)m * out_per + v], a);
    }
    if (planes)
      for (uint32_t m = threadIdx.x; m < merges; m += blockDim.x) nxt[(size_t)m * out_per + k + 1] = cur[(size_t)(2 * m + 1) * in_per];   // T_k = A_r
    __syncthreads();
    proj28_slot* t = cur; cur = nxt; nxt = t;
  }
  return cur;
}

// One level of the tree through HBM, one addition per lane ("wide" levels: at least a wave round of additions, where the chip is
// throughput bound and the cooperative form's doubled work would be the cost).  in: nodes of 2^k buckets, k + 1 values each
// (LEAF: k = 0, the bucket sums themselves, empty buckets read as the identity); out: n_out nodes of k + 2 values.
// Lane (node, v), v <= k, adds value v of the node's two children; the lane of v = 0 also moves A_r into the new plane T_k.
template <bool LEAF>
__global__ void __launch_bounds__(256, 2)
msm_planes_level(const uint32_t* __restrict__ offsets, const proj28_slot* __restrict__ in, uint32_t k, uint32_t n_out,
                 proj28_slot* __restrict__ out) {
  const uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per = k + 1, node = (uint32_t)(item / per), v = (uint32_t)(item - (uint64_t)node * per);
  if (node >= n_out) return;
  g1_proj28 a, b;
  if (LEAF) {
    const size_t g = 2 * (size_t)node;
    a = offsets[g + 1] != offsets[g] ? load_proj28(&in[g]) : g1_identity28();             // empty bucket: slot never written
    b = offsets[g + 2] != offsets[g + 1] ? load_proj28(&in[g + 1]) : g1_identity28();
  } else {
    const proj28_slot* L = in + (size_t)(2 * (size_t)node) * per;
    a = load_proj28(&L[v]);
    b = load_proj28(&L[per + v]);
  }
  proj28_slot* o = out + (size_t)node * (per + 1);
  if (v == 0) store_proj28(&o[per], b);                                                    // T_k = A_r
  if (!(is_blank28(a) && is_blank28(b))) g1_add28(a, a, b);                                // empty regions: whole waves skip
  store_proj28(&o[v], a);
}

// Levels 0 and 1 in one launch: a lane takes four consecutive buckets S0..S3 and writes the node (A, T0, T1) = (S0+S1+S2+S3,
// S1+S3, S2+S3) -- the same four additions as two msm_planes_level launches, without the round trip of the level-1 nodes
// through HBM and with one launch less (2^19 buckets: 83 + 75 us -> see profiles/r03_window_width_ab.txt).  n_out = buckets / 4.
__global__ void __launch_bounds__(256, 2)       // (five points live: 177 registers spill at two waves per SIMD -- 138 us; one wave per SIMD without spills measured 151)
msm_planes_level01(const uint32_t* __restrict__ offsets, const proj28_slot* __restrict__ in, uint32_t n_out, proj28_slot* __restrict__ out) {
  const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n_out) return;
  const size_t g = 4 * (size_t)node;
  const uint32_t o0 = offsets[g], o1 = offsets[g + 1], o2 = offsets[g + 2], o3 = offsets[g + 3], o4 = offsets[g + 4];
  proj28_slot* o = out + (size_t)node * 3;
  if (o4 == o0) {                                           // four empty buckets (8-bit or sparse scalars leave most of 2^19 empty): no additions
    const g1_proj28 blank = g1_identity28();
    store_proj28(&o[0], blank);
    store_proj28(&o[1], blank);
    store_proj28(&o[2], blank);
    return;
  }
  // at most three points live at a time (five would spill ~180 registers at two waves per SIMD): S3 is read twice and S0 + S1
  // makes a round trip through its output slot instead
  {
    g1_proj28 s1 = o2 != o1 ? load_proj28(&in[g + 1]) : g1_identity28();
    {
      g1_proj28 a01 = o1 != o0 ? load_proj28(&in[g]) : g1_identity28();
      g1_add28(a01, a01, s1);                                 // S0 + S1
      store_proj28(&o[0], a01);
    }
    const g1_proj28 s3 = o4 != o3 ? load_proj28(&in[g + 3]) : g1_identity28();
    g1_add28(s1, s1, s3);                                     // T0 = S1 + S3
    store_proj28(&o[1], s1);
  }
  g1_proj28 a23 = o3 != o2 ? load_proj28(&in[g + 2]) : g1_identity28();
  {
    const g1_proj28 s3 = o4 != o3 ? load_proj28(&in[g + 3]) : g1_identity28();
    g1_add28(a23, a23, s3);                                   // T1 = S2 + S3
  }
  store_proj28(&o[2], a23);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this lane's own store of S0 + S1 has landed before it is read back
  g1_proj28 a01 = load_proj28(&o[0]);
  g1_add28(a01, a01, a23);                                    // A
  store_proj28(&o[0], a01);
}

// m levels of the tree inside one workgroup, every addition shared by a group of COOP lanes ("narrow" levels: fewer pending
// additions than the chip has lanes, so the length of the dependent chain is the cost: ~1 600 instructions per level instead of
// ~6 600).  Generic step: 2^m input nodes of k + 1 values each (A and k planes) become one output node of k + m + 1 values.
// grid (nodes out, k + 1): workgroup (w, v) folds value v of the 2^m input nodes of output node w --
//   v = 0: the sums A -> the node's A and its planes k .. k + m - 1 (the input node's index supplies those bits);
//   v > 0: plane v - 1, a plain sum over the input nodes.
// 256 lanes = 32 cooperative groups: with m <= 6 no level has more than one round of additions (32, 32, 24, 16, 10, 6 for v = 0).
// LEAF: k = 0, the input nodes are the bucket sums (empty buckets read as the identity).
// status_out != null on the last step only: the scalar-status word and the entry count ride to the host behind the sums.
// LDS: two arrays of 2^m slots.   out[w * (k + m + 1) + {0: A, 1 + j: T_j}]
constexpr uint32_t PLANES_STEP_LOG = 6;
template <bool LEAF>
__global__ void __launch_bounds__(256, 2)
msm_planes_step(const uint32_t* __restrict__ offsets, const proj28_slot* __restrict__ in, uint32_t k, uint32_t m,
                proj28_slot* __restrict__ out, const uint32_t* __restrict__ status_in, const uint32_t* __restrict__ entries_in,
                uint32_t* __restrict__ status_out) {
  proj28_slot* buf = reinterpret_cast<proj28_slot*>(msm_lds_tree);
  const uint32_t w = blockIdx.x, v = blockIdx.y, nin = 1u << m, c = k + m + 1;
  if (status_out && v == 0 && w == 0 && threadIdx.x == 0) { status_out[0] = *status_in; status_out[1] = *entries_in; }
  if (threadIdx.x < nin) {
    const size_t node = ((size_t)w << m) + threadIdx.x;
    if (LEAF) {
      if (offsets[node + 1] != offsets[node]) buf[threadIdx.x] = in[node];
      else store_proj28(&buf[threadIdx.x], g1_identity28());
    } else {
      buf[threadIdx.x] = in[node * (k + 1) + v];
    }
  }
  __syncthreads();
  const proj28_slot* root = planes_tree(buf, buf + nin, m, v == 0);
  if (v == 0) {
    if (threadIdx.x == 0) out[(size_t)w * c] = root[0];
    else if (threadIdx.x <= m) out[(size_t)w * c + k + threadIdx.x] = root[threadIdx.x];       // T'_j -> plane k + j
  } else if (threadIdx.x == 0) {
    out[(size_t)w * c + v] = root[0];
  }
}

// Table-free path (per-window bucket sets): the forest's W roots, c values each (A, T_0 .. T_{c-2}), on their way to
//   sum_b (b + 1) S_{w,b} = A_w + sum_j 2^j T_{w,j}      (what msm.rs:42-46 computes per window).
// Every factor 2^j is j DEPENDENT doublings, whoever performs them, and the host performs one in ~0.5 us where a lone wave needs ~7 us
// per cooperative operation: round 5 evaluated the whole form here (22 dependent operations at c = 16: 148 us on 16 waves) and the host
// then doubled its way through the 16 window sums anyway.  Round 6 stops after two levels: a window's planes are folded in QUADS
//   Q_j = (T_4j + 2 T_4j+1) + 4 (T_4j+2 + 2 T_4j+3)     (+ A in quad 0),   j < ceil((c - 1) / 4)
// -- five dependent operations, every quad of every window side by side -- and the host's one Horner pass runs over the quads (the same
// c (W - 1) + ... doublings as before, 48 more additions at c = 16).  One 64-lane workgroup per window: eight cooperative groups =
// four quads x two halves; every group issues the same five operations (a half that has nothing to add adds the identity: the complete
// formulas make that a no-op), so the wave never diverges.  out[w * nq + j].
// The last kernel of the MSM: it also moves the scalar-status word and the entry count behind the values.
__global__ void __launch_bounds__(64) msm_planes_window_quads(const proj28_slot* __restrict__ roots, uint32_t c, uint32_t nq, proj28_slot* __restrict__ window_sum,
                                                              const uint32_t* __restrict__ status_in, const uint32_t* __restrict__ entries_in,
                                                              uint32_t* __restrict__ status_out) {
  __shared__ proj28_slot part[4];
  const uint32_t w = blockIdx.x, grp = threadIdx.x / COOP, j = grp >> 1, h = grp & 1;
  const bool lead = (threadIdx.x & (COOP - 1)) == 0;
  if (w == 0 && threadIdx.x == 0) { status_out[0] = *status_in; status_out[1] = *entries_in; }
  const proj28_slot* r = roots + (size_t)w * c;
  const uint32_t np = c - 1, p0 = 4 * j + 2 * h, p1 = p0 + 1;       // this half: T_p0 + 2 T_p1 (planes that do not exist are the identity)
  // (operands are picked limb by limb: a ?: between whole points goes through a stack frame)
  auto pick = [](bool c, const g1_proj28& a, const g1_proj28& b) {
    g1_proj28 o;
#pragma unroll
    for (int i = 0; i < N28; i++) {
      o.x.l[i] = c ? a.x.l[i] : b.x.l[i];
      o.y.l[i] = c ? a.y.l[i] : b.y.l[i];
      o.z.l[i] = c ? a.z.l[i] : b.z.l[i];
    }
    return o;
  };
  const g1_proj28 blank = g1_identity28();
  const bool have1 = j < nq && p1 < np, have0 = j < nq && p0 < np;
  g1_proj28 t = pick(have1, load_proj28(&r[have1 ? 1 + p1 : 0]), blank);
  t = g1_add28_coop(t, t);
  t = g1_add28_coop(t, pick(have0, load_proj28(&r[have0 ? 1 + p0 : 0]), blank));
  // upper half: x 4; lower half: + A (quad 0 only), then nothing
  t = g1_add28_coop(t, pick(h != 0, t, pick(j == 0, load_proj28(&r[0]), blank)));
  t = g1_add28_coop(t, pick(h != 0, t, blank));
  if (h && lead && j < 4) store_proj28(&part[j], t);
  __syncthreads();
  if (h) return;
  t = g1_add28_coop(t, load_proj28(&part[j]));
  if (lead && j < nq) store_proj28(&window_sum[(size_t)w * nq + j], t);
}

}  // namespace bp
