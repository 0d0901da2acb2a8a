// ntt_kernels.hpp -- Fr number-theoretic transform on gfx950.
// Replaces ntt_381 / i_ntt_381 (src/utils.rs:63-81, 106-129): out[x] = sum_y in[y] * w^(x*y),
// w = ROOT_OF_UNITY^(2^32/N) (inverse: ROOT_OF_UNITY_INV, then * N^-1); natural order in and out.
// The reference evaluates that sum literally (O(N^2), one 256-bit pow per term); every value is a
// unique reduced residue, so an O(N log N) factorisation produces bit-identical output.
//
// Factorisation (decimation by index digits, "four-step" applied recursively):
//   N = 2^k, k = l_1 + ... + l_P, each l_i <= 10 (two passes up to 2^19, three from 2^20 to 2^28).  Input index n = (d_1, ..., d_P), d_1 most significant.
//   pass i < P : length-2^(l_i) transforms over digit d_i (stride 2^(s_i), s_i = bits below d_i), all
//                butterflies in LDS, then one multiply by w_{M}^(e_i * r) (M = 2^(l_i+s_i), r = low part)
//                from a two-level precomputed table; written back in place of d_i.
//   pass P     : length-2^(l_P) transforms over contiguous rows; the result goes to the digit-reversed
//                position e_1 + 2^(l_1) e_2 + ..., i.e. natural order, written in coalesced runs.
//   A tile is 2^l x C elements (C = 8 adjacent columns = 256-B global runs) staged in LDS limb-major
//   (9 x 29-bit limbs per element, fr29.hpp: one v_mad_u64_u32 per partial product, lazy butterflies); a lane keeps a
//   radix-4 group of four elements in registers over two stages (lds_ntt_dif).
//   HBM traffic: P reads + P writes of the vector (P = 1 up to 2^10, 2 up to 2^19, 3 beyond).
#pragma once
#include "fr_io.hpp"
#include "fr29.hpp"

namespace bp {

constexpr int NTT_MAX_PASS_LOG = 10;    // per-pass transform length up to 2^10
constexpr int NTT_SMALL_MAX_LOG = 10;   // single-workgroup transform up to 2^10
// tile columns C = 2^cl shrink as the per-pass length grows so that TWO tiles (2^l x (C+1) x 36 B each, plus stage twiddles)
// fit the 160 KiB of LDS wherever possible: with one 1 024-lane workgroup per CU nothing overlaps a tile's global loads and
// stores, with two the other one computes meanwhile (measured, profiles/r02_ntt_tile_width_ab.txt: l = 8: C = 8 -> 4 is -8 %
// at 2^22 and 2^24; l = 9: C = 4 -> 2 is -25 % at 2^18).  l <= 7 -> C = 8 (256-B global runs), l = 8 -> 4, l = 9 -> 2,
// l = 10 -> 2 (one tile per CU: a single column would halve the run to 32 B and measured +32 %)
__host__ __device__ constexpr uint32_t ntt_tile_cols_log(uint32_t l) { return l <= 7 ? 3u : (l == 8 ? 2u : 1u); }

struct NttPlan {
  uint32_t k;            // log2 N
  uint32_t P;            // passes
  uint32_t l[4];         // digit widths
  uint32_t cl[4];        // log2 of the tile's column count per pass (ntt_tile_cols_log unless overridden)
  uint32_t h;            // low table has 2^h entries, high table 2^(k-h)
};

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

// Twiddle tables live in HBM as 48-byte records: 9 x 29-bit limbs of w * 2^261 (fr29.hpp) + 3 pad words.
struct tw29_t { uint4 q[3]; };
__device__ __forceinline__ fr29 load_tw29(const tw29_t* __restrict__ p) {
  uint4 a = p->q[0], b = p->q[1], c = p->q[2];
  fr29 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x;
  return r;
}
__device__ __forceinline__ void store_tw29(tw29_t* __restrict__ p, const fr29& v) {
  p->q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  p->q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  p->q[2] = make_uint4(v.l[8], 0, 0, 0);
}

// out[j] = base^(j << shift_), j < count  (table builder; base is a Montgomery Fr).  If scale != null every
// entry is additionally multiplied by *scale (folds N^-1 into a table).  Output either as plain Montgomery
// Fr (out_fr: roots_of_unity) or as a 29-bit twiddle record (out_tw).
__global__ void __launch_bounds__(256) ntt_make_table(fr_t base, uint32_t count, uint32_t shift_, const fr_t* scale,
                                                       fr_t* __restrict__ out_fr, tw29_t* __restrict__ out_tw) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  uint64_t e = (uint64_t)j << shift_;
  fr_t acc = Fr::one(), b = base;
  while (e) {
    if (e & 1) Fr::mul(acc, acc, b);
    Fr::sqr(b, b);
    e >>= 1;
  }
  if (scale) Fr::mul(acc, acc, *scale);
  if (out_fr) store_fr(&out_fr[j], acc);
  if (out_tw) store_tw29(&out_tw[j], fr29_twiddle_from_mont(acc));
}

// LDS tiles hold fr29 elements limb-major (SoA) in 64-bit pairs: limbs (2j, 2j+1) of element e at
// ((uint2*)lds)[j * stride + e] for j < 4, limb 8 at the dword array behind them.  Adjacent lanes touch adjacent
// 8-byte words: every ds_read_b64 / ds_write_b64 is conflict-free, 5 LDS instructions per element.
extern __shared__ uint32_t ntt_lds_raw[];
__device__ __forceinline__ fr29 lds_ld29(const uint32_t* base, uint32_t stride, uint32_t e) {
  fr29 r;
  const uint2* p2 = reinterpret_cast<const uint2*>(base);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint2 v = p2[j * stride + e];
    r.l[2 * j] = v.x;
    r.l[2 * j + 1] = v.y;
  }
  r.l[8] = base[8 * stride + e];
  return r;
}
__device__ __forceinline__ void lds_st29(uint32_t* base, uint32_t stride, uint32_t e, const fr29& v) {
  uint2* p2 = reinterpret_cast<uint2*>(base);
#pragma unroll
  for (int j = 0; j < 4; j++) p2[j * stride + e] = make_uint2(v.l[2 * j], v.l[2 * j + 1]);
  base[8 * stride + e] = v.l[8];
}
// (u, v) <- (u + v, u - v), both back to the < 2q / normalised-limb invariant (twiddle = 1, no multiplication)
__device__ __forceinline__ void fr29_butterfly_notwiddle(fr29& u, fr29& v) {
  fr29 s = fr29_add_lazy(u, v);
  fr29 d, d2;
  uint32_t borrow = fr29_sub_exact(d, u, [&](int i) { return v.l[i]; });         // u - v, exact
  fr29 t;                                                                         // u - v + 2q for the negative case
#pragma unroll
  for (int i = 0; i < N29; i++) t.l[i] = d.l[i] + Q29::two_q(i);
  d2 = fr29_carry(t);
  d2.l[N29 - 1] &= MASK29;                                                        // drop the 2^261 wrap of the borrow
#pragma unroll
  for (int i = 0; i < N29; i++) v.l[i] = borrow ? d2.l[i] : d.l[i];
  u = s;
}

// Tile addressing, element (row, c) of a tile of 2^l rows x C columns:
//   padded   (SWZ = false): row * CP + c with CP = C + 1 (a single column needs no pad);
//   swizzled (SWZ = true, C = 8 and l <= 7 only): no pad -- ((row ^ z(row)) << 3) | c, where z puts the parity of row bits
//            2, 4, 6 into bit 0 and the parity of bits 3, 5 into bit 1.  Every access pattern of a pass (the fill: consecutive
//            rows; a stage: rows j, j + 1, .. of one block, or -- last stage pairs -- rows 4 or 8 apart; the drain: rows in
//            bit-reversed order, i.e. 64, 32, 96 apart) then spreads the 8-element groups of the lanes of one LDS cycle over
//            different banks, which the one-element pad does not do for the drain (row pitch 72 B: rows 16 apart collide).
//            Without the pad a 2^7 x 8 tile plus its stage twiddles is 39 KiB (four would fit a CU; the butterflies' 129 registers
//            allow three workgroups = three waves per SIMD, and tools/ntt_occupancy_probe.sh shows the pass time does not depend on
//            the residency: T = 10 us + 11 us per tile per CU with one, three or four resident -- the VALU is the bound).
template <bool SWZ>
__device__ __forceinline__ uint32_t tile_at(uint32_t row, uint32_t c, uint32_t CP) {
  if (SWZ) {
    const uint32_t z = (__popc(row & 0x54u) & 1u) | ((__popc(row & 0x28u) & 1u) << 1);
    return ((row ^ z) << 3) | c;
  }
  return row * CP + c;
}
__host__ __device__ constexpr uint32_t ntt_tw_slots(uint32_t l) { return (1u << l) < 4 ? 2u : (1u << l) >> 1; }

// In-LDS radix-2 decimation-in-frequency transform of length L = 2^l on every column of a tile (C = 2^cl live columns).
// Result of output index e sits at row bitrev(e).
// tw (LDS, SoA with stride L/2): the twiddles of the stages s >= 1, w_L^(j << s), j < half = L >> (s+1), as the contiguous run
// starting at L/2 - 2*half, so the lanes of a wave read adjacent words whatever the stage (a single table indexed j << s puts
// every j of a late stage on one bank).  Stage 0 reads its L/2 twiddles straight from the 24-KiB global table (once per tile,
// issued with the tile's first LDS reads): keeping them in LDS would cost as much again as all other stages together.
__device__ __forceinline__ void lds_fill_stage_twiddles(uint32_t* tw, uint32_t l, const tw29_t* __restrict__ small_tw) {
  const uint32_t L = 1u << l, twn = ntt_tw_slots(l);
  for (uint32_t x = (L >> 1) + threadIdx.x; x + 1 < L; x += blockDim.x) {
    // x in [L - 2*half, L - half)  <=>  stage s with half = L >> (s+1); j = x - (L - 2*half)
    const uint32_t rem = L - x;                           // in (half, 2*half]
    const uint32_t hl = 31 - __clz(rem - 1);              // floor(log2(rem - 1)) : half = 2^hl when rem - 1 >= half
    const uint32_t half = 1u << hl, s = l - hl - 1, j = x - (L - 2 * half);
    lds_st29(tw, twn, x - (L >> 1), load_tw29(&small_tw[(j << s) << (NTT_SMALL_MAX_LOG - l)]));
  }
}
// stages s and s + 1 on the radix-4 groups of a tile.  GLOBAL0: s = 0, whose twiddles come from the global table.
template <bool SWZ, bool GLOBAL0>
__device__ __forceinline__ void ntt_stage_pair(uint32_t* tile, uint32_t tstride, const uint32_t* tw, const tw29_t* __restrict__ small_tw,
                                               uint32_t l, uint32_t cl, uint32_t CP, uint32_t s) {
  const uint32_t L = 1u << l, C = 1u << cl, twn = ntt_tw_slots(l), g0 = NTT_SMALL_MAX_LOG - l, tw0 = L >> 1;
  const uint32_t nquad = (L >> 2) << cl;
  const uint32_t hl = l - s - 1, half = 1u << hl, quarter = half >> 1;      // hl >= 1
  for (uint32_t b = threadIdx.x; b < nquad; b += blockDim.x) {
    const uint32_t c = b & (C - 1), jp = b >> cl;
    const uint32_t blk = jp >> (hl - 1), j = jp & (quarter - 1), r0 = blk * 2 * half + j;
    const uint32_t i0 = tile_at<SWZ>(r0, c, CP), i1 = tile_at<SWZ>(r0 + quarter, c, CP);
    const uint32_t i2 = tile_at<SWZ>(r0 + half, c, CP), i3 = tile_at<SWZ>(r0 + half + quarter, c, CP);
    fr29 a0 = lds_ld29(tile, tstride, i0), a2 = lds_ld29(tile, tstride, i2);
    if (quarter == 1) {                            // stages l-2 and l-1 (uniform branch): twiddles (1, w_4) and (1, 1)
      fr29 a1 = lds_ld29(tile, tstride, i1), a3 = lds_ld29(tile, tstride, i3);
      fr29_butterfly_notwiddle(a0, a2);
      fr29_butterfly(a1, a3, GLOBAL0 ? load_tw29(&small_tw[1u << g0]) : lds_ld29(tw, twn, tw0 - 3));
      fr29_butterfly_notwiddle(a0, a1);
      fr29_butterfly_notwiddle(a2, a3);
      lds_st29(tile, tstride, i0, a0);
      lds_st29(tile, tstride, i1, a1);
      lds_st29(tile, tstride, i2, a2);
      lds_st29(tile, tstride, i3, a3);
    } else {
#ifdef BP_NTT_PLAIN_BUTTERFLIES
      fr29_butterfly(a0, a2, GLOBAL0 ? load_tw29(&small_tw[j << g0]) : lds_ld29(tw, twn, tw0 - 2 * half + j));
      fr29 a1 = lds_ld29(tile, tstride, i1), a3 = lds_ld29(tile, tstride, i3);
      fr29_butterfly(a1, a3, GLOBAL0 ? load_tw29(&small_tw[(j + quarter) << g0]) : lds_ld29(tw, twn, tw0 - 2 * half + j + quarter));
      const fr29 w = lds_ld29(tw, twn, tw0 - 2 * quarter + j);
      fr29_butterfly(a0, a1, w);
      lds_st29(tile, tstride, i0, a0);
      lds_st29(tile, tstride, i1, a1);
      fr29_butterfly(a2, a3, w);
      lds_st29(tile, tstride, i2, a2);
      lds_st29(tile, tstride, i3, a3);
#else
      // fr29_radix4 written out so that every value dies as early as possible (the kernels live within 128 registers):
      // first-stage sums stay unreduced (limbs < 2^30), their sum is reduced once, their difference enters the product over 8q
      fr29 a1 = lds_ld29(tile, tstride, i1), a3 = lds_ld29(tile, tstride, i3);
      fr29 s02, s13;
#pragma unroll
      for (int i = 0; i < N29; i++) { s02.l[i] = a0.l[i] + a2.l[i]; s13.l[i] = a1.l[i] + a3.l[i]; }
      fr29 d02 = fr29_mul(fr29_sub_lazy(a0, a2), GLOBAL0 ? load_tw29(&small_tw[j << g0]) : lds_ld29(tw, twn, tw0 - 2 * half + j));
      fr29 d13 = fr29_mul(fr29_sub_lazy(a1, a3), GLOBAL0 ? load_tw29(&small_tw[(j + quarter) << g0]) : lds_ld29(tw, twn, tw0 - 2 * half + j + quarter));
      const fr29 w = lds_ld29(tw, twn, tw0 - 2 * quarter + j);
      lds_st29(tile, tstride, i2, fr29_add_lazy(d02, d13));
      lds_st29(tile, tstride, i3, fr29_mul(fr29_sub_lazy(d02, d13), w));
      fr29 x, d;
#pragma unroll
      for (int i = 0; i < N29; i++) { x.l[i] = s02.l[i] + s13.l[i]; d.l[i] = s02.l[i] + (Q29::eight_q_spread(i) - s13.l[i]); }
      lds_st29(tile, tstride, i0, fr29_reduce8(x));
      lds_st29(tile, tstride, i1, fr29_mul(d, w));
#endif
    }
  }
}
// One radix-2 stage (only the first stage of an odd l), then PAIRS of stages with the four elements of a radix-4 group held in
// registers: a group (rows j, j + quarter, j + half, j + half + quarter of a block of 2 * half rows) is closed under stages s
// and s + 1, so a lane reads 4 elements + 3 twiddles and writes 4 elements per two stages where one butterfly per lane per stage
// moved 4 + 2 + 4 per ONE stage -- half the LDS traffic and half the barriers; the multiplications are the same (a prime field has
// no free fourth root of unity).  The last pair knows its twiddles: (1, i) then (1, 1): one product per group instead of two.
template <bool SWZ>
__device__ __forceinline__ void lds_ntt_dif(uint32_t* tile, uint32_t tstride, const uint32_t* tw, const tw29_t* __restrict__ small_tw,
                                            uint32_t l, uint32_t cl, uint32_t CP) {
  const uint32_t C = 1u << cl, g0 = NTT_SMALL_MAX_LOG - l;
  uint32_t s = 0;
  if (l & 1) {
    const uint32_t hl = l - 1, half = 1u << hl, nbf = half << cl;
    for (uint32_t b = threadIdx.x; b < nbf; b += blockDim.x) {
      const uint32_t c = b & (C - 1), j = b >> cl;
      const uint32_t i0 = tile_at<SWZ>(j, c, CP), i1 = tile_at<SWZ>(j + half, c, CP);
      fr29 u = lds_ld29(tile, tstride, i0), v = lds_ld29(tile, tstride, i1);
      if (half == 1) {                               // l = 1: the twiddle is 1 (uniform branch)
        fr29_butterfly_notwiddle(u, v);
      } else {
        fr29_butterfly(u, v, load_tw29(&small_tw[j << g0]));
      }
      lds_st29(tile, tstride, i0, u);
      lds_st29(tile, tstride, i1, v);
    }
    __syncthreads();
    s = 1;
  }
  for (; s < l; s += 2) {
    if (s == 0) ntt_stage_pair<SWZ, true>(tile, tstride, tw, small_tw, l, cl, CP, s);     // own code: the global reads of stage 0
    else ntt_stage_pair<SWZ, false>(tile, tstride, tw, small_tw, l, cl, CP, s);          //   must not cost the other pairs registers
    __syncthreads();
  }
}

// Single-pass transform: N = 2^k <= 2^10, one workgroup per transform (blockIdx.x = batch index).
// small_tw[j] = w_1024^j (forward or inverse table), scale = N^-1 as a twiddle record, or null.
__global__ void __launch_bounds__(256) ntt_small(fr_t* __restrict__ data, size_t stride, uint32_t k,
                                                  const tw29_t* __restrict__ small_tw, const tw29_t* scale) {
  const uint32_t N = 1u << k, tstride = (N + 1) & ~1u;
  uint32_t* tile = ntt_lds_raw;
  uint32_t* tw = tile + N29 * tstride;
  fr_t* base = data + (size_t)blockIdx.x * stride;
  for (uint32_t i = threadIdx.x; i < N; i += blockDim.x) lds_st29(tile, tstride, i, fr29_from_sat(load_fr(&base[i])));
  lds_fill_stage_twiddles(tw, k, small_tw);
  __syncthreads();
  lds_ntt_dif<false>(tile, tstride, tw, small_tw, k, 0, 1);
  fr29 sc;
  if (scale) sc = load_tw29(scale);
  for (uint32_t e = threadIdx.x; e < N; e += blockDim.x) {
    fr29 v = lds_ld29(tile, tstride, bitrev(e, k));
    if (scale) v = fr29_mul(v, sc);
    store_fr(&base[e], fr29_to_sat_canonical(v));
  }
}

// w_N^E from the two-level table (lo[E & (2^h-1)], hi[E >> h]), 29-bit twiddle form
__device__ __forceinline__ fr29 twiddle_lookup(const tw29_t* __restrict__ lo, const tw29_t* __restrict__ hi, uint32_t h, uint64_t E) {
  return fr29_mul(load_tw29(&lo[E & ((1u << h) - 1u)]), load_tw29(&hi[E >> h]));
}

// full inter-pass twiddle table of one strided pass: out[(e << s) + r] = w_N^((e r) << tshift) (from the two-level table; hi may
// carry N^-1), e < 2^l, r < 2^s
__global__ void __launch_bounds__(256) ntt_make_pass_table(const tw29_t* __restrict__ lo, const tw29_t* __restrict__ hi, uint32_t h, uint32_t l,
                                                            uint32_t s, uint32_t tshift, tw29_t* __restrict__ out) {
  const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >> (l + s)) return;
  const uint64_t e = x >> s, r = x & (((size_t)1 << s) - 1);
  store_tw29(&out[x], twiddle_lookup(lo, hi, h, (e * r) << tshift));
}

// Kernel arguments of a pass travel as ONE struct (= the kernarg segment).  The fields of the FILL phase are read the usual way; the
// fields only the DRAIN phase needs (destination, inter-pass twiddle tables) are read back from the kernarg segment AFTER the
// butterflies, behind an opaque barrier the compiler cannot hoist them over: rounds 2-5 kept all fourteen arguments in scalar
// registers through the whole pass, which put every ntt_pass_* at the 106-SGPR ceiling with 4-6 spilled and a (never used) 36-byte
// scratch frame per lane.
// (a pointer read back from the kernarg segment would be a GENERIC pointer to the compiler -- flat_load / flat_store, which also tie up
// the LDS counter; the device pass therefore sees the drain-phase pointers typed as global memory, the host pass as plain pointers: same bytes)
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_GLOBAL_PTR(T) T __attribute__((address_space(1)))*
#define BP_KERNARG_PTR(T) const T __attribute__((address_space(4)))*
#else
#define BP_GLOBAL_PTR(T) T*
#define BP_KERNARG_PTR(T) const T*
#endif
struct NttStridedArgs {
  const fr_t* src;
  size_t src_stride;
  uint32_t k, l, s, cl;          // l digit width, s = bits below the digit, k = log2 N, cl = log2 of the tile's columns
  const tw29_t* small_tw;
  uint32_t tile0, h;             // tile0: first tile of this launch (a member of a group context runs a slice of the tiles)
  // drain phase
  BP_GLOBAL_PTR(fr_t) dst;
  size_t dst_stride;
  BP_GLOBAL_PTR(const tw29_t) tw_lo;
  BP_GLOBAL_PTR(const tw29_t) tw_hi;
  BP_GLOBAL_PTR(const tw29_t) tw_full;
};
template <class A>
__device__ __forceinline__ BP_KERNARG_PTR(A) ntt_late_args() {
#if defined(__HIP_DEVICE_COMPILE__)
  BP_KERNARG_PTR(A) p = (BP_KERNARG_PTR(A))__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));    // the (scalar) loads through p stay behind this point
  return p;
#else
  return nullptr;                // (host pass of the compiler: never executed)
#endif
}

// Strided pass (every pass but the last).  grid.x = tiles, grid.y = batch.
//   element (hi, d, r): address hi * 2^mlog + d * 2^s + r (mlog = l + s); tile = all d x C consecutive r.
template <bool SWZ>
__device__ __forceinline__ void ntt_pass_strided_body(const NttStridedArgs& a) {
  const fr_t* src = a.src;
  const tw29_t* __restrict__ small_tw = a.small_tw;
  const uint32_t k = a.k, l = a.l, s = a.s, cl = a.cl;
  const uint32_t C = 1u << cl, CP = SWZ || C == 1 ? C : C + 1;        // a single column needs no row pad
  const uint32_t bx = blockIdx.x + a.tile0;
  const uint32_t L = 1u << l, mlog = l + s, tstride = (L * CP + 1) & ~1u;      // even: the limb-pair arrays stay 8-byte aligned
  uint32_t* tile = ntt_lds_raw;
  uint32_t* tw = tile + N29 * tstride;
  const uint32_t tiles_per_hi = 1u << (s - cl);
  const uint32_t hi = bx / tiles_per_hi, r0 = (bx % tiles_per_hi) << cl;
  {
    const size_t soff = (size_t)blockIdx.y * a.src_stride, base = ((size_t)hi << mlog) + r0;
    for (uint32_t x = threadIdx.x; x < L * C; x += blockDim.x) {
      const uint32_t c = x & (C - 1), d = x >> cl;
      lds_st29(tile, tstride, tile_at<SWZ>(d, c, CP), fr29_from_sat(load_fr(&src[soff + base + ((size_t)d << s) + c])));
    }
  }
  lds_fill_stage_twiddles(tw, l, small_tw);
  __syncthreads();
  lds_ntt_dif<SWZ>(tile, tstride, tw, small_tw, l, cl, CP);
  BP_KERNARG_PTR(NttStridedArgs) late = ntt_late_args<NttStridedArgs>();
  fr_t* dst = (fr_t*)late->dst;
  const tw29_t* __restrict__ tw_full = (const tw29_t*)late->tw_full;
  const tw29_t* __restrict__ tw_lo = (const tw29_t*)late->tw_lo;
  const tw29_t* __restrict__ tw_hi = (const tw29_t*)late->tw_hi;
  const size_t base = ((size_t)hi << mlog) + r0, doff = (size_t)blockIdx.y * late->dst_stride;
  const uint32_t tshift = k - mlog;                 // w_M^x = w_N^(x << tshift)
  if (tw_full) {                                               // precomputed w_M^(e r): one product per element (uniform branch)
    // The inter-pass twiddles of up to FOUR of a lane's elements are requested together, ahead of the products that use them (a lane
    // drains L C / lanes = 2 .. 4 elements; round 5 requested each twiddle right in front of its product: a memory latency per element,
    // the previous element's store in front of it on the same counter)
    for (uint32_t x0 = threadIdx.x; x0 < L * C; x0 += 4 * blockDim.x) {
      fr29 w[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {                        // (unconditional: an element beyond the tile re-reads the lane's first twiddle)
        const uint32_t xu = x0 + u * blockDim.x, x = xu < L * C ? xu : x0;
        w[u] = load_tw29(&tw_full[((size_t)(x >> cl) << s) + r0 + (x & (C - 1))]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t x = x0 + u * blockDim.x;
        if (x < L * C) {
          const uint32_t c = x & (C - 1), e = x >> cl;
          fr29 v = fr29_mul(lds_ld29(tile, tstride, tile_at<SWZ>(bitrev(e, l), c, CP)), w[u]);
          // dst is the transform's own intermediate buffer: the value (< 2q < 2^256 after the product) is only packed, not brought
          // below q -- the next pass accepts anything below 2q; the last pass writes canonical residues
          store_fr(&dst[doff + base + ((size_t)e << s) + c], fr29_pack(v));
        }
      }
    }
    return;
  }
  for (uint32_t x = threadIdx.x; x < L * C; x += blockDim.x) {
    const uint32_t c = x & (C - 1), e = x >> cl;
    fr29 v = lds_ld29(tile, tstride, tile_at<SWZ>(bitrev(e, l), c, CP));
    const uint64_t E = ((uint64_t)e * (r0 + c)) << tshift;
    v = fr29_mul(v, twiddle_lookup(tw_lo, tw_hi, late->h, E));   // tw_hi may carry the folded N^-1 (first pass of an inverse)
    store_fr(&dst[doff + base + ((size_t)e << s) + c], fr29_pack(v));
  }
}
__global__ void __launch_bounds__(512) ntt_pass_strided(NttStridedArgs a) { ntt_pass_strided_body<false>(a); }
// 2^l x 8 tiles, l <= 7, unpadded with swizzled rows: 256 lanes, three workgroups per CU
__global__ void __launch_bounds__(256, 3) ntt_pass_strided_swz(NttStridedArgs a) { ntt_pass_strided_body<true>(a); }

// Last pass: contiguous rows of length L = 2^l (l = l_P); tile = C rows with consecutive e_1.
// Row (e_1, mid): src address (e_1 * 2^(s1 - l) + mid) * L + d, s1 = k - l_1.
// Output index = e_1 + 2^(l_1) * rev_digits(mid) + 2^(k - l) * e_P, where mid = (e_2..e_{P-1}) is re-ordered
// digit by digit (least significant output digit first).
struct NttLastArgs {
  const fr_t* src;
  size_t src_stride;
  const tw29_t* small_tw;
  uint32_t tile0;
  NttPlan plan;
  // drain phase
  BP_GLOBAL_PTR(fr_t) dst;
  size_t dst_stride;
};
template <bool SWZ>
__device__ __forceinline__ void ntt_pass_last_body(const NttLastArgs& a) {
  const fr_t* __restrict__ src = a.src;
  const tw29_t* __restrict__ small_tw = a.small_tw;
  const uint32_t k = a.plan.k, P = a.plan.P, l = a.plan.l[P - 1], l1 = a.plan.l[0], L = 1u << l, bx = blockIdx.x + a.tile0;
  const uint32_t cl = a.plan.cl[P - 1], C = 1u << cl, CP = SWZ || C == 1 ? C : C + 1, tstride = (L * CP + 1) & ~1u;
  uint32_t* tile = ntt_lds_raw;
  uint32_t* tw = tile + N29 * tstride;
  const uint32_t midbits = k - l1 - l;              // bits of (e_2 .. e_{P-1})
  // blockIdx.x enumerates (e1_tile, mid): e_1 = e1_tile * C + c
  const uint32_t mid = bx & ((1u << midbits) - 1u), e1_0 = (bx >> midbits) << cl;
  {
    const size_t soff = (size_t)blockIdx.y * a.src_stride;
    for (uint32_t x = threadIdx.x; x < L * C; x += blockDim.x) {
      const uint32_t d = x % L, c = x / L;
      const size_t row = ((size_t)(e1_0 + c) << midbits) + mid;
      lds_st29(tile, tstride, tile_at<SWZ>(d, c, CP), fr29_from_sat(load_fr(&src[soff + (row << l) + d])));
    }
  }
  lds_fill_stage_twiddles(tw, l, small_tw);
  __syncthreads();
  lds_ntt_dif<SWZ>(tile, tstride, tw, small_tw, l, cl, CP);
  BP_KERNARG_PTR(NttLastArgs) late = ntt_late_args<NttLastArgs>();
  fr_t* __restrict__ dst = (fr_t*)late->dst;
  const size_t doff = (size_t)blockIdx.y * late->dst_stride;
  // digit-reverse mid: mid = e_2 * 2^(l_3+..+l_{P-1}) + ... + e_{P-1}; output wants e_2 lowest.
  uint32_t mid_out = 0, shift_out = 0, rem = midbits;
  for (uint32_t i = 1; i + 1 < P; i++) {
    const uint32_t li = late->plan.l[i];
    rem -= li;
    const uint32_t digit = (mid >> rem) & ((1u << li) - 1u);
    mid_out |= digit << shift_out;
    shift_out += li;
  }
  const size_t obase = ((size_t)mid_out << l1) + e1_0;
  for (uint32_t x = threadIdx.x; x < L * C; x += blockDim.x) {
    const uint32_t c = x & (C - 1), e = x >> cl;
    fr29 v = lds_ld29(tile, tstride, tile_at<SWZ>(bitrev(e, l), c, CP));
    store_fr(&dst[doff + obase + ((size_t)e << (k - l)) + c], fr29_to_sat_canonical(v));
  }
}
__global__ void __launch_bounds__(512) ntt_pass_last(NttLastArgs a) { ntt_pass_last_body<false>(a); }
__global__ void __launch_bounds__(256, 3) ntt_pass_last_swz(NttLastArgs a) { ntt_pass_last_body<true>(a); }

// ---- element-wise helpers used by the Polynomial layer (src/polynomial.rs) -------------------------
// op: 0 a+b, 1 a-b, 2 a*b (pointwise), with broadcast of a single scalar when nb == 1 is not done here.
__global__ void __launch_bounds__(256) fr_binary(const fr_t* a, size_t na, const fr_t* b, size_t nb,
                                                  fr_t* out, size_t n, int op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t x = i < na ? load_fr(&a[i]) : Fr::zero(), y = i < nb ? load_fr(&b[i]) : Fr::zero(), r;
  if (op == 0) Fr::add(r, x, y);
  else if (op == 1) Fr::sub(r, x, y);
  else Fr::mul(r, x, y);
  store_fr(&out[i], r);
}
// out[i] = a[i] (op) s for i < n;  op: 0 add, 1 sub, 2 mul
__global__ void __launch_bounds__(256) fr_scalar_op(const fr_t* a, fr_t s, fr_t* out, size_t n, int op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t x = load_fr(&a[i]), r;
  if (op == 0) Fr::add(r, x, s);
  else if (op == 1) Fr::sub(r, x, s);
  else Fr::mul(r, x, s);
  store_fr(&out[i], r);
}
// Synthetic uniform scalars: element i = from_u512 (scalar.rs:323-339) of eight
// SplitMix64 outputs of the stream seeded with seed + 8*i*golden; Montgomery limbs out.
__device__ __forceinline__ uint64_t splitmix64_next(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) fr_synthetic(fr_t* __restrict__ out, size_t n, uint64_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t st = seed + 0x9e3779b97f4a7c15ull * 8ull * i;
  fr_t d0, d1, r3;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint64_t z = splitmix64_next(st);
    d0.l[2 * k] = (uint32_t)z;
    d0.l[2 * k + 1] = (uint32_t)(z >> 32);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint64_t z = splitmix64_next(st);
    d1.l[2 * k] = (uint32_t)z;
    d1.l[2 * k + 1] = (uint32_t)(z >> 32);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) r3.l[k] = FrParams::r3(k);
  // the Montgomery product needs a*b < q*R only, which holds for any 256-bit a and b < q
  Fr::mul(d0, d0, Fr::r2());
  Fr::mul(d1, d1, r3);
  Fr::add(d0, d0, d1);
  store_fr(&out[i], d0);
}
// in-place conversion between 32-byte LE canonical and Montgomery limbs (dir 0: to Montgomery, 1: from)
__global__ void __launch_bounds__(256) fr_convert(fr_t* __restrict__ a, size_t n, int dir) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t x = load_fr(&a[i]), r;
  if (dir == 0) Fr::to_mont(r, x); else Fr::from_mont(r, x);
  store_fr(&a[i], r);
}

}  // namespace bp
