// fp28.hpp -- BLS12-381 base field in an UNSATURATED radix for the MSM hot loop: 14 limbs x 28 bits in
// u32 registers, Montgomery radix R' = 2^392.
//
// Why (measured on MI355X, profiles/r01_ubench_int_issue_rates.txt): v_mad_u64_u32 issues at the same rate
// as a plain carry add (~57 lanes/clk/CU), so on saturated 32-bit limbs every partial product costs TWO
// instructions (the mad plus the carry into a third accumulator word) and every modular add/sub is a carry
// chain plus a conditional subtract.  With 28-bit limbs a column of 14 + 14 products fits a 64-bit
// accumulator, so a product is exactly ONE v_mad_u64_u32 emitted from plain C (no inline asm, no carry
// instruction), and additions are 14 independent v_add_u32 with no carries at all ("lazy" limbs).
//
// Safety: every value carries compile-time bounds in its type, F28<LB, VB>:
//     every limb <= LB      and      0 <= value < VB * p
// Operations compute the bounds of their result and static_assert the preconditions (no u32 limb overflow,
// no u64 column overflow in mul, subtrahend covered limb-by-limb by the added multiple of p).  A kernel that
// compiles cannot overflow.
#pragma once
#include "fields.hpp"

namespace bp {

constexpr int N28 = 14;
constexpr uint32_t MASK28 = (1u << 28) - 1;
typedef unsigned __int128 u128_t;

// p in radix 2^28 (little endian) and -p^-1 mod 2^28
struct P28 {
  static BP_HD constexpr uint32_t mod(int i) {
    // bits [28 i, 28 i + 28) of p, from the 32-bit limbs of FpParams
    const int bit = 28 * i, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)FpParams::mod(w) | (w + 1 < 12 ? (uint64_t)FpParams::mod(w + 1) << 32 : 0);
    return (uint32_t)(two >> s) & MASK28;
  }
  static constexpr uint32_t INV = FpParams::INV32 & MASK28;
};

// digit i (radix 2^28) of K * p, exact (top digit absorbs the overflow), K < 2^20
BP_HD constexpr uint32_t kp_digit(uint32_t K, int i) {
  uint64_t carry = 0, d = 0;
  for (int j = 0; j <= i; j++) {
    uint64_t t = (uint64_t)P28::mod(j) * K + carry;
    d = j < N28 - 1 ? (t & MASK28) : t;
    carry = t >> 28;
  }
  return (uint32_t)d;
}
// K*p written with every limb >= 2^S - 2^(S-28) ("spread"), value unchanged:
//   c_0 = d_0 + 2^S,  c_i = d_i + 2^S - 2^(S-28) (0 < i < 13),  c_13 = d_13 - 2^(S-28)
template <uint32_t K, int S>
BP_HD constexpr uint32_t kp_spread(int i) {
  const uint32_t d = kp_digit(K, i), up = 1u << S, down = 1u << (S - 28);
  return i == 0 ? d + up : (i < N28 - 1 ? d + up - down : d - down);
}
template <uint32_t K, int S>
constexpr uint64_t kp_spread_max() {
  uint64_t m = 0;
  for (int i = 0; i < N28; i++) m = kp_spread<K, S>(i) > m ? kp_spread<K, S>(i) : m;
  return m;
}
// top limb of any value < V*p is below this (limbs are non-negative): floor(V*p / 2^364) + 1
constexpr uint64_t top_limb_bound(uint32_t V) { return (uint64_t)kp_digit(V, N28 - 1) + 1; }

template <uint64_t LB, uint32_t VB>
struct F28 {
  static constexpr uint64_t limb_bound = LB;
  static constexpr uint32_t value_bound = VB;
  uint32_t l[N28];
};
using F28n = F28<MASK28, 1>;        // fully normalised, value < p (stored SRS coordinates)

template <uint64_t A, uint32_t VA, uint64_t B, uint32_t VB>
BP_HD F28<A + B, VA + VB> add28(const F28<A, VA>& a, const F28<B, VB>& b) {
  static_assert(A + B < (1ull << 32), "limb overflow in add28");
  F28<A + B, VA + VB> r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <uint32_t K, uint64_t A, uint32_t VA>
BP_HD F28<A * K, VA * K> mulk28(const F28<A, VA>& a) {
  static_assert(A * K < (1ull << 32), "limb overflow in mulk28");
  F28<A * K, VA * K> r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = a.l[i] * K;
  return r;
}
// parallel (one-hop) carry: limbs drop to 28 bits + the neighbour's overflow; value unchanged
template <uint64_t A, uint32_t VA>
BP_HD F28<MASK28 + (A >> 28), VA> norm28(const F28<A, VA>& a) {
  static_assert(top_limb_bound(VA) + (A >> 28) <= MASK28 + (A >> 28), "top limb too large");
  F28<MASK28 + (A >> 28), VA> r;
  r.l[0] = a.l[0] & MASK28;
#pragma unroll
  for (int i = 1; i < N28 - 1; i++) r.l[i] = (a.l[i] & MASK28) + (a.l[i - 1] >> 28);
  r.l[N28 - 1] = a.l[N28 - 1] + (a.l[N28 - 2] >> 28);
  return r;
}
// a - b + K*p  (K*p in spread form with shift S covers b limb by limb)
template <uint32_t K, int S, uint64_t A, uint32_t VA, uint64_t B, uint32_t VB>
BP_HD F28<A + kp_spread_max<K, S>(), VA + K> sub28(const F28<A, VA>& a, const F28<B, VB>& b) {
  static_assert(K >= VB, "K*p must dominate the subtrahend's value");
  static_assert((1ull << S) - (1ull << (S - 28)) >= B, "spread limbs must dominate the subtrahend's limbs");
  static_assert(kp_digit(K, N28 - 1) >= (1u << (S - 28)) + top_limb_bound(VB), "top limb of K*p too small");
  static_assert(A + kp_spread_max<K, S>() < (1ull << 32), "limb overflow in sub28");
  F28<A + kp_spread_max<K, S>(), VA + K> r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = a.l[i] + (kp_spread<K, S>(i) - b.l[i]);
  return r;
}

constexpr uint32_t mul28_out_v(uint32_t VA, uint32_t VB) { return 1 + (VA * VB + 2047) / 2048; }   // R'/p > 2^11

// Montgomery product a*b / 2^392 mod p; result limbs <= 2^28 - 1 (top limb smaller), value < (1 + VA*VB/2^11) p
template <uint64_t A, uint32_t VA, uint64_t B, uint32_t VB>
BP_HD F28<MASK28, mul28_out_v(VA, VB)> mul28(const F28<A, VA>& a, const F28<B, VB>& b) {
  static_assert((u128_t)14 * A * B + ((u128_t)14 << 56) + ((u128_t)1 << 40) < ((u128_t)1 << 64), "column overflow in mul28");
  static_assert(top_limb_bound(mul28_out_v(VA, VB)) <= MASK28, "result top limb");
  F28<MASK28, mul28_out_v(VA, VB)> r;
  uint32_t m[N28];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N28; k++) {
    // two independent accumulation chains per column (products | reduction terms) halve the dependent-mad depth
    uint64_t red = 0;
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) red += (uint64_t)m[i] * P28::mod(k - i);
    acc += red;
    m[k] = ((uint32_t)acc * P28::INV) & MASK28;
    acc += (uint64_t)m[k] * P28::mod(0);
    acc >>= 28;
  }
#pragma unroll
  for (int k = N28; k < 2 * N28 - 1; k++) {
    uint64_t red = 0;
#pragma unroll
    for (int i = k - N28 + 1; i < N28; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - N28 + 1; i < N28; i++) red += (uint64_t)m[i] * P28::mod(k - i);
    acc += red;
    r.l[k - N28] = (uint32_t)acc & MASK28;
    acc >>= 28;
  }
  r.l[N28 - 1] = (uint32_t)acc;
  return r;
}

// Fused a*b + c*d with ONE Montgomery reduction: (a*b + c*d) / 2^392 mod p.  Saves the 14 x 14 reduction
// products (and the m computations) of the second product; used for the three two-term outputs of the
// complete addition formulas.
template <uint64_t A, uint32_t VA, uint64_t B, uint32_t VB, uint64_t C, uint32_t VC, uint64_t D, uint32_t VD>
BP_HD F28<MASK28, 1 + (VA * VB + VC * VD + 2047) / 2048> mul28_2(const F28<A, VA>& a, const F28<B, VB>& b, const F28<C, VC>& c,
                                                                  const F28<D, VD>& d) {
  static_assert((u128_t)14 * A * B + (u128_t)14 * C * D + ((u128_t)14 << 56) + ((u128_t)1 << 40) < ((u128_t)1 << 64),
                "column overflow in mul28_2");
  constexpr uint32_t VO = 1 + (VA * VB + VC * VD + 2047) / 2048;
  static_assert(top_limb_bound(VO) <= MASK28, "result top limb");
  F28<MASK28, VO> r;
  uint32_t m[N28];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N28; k++) {
    uint64_t acc2 = 0, red = 0;       // three independent chains per column
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i <= k; i++) acc2 += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) red += (uint64_t)m[i] * P28::mod(k - i);
    acc += acc2 + red;
    m[k] = ((uint32_t)acc * P28::INV) & MASK28;
    acc += (uint64_t)m[k] * P28::mod(0);
    acc >>= 28;
  }
#pragma unroll
  for (int k = N28; k < 2 * N28 - 1; k++) {
    uint64_t acc2 = 0, red = 0;
#pragma unroll
    for (int i = k - N28 + 1; i < N28; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - N28 + 1; i < N28; i++) acc2 += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
    for (int i = k - N28 + 1; i < N28; i++) red += (uint64_t)m[i] * P28::mod(k - i);
    acc += acc2 + red;
    r.l[k - N28] = (uint32_t)acc & MASK28;
    acc >>= 28;
  }
  r.l[N28 - 1] = (uint32_t)acc;
  return r;
}
// K*p - b (lazy negation, value in (0, K p])
template <uint32_t K, int S, uint64_t B, uint32_t VB>
BP_HD F28<kp_spread_max<K, S>(), K> neg28(const F28<B, VB>& b) {
  F28<0, 0> zero;
#pragma unroll
  for (int i = 0; i < N28; i++) zero.l[i] = 0;
  return sub28<K, S>(zero, b);
}

// ---- conversions between the saturated 12 x 32 form (fp_t) and 14 x 28 -----------------------------------
// plain re-slicing of a 384-bit integer < 2^384 (no change of Montgomery radix)
BP_HD F28n reslice_to28(const fp_t& a) {
  F28n r;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    const int bit = 28 * i, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)a.l[w] | (w + 1 < 12 ? (uint64_t)a.l[w + 1] << 32 : 0);
    r.l[i] = (uint32_t)(two >> s) & MASK28;
  }
  return r;
}
// full sequential carry, then pack; the value must be < 2^384 (true for VB <= 8)
template <uint64_t A, uint32_t VA>
BP_HD fp_t reslice_from28(const F28<A, VA>& a) {
  static_assert(VA <= 8, "value must fit 384 bits");
  uint32_t t[N28];
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    uint64_t v = (uint64_t)a.l[i] + carry;
    t[i] = (uint32_t)v & MASK28;
    carry = (uint32_t)(v >> 28);
  }
  fp_t r;
  uint64_t buf = 0;
  int bits = 0, w = 0;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    buf |= (uint64_t)t[i] << bits;
    bits += 28;
    if (bits >= 32 && w < 12) {
      r.l[w++] = (uint32_t)buf;
      buf >>= 32;
      bits -= 32;
    }
  }
  return r;
}

// constants 2^392 mod p and 2^376 mod p as plain integers in saturated limbs (domain changes, see below)
struct FpDomain {
  BP_TABLE(two392, 0x0347fcb8u, 0x19d80000u, 0x6d2002b1u, 0x12e00cdeu, 0xa2090c72u, 0x37669f83u, 0xda0f73e0u, 0x09b09b42u,
           0x8f1297bbu, 0xa7c515d9u, 0xfcfa012cu, 0x0577a659u)
  BP_TABLE(two376, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u,
           0x00000000u, 0x00000000u, 0x00000000u, 0x01000000u)
};
// stored SRS coordinate x~ = x * 2^384 (reference Montgomery form)  ->  x * 2^392 mod p in 14 x 28 limbs
BP_HD F28n fp_to_28(const fp_t& x_mont384) {
  fp_t c, t;
#pragma unroll
  for (int i = 0; i < 12; i++) c.l[i] = FpDomain::two392(i);
  Fp::mul(t, x_mont384, c);             // x*2^384 * 2^392 / 2^384
  return reslice_to28(t);
}
// x * 2^392 (lazy, < 8p) -> canonical reference Montgomery form x * 2^384 mod p
template <uint64_t A, uint32_t VA>
BP_HD fp_t fp_from_28(const F28<A, VA>& a) {
  fp_t c, t = reslice_from28(a), r;
#pragma unroll
  for (int i = 0; i < 12; i++) c.l[i] = FpDomain::two376(i);
  Fp::mul(r, t, c);                     // (x*2^392) * 2^376 / 2^384 = x * 2^384; inputs 8p * p < p * 2^384
  return r;
}

}  // namespace bp
