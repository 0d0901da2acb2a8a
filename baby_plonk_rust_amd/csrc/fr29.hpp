// fr29.hpp -- BLS12-381 scalar field in an unsaturated radix for the NTT butterflies: 9 limbs x 29 bits.
//
// Same reasoning as fp28.hpp (v_mad_u64_u32 issues like an add, so carries cost as much as products): with
// 29-bit limbs a column of 9 + 9 products fits a 64-bit accumulator, one v_mad_u64_u32 per product from plain
// C, no carry instructions.
//
// Domain trick: data in HBM stays in the reference's Montgomery form x~ = x * 2^256 (scalar.rs:16-22) and is
// only RE-SLICED (8 x 32 -> 9 x 29 bits).  Twiddles are stored as w * 2^261 (true Montgomery form for the
// radix R' = 2^261 of this limb layout), so  mont'(x~, w') = x~ * w * 2^261 / 2^261 = (x w)~ : products land
// back in the data's own domain and additions/subtractions are domain-agnostic.  No conversion pass exists.
//
// Invariants between butterfly stages (Harvey-style lazy butterflies):
//     value < 2q,   every limb < 2^29 (top limb < 2^24)
//   sum : u + v  (< 4q), exact conditional subtraction of 2q          -> < 2q, limbs normalised
//   diff: u - v + 4q (limb-wise, 4q in a spread form that dominates v's limbs), then * w -> < 2q, normalised
//         Montgomery output bound: (6q * q) / 2^261 + q < 2q since 2^261 / q > 2^6.
#pragma once
#include "fields.hpp"

namespace bp {

constexpr int N29 = 9;
constexpr uint32_t MASK29 = (1u << 29) - 1;

struct fr29 {
  uint32_t l[N29];
};

struct Q29 {
  static BP_HD constexpr uint32_t mod(int i) {       // bits [29 i, 29 i + 29) of q
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)FrParams::mod(w) | (w + 1 < 8 ? (uint64_t)FrParams::mod(w + 1) << 32 : 0);
    return (uint32_t)(two >> s) & MASK29;
  }
  // -q^-1 mod 2^29: Newton iteration on the low limb (q is odd)
  static BP_HD constexpr uint32_t inv() {
    uint32_t q0 = FrParams::mod(0), y = q0;
    for (int i = 0; i < 5; i++) y *= 2 - q0 * y;
    return (0u - y) & MASK29;
  }
  // digit i of 2q in radix 2^29 (2q < 2^256 fits 9 digits exactly)
  static BP_HD constexpr uint32_t two_q(int i) {
    uint64_t carry = 0, d = 0;
    for (int j = 0; j <= i; j++) {
      uint64_t t = (uint64_t)mod(j) * 2 + carry;
      d = j < N29 - 1 ? (t & MASK29) : t;
      carry = t >> 29;
    }
    return (uint32_t)d;
  }
  static BP_HD constexpr uint32_t four_q(int i) {
    uint64_t carry = 0, d = 0;
    for (int j = 0; j <= i; j++) {
      uint64_t t = (uint64_t)mod(j) * 4 + carry;
      d = j < N29 - 1 ? (t & MASK29) : t;
      carry = t >> 29;
    }
    return (uint32_t)d;
  }
  static BP_HD constexpr uint32_t eight_q(int i) {
    uint64_t carry = 0, d = 0;
    for (int j = 0; j <= i; j++) {
      uint64_t t = (uint64_t)mod(j) * 8 + carry;
      d = j < N29 - 1 ? (t & MASK29) : t;
      carry = t >> 29;
    }
    return (uint32_t)d;
  }
  // 8q with every limb >= 2^30 - 2 (dominates the limb-wise sum of two normalised values) and a top limb above any value < 4q:
  //   c_0 = d_0 + 2^30,  c_i = d_i + 2^30 - 2 (0 < i < 8),  c_8 = d_8 - 2,   d = digits of 8q
  static BP_HD constexpr uint32_t eight_q_spread(int i) {
    return i == 0 ? eight_q(0) + (1u << 30) : (i < N29 - 1 ? eight_q(i) + (1u << 30) - 2 : eight_q(i) - 2);
  }
  // 4q with every limb >= 2^29 - 1 (dominates any normalised limb) and a top limb above any value < 2q:
  //   c_0 = d_0 + 2^29,  c_i = d_i + 2^29 - 1 (0 < i < 8),  c_8 = d_8 - 1,   d = digits of 4q
  static BP_HD constexpr uint32_t four_q_spread(int i) {
    return i == 0 ? four_q(0) + (1u << 29) : (i < N29 - 1 ? four_q(i) + (1u << 29) - 1 : four_q(i) - 1);
  }
};

// 8 x 32-bit limbs (value < 2^256) -> 9 x 29-bit limbs, same integer
BP_HD fr29 fr29_from_sat(const fr_t& a) {
  fr29 r;
#pragma unroll
  for (int i = 0; i < N29; i++) {
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)a.l[w] | (w + 1 < 8 ? (uint64_t)a.l[w + 1] << 32 : 0);
    r.l[i] = (uint32_t)(two >> s) & MASK29;
  }
  return r;
}
// normalised limbs (each < 2^29, value < 2^256) -> 8 x 32
BP_HD fr_t fr29_pack(const fr29& a) {
  fr_t r;
  uint64_t buf = 0;
  int bits = 0, w = 0;
#pragma unroll
  for (int i = 0; i < N29; i++) {
    buf |= (uint64_t)a.l[i] << bits;
    bits += 29;
    if (bits >= 32 && w < 8) {
      r.l[w++] = (uint32_t)buf;
      buf >>= 32;
      bits -= 32;
    }
  }
  return r;
}
// a - k (k given by digit function) with exact borrow propagation; returns the borrow-out (1 when a < k)
template <class DigitFn>
BP_HD uint32_t fr29_sub_exact(fr29& r, const fr29& a, DigitFn digit) {
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N29; i++) {
    int32_t t = (int32_t)a.l[i] - (int32_t)digit(i) + borrow;      // |t| < 2^31: limbs and digits are < 2^30
    r.l[i] = (uint32_t)t & MASK29;
    borrow = t >> 29;                                               // arithmetic shift: 0 or -1 (or small negative)
  }
  return borrow < 0 ? 1u : 0u;
}
// full sequential carry: any lazy limbs (value < 2^261) -> every limb < 2^29
BP_HD fr29 fr29_carry(const fr29& a) {
  fr29 r;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < N29; i++) {
    uint32_t t = a.l[i] + carry;          // lazy limbs stay below 2^31, carry below 2^3
    r.l[i] = i < N29 - 1 ? (t & MASK29) : t;
    carry = t >> 29;
  }
  return r;
}
// value < 2q with normalised limbs -> canonical representative in [0, q), packed
BP_HD fr_t fr29_to_sat_canonical(const fr29& a) {
  fr29 t;
  uint32_t borrow = fr29_sub_exact(t, a, [](int i) { return Q29::mod(i); });
  fr29 r;
#pragma unroll
  for (int i = 0; i < N29; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
  return fr29_pack(r);
}
// (u + v) mod-ish 2q: inputs < 2q normalised, output < 2q normalised
BP_HD fr29 fr29_add_lazy(const fr29& u, const fr29& v) {
  fr29 s, t;
#pragma unroll
  for (int i = 0; i < N29; i++) s.l[i] = u.l[i] + v.l[i];                 // < 2^30 per limb, value < 4q
  uint32_t borrow = fr29_sub_exact(t, s, [](int i) { return Q29::two_q(i); });   // exact s - 2q (normalised when >= 0)
  fr29 c = fr29_carry(s), r;                                               // s itself, normalised (used when s < 2q)
#pragma unroll
  for (int i = 0; i < N29; i++) r.l[i] = borrow ? c.l[i] : t.l[i];
  return r;
}
// u - v + 4q limb-wise (no borrows): limbs < 2^31, value in (2q, 6q)
BP_HD fr29 fr29_sub_lazy(const fr29& u, const fr29& v) {
  fr29 r;
#pragma unroll
  for (int i = 0; i < N29; i++) r.l[i] = u.l[i] + (Q29::four_q_spread(i) - v.l[i]);
  return r;
}
// Montgomery product a * w / 2^261 mod q.  a: limbs < 1.5 * 2^31, value < 2^261 (= 70 q).  w: limbs < 2^29, value < q.
// Output: limbs < 2^29 (top limb < 2^24), value < 2q   ((a w + m q) / 2^261 < q (a / 2^261 + 1)).
//   column bound: 9 * (1.5 * 2^31) * 2^29 + 9 * 2^58 + carry = (13.5 + 2.25) * 2^60 + 2^35 < 2^64
#ifndef BP_FR29_CHECK
#define BP_FR29_CHECK(cond)
#endif
BP_HD fr29 fr29_mul(const fr29& a, const fr29& w) {
  fr29 r;
  uint32_t m[N29];
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < N29; i++) { BP_FR29_CHECK(a.l[i] < 0xC0000000u); BP_FR29_CHECK(w.l[i] <= MASK29); }
#pragma unroll
  for (int k = 0; k < N29; k++) {
    uint64_t red = 0;
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * w.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) red += (uint64_t)m[i] * Q29::mod(k - i);
    acc += red;
    m[k] = ((uint32_t)acc * Q29::inv()) & MASK29;
    acc += (uint64_t)m[k] * Q29::mod(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = N29; k < 2 * N29 - 1; k++) {
    uint64_t red = 0;
#pragma unroll
    for (int i = k - N29 + 1; i < N29; i++) acc += (uint64_t)a.l[i] * w.l[k - i];
#pragma unroll
    for (int i = k - N29 + 1; i < N29; i++) red += (uint64_t)m[i] * Q29::mod(k - i);
    acc += red;
    r.l[k - N29] = (uint32_t)acc & MASK29;
    acc >>= 29;
  }
  r.l[N29 - 1] = (uint32_t)acc;
  return r;
}
// decimation-in-frequency butterfly: (u, v) <- (u + v, (u - v) * w)
BP_HD void fr29_butterfly(fr29& u, fr29& v, const fr29& w) {
  fr29 s = fr29_add_lazy(u, v);
  v = fr29_mul(fr29_sub_lazy(u, v), w);
  u = s;
}
// x with limbs < 2^31 and value < 8q  ->  value < 2q, normalised (the same residue): carry, then subtract 4q and 2q where they fit
BP_HD fr29 fr29_reduce8(const fr29& x) {
  const fr29 c = fr29_carry(x);
  fr29 t, y, r;
  uint32_t borrow = fr29_sub_exact(t, c, [](int i) { return Q29::four_q(i); });
#pragma unroll
  for (int i = 0; i < N29; i++) y.l[i] = borrow ? c.l[i] : t.l[i];
  borrow = fr29_sub_exact(t, y, [](int i) { return Q29::two_q(i); });
#pragma unroll
  for (int i = 0; i < N29; i++) r.l[i] = borrow ? y.l[i] : t.l[i];
  return r;
}
// Two decimation-in-frequency stages on the four elements of a radix-4 group (rows j, j + quarter, j + half, j + half + quarter):
//   stage s    : (a0, a2) with w0, (a1, a3) with w1;      stage s + 1 : (a0, a1) with w2, (a2, a3) with w2.
// Same results as four fr29_butterfly calls, but the two first-stage sums stay unreduced (limbs < 2^30, values < 4q): their sum
// is reduced once from < 8q, and their difference goes into the product over 8q in a spread form that dominates such limbs
// (limbs of the product's input < 1.25 * 2^31, value < 12q).  One exact reduction instead of three: -70 of ~1 230 instructions.
// Inputs and outputs: values < 2q, normalised limbs.
BP_HD void fr29_radix4(fr29& a0, fr29& a1, fr29& a2, fr29& a3, const fr29& w0, const fr29& w1, const fr29& w2) {
  fr29 s02, s13, x, d;
#pragma unroll
  for (int i = 0; i < N29; i++) { s02.l[i] = a0.l[i] + a2.l[i]; s13.l[i] = a1.l[i] + a3.l[i]; }
  const fr29 d02 = fr29_mul(fr29_sub_lazy(a0, a2), w0);
  const fr29 d13 = fr29_mul(fr29_sub_lazy(a1, a3), w1);
#pragma unroll
  for (int i = 0; i < N29; i++) { x.l[i] = s02.l[i] + s13.l[i]; d.l[i] = s02.l[i] + (Q29::eight_q_spread(i) - s13.l[i]); }
  a0 = fr29_reduce8(x);
  a1 = fr29_mul(d, w2);
  a2 = fr29_add_lazy(d02, d13);
  a3 = fr29_mul(fr29_sub_lazy(d02, d13), w2);
}
// twiddle in the reference's Montgomery form (w * 2^256) -> w * 2^261 mod q, re-sliced
BP_HD fr29 fr29_twiddle_from_mont(const fr_t& w_mont256) {
  fr_t t = w_mont256;
#pragma unroll
  for (int i = 0; i < 5; i++) Fr::dbl(t, t);
  return fr29_from_sat(t);
}

}  // namespace bp
