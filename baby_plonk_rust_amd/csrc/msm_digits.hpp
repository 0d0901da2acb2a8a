// msm_digits.hpp -- the digit radix of fixed-base MSM tables: how it is chosen (host) and how a scalar is cut into its digits (host and
// device: the kernels of msm_kernels.hpp and the CPU check of tests/hostcheck run the same lines).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "bigint.hpp"

namespace bp {

// ---- digit radix of fixed-base tables (MsmPlan::radix) -----------------------------------------------------------------------
// W windows, radix R: every scalar k < q < 2^255 needs signed digits |d| <= R / 2 with sum_w d_w R^w = k; with the bias trick of the
// kernels (k + sum_{w < W-1} (R / 2) R^w, unsigned digits, top window unsigned) that works as soon as R^W > 2^256.  Among the even R
// from the smallest such number up to 2 % above it the one with the fewest one bits is taken (the table rows are R^w P_i: one
// double-and-add by R per row), provided R^W < 1.9 * 2^256 (the error bound of the kernels' fixed-point division).  Widths whose
// smallest radix is within 10 % of 2^c keep power-of-two windows (radix 0): c = 16 with 16 windows wastes nothing.
namespace radix_detail {
struct Wide {                       // little-endian 32-bit limbs, enough for 2^512 and R^W < 2^258
  uint32_t l[18];
};
inline void wide_set(Wide& a, uint32_t v) { memset(&a, 0, sizeof a); a.l[0] = v; }
inline void wide_mul_small(Wide& a, uint32_t m) {
  uint64_t c = 0;
  for (int i = 0; i < 18; i++) { c += (uint64_t)a.l[i] * m; a.l[i] = (uint32_t)c; c >>= 32; }
}
inline int wide_cmp(const Wide& a, const Wide& b) {
  for (int i = 17; i >= 0; i--) if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
  return 0;
}
inline bool wide_sub_if_ge(Wide& a, const Wide& b) {      // a -= b when a >= b
  if (wide_cmp(a, b) < 0) return false;
  uint64_t br = 0;
  for (int i = 0; i < 18; i++) { const uint64_t d = (uint64_t)a.l[i] - b.l[i] - br; a.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
  return true;
}
inline Wide wide_pow(uint32_t R, uint32_t W) {           // R^W; W log2 R < 290 for every caller
  Wide r;
  wide_set(r, 1);
  for (uint32_t i = 0; i < W; i++) wide_mul_small(r, R);
  return r;
}
inline bool pow_gt_2p256(uint32_t R, uint32_t W) {
  const Wide p = wide_pow(R, W);
  for (int i = 17; i >= 9; i--) if (p.l[i]) return true;
  if (p.l[8] > 1) return true;
  if (p.l[8] == 0) return false;
  for (int i = 7; i >= 0; i--) if (p.l[i]) return true;
  return false;                                 // exactly 2^256: not greater
}
}  // namespace radix_detail


struct MsmRadix {
  uint32_t R = 0;            // 0: power-of-two windows
  uint32_t m[8] = {0};       // ceil(2^512 / R^W)
  uint32_t bias[9] = {0};    // sum_{w < W-1} (R / 2) R^w
};

// the radix of table width c with W windows (host only)
inline MsmRadix msm_radix_compute(uint32_t c, uint32_t W) {
  using namespace radix_detail;
  MsmRadix rc;
  if (c < 8 || c > 30 || W < 2 || W > 40) return rc;
  // smallest even R with R^W > 2^256
  uint32_t R = (uint32_t)floor(pow(2.0, 256.0 / W));
  R &= ~1u;
  if (R < 4) return rc;
  while (R > 4 && pow_gt_2p256(R - 2, W)) R -= 2;
  while (!pow_gt_2p256(R, W)) R += 2;
  if ((double)R > 0.9 * (double)(1u << c)) return rc;               // nothing to gain: the windows waste less than a tenth of the bucket range
  uint32_t best = 0, best_bits = 99;
  for (uint32_t cand = R; cand <= R + R / 50; cand += 2) {
    const Wide p = wide_pow(cand, W);
    bool small = true;                                              // R^W < 1.9 * 2^256
    for (int i = 17; i >= 9; i--) if (p.l[i]) small = false;
    if (p.l[8] != 1 || p.l[7] >= 0xE6666666u) small = false;
    if (!small) break;
    const uint32_t bits = (uint32_t)__builtin_popcount(cand);
    if (bits < best_bits) { best_bits = bits; best = cand; }
  }
  if (!best) return rc;
  // m = ceil(2^512 / R^W) by binary long division; the quotient is < 2^256 because R^W > 2^256
  const Wide d = wide_pow(best, W);
  Wide rem, q;
  wide_set(rem, 0);
  wide_set(q, 0);
  for (int bit = 512; bit >= 0; bit--) {
    // rem = 2 rem + bit of 2^512
    uint32_t carry = bit == 512 ? 1u : 0u;
    for (int i = 0; i < 18; i++) { const uint32_t nc = rem.l[i] >> 31; rem.l[i] = (rem.l[i] << 1) | carry; carry = nc; }
    if (wide_sub_if_ge(rem, d)) {
      if (bit >= 256) return rc;                                    // cannot happen (R^W > 2^256)
      q.l[bit >> 5] |= 1u << (bit & 31);
    }
  }
  bool exact = true;
  for (int i = 0; i < 18; i++) if (rem.l[i]) exact = false;
  if (!exact) {                                                     // ceil
    uint64_t cy = 1;
    for (int i = 0; i < 18 && cy; i++) { cy += q.l[i]; q.l[i] = (uint32_t)cy; cy >>= 32; }
  }
  for (int i = 8; i < 18; i++) if (q.l[i]) return rc;               // would not fit 8 limbs: cannot happen
  // bias = sum_{w < W - 1} (R / 2) R^w
  Wide bias, term;
  wide_set(bias, 0);
  wide_set(term, best >> 1);
  for (uint32_t w = 0; w + 1 < W; w++) {
    uint64_t cy = 0;
    for (int i = 0; i < 18; i++) { cy += (uint64_t)bias.l[i] + term.l[i]; bias.l[i] = (uint32_t)cy; cy >>= 32; }
    wide_mul_small(term, best);
  }
  for (int i = 8; i < 18; i++) if (bias.l[i]) return rc;            // bias < R^(W-1) < 2^256
  rc.R = best;
  for (int i = 0; i < 8; i++) { rc.m[i] = q.l[i]; rc.bias[i] = bias.l[i]; }
  return rc;
}

// Signed radix-R digits of the canonical integer k (8 limbs): emit(w, d) for every window w (top window first), |d| <= R / 2,
// sum_w d_w R^w = k.  Returns false -- before any emit -- when k is not below 2^255-ish (only a canonical-bytes input >= q, which the
// status word rejects): such a scalar contributes no entries.
//   k' = k + bias;  f = k' / R^W as a 288-bit fraction = limbs 7..15 of k' * m, + 2 units;  W times: f *= R, digit = integer part;
//   signed digit = digit - R / 2 below the top window (the bias put R / 2 into each of them).
// Exactness.  k < q < 0.906 * 2^255 and bias < R^W / (2 R) * 1.001 < 2^245 give k' < 0.455 * 2^256.  m = ceil(2^512 / R^W) makes
// k' m / 2^512 exceed k' / R^W by less than k' / 2^512 < 0.455 * 2^-256; leaving out the partial products below column 5 (< 2^195) and
// the limbs below limb 7 (< 2^224) loses less than (1 + 2^-26) 2^-288 of the fraction, the two units put back 2^-287: the fraction's
// error lies in (0, 0.455 * 2^-256 + 2^-287).  The digits of a fraction are those of k' exactly when that error is non-negative and
// smaller than 1 / R^W, and msm_radix_compute only accepts R^W < 1.9 * 2^256, i.e. 1 / R^W > 0.526 * 2^-256.  (tests/test_radix_digits.py
// runs these very lines on the CPU over edge values and random scalars for every width.)
template <class F>
BP_HD bool msm_radix_digits(const uint32_t k[8], uint32_t R, const uint32_t m[8], const uint32_t bias[8], uint32_t W, F&& emit) {
  const uint32_t half = R >> 1;
  uint32_t kb[8];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    carry += (uint64_t)k[j] + bias[j];
    kb[j] = (uint32_t)carry;
    carry >>= 32;
  }
  if (carry) return false;
  uint32_t prod[16];
#pragma unroll
  for (int j = 0; j < 16; j++) prod[j] = 0;
#pragma unroll
  for (int a = 0; a < 8; a++) {
    uint64_t c = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      if (a + b < 5) continue;
      const uint64_t t = (uint64_t)kb[a] * m[b] + prod[a + b] + c;
      prod[a + b] = (uint32_t)t;
      c = t >> 32;
    }
    prod[a + 8] = (uint32_t)c;
  }
  uint32_t f[9];
  uint64_t up = 2;
#pragma unroll
  for (int l = 0; l < 9; l++) {
    up += prod[7 + l];
    f[l] = (uint32_t)up;
    up >>= 32;
  }
  for (uint32_t w = W; w-- > 0;) {
    uint64_t c = 0;
#pragma unroll
    for (int l = 0; l < 9; l++) {
      const uint64_t t = (uint64_t)f[l] * R + c;
      f[l] = (uint32_t)t;
      c = t >> 32;
    }
    const uint32_t u = (uint32_t)c;              // digit w of k + bias, in [0, R)
    const bool top = w + 1 == W;
    if (top && u > half) return false;           // k >= q again (k < q keeps the top digit <= R / 2)
    emit(w, (int32_t)u - (top ? 0 : (int32_t)half));
  }
  return true;
}

}  // namespace bp
