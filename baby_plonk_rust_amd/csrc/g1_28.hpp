// g1_28.hpp -- the bucket-accumulation group law on the unsaturated field of fp28.hpp.
// Same complete mixed addition as the reference (Renes-Costello-Batina Algorithm 8, g1.rs:715-752), with
// lazy limbs; every intermediate's limb and value bounds are checked at compile time by the F28 types.
#pragma once
#include "fp28.hpp"
#include "g1.hpp"

namespace bp {

// widen bounds (a value that satisfies tighter bounds also satisfies looser ones)
template <class To, uint64_t A, uint32_t VA>
BP_HD To widen28(const F28<A, VA>& a) {
  static_assert(A <= To::limb_bound && VA <= To::value_bound, "cannot narrow bounds");
  To r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = a.l[i];
  return r;
}

// every projective coordinate that lives in a register file or in HBM between kernels has this type:
// limbs <= 2^28 + 7, value < 6p.  All three group operations below map (C28, C28, C28) to itself.
using C28 = F28<MASK28 + 8, 6>;
using PtY28 = F28<MASK28 + 8, 2>;     // y or 2p - y after one normalisation

struct alignas(128) g1_affine28 {     // x | y, 14 + 14 limbs (112 B) in a 128-byte HBM slot: one cache line per gathered point; identity = all zero
  F28n x, y;
};
struct g1_proj28 {                    // 168 bytes of limbs (stored padded to 176 B = 11 x 16 B)
  C28 x, y, z;
};
constexpr int PROJ28_WORDS = 44;

// 1 in the 28-bit Montgomery domain: 2^392 mod p, radix 2^28 (a literal table: the conversion through the saturated
// multiplier would otherwise run on every bucket flush of every wave)
struct One28 {
  BP_TABLE(limb, 0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu, 0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au)
};
BP_HD g1_proj28 g1_identity28() {     // (0 : 1 : 0)
  g1_proj28 r;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    r.x.l[i] = 0;
    r.y.l[i] = One28::limb(i);
    r.z.l[i] = 0;
  }
  return r;
}
BP_HD bool g1_is_identity28(const g1_proj28& p) {     // Z == 0 mod p, Z lazy (< 6p): compare after full reduction
  fp_t z = fp_from_28(p.z);
  return big_is_zero(z);
}
BP_HD g1_affine28 g1_affine_to_28(const g1_affine& p) {
  g1_affine28 r;
  r.x = fp_to_28(p.x);                // (0,0) stays (0,0)
  r.y = fp_to_28(p.y);
  return r;
}
BP_HD g1_proj g1_proj_from_28(const g1_proj28& p) {
  g1_proj r;
  r.x = fp_from_28(p.x);
  r.y = fp_from_28(p.y);
  r.z = fp_from_28(p.z);
  return r;
}
// y -> 2p - y when neg (kept lazy, one normalisation); limb-select keeps the wave convergent
BP_HD PtY28 pt_y_signed(const F28n& y, bool neg) {
  F28<0, 0> zero;
#pragma unroll
  for (int i = 0; i < N28; i++) zero.l[i] = 0;
  auto ny = norm28(sub28<2, 29>(zero, y));          // 2p - y  in (p, 2p]
  PtY28 a = widen28<PtY28>(ny), b = widen28<PtY28>(y), r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = neg ? a.l[i] : b.l[i];
  return r;
}

// acc += (x2, y2)   -- RCB Algorithm 8 (g1.rs:715-752); (x2, y2) must not be the identity (caller selects)
BP_HD void g1_add_mixed28(g1_proj28& acc, const F28n& x2, const PtY28& y2) {
  const C28 &X1 = acc.x, &Y1 = acc.y, &Z1 = acc.z;
  auto t0 = mul28(X1, x2);
  auto t1 = mul28(Y1, y2);
  auto t3 = mul28(add28(x2, y2), add28(X1, Y1));
  auto t3s = norm28(sub28<8, 30>(t3, add28(t0, t1)));         // t3 - (t0 + t1)
  auto t4 = add28(mul28(y2, Z1), Y1);                          // y2 Z1 + Y1
  auto y3a = norm28(add28(mul28(x2, Z1), X1));                 // x2 Z1 + X1
  auto t0x3 = mulk28<3>(t0);                                   // 3 t0
  auto t2 = norm28(mulk28<12>(Z1));                            // 3b Z1
  auto z3 = add28(t1, t2);
  auto t1s = sub28<80, 29>(t1, t2);
  auto y3 = norm28(mulk28<12>(y3a));                           // 3b (x2 Z1 + X1)
  // the three outputs are two-term products: one fused reduction each (mul28_2)
  auto x3 = mul28_2(t3s, t1s, t4, neg28<128, 29>(y3));         // t3 t1 - t4 y3
  auto yo = mul28_2(t1s, z3, y3, t0x3);                        // t1 z3 + y3 t0
  auto zo = mul28_2(z3, t4, t0x3, t3s);                        // z3 t4 + t0 t3
  acc.x = widen28<C28>(x3);
  acc.y = widen28<C28>(yo);
  acc.z = widen28<C28>(zo);
}

// r = a + b   -- RCB Algorithm 7 (g1.rs:670-712), complete
BP_HD void g1_add28(g1_proj28& r, const g1_proj28& a, const g1_proj28& b) {
  auto t0 = mul28(a.x, b.x);
  auto t1 = mul28(a.y, b.y);
  auto t2 = mul28(a.z, b.z);
  auto t3 = norm28(sub28<8, 30>(mul28(add28(a.x, a.y), add28(b.x, b.y)), add28(t0, t1)));
  auto t4 = norm28(sub28<8, 30>(mul28(add28(a.y, a.z), add28(b.y, b.z)), add28(t1, t2)));
  auto y3a = norm28(sub28<8, 30>(mul28(add28(a.x, a.z), add28(b.x, b.z)), add28(t0, t2)));
  auto t0x3 = mulk28<3>(t0);
  auto t2b = norm28(mulk28<12>(t2));                            // 3b t2
  auto z3 = add28(t1, t2b);
  auto t1s = sub28<32, 29>(t1, t2b);
  auto y3 = norm28(mulk28<12>(y3a));
  auto x3 = mul28_2(t3, t1s, t4, neg28<128, 29>(y3));
  auto yo = mul28_2(t1s, z3, y3, t0x3);
  auto zo = mul28_2(z3, t4, t0x3, t3);
  r.x = widen28<C28>(x3);
  r.y = widen28<C28>(yo);
  r.z = widen28<C28>(zo);
}

// ---- the same complete addition split over the lanes of a group ("cooperative" form) ---------------------------------
// The bucket reduction and fix-up kernels are latency bound: few independent additions, each a ~6 600-instruction chain
// on one lane.  Algorithm 7 is six independent products followed by three independent two-term products, so a group of
// lanes that all hold both points can run it as  stage A: lane r computes product r (r < 6);  stage B: lane r computes
// output coordinate r (r < 3)  -- a chain of one mul28 + one mul28_2 instead of 6 + 3.  The stages are pure functions of
// (role, operands) with role-independent types (operands are widened to the union of the bounds of the candidates a slot
// can receive, selected limb by limb), so the lanes of a wave stay convergent; the exchange of the six products between
// the lanes is the caller's (shuffles on the GPU, a loop over roles in the host check).
using CoopProd = F28<MASK28, 2>;                                   // a stage-A product
using CoopSum = decltype(add28(C28(), C28()));                      // a coordinate or a sum of two

template <class U, class A, class B, class C>
BP_HD U select3_28(uint32_t which, const A& a, const B& b, const C& c) {
  const U ua = widen28<U>(a), ub = widen28<U>(b), uc = widen28<U>(c);
  U r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = which == 0 ? ua.l[i] : (which == 1 ? ub.l[i] : uc.l[i]);
  return r;
}
template <class A, class B, class C>
struct Union28 {
  static constexpr uint64_t lb = A::limb_bound > B::limb_bound ? (A::limb_bound > C::limb_bound ? A::limb_bound : C::limb_bound)
                                                               : (B::limb_bound > C::limb_bound ? B::limb_bound : C::limb_bound);
  static constexpr uint32_t vb = A::value_bound > B::value_bound ? (A::value_bound > C::value_bound ? A::value_bound : C::value_bound)
                                                                 : (B::value_bound > C::value_bound ? B::value_bound : C::value_bound);
  using type = F28<lb, vb>;
};

// stage A: role 0..5 -> X1 X2, Y1 Y2, Z1 Z2, (X1+Y1)(X2+Y2), (Y1+Z1)(Y2+Z2), (X1+Z1)(X2+Z2); other roles -> 0
BP_HD CoopProd g1_add28_coop_a(uint32_t role, const g1_proj28& a, const g1_proj28& b) {
  const bool ux = role == 0 || role == 3 || role == 5, uy = role == 1 || role == 3 || role == 4, uz = role == 2 || role == 4 || role == 5;
  CoopSum u, v;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    u.l[i] = (ux ? a.x.l[i] : 0u) + (uy ? a.y.l[i] : 0u) + (uz ? a.z.l[i] : 0u);      // at most two of the three are selected
    v.l[i] = (ux ? b.x.l[i] : 0u) + (uy ? b.y.l[i] : 0u) + (uz ? b.z.l[i] : 0u);
  }
  return widen28<CoopProd>(mul28(u, v));
}
// stage B: role 0, 1, 2 -> X3, Y3, Z3 from the six products p[0..5] of stage A
BP_HD C28 g1_add28_coop_b(uint32_t role, const CoopProd p[6]) {
  const CoopProd &t0 = p[0], &t1 = p[1], &t2 = p[2];
  auto t3 = norm28(sub28<8, 30>(p[3], add28(t0, t1)));
  auto t4 = norm28(sub28<8, 30>(p[4], add28(t1, t2)));
  auto y3a = norm28(sub28<8, 30>(p[5], add28(t0, t2)));
  auto t0x3 = norm28(mulk28<3>(t0));
  auto t2b = norm28(mulk28<12>(t2));                               // 3b t2
  auto z3 = norm28(add28(t1, t2b));
  auto t1s = norm28(sub28<32, 29>(t1, t2b));
  auto y3 = norm28(mulk28<12>(y3a));
  auto ny3 = norm28(neg28<128, 29>(y3));
  //            first product      second product (the large-valued factors share slot s)
  // role 0:  X3 = t3 t1s     +   t4 (-y3)
  // role 1:  Y3 = z3 t1s     +   t0x3 y3
  // role 2:  Z3 = t0x3 t3    +   t4 z3
  using P = typename Union28<decltype(t3), decltype(z3), decltype(t0x3)>::type;
  using Q = typename Union28<decltype(t1s), decltype(t1s), decltype(t3)>::type;
  using R = typename Union28<decltype(t4), decltype(t0x3), decltype(t4)>::type;
  using S = typename Union28<decltype(ny3), decltype(y3), decltype(z3)>::type;
  const uint32_t w = role < 3 ? role : 2;
  return widen28<C28>(mul28_2(select3_28<P>(w, t3, z3, t0x3), select3_28<Q>(w, t1s, t1s, t3), select3_28<R>(w, t4, t0x3, t4),
                              select3_28<S>(w, ny3, y3, z3)));
}

// r = 2 p   -- RCB Algorithm 9 (g1.rs:638-667)
BP_HD void g1_double28(g1_proj28& r, const g1_proj28& p) {
  auto t0 = mul28(p.y, p.y);
  auto z8 = mulk28<8>(t0);
  auto t1 = mul28(p.y, p.z);
  auto t2 = norm28(mulk28<12>(mul28(p.z, p.z)));               // 3b Z^2
  auto x3 = mul28(t2, z8);
  auto y3 = add28(t0, t2);
  auto zo = mul28(t1, z8);
  auto t0s = sub28<80, 30>(t0, mulk28<3>(t2));                 // t0 - 3 t2
  auto yo = add28(x3, mul28(t0s, y3));
  auto xo = mulk28<2>(mul28(t0s, mul28(p.x, p.y)));
  r.x = widen28<C28>(norm28(xo));
  r.y = widen28<C28>(norm28(yo));
  r.z = widen28<C28>(norm28(zo));
}

// ---- canonical form and inversion in the 28-bit Montgomery domain (table builder: srs_window_tables) ---------------------------
using M28 = F28<MASK28, 2>;                                         // a Montgomery product: limbs <= 2^28 - 1, value < 2p
// value < 2p with limbs < 2^28 (a mul28 result)  ->  the unique representative in [0, p), limbs < 2^28
BP_HD F28n canon28(const M28& a) {
  uint32_t t[N28];
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N28; i++) {
    const int32_t d = (int32_t)a.l[i] - (int32_t)P28::mod(i) + borrow;      // |d| < 2^29
    t[i] = (uint32_t)d & MASK28;
    borrow = d >> 28;                                                       // 0 or -1
  }
  F28n r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = borrow < 0 ? a.l[i] : t[i];         // a < p: keep a; else a - p
  return r;
}
// a^(p-2) (fp.rs:346-358): every lane runs the same exponent, so the multiply steps are wave-uniform branches.  0 -> 0.
BP_HD M28 fp28_invert(const M28& a) {
  M28 r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = One28::limb(i);
  for (int w = 11; w >= 0; w--) {
    const uint32_t e = FpParams::mod_minus_2(w);
    for (int b = (w == 11 ? 28 : 31); b >= 0; b--) {
      r = mul28(r, r);
      if ((e >> b) & 1) r = mul28(r, a);
    }
  }
  return r;
}

// inversion of the reference's Montgomery limbs (x 2^384) through the 28-bit domain: one conversion each way, the 381-bit power itself on
// mul28 (~500 instructions per product against ~700 on saturated limbs).  0 -> 0.
BP_HD void fp_invert_via28(fp_t& r, const fp_t& a) { r = fp_from_28(fp28_invert(widen28<M28>(fp_to_28(a)))); }

// r = k * p for a small non-negative integer k (< 2^nbits), MSB-first double-and-add
BP_HD void g1_mul_small28(g1_proj28& r, const g1_proj28& p, uint32_t k, int nbits) {
  g1_proj28 acc = g1_identity28();
  for (int i = nbits - 1; i >= 0; i--) {
    g1_double28(acc, acc);
    if ((k >> i) & 1) g1_add28(acc, acc, p);
  }
  r = acc;
}

}  // namespace bp
