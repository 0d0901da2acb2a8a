// g1.hpp -- BLS12-381 G1 on y^2 = x^3 + 4 with the complete homogeneous-projective formulas of
// Renes-Costello-Batina (eprint 2015/1060), the same formulas the reference uses
// (lib/bls12_381/src/g1.rs:638-752).  Complete formulas are branch-free: identity inputs, P+P and
// P+(-P) need no special cases, which is what a 64-lane wavefront wants (SURVEY.md section 7,
// "complete vs incomplete formulas": an SRS with tau = 1 makes every bucket add a doubling).
//
// Device layout: affine point = x | y, 24 x u32 = 96 B, Montgomery limbs; the affine identity is
// stored as x = y = 0 (not on the curve, so unambiguous).  Projective = x | y | z, 144 B, identical
// to the reference's G1Projective memory image (g1.rs:442-446).
#pragma once
#include "fields.hpp"

namespace bp {

struct g1_affine {
  fp_t x, y;
};
struct g1_proj {
  fp_t x, y, z;
};

BP_HD bool g1_affine_is_identity(const g1_affine& p) { return big_is_zero(p.x) && big_is_zero(p.y); }
BP_HD g1_proj g1_identity() {   // g1.rs:605-611
  g1_proj r;
  r.x = Fp::zero();
  r.y = Fp::one();
  r.z = Fp::zero();
  return r;
}
BP_HD g1_affine g1_affine_generator() {
  g1_affine g;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    g.x.l[i] = FpParams::gen_x(i);
    g.y.l[i] = FpParams::gen_y(i);
  }
  return g;
}
BP_HD bool g1_is_identity(const g1_proj& p) { return big_is_zero(p.z); }

// g1.rs:597-601
BP_HD void fp_mul_by_3b(fp_t& r, const fp_t& a) {
  fp_t t, u;
  Fp::dbl(t, a);
  Fp::dbl(t, t);
  Fp::dbl(u, t);
  Fp::add(r, u, t);
}

// Algorithm 8 (mixed), g1.rs:715-752.  `b` must not be the identity (caller selects).
BP_HD void g1_add_mixed_nz(g1_proj& r, const g1_proj& a, const g1_affine& b) {
  fp_t t0, t1, t2, t3, t4, x3, y3, z3;
  Fp::mul(t0, a.x, b.x);
  Fp::mul(t1, a.y, b.y);
  Fp::add(t3, b.x, b.y);
  Fp::add(t4, a.x, a.y);
  Fp::mul(t3, t3, t4);
  Fp::add(t4, t0, t1);
  Fp::sub(t3, t3, t4);
  Fp::mul(t4, b.y, a.z);
  Fp::add(t4, t4, a.y);
  Fp::mul(y3, b.x, a.z);
  Fp::add(y3, y3, a.x);
  Fp::dbl(x3, t0);
  Fp::add(t0, x3, t0);
  fp_mul_by_3b(t2, a.z);
  Fp::add(z3, t1, t2);
  Fp::sub(t1, t1, t2);
  fp_mul_by_3b(y3, y3);
  Fp::mul(x3, t4, y3);
  Fp::mul(t2, t3, t1);
  Fp::sub(x3, t2, x3);
  Fp::mul(y3, y3, t0);
  Fp::mul(t1, t1, z3);
  Fp::add(y3, t1, y3);
  Fp::mul(t0, t0, t3);
  Fp::mul(z3, z3, t4);
  Fp::add(z3, z3, t0);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}
BP_HD void g1_add_mixed(g1_proj& r, const g1_proj& a, const g1_affine& b) {
  g1_proj t;
  g1_add_mixed_nz(t, a, b);
  bool skip = g1_affine_is_identity(b);
  big_select(r.x, skip, a.x, t.x);
  big_select(r.y, skip, a.y, t.y);
  big_select(r.z, skip, a.z, t.z);
}

// Algorithm 7 (complete projective add), g1.rs:670-712
BP_HD void g1_add(g1_proj& r, const g1_proj& a, const g1_proj& b) {
  fp_t t0, t1, t2, t3, t4, x3, y3, z3;
  Fp::mul(t0, a.x, b.x);
  Fp::mul(t1, a.y, b.y);
  Fp::mul(t2, a.z, b.z);
  Fp::add(t3, a.x, a.y);
  Fp::add(t4, b.x, b.y);
  Fp::mul(t3, t3, t4);
  Fp::add(t4, t0, t1);
  Fp::sub(t3, t3, t4);
  Fp::add(t4, a.y, a.z);
  Fp::add(x3, b.y, b.z);
  Fp::mul(t4, t4, x3);
  Fp::add(x3, t1, t2);
  Fp::sub(t4, t4, x3);
  Fp::add(x3, a.x, a.z);
  Fp::add(y3, b.x, b.z);
  Fp::mul(x3, x3, y3);
  Fp::add(y3, t0, t2);
  Fp::sub(y3, x3, y3);
  Fp::dbl(x3, t0);
  Fp::add(t0, x3, t0);
  fp_mul_by_3b(t2, t2);
  Fp::add(z3, t1, t2);
  Fp::sub(t1, t1, t2);
  fp_mul_by_3b(y3, y3);
  Fp::mul(x3, t4, y3);
  Fp::mul(t2, t3, t1);
  Fp::sub(x3, t2, x3);
  Fp::mul(y3, y3, t0);
  Fp::mul(t1, t1, z3);
  Fp::add(y3, t1, y3);
  Fp::mul(t0, t0, t3);
  Fp::mul(z3, z3, t4);
  Fp::add(z3, z3, t0);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}

// Algorithm 9 (doubling), g1.rs:638-667.  For Z = 0 the formulas give (0 : Y^3*.. : 0), still the
// identity class, so no select is needed for group correctness.
BP_HD void g1_double(g1_proj& r, const g1_proj& p) {
  fp_t t0, t1, t2, x3, y3, z3;
  Fp::sqr(t0, p.y);
  Fp::dbl(z3, t0);
  Fp::dbl(z3, z3);
  Fp::dbl(z3, z3);
  Fp::mul(t1, p.y, p.z);
  Fp::sqr(t2, p.z);
  fp_mul_by_3b(t2, t2);
  Fp::mul(x3, t2, z3);
  Fp::add(y3, t0, t2);
  Fp::mul(z3, t1, z3);
  Fp::dbl(t1, t2);
  Fp::add(t2, t1, t2);
  Fp::sub(t0, t0, t2);
  Fp::mul(y3, t0, y3);
  Fp::add(y3, x3, y3);
  Fp::mul(t1, p.x, p.y);
  Fp::mul(x3, t0, t1);
  Fp::dbl(x3, x3);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}

BP_HD void g1_neg_affine(g1_affine& r, const g1_affine& p) {
  r.x = p.x;
  Fp::neg(r.y, p.y);     // (0,0) stays (0,0)
}
BP_HD g1_proj g1_from_affine(const g1_affine& p) {
  g1_proj r;
  bool inf = g1_affine_is_identity(p);
  r.x = p.x;
  big_select(r.y, inf, Fp::one(), p.y);
  big_select(r.z, inf, Fp::zero(), Fp::one());
  return r;
}
// r = k * p for a small non-negative integer k (< 2^nbits), MSB-first double-and-add
BP_HD void g1_mul_small(g1_proj& r, const g1_proj& p, uint32_t k, int nbits) {
  g1_proj acc = g1_identity();
  for (int i = nbits - 1; i >= 0; i--) {
    g1_double(acc, acc);
    if ((k >> i) & 1) g1_add(acc, acc, p);
  }
  r = acc;
}
// r = k * p for a canonical 256-bit scalar given as 8 limbs (g1.rs:754-774: 255 double-and-add steps)
BP_HD void g1_mul_scalar(g1_proj& r, const g1_proj& p, const fr_t& k_canonical) {
  g1_proj acc = g1_identity();
  for (int i = 254; i >= 0; i--) {
    g1_double(acc, acc);
    if ((k_canonical.l[i >> 5] >> (i & 31)) & 1) g1_add(acc, acc, p);
  }
  r = acc;
}
// affine normalisation: one inversion (g1.rs:49-63).  identity -> (0,0)
BP_HD g1_affine g1_to_affine(const g1_proj& p) {
  fp_t zinv;
  fp_invert(zinv, p.z);          // 0 -> 0, so the identity maps to (0,0)
  g1_affine r;
  Fp::mul(r.x, p.x, zinv);
  Fp::mul(r.y, p.y, zinv);
  return r;
}

}  // namespace bp
