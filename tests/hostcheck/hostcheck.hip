// hostcheck.hip -- TEST-ONLY host build of the product's __host__ __device__ arithmetic headers
// (baby_plonk_rust_amd/csrc/{bigint,fields,g1}.hpp) so their logic can be checked against the
// oracle on a machine without a GPU.  Never linked into the product library.
#define BP_HOST_USE_DEVICE_ALGO 1   /* run the 32-bit column multiplier (the device algorithm) on the CPU */
#include "../../baby_plonk_rust_amd/csrc/g1.hpp"
#include <string.h>
using namespace bp;

extern "C" {
void hc_fr_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::mul(z, x, y); memcpy(r, &z, 32); }
void hc_fr_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::add(z, x, y); memcpy(r, &z, 32); }
void hc_fr_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::sub(z, x, y); memcpy(r, &z, 32); }
void hc_fr_neg(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); Fr::neg(z, x); memcpy(r, &z, 32); }
void hc_fr_inv(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); fr_invert(z, x); memcpy(r, &z, 32); }
void hc_fr_from_mont(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); Fr::from_mont(z, x); memcpy(r, &z, 32); }
void hc_fp_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::mul(z, x, y); memcpy(r, &z, 48); }
void hc_fp_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::add(z, x, y); memcpy(r, &z, 48); }
void hc_fp_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::sub(z, x, y); memcpy(r, &z, 48); }
void hc_fp_neg(uint32_t* r, const uint32_t* a) { fp_t x, z; memcpy(&x, a, 48); Fp::neg(z, x); memcpy(r, &z, 48); }
void hc_fp_inv(uint32_t* r, const uint32_t* a) { fp_t x, z; memcpy(&x, a, 48); fp_invert(z, x); memcpy(r, &z, 48); }
void hc_g1_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { g1_proj x, y, z; memcpy(&x, a, 144); memcpy(&y, b, 144); g1_add(z, x, y); memcpy(r, &z, 144); }
void hc_g1_double(uint32_t* r, const uint32_t* a) { g1_proj x, z; memcpy(&x, a, 144); g1_double(z, x); memcpy(r, &z, 144); }
void hc_g1_add_mixed(uint32_t* r, const uint32_t* a, const uint32_t* b) { g1_proj x, z; g1_affine y; memcpy(&x, a, 144); memcpy(&y, b, 96); g1_add_mixed(z, x, y); memcpy(r, &z, 144); }
void hc_g1_mul_scalar(uint32_t* r, const uint32_t* a, const uint32_t* k) { g1_proj x, z; fr_t s; memcpy(&x, a, 144); memcpy(&s, k, 32); g1_mul_scalar(z, x, s); memcpy(r, &z, 144); }
void hc_g1_mul_small(uint32_t* r, const uint32_t* a, uint32_t k, int nbits) { g1_proj x, z; memcpy(&x, a, 144); g1_mul_small(z, x, k, nbits); memcpy(r, &z, 144); }
void hc_g1_to_affine(uint32_t* r, const uint32_t* a) { g1_proj x; memcpy(&x, a, 144); g1_affine z = g1_to_affine(x); memcpy(r, &z, 96); }
}

// ---- unsaturated 14 x 28 field (fp28.hpp / g1_28.hpp) ----
#include "../../baby_plonk_rust_amd/csrc/g1_28.hpp"
extern "C" {
// r (saturated Montgomery) = a * b through the 28-bit path: convert in, mul28, convert out
void hc_fp28_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  fp_t x, y; memcpy(&x, a, 48); memcpy(&y, b, 48);
  fp_t z = fp_from_28(mul28(fp_to_28(x), fp_to_28(y)));
  memcpy(r, &z, 48);
}
void hc_fp28_roundtrip(uint32_t* r, const uint32_t* a) { fp_t x; memcpy(&x, a, 48); fp_t z = fp_from_28(fp_to_28(x)); memcpy(r, &z, 48); }
// acc (144-byte saturated projective) += sign * affine point (96-byte saturated), n_rep times, through g1_add_mixed28
void hc_g1_28_add_mixed(uint32_t* r, const uint32_t* acc_in, const uint32_t* pt, int neg, int reps) {
  g1_proj a; g1_affine p; memcpy(&a, acc_in, 144); memcpy(&p, pt, 96);
  g1_proj28 acc;
  acc.x = widen28<C28>(fp_to_28(a.x)); acc.y = widen28<C28>(fp_to_28(a.y)); acc.z = widen28<C28>(fp_to_28(a.z));
  g1_affine28 q = g1_affine_to_28(p);
  for (int i = 0; i < reps; i++) g1_add_mixed28(acc, q.x, pt_y_signed(q.y, neg != 0));
  g1_proj out = g1_proj_from_28(acc);
  memcpy(r, &out, 144);
}
static g1_proj28 to28(const g1_proj& a) { g1_proj28 r; r.x = widen28<C28>(fp_to_28(a.x)); r.y = widen28<C28>(fp_to_28(a.y)); r.z = widen28<C28>(fp_to_28(a.z)); return r; }
// r = (((a + b) + b) ... reps times) through g1_add28; doubling chain through g1_double28; small multiples
void hc_g1_28_add(uint32_t* r, const uint32_t* a_in, const uint32_t* b_in, int reps) {
  g1_proj a, b; memcpy(&a, a_in, 144); memcpy(&b, b_in, 144);
  g1_proj28 x = to28(a), y = to28(b);
  for (int i = 0; i < reps; i++) g1_add28(x, x, y);
  g1_proj out = g1_proj_from_28(x); memcpy(r, &out, 144);
}
// the same chain through the cooperative form: stage A per role, "exchange", stage B per role (what 8 GPU lanes do together)
void hc_g1_28_add_coop(uint32_t* r, const uint32_t* a_in, const uint32_t* b_in, int reps) {
  g1_proj a, b; memcpy(&a, a_in, 144); memcpy(&b, b_in, 144);
  g1_proj28 x = to28(a), y = to28(b);
  for (int i = 0; i < reps; i++) {
    CoopProd prod[6];
    for (uint32_t role = 0; role < 6; role++) prod[role] = g1_add28_coop_a(role, x, y);
    CoopProd idle = g1_add28_coop_a(7, x, y);                      // lanes 6, 7 select nothing
    for (int j = 0; j < N28; j++) if (idle.l[j] != 0) prod[0].l[0] ^= 1;   // would corrupt the result
    g1_proj28 nx;
    nx.x = g1_add28_coop_b(0, prod); nx.y = g1_add28_coop_b(1, prod); nx.z = g1_add28_coop_b(2, prod);
    x = nx;
  }
  g1_proj out = g1_proj_from_28(x); memcpy(r, &out, 144);
}
void hc_g1_28_double(uint32_t* r, const uint32_t* a_in, int reps) {
  g1_proj a; memcpy(&a, a_in, 144);
  g1_proj28 x = to28(a);
  for (int i = 0; i < reps; i++) g1_double28(x, x);
  g1_proj out = g1_proj_from_28(x); memcpy(r, &out, 144);
}
void hc_g1_28_mul_small(uint32_t* r, const uint32_t* a_in, uint32_t k, int nbits) {
  g1_proj a; memcpy(&a, a_in, 144);
  g1_proj28 x; g1_mul_small28(x, to28(a), k, nbits);
  g1_proj out = g1_proj_from_28(x); memcpy(r, &out, 144);
}
int hc_g1_28_is_identity(const uint32_t* a_in) { g1_proj a; memcpy(&a, a_in, 144); return g1_is_identity28(to28(a)) ? 1 : 0; }
void hc_g1_28_identity(uint32_t* r) { g1_proj out = g1_proj_from_28(g1_identity28()); memcpy(r, &out, 144); }
}

// ---- unsaturated 9 x 29 scalar field (fr29.hpp) ----
#include <stdio.h>
#include <stdlib.h>
#define BP_FR29_CHECK(cond) do { if (!(cond)) { fprintf(stderr, "fr29 bound violated: %s\n", #cond); abort(); } } while (0)
#include "../../baby_plonk_rust_amd/csrc/fr29.hpp"
extern "C" {
// (u, v) in the reference's Montgomery form, w in Montgomery form; outputs canonical Montgomery: u+v, (u-v)*w after `reps` chained butterflies
void hc_fr29_butterfly(uint32_t* ru, uint32_t* rv, const uint32_t* u_in, const uint32_t* v_in, const uint32_t* w_in, int reps) {
  fr_t u, v, w; memcpy(&u, u_in, 32); memcpy(&v, v_in, 32); memcpy(&w, w_in, 32);
  fr29 a = fr29_from_sat(u), b = fr29_from_sat(v), t = fr29_twiddle_from_mont(w);
  for (int i = 0; i < reps; i++) fr29_butterfly(a, b, t);
  fr_t x = fr29_to_sat_canonical(a), y = fr29_to_sat_canonical(b);
  memcpy(ru, &x, 32); memcpy(rv, &y, 32);
}
// radix-4 group: a[4] any 256-bit values < 2q (NOT necessarily canonical), w[3] Montgomery twiddles; lazy != 0: fr29_radix4, else four
// fr29_butterfly calls; outputs canonical
void hc_fr29_radix4(uint32_t* out, const uint32_t* a_in, const uint32_t* w_in, int lazy) {
  fr29 a[4], w[3];
  for (int i = 0; i < 4; i++) { fr_t t; memcpy(&t, a_in + 8 * i, 32); a[i] = fr29_from_sat(t); }
  for (int i = 0; i < 3; i++) { fr_t t; memcpy(&t, w_in + 8 * i, 32); w[i] = fr29_twiddle_from_mont(t); }
  if (lazy) {
    fr29_radix4(a[0], a[1], a[2], a[3], w[0], w[1], w[2]);
  } else {
    fr29_butterfly(a[0], a[2], w[0]); fr29_butterfly(a[1], a[3], w[1]);
    fr29_butterfly(a[0], a[1], w[2]); fr29_butterfly(a[2], a[3], w[2]);
  }
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < N29; j++) BP_FR29_CHECK(a[i].l[j] <= MASK29);          // normalised, and below 2q:
    fr29 t; BP_FR29_CHECK(fr29_sub_exact(t, a[i], [](int k) { return Q29::two_q(k); }) == 1);
    fr_t z = fr29_to_sat_canonical(a[i]); memcpy(out + 8 * i, &z, 32);
  }
}
void hc_fr29_roundtrip(uint32_t* r, const uint32_t* a_in) { fr_t a; memcpy(&a, a_in, 32); fr_t z = fr29_to_sat_canonical(fr29_from_sat(a)); memcpy(r, &z, 32); }
}

// ---- digit radix of fixed-base MSM tables (msm_digits.hpp): the host's choice and the kernels' digit cutter, on the CPU ----
#include "../../baby_plonk_rust_amd/csrc/msm_digits.hpp"
extern "C" {
// out[0] = R (0: power-of-two windows), out[1..8] = m, out[9..16] = bias
void hc_radix_info(uint32_t* out, uint32_t c, uint32_t W) {
  const MsmRadix r = msm_radix_compute(c, W);
  out[0] = r.R;
  for (int i = 0; i < 8; i++) { out[1 + i] = r.m[i]; out[9 + i] = r.bias[i]; }
}
// digits[w] (int32, w < W) of the 8-limb integer k for the radix of (c, W); returns 1 when the cutter accepted k, 0 when it emitted nothing
int hc_radix_digits(int32_t* digits, const uint32_t* k, uint32_t c, uint32_t W) {
  const MsmRadix r = msm_radix_compute(c, W);
  if (!r.R) return -1;
  for (uint32_t w = 0; w < W; w++) digits[w] = 0x7fffffff;
  return msm_radix_digits(k, r.R, r.m, r.bias, W, [&](uint32_t w, int32_t d) { digits[w] = d; }) ? 1 : 0;
}
}
