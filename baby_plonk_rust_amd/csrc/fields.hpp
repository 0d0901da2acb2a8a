// fields.hpp -- the two BLS12-381 prime fields as parameter packs for bigint.hpp.
//   Fr : scalar field, 8 x 32-bit limbs,  Montgomery R = 2^256  (lib/bls12_381/src/scalar.rs:83-221)
//   Fp : base field,  12 x 32-bit limbs,  Montgomery R = 2^384  (lib/bls12_381/src/fp.rs:70-110)
// The 32-bit limb tables are the reference's 64-bit constants split in halves.
#pragma once
#include "bigint.hpp"

namespace bp {

#define BP_TABLE(name, ...)                                  \
  static BP_HD constexpr uint32_t name(int i) {              \
    constexpr uint32_t t[] = {__VA_ARGS__};                  \
    return t[i];                                             \
  }

struct FrParams {
  static constexpr int N = 8;
  static constexpr uint32_t INV32 = 0xffffffffu;             // low half of scalar.rs:164
  BP_TABLE(mod, 0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u)
  BP_TABLE(one, 0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u)
  BP_TABLE(r2, 0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u)
  // R^3 (scalar.rs:183-188), used by from_u512 (scalar.rs:323-339)
  BP_TABLE(r3, 0x439b73afu, 0xc62c1807u, 0x8cf06990u, 0x1b3e0d18u, 0xc7b5f418u, 0x73d13c71u, 0xc8db33e9u, 0x6e2a5bb9u)
  // scalar.rs:208-221
  BP_TABLE(root_of_unity, 0x5f0e466au, 0xb9b58d8cu, 0x1819d7ecu, 0x5b1b4c80u, 0x52a31e64u, 0x0af53ae3u, 0x19e9b27bu, 0x5bf3addau)
  BP_TABLE(root_of_unity_inv, 0xdcf3219au, 0x4256481au, 0x96b6cad3u, 0x45f37b7fu, 0x5f7a3b27u, 0xf9c3f1d7u, 0x658afd43u, 0x2d2fc049u)
  BP_TABLE(mod_minus_2, 0xffffffffu, 0xfffffffeu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u)
};

struct FpParams {
  static constexpr int N = 12;
  static constexpr uint32_t INV32 = 0xfffcfffdu;             // low half of fp.rs:80
  BP_TABLE(mod, 0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
           0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau)
  BP_TABLE(one, 0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
           0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u)
  BP_TABLE(r2, 0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
           0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u)
  BP_TABLE(mod_minus_2, 0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu,
           0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau)
  // generator, Montgomery form (g1.rs:197-217)
  BP_TABLE(gen_x, 0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u, 0xf3d0e747u, 0xf0ae6acdu,
           0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u)
  BP_TABLE(gen_y, 0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u, 0x50405194u, 0x51ac5829u,
           0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu)
};

using Fr = Mont<FrParams>;
using Fp = Mont<FpParams>;
using fr_t = Big<8>;
using fp_t = Big<12>;

BP_HD fr_t fr_root_of_unity(bool inverse) {
  fr_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = inverse ? FrParams::root_of_unity_inv(i) : FrParams::root_of_unity(i);
  return r;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// Host path of the two inversions below: binary extended Euclid on 64-bit limbs instead of the 255- / 381-bit power (65 us per Fp
// inversion on the build container's CPU, one per MSM result and per transcript point: most of the host epilogue).  The value
// inverted is the Montgomery image a R; (a R)^-1 = a^-1 R^-1 as a plain residue, and two products with R^2 bring it back to a^-1 R.
// Same unique representative in [0, p) as the power gives; 0 -> 0.
template <class F>
inline void invert_host(typename F::V& r, const typename F::V& a) {
  constexpr int N = F::N, M = N / 2;
  typedef unsigned __int128 u128;
  struct W { uint64_t w[M]; };
  auto is_zero = [](const W& x) { uint64_t o = 0; for (int i = 0; i < M; i++) o |= x.w[i]; return o == 0; };
  auto is_one = [](const W& x) { uint64_t o = x.w[0] ^ 1; for (int i = 1; i < M; i++) o |= x.w[i]; return o == 0; };
  auto shr1 = [](W& x, uint64_t top) {
    for (int i = 0; i + 1 < M; i++) x.w[i] = (x.w[i] >> 1) | (x.w[i + 1] << 63);
    x.w[M - 1] = (x.w[M - 1] >> 1) | (top << 63);
  };
  auto add = [](W& x, const W& y) -> uint64_t {
    u128 c = 0;
    for (int i = 0; i < M; i++) { c += (u128)x.w[i] + y.w[i]; x.w[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
  };
  auto sub = [](W& x, const W& y) -> uint64_t {                // x -= y, returns the borrow
    uint64_t b = 0;
    for (int i = 0; i < M; i++) {
      const u128 d = (u128)x.w[i] - y.w[i] - b;
      x.w[i] = (uint64_t)d;
      b = (uint64_t)(d >> 64) & 1;
    }
    return b;
  };
  auto geq = [](const W& x, const W& y) {
    for (int i = M - 1; i >= 0; i--)
      if (x.w[i] != y.w[i]) return x.w[i] > y.w[i];
    return true;
  };
  W u, v, x1, x2, p;
  const typename F::V mod = F::modulus();
  for (int i = 0; i < M; i++) {
    u.w[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
    p.w[i] = (uint64_t)mod.l[2 * i] | ((uint64_t)mod.l[2 * i + 1] << 32);
    x1.w[i] = i == 0;
    x2.w[i] = 0;
  }
  v = p;
  // caller-supplied limbs are not always canonical (a peer's record, a divisor's leading coefficient): reduce first, so that
  // a non-zero multiple of the modulus is the 0 it represents and gcd(u, p) = 1 holds below (u = p made the loop spin forever)
  while (geq(u, p)) sub(u, p);
  if (is_zero(u)) {
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return;
  }
  auto halve = [&](W& x) {                                      // x / 2 mod p, x in [0, p)
    if (x.w[0] & 1) { const uint64_t c = add(x, p); shr1(x, c); } else shr1(x, 0);
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u.w[0] & 1)) { shr1(u, 0); halve(x1); }
    while (!(v.w[0] & 1)) { shr1(v, 0); halve(x2); }
    if (geq(u, v)) {
      sub(u, v);
      if (sub(x1, x2)) add(x1, p);
    } else {
      sub(v, u);
      if (sub(x2, x1)) add(x2, p);
    }
  }
  const W& res = is_one(u) ? x1 : x2;
  typename F::V t;
  for (int i = 0; i < M; i++) {
    t.l[2 * i] = (uint32_t)res.w[i];
    t.l[2 * i + 1] = (uint32_t)(res.w[i] >> 32);
  }
  F::to_mont(t, t);
  F::to_mont(r, t);
}
#endif
// a^-1 by Fermat (scalar.rs:416-511 uses an addition chain for the same exponent q-2); on the host by binary Euclid (above)
BP_HD void fr_invert(fr_t& r, const fr_t& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
  invert_host<Fr>(r, a);
#else
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = FrParams::mod_minus_2(i);
  Fr::pow(r, a, e, 8);
#endif
}
// fp.rs:346-358
BP_HD void fp_invert(fp_t& r, const fp_t& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
  invert_host<Fp>(r, a);
#else
  uint32_t e[12];
#pragma unroll
  for (int i = 0; i < 12; i++) e[i] = FpParams::mod_minus_2(i);
  Fp::pow(r, a, e, 12);
#endif
}

}  // namespace bp
