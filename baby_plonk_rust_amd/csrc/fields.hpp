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
// a^-1 by Fermat (scalar.rs:416-511 uses an addition chain for the same exponent q-2)
BP_HD void fr_invert(fr_t& r, const fr_t& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = FrParams::mod_minus_2(i);
  Fr::pow(r, a, e, 8);
}
// fp.rs:346-358
BP_HD void fp_invert(fp_t& r, const fp_t& a) {
  uint32_t e[12];
#pragma unroll
  for (int i = 0; i < 12; i++) e[i] = FpParams::mod_minus_2(i);
  Fp::pow(r, a, e, 12);
}

}  // namespace bp
