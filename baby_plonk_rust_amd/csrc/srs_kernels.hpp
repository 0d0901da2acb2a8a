// srs_kernels.hpp -- SRS ingest / export / generation on the GPU.
//   srs_decode96   : 96-byte zkcrypto uncompressed encoding (g1.rs:246-322) -> device Montgomery affine
//   srs_encode96   : the inverse (G1Affine::to_uncompressed)
//   srs_generate   : P_i = s_i * G with s_i = tau^i (Setup::generate_srs, setup.rs:12-31) or
//                    s_i = a + i*d (synthetic benchmark points)
#pragma once
#include "g1.hpp"
#include "g1_28.hpp"

namespace bp {

__device__ __forceinline__ bool fp_lt_modulus(const fp_t& a) {
  fp_t t;
  return big_sub(t, a, Fp::modulus()) != 0;
}
// 48 big-endian bytes -> canonical limbs (fp.rs:179-190)
__device__ __forceinline__ fp_t fp_from_be48(const uint8_t* b) {
  fp_t r;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const uint8_t* p = b + 4 * (11 - i);
    r.l[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
  }
  return r;
}
__device__ __forceinline__ void fp_to_be48(uint8_t* b, const fp_t& a) {
#pragma unroll
  for (int i = 0; i < 12; i++) {
    uint8_t* p = b + 4 * (11 - i);
    p[0] = (uint8_t)(a.l[i] >> 24); p[1] = (uint8_t)(a.l[i] >> 16); p[2] = (uint8_t)(a.l[i] >> 8); p[3] = (uint8_t)a.l[i];
  }
}

// status: bit 0 = non-canonical coordinate or bad flag bits, bit 1 = point not on the curve
__global__ void __launch_bounds__(256) srs_decode96(const uint8_t* __restrict__ in, size_t n, g1_affine* __restrict__ out,
                                                     uint32_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t buf[96];
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + 96 * i);   // 96*i is 4-byte aligned
#pragma unroll
  for (int j = 0; j < 24; j++) {
    uint32_t w = src[j];
    buf[4 * j] = (uint8_t)w; buf[4 * j + 1] = (uint8_t)(w >> 8); buf[4 * j + 2] = (uint8_t)(w >> 16); buf[4 * j + 3] = (uint8_t)(w >> 24);
  }
  const uint32_t flags = buf[0] >> 5;              // compression | infinity | sort   (g1.rs:275-277)
  buf[0] &= 0x1f;
  fp_t x = fp_from_be48(buf), y = fp_from_be48(buf + 48);
  uint32_t bad = 0;
  if (!fp_lt_modulus(x) || !fp_lt_modulus(y)) bad |= 1;
  if (flags & 0b101) bad |= 1;                     // compressed / sort flag on an uncompressed point
  g1_affine p;
  if (flags & 0b010) {                             // infinity: coordinates must be zero (g1.rs:313-314)
    if (!big_is_zero(x) || !big_is_zero(y)) bad |= 1;
    p.x = Fp::zero();
    p.y = Fp::zero();
  } else {
    Fp::to_mont(p.x, x);
    Fp::to_mont(p.y, y);
    fp_t lhs, rhs, b4;                             // y^2 == x^3 + 4  (g1.rs:101-106)
    Fp::sqr(lhs, p.y);
    Fp::sqr(rhs, p.x);
    Fp::mul(rhs, rhs, p.x);
    b4 = Fp::one();
    Fp::dbl(b4, b4);
    Fp::dbl(b4, b4);
    Fp::add(rhs, rhs, b4);
    if (!big_eq(lhs, rhs)) bad |= 2;
  }
  if (bad) atomicOr(status, bad);
  out[i] = p;
}

__global__ void __launch_bounds__(256) srs_encode96(const g1_affine* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_affine p = in[i];
  uint8_t buf[96];
  if (g1_affine_is_identity(p)) {
#pragma unroll
    for (int j = 0; j < 96; j++) buf[j] = 0;
    buf[0] = 0x40;
  } else {
    fp_t x, y;
    Fp::from_mont(x, p.x);
    Fp::from_mont(y, p.y);
    fp_to_be48(buf, x);
    fp_to_be48(buf + 48, y);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(out + 96 * i);
#pragma unroll
  for (int j = 0; j < 24; j++)
    dst[j] = (uint32_t)buf[4 * j] | ((uint32_t)buf[4 * j + 1] << 8) | ((uint32_t)buf[4 * j + 2] << 16) | ((uint32_t)buf[4 * j + 3] << 24);
}

// G1Projective memory images (x | y | z Montgomery limbs, g1.rs:442-446) -> affine, the device side of
// bp_srs_load_projective144.  One lane normalises `group` points with one shared inversion (Montgomery's trick,
// G1Projective::batch_normalize g1.rs:806-839); z = 0 (the identity) becomes (0, 0).  The inversion (a 381-bit power, ~570
// multiplications) is most of what the kernel costs: one per 8 points made 131 072 of them at 2^20 points, 1.77 ms; one per 16: 1.1 ms.  A lane's points are
// `lanes` apart (lane, lane + lanes, ...: neighbouring lanes read neighbouring points) and the running products z'_0 ... z'_j wait in
// the x field of the output slot j, so the group can be as long as the launch wants.
__global__ void __launch_bounds__(256) srs_from_projective(const g1_proj* __restrict__ in, size_t n, uint32_t group, g1_affine* __restrict__ out) {
  const size_t lanes = (n + group - 1) / group;
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= lanes) return;
  fp_t run = Fp::one();
  uint32_t cnt = 0;
  for (size_t i = lane; i < n; i += lanes, cnt++) {
    fp_t z = in[i].z;
    if (big_is_zero(z)) z = Fp::one();                  // z' = z, or 1 where z = 0
    if (cnt == 0) run = z; else Fp::mul(run, run, z);
    out[i].x = run;
  }
  fp_t inv;
  fp_invert_via28(inv, run);
  for (uint32_t j = cnt; j-- > 0;) {
    const size_t i = lane + (size_t)j * lanes;
    const g1_proj p = in[i];
    const bool inf = big_is_zero(p.z);
    fp_t z = p.z, zinv;
    if (inf) z = Fp::one();
    if (j) {
      const fp_t before = out[i - lanes].x;
      Fp::mul(zinv, inv, before);
    } else {
      zinv = inv;
    }
    Fp::mul(inv, inv, z);
    g1_affine r;
    Fp::mul(r.x, p.x, zinv);
    Fp::mul(r.y, p.y, zinv);
    if (inf) { r.x = Fp::zero(); r.y = Fp::zero(); }
    out[i] = r;
  }
}

// ---- fixed-base generation (round 4) --------------------------------------------------------------------------------------------
// P = s G as 32 mixed additions from a table of the generator's multiples,  gen_table[j * 255 + (d - 1)] = d 2^(8 j) G  (j < 32, d = 1 .. 255;
// 28-bit affine slots, 1 MiB, built once per context by srs_gen_table), instead of 255 doublings + ~128 additions on saturated limbs per point;
// a lane makes SRS_GEN_GROUP consecutive points and normalises them with ONE inversion (Montgomery's trick, on the same 28-bit limbs).
// Same points as srs_generate (the unique affine representatives), 82 -> see profiles/r04_table_build_ab.txt.
constexpr uint32_t GEN_WINDOWS = 32, GEN_DIGITS = 255, SRS_GEN_GROUP = 8;
__device__ __forceinline__ void store_affine28_slot(g1_affine28* __restrict__ dst, const F28n& x, const F28n& y) {
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < N28; j++) { w[j] = x.l[j]; w[N28 + j] = y.l[j]; }
  uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int j = 0; j < 7; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
__device__ __forceinline__ g1_affine28 load_affine28_slot(const g1_affine28* __restrict__ p) {
  g1_affine28 r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < 7; j++) { const uint4 v = q[j]; w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w; }
#pragma unroll
  for (int j = 0; j < N28; j++) { r.x.l[j] = w[j]; r.y.l[j] = w[N28 + j]; }
  return r;
}
__global__ void __launch_bounds__(256) srs_gen_table(g1_affine28* __restrict__ table) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= GEN_WINDOWS * GEN_DIGITS) return;
  const uint32_t j = t / GEN_DIGITS, d = t % GEN_DIGITS + 1;
  fr_t s = Fr::zero();
  s.l[j >> 2] = d << (8 * (j & 3));                            // the integer d 2^(8 j)
  g1_proj r;
  g1_mul_scalar(r, g1_from_affine(g1_affine_generator()), s);
  const g1_affine28 a = g1_affine_to_28(g1_to_affine(r));
  store_affine28_slot(&table[t], a.x, a.y);
}
// mode 0: s_i = a^i (a = tau, Montgomery);  mode 1: s_i = a + i*d (Montgomery).  out[j] = s_{first + j} G, j < n.
__global__ void __launch_bounds__(256, 2) srs_generate_fb(fr_t a, fr_t d, int mode, size_t first, size_t n, const g1_affine28* __restrict__ gen_table,
                                                          g1_affine* __restrict__ out) {
  const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * SRS_GEN_GROUP;
  if (base >= n) return;
  const uint32_t cnt = n - base < SRS_GEN_GROUP ? (uint32_t)(n - base) : SRS_GEN_GROUP;
  g1_proj28 pts[SRS_GEN_GROUP];
  M28 prefix[SRS_GEN_GROUP];
  M28 one;
#pragma unroll
  for (int j = 0; j < N28; j++) one.l[j] = One28::limb(j);
  for (uint32_t k = 0; k < cnt; k++) {
    const size_t i = first + base + k;
    fr_t s;
    if (mode == 0) {
      uint32_t e[2] = {(uint32_t)i, (uint32_t)(i >> 32)};
      Fr::pow(s, a, e, 2);
    } else {
      fr_t idx = Fr::zero(), t;
      idx.l[0] = (uint32_t)i;
      idx.l[1] = (uint32_t)(i >> 32);
      Fr::to_mont(idx, idx);
      Fr::mul(t, idx, d);
      Fr::add(s, a, t);
    }
    Fr::from_mont(s, s);
    g1_proj28 acc = g1_identity28();
    for (uint32_t j = 0; j < GEN_WINDOWS; j++) {
      const uint32_t dg = (s.l[j >> 2] >> (8 * (j & 3))) & 255u;
      if (dg) {
        const g1_affine28 q = load_affine28_slot(&gen_table[j * GEN_DIGITS + dg - 1]);
        g1_add_mixed28(acc, q.x, widen28<PtY28>(q.y));
      }
    }
    pts[k] = acc;
    uint32_t zz = 0;
#pragma unroll
    for (int j = 0; j < N28; j++) zz |= acc.z.l[j];              // s = 0: the accumulator is still (0 : 1 : 0) as written
    prefix[k] = zz ? mul28(k ? prefix[k - 1] : one, acc.z) : (k ? prefix[k - 1] : one);
  }
  M28 inv = fp28_invert(prefix[cnt - 1]);
  for (uint32_t k = cnt; k-- > 0;) {
    uint32_t zz = 0;
#pragma unroll
    for (int j = 0; j < N28; j++) zz |= pts[k].z.l[j];
    g1_affine r;
    if (zz) {
      const M28 zinv = mul28(inv, k ? prefix[k - 1] : one);
      inv = mul28(inv, pts[k].z);
      r.x = fp_from_28(mul28(pts[k].x, zinv));
      r.y = fp_from_28(mul28(pts[k].y, zinv));
    } else {
      r.x = Fp::zero();
      r.y = Fp::zero();
    }
    out[base + k] = r;
  }
}

}  // namespace bp
