// bigint.hpp -- fixed-width multi-limb arithmetic on 32-bit limbs for gfx950.
//
// Representation: little-endian 32-bit limbs.  A field element is the SAME bit pattern as the
// reference's 64-bit-limb Montgomery value (lib/bls12_381/src/scalar.rs:16-22, fp.rs:11-15)
// reinterpreted as twice as many 32-bit limbs, so device buffers, the C ABI and the test checker all
// share one encoding and no conversion pass exists anywhere.
//
// The Montgomery product is a product-scanning (column-wise) loop: each 32x32 partial product is one
// v_mad_u64_u32 (64-bit accumulate, carry to VCC) plus one v_addc_co_u32 (carry into the third
// accumulator word).  tools/ubench_int.hip measures the issue rate of exactly that pair.
//
// Everything here is __host__ __device__: the host build (plain C++) is what the C-ABI's O(1)
// epilogues and the CPU-side unit tests of this header use; the device build swaps in inline asm
// for the multiply-accumulate step only.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define BP_HD __host__ __device__ __forceinline__

namespace bp {

// acc (96 bit: lo64, hi32) += x0*y0 (+ x1*y1 ...).  One asm statement per group: hipcc pads an s_nop
// after every asm statement whose result the next instruction consumes, so products are grouped to
// amortise that pad (cdna_hip_programming.md section 5.7 item 2).
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_MAC_ASM(X, Y) "v_mad_u64_u32 %0, vcc, " X ", " Y ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
BP_HD void mac96(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0) {
  asm(BP_MAC_ASM("%2", "%3") : "+v"(lo), "+v"(hi) : "v"(x0), "v"(y0) : "vcc");
}
BP_HD void mac96x2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  asm(BP_MAC_ASM("%2", "%3") BP_MAC_ASM("%4", "%5")
      : "+v"(lo), "+v"(hi) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
BP_HD void mac96x4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  asm(BP_MAC_ASM("%2", "%3") BP_MAC_ASM("%4", "%5") BP_MAC_ASM("%6", "%7") BP_MAC_ASM("%8", "%9")
      : "+v"(lo), "+v"(hi)
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
// same with the second factor in SGPRs (field-modulus limbs: wave-uniform constants)
BP_HD void mac96s(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0) {
  asm(BP_MAC_ASM("%2", "%3") : "+v"(lo), "+v"(hi) : "v"(x0), "s"(y0) : "vcc");
}
BP_HD void mac96sx2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  asm(BP_MAC_ASM("%2", "%3") BP_MAC_ASM("%4", "%5")
      : "+v"(lo), "+v"(hi) : "v"(x0), "s"(y0), "v"(x1), "s"(y1) : "vcc");
}
BP_HD void mac96sx4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                    uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  asm(BP_MAC_ASM("%2", "%3") BP_MAC_ASM("%4", "%5") BP_MAC_ASM("%6", "%7") BP_MAC_ASM("%8", "%9")
      : "+v"(lo), "+v"(hi)
      : "v"(x0), "s"(y0), "v"(x1), "s"(y1), "v"(x2), "s"(y2), "v"(x3), "s"(y3) : "vcc");
}
#else
BP_HD void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
  uint64_t p = (uint64_t)a * b;
  lo += p;
  hi += (lo < p);
}
BP_HD void mac96x2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  mac96(lo, hi, x0, y0);
  mac96(lo, hi, x1, y1);
}
BP_HD void mac96x4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  mac96x2(lo, hi, x0, y0, x1, y1);
  mac96x2(lo, hi, x2, y2, x3, y3);
}
BP_HD void mac96s(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) { mac96(lo, hi, a, b); }
BP_HD void mac96sx2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
  mac96x2(lo, hi, x0, y0, x1, y1);
}
BP_HD void mac96sx4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                    uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  mac96x4(lo, hi, x0, y0, x1, y1, x2, y2, x3, y3);
}
#endif
// acc >>= 32
BP_HD void shift96(uint64_t& lo, uint32_t& hi) {
  lo = (lo >> 32) | ((uint64_t)hi << 32);
  hi = 0;
}

// acc += sum_{i=I0}^{I1} x[i] * y[K-i]   (compile-time bounds; groups of 4, 2, 1)
template <int K, int I0, int I1>
BP_HD void mac_col(uint64_t& lo, uint32_t& hi, const uint32_t* x, const uint32_t* y) {
  if constexpr (I1 - I0 + 1 >= 4) {
    mac96x4(lo, hi, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1], x[I0 + 2], y[K - I0 - 2], x[I0 + 3], y[K - I0 - 3]);
    mac_col<K, I0 + 4, I1>(lo, hi, x, y);
  } else if constexpr (I1 - I0 + 1 >= 2) {
    mac96x2(lo, hi, x[I0], y[K - I0], x[I0 + 1], y[K - I0 - 1]);
    mac_col<K, I0 + 2, I1>(lo, hi, x, y);
  } else if constexpr (I1 - I0 + 1 == 1) {
    mac96(lo, hi, x[I0], y[K - I0]);
  }
}
// acc += sum_{i=I0}^{I1} m[i] * mod[K-i]
template <class P, int K, int I0, int I1>
BP_HD void mac_col_mod(uint64_t& lo, uint32_t& hi, const uint32_t* m) {
  if constexpr (I1 - I0 + 1 >= 4) {
    mac96sx4(lo, hi, m[I0], P::mod(K - I0), m[I0 + 1], P::mod(K - I0 - 1), m[I0 + 2], P::mod(K - I0 - 2), m[I0 + 3],
             P::mod(K - I0 - 3));
    mac_col_mod<P, K, I0 + 4, I1>(lo, hi, m);
  } else if constexpr (I1 - I0 + 1 >= 2) {
    mac96sx2(lo, hi, m[I0], P::mod(K - I0), m[I0 + 1], P::mod(K - I0 - 1));
    mac_col_mod<P, K, I0 + 2, I1>(lo, hi, m);
  } else if constexpr (I1 - I0 + 1 == 1) {
    mac96s(lo, hi, m[I0], P::mod(K - I0));
  }
}

template <int N>
struct Big {
  uint32_t l[N];
};

template <int N>
BP_HD bool big_is_zero(const Big<N>& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) acc |= a.l[i];
  return acc == 0;
}
template <int N>
BP_HD bool big_eq(const Big<N>& a, const Big<N>& b) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; i++) acc |= a.l[i] ^ b.l[i];
  return acc == 0;
}
// r = a + b, returns carry-out
template <int N>
BP_HD uint32_t big_add(Big<N>& r, const Big<N>& a, const Big<N>& b) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
  return c;
}
// r = a - b, returns borrow-out (1 when a < b)
template <int N>
BP_HD uint32_t big_sub(Big<N>& r, const Big<N>& a, const Big<N>& b) {
  unsigned borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &borrow);
  return borrow;
}
template <int N>
BP_HD void big_select(Big<N>& r, bool take_a, const Big<N>& a, const Big<N>& b) {
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = take_a ? a.l[i] : b.l[i];
}

// Field parameter pack: P::N limbs, P::mod(i), P::INV32 = -mod^{-1} mod 2^32.
// Constants are constexpr functions so they fold to literals in both host and device code.

template <class P>
struct Mont {
  static constexpr int N = P::N;
  using V = Big<N>;

  static BP_HD V modulus() {
    V m;
#pragma unroll
    for (int i = 0; i < N; i++) m.l[i] = P::mod(i);
    return m;
  }
  // reduce a value known to be < 2*mod
  static BP_HD void reduce_once(V& r, const V& a, uint32_t extra_carry = 0) {
    V s;
    uint32_t borrow = big_sub(s, a, modulus());
    big_select(r, (extra_carry != 0) || !borrow, s, a);
  }
  static BP_HD void add(V& r, const V& a, const V& b) {
    V t;
    uint32_t c = big_add(t, a, b);
    reduce_once(r, t, c);
  }
  static BP_HD void sub(V& r, const V& a, const V& b) {
    V t, u;
    uint32_t borrow = big_sub(t, a, b);
    big_add(u, t, modulus());
    big_select(r, borrow != 0, u, t);
  }
  static BP_HD void neg(V& r, const V& a) {
    V t;
    big_sub(t, modulus(), a);
    V z;
#pragma unroll
    for (int i = 0; i < N; i++) z.l[i] = 0;
    big_select(r, big_is_zero(a), z, t);
  }
  static BP_HD void dbl(V& r, const V& a) { add(r, a, a); }

  // Montgomery product a*b*R^-1 mod p, R = 2^(32N); inputs < p, output < p (unique representative,
  // as scalar.rs:514-586 / fp.rs:487-609 produce).
  template <int K>
  static BP_HD void mul_cols(uint64_t& lo, uint32_t& hi, uint32_t* m, V& t, const V& a, const V& b) {
    if constexpr (K < N) {
      mac_col<K, 0, K>(lo, hi, a.l, b.l);
      mac_col_mod<P, K, 0, K - 1>(lo, hi, m);
      m[K] = (uint32_t)lo * P::INV32;
      mac96s(lo, hi, m[K], P::mod(0));
      shift96(lo, hi);
      mul_cols<K + 1>(lo, hi, m, t, a, b);
    } else if constexpr (K < 2 * N) {
      mac_col<K, K - N + 1, N - 1>(lo, hi, a.l, b.l);
      mac_col_mod<P, K, K - N + 1, N - 1>(lo, hi, m);
      t.l[K - N] = (uint32_t)lo;
      shift96(lo, hi);
      mul_cols<K + 1>(lo, hi, m, t, a, b);
    }
  }
#if !defined(__HIP_DEVICE_COMPILE__)
  // Host path: the same Montgomery product on 64-bit limbs (a Big<N> is bit-identical to N/2 little-endian
  // u64 limbs on the little-endian hosts this runs on).  Used by the O(W) host epilogues only.
  static constexpr uint64_t inv64() {
    uint64_t p0 = (uint64_t)P::mod(0) | ((uint64_t)P::mod(1) << 32), y = p0;   // y = p0^-1 mod 2^k, Newton
    for (int i = 0; i < 6; i++) y *= 2 - p0 * y;
    return ~y + 1;
  }
  static void mul_host(V& r, const V& a, const V& b) {
    constexpr int M = N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], Bv[M], Pm[M], T[M + 2];
    for (int i = 0; i < M; i++) {
      A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
      Bv[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
      Pm[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    }
    for (int i = 0; i < M + 2; i++) T[i] = 0;
    for (int i = 0; i < M; i++) {
      uint64_t c = 0;
      for (int j = 0; j < M; j++) {
        u128 x = (u128)A[j] * Bv[i] + T[j] + c;
        T[j] = (uint64_t)x;
        c = (uint64_t)(x >> 64);
      }
      u128 x = (u128)T[M] + c;
      T[M] = (uint64_t)x;
      T[M + 1] = (uint64_t)(x >> 64);
      uint64_t mq = T[0] * inv64();
      x = (u128)mq * Pm[0] + T[0];
      c = (uint64_t)(x >> 64);
      for (int j = 1; j < M; j++) {
        x = (u128)mq * Pm[j] + T[j] + c;
        T[j - 1] = (uint64_t)x;
        c = (uint64_t)(x >> 64);
      }
      x = (u128)T[M] + c;
      T[M - 1] = (uint64_t)x;
      T[M] = T[M + 1] + (uint64_t)(x >> 64);
    }
    V t;
    for (int i = 0; i < M; i++) {
      t.l[2 * i] = (uint32_t)T[i];
      t.l[2 * i + 1] = (uint32_t)(T[i] >> 32);
    }
    reduce_once(r, t, (uint32_t)T[M]);
  }
#endif
  static BP_HD void mul(V& r, const V& a, const V& b) {
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(BP_HOST_USE_DEVICE_ALGO)
    mul_host(r, a, b);     // tests/hostcheck defines BP_HOST_USE_DEVICE_ALGO to run the column code below on the CPU
    return;
#endif
    uint32_t m[N];
    V t;
    uint64_t lo = 0;
    uint32_t hi = 0;
    mul_cols<0>(lo, hi, m, t, a, b);
    reduce_once(r, t, (uint32_t)lo);
  }
  static BP_HD void sqr(V& r, const V& a) { mul(r, a, a); }

  // a * R^-1 (Montgomery -> canonical integer), scalar.rs:292-304 / fp.rs:212-227
  static BP_HD void from_mont(V& r, const V& a) {
    V one;
#pragma unroll
    for (int i = 0; i < N; i++) one.l[i] = (i == 0);
    mul(r, a, one);
  }
  static BP_HD V r2() {
    V m;
#pragma unroll
    for (int i = 0; i < N; i++) m.l[i] = P::r2(i);
    return m;
  }
  static BP_HD V one() {
    V m;
#pragma unroll
    for (int i = 0; i < N; i++) m.l[i] = P::one(i);
    return m;
  }
  static BP_HD V zero() {
    V m;
#pragma unroll
    for (int i = 0; i < N; i++) m.l[i] = 0;
    return m;
  }
  // canonical integer -> Montgomery
  static BP_HD void to_mont(V& r, const V& a) { mul(r, a, r2()); }

  // r = a^e, e given as nlimbs 32-bit limbs (little endian), MSB-first square-and-multiply
  static BP_HD void pow(V& r, const V& a, const uint32_t* e, int nlimbs) {
    V res = one();
    for (int i = nlimbs - 1; i >= 0; i--) {
      for (int j = 31; j >= 0; j--) {
        sqr(res, res);
        if ((e[i] >> j) & 1) mul(res, res, a);
      }
    }
    r = res;
  }
};

}  // namespace bp
