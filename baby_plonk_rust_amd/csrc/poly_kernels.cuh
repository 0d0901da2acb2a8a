// poly_kernels.cuh -- device kernels behind the Polynomial operators of src/polynomial.rs:
//   evaluation at a point (coeffs_evaluate :34-45), division by a binomial b0 + bm x^m (the only
//   divisors the prover uses: Z_H = x^n - 1 at prover.rs:450 and x - zeta / x - zeta*omega at :623-638),
//   and a general long division for everything else.  Element-wise ops live in ntt_kernels.cuh.
#pragma once
#include "fr_io.cuh"

namespace bp {

extern __shared__ uint4 poly_lds_raw[];

__device__ __forceinline__ fr_t fr_pow_u64(fr_t base, uint64_t e) {
  fr_t acc = Fr::one();
  while (e) {
    if (e & 1) Fr::mul(acc, acc, base);
    Fr::sqr(base, base);
    e >>= 1;
  }
  return acc;
}
// block-wide sum of one fr_t per lane (256 lanes) through LDS; result valid on lane 0
__device__ __forceinline__ fr_t block_sum_fr(fr_t v) {
  fr_t* buf = reinterpret_cast<fr_t*>(poly_lds_raw);
  buf[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t stride = blockDim.x >> 1; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) {
      fr_t a = buf[threadIdx.x], b = buf[threadIdx.x + stride];
      Fr::add(a, a, b);
      buf[threadIdx.x] = a;
    }
    __syncthreads();
  }
  return buf[0];
}
// partial[blockIdx.x] = sum over this block's lanes of (sum_{j<K} c[t*K+j] x^j) * x^(t*K)
__global__ void __launch_bounds__(256) poly_eval_partial(const fr_t* __restrict__ c, size_t n, fr_t x, uint32_t K,
                                                          fr_t* __restrict__ partial) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * K, hi = lo + K < n ? lo + K : n;
  fr_t acc = Fr::zero();
  if (lo < n) {
    for (size_t i = hi; i-- > lo;) {
      Fr::mul(acc, acc, x);
      fr_t ci = load_fr(&c[i]);
      Fr::add(acc, acc, ci);
    }
    fr_t xp = fr_pow_u64(x, lo);
    Fr::mul(acc, acc, xp);
  }
  fr_t s = block_sum_fr(acc);
  if (threadIdx.x == 0) store_fr(&partial[blockIdx.x], s);
}
__global__ void __launch_bounds__(256) fr_sum_small(const fr_t* __restrict__ in, uint32_t n, fr_t* __restrict__ out) {
  fr_t acc = Fr::zero();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    fr_t v = load_fr(&in[i]);
    Fr::add(acc, acc, v);
  }
  fr_t s = block_sum_fr(acc);
  if (threadIdx.x == 0) store_fr(out, s);
}

// Division of a (na coefficients, leading one non-zero) by b0 + bm x^m, quotient length nq = na - m:
//   q_i = (a_{i+m} - b0 * q_{i+m}) / bm           (q_j = 0 for j >= nq)
// i.e. m independent chains i = r (mod m), each the recurrence q <- a' + f * q with f = -b0/bm.
// Chain r has len_r = ceil((nq - r) / m) elements.  Lanes own (chain, chunk) pairs of K consecutive chain
// elements: phase 1 evaluates each chunk with carry-in 0 and records it; phase 2 propagates carries chunk
// to chunk (sequential in the chunk index, parallel over chains); phase 3 adds carry * f^(distance).
__global__ void __launch_bounds__(256) poly_div_binomial_local(const fr_t* __restrict__ a, size_t nq, size_t m, fr_t f, fr_t bm_inv,
                                                                uint32_t K, size_t chunks_per_chain, fr_t* __restrict__ q,
                                                                fr_t* __restrict__ chunk_head) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * chunks_per_chain) return;
  const size_t r = t % m, ck = t / m;              // adjacent lanes = adjacent chains -> coalesced
  if (r >= nq) return;
  const size_t len = (nq - r + m - 1) / m;         // elements j = 0..len-1 at index r + j*m; recurrence runs from j = len-1 down
  const size_t j_hi = len > ck * K ? len - ck * K : 0;            // this chunk covers j in [j_lo, j_hi)
  const size_t j_lo = j_hi > K ? j_hi - K : 0;
  fr_t acc = Fr::zero();
  for (size_t j = j_hi; j-- > j_lo;) {
    const size_t i = r + j * m;
    fr_t ai = load_fr(&a[i + m]), v;
    Fr::mul(ai, ai, bm_inv);
    Fr::mul(v, acc, f);
    Fr::add(acc, v, ai);
    store_fr(&q[i], acc);
  }
  if (j_hi > j_lo) store_fr(&chunk_head[ck * m + r], acc);       // value at the chunk's lowest j with zero carry-in
}
// carry[ck][r] = true q just above chunk ck (i.e. q at j_hi of that chunk), sequential over ck
__global__ void __launch_bounds__(256) poly_div_binomial_carry(size_t nq, size_t m, fr_t f, uint32_t K, size_t chunks_per_chain,
                                                                const fr_t* __restrict__ chunk_head, fr_t* __restrict__ carry) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m || r >= nq) return;
  const size_t len = (nq - r + m - 1) / m;
  fr_t c = Fr::zero();
  fr_t fK = fr_pow_u64(f, K);
  for (size_t ck = 0; ck < chunks_per_chain; ck++) {
    store_fr(&carry[ck * m + r], c);
    const size_t j_hi = len > ck * K ? len - ck * K : 0;
    if (j_hi == 0) break;
    const size_t j_lo = j_hi > K ? j_hi - K : 0;
    // true head of this chunk = local head + carry * f^(chunk length)
    fr_t head = load_fr(&chunk_head[ck * m + r]), fp = (j_hi - j_lo == K) ? fK : fr_pow_u64(f, j_hi - j_lo), t;
    Fr::mul(t, c, fp);
    Fr::add(c, head, t);
  }
}
__global__ void __launch_bounds__(256) poly_div_binomial_apply(size_t nq, size_t m, fr_t f, uint32_t K, size_t chunks_per_chain,
                                                                const fr_t* __restrict__ carry, fr_t* __restrict__ q) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * chunks_per_chain) return;
  const size_t r = t % m, ck = t / m;
  if (r >= nq || ck == 0) return;                  // chunk 0 has zero carry-in
  const size_t len = (nq - r + m - 1) / m;
  const size_t j_hi = len > ck * K ? len - ck * K : 0;
  const size_t j_lo = j_hi > K ? j_hi - K : 0;
  fr_t c = load_fr(&carry[ck * m + r]);
  for (size_t j = j_hi; j-- > j_lo;) {
    Fr::mul(c, c, f);                              // carry * f^(j_hi - j)
    const size_t i = r + j * m;
    fr_t v = load_fr(&q[i]);
    Fr::add(v, v, c);
    store_fr(&q[i], v);
  }
}

// General long division, one workgroup: rem (na values, modified in place) / b (nb values, lead != 0).
// q has na - nb + 1 slots.  Sequential over quotient coefficients, parallel over the divisor.
__global__ void __launch_bounds__(1024) poly_div_general(fr_t* __restrict__ rem, size_t na, const fr_t* __restrict__ b, size_t nb,
                                                          fr_t lead_inv, fr_t* __restrict__ q) {
  __shared__ fr_t coeff_s;
  for (size_t top = na; top >= nb; top--) {
    if (threadIdx.x == 0) {
      fr_t lead = load_fr(&rem[top - 1]), c;
      Fr::mul(c, lead, lead_inv);
      coeff_s = c;
      store_fr(&q[top - nb], c);
    }
    __syncthreads();
    fr_t c = coeff_s;
    for (size_t i = threadIdx.x; i < nb; i += blockDim.x) {
      fr_t bi = load_fr(&b[i]), r = load_fr(&rem[top - nb + i]), t;
      Fr::mul(t, bi, c);
      Fr::sub(r, r, t);
      store_fr(&rem[top - nb + i], r);
    }
    __syncthreads();
  }
}

}  // namespace bp
