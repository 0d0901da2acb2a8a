// poly_kernels.hpp -- device kernels behind the Polynomial operators of src/polynomial.rs:
//   evaluation at a point (coeffs_evaluate :34-45), division by a binomial b0 + bm x^m (the only
//   divisors the prover uses: Z_H = x^n - 1 at prover.rs:450 and x - zeta / x - zeta*omega at :623-638),
//   and a general long division for everything else.  Element-wise ops live in ntt_kernels.hpp.
#pragma once
#include "fr_io.hpp"

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
// the same for up to 8 polynomials in ONE launch (blockIdx.y = polynomial): each alone is half a wave round of lanes and therefore
// latency bound (43 us: 16 Horner steps + x^(16 t) per lane); together they fill the chip
struct PolyEvalMany {
  const fr_t* c[8];
  size_t n[8];
  fr_t x[8];
  uint32_t K[8];
  uint32_t off[8];        // first partial slot of polynomial j
  uint32_t blocks[8];
};
__global__ void __launch_bounds__(256) poly_eval_partial_many(PolyEvalMany a, fr_t* __restrict__ partial) {
  const uint32_t j = blockIdx.y;
  if (blockIdx.x >= a.blocks[j]) return;
  const fr_t* __restrict__ c = a.c[j];
  const size_t n = a.n[j];
  const uint32_t K = a.K[j];
  const fr_t x = a.x[j];
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
  if (threadIdx.x == 0) store_fr(&partial[a.off[j] + blockIdx.x], s);
}
__global__ void __launch_bounds__(256) fr_sum_small_many(PolyEvalMany a, const fr_t* __restrict__ partial, fr_t* __restrict__ out) {
  const uint32_t j = blockIdx.x;
  fr_t acc = Fr::zero();
  for (uint32_t i = threadIdx.x; i < a.blocks[j]; i += blockDim.x) {
    fr_t v = load_fr(&partial[a.off[j] + i]);
    Fr::add(acc, acc, v);
  }
  fr_t s = block_sum_fr(acc);
  if (threadIdx.x == 0) store_fr(&out[j], s);
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
// Long chains (small m, e.g. the linear divisors x - zeta of prover.rs:623-638): one workgroup per chain.  Lane t owns
// the run of chunks [t*S, (t+1)*S): it first composes its run into an affine map carry_out = A * carry_in + B, the maps
// are scanned across the lanes, then every lane replays its run writing the per-chunk carries.  Sequential depth
// 2S + log2(lanes) compositions instead of the number of chunks.
__global__ void __launch_bounds__(1024) poly_div_binomial_carry_wg(size_t nq, size_t m, fr_t f, uint32_t K, size_t chunks_per_chain,
                                                                   const fr_t* __restrict__ chunk_head, fr_t* __restrict__ carry) {
  const size_t r = blockIdx.x;
  if (r >= m || r >= nq) return;
  const size_t len = (nq - r + m - 1) / m;
  const size_t S = (chunks_per_chain + blockDim.x - 1) / blockDim.x;
  const size_t ck0 = (size_t)threadIdx.x * S, ck1 = ck0 + S < chunks_per_chain ? ck0 + S : chunks_per_chain;
  const fr_t fK = fr_pow_u64(f, K);
  fr_t A = Fr::one(), B = Fr::zero();
  for (size_t ck = ck0; ck < ck1; ck++) {
    const size_t j_hi = len > ck * K ? len - ck * K : 0;
    if (j_hi == 0) break;
    const size_t j_lo = j_hi > K ? j_hi - K : 0;
    fr_t head = load_fr(&chunk_head[ck * m + r]), fp = (j_hi - j_lo == K) ? fK : fr_pow_u64(f, j_hi - j_lo);
    Fr::mul(A, A, fp);                       // carry_out = head + fp * (A_prev * c + B_prev)
    Fr::mul(B, B, fp);
    Fr::add(B, B, head);
  }
  // inclusive scan of the lanes' affine maps (Hillis-Steele in LDS): after it, B of lane t is the carry leaving lane t's run
  fr_t* bufA = reinterpret_cast<fr_t*>(poly_lds_raw);
  fr_t* bufB = bufA + blockDim.x;
  bufA[threadIdx.x] = A;
  bufB[threadIdx.x] = B;
  __syncthreads();
  for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
    const bool has = threadIdx.x >= d;
    fr_t a1, b1;
    if (has) { a1 = bufA[threadIdx.x - d]; b1 = bufB[threadIdx.x - d]; }
    __syncthreads();
    if (has) {                                  // (A, B) after (a1, b1):  x -> A (a1 x + b1) + B
      fr_t t;
      Fr::mul(t, A, b1);
      Fr::add(B, B, t);
      Fr::mul(A, A, a1);
      bufA[threadIdx.x] = A;
      bufB[threadIdx.x] = B;
    }
    __syncthreads();
  }
  fr_t c = threadIdx.x ? bufB[threadIdx.x - 1] : Fr::zero();          // carry into this lane's run
  for (size_t ck = ck0; ck < ck1; ck++) {
    store_fr(&carry[ck * m + r], c);
    const size_t j_hi = len > ck * K ? len - ck * K : 0;
    if (j_hi == 0) break;
    const size_t j_lo = j_hi > K ? j_hi - K : 0;
    fr_t head = load_fr(&chunk_head[ck * m + r]), fp = (j_hi - j_lo == K) ? fK : fr_pow_u64(f, j_hi - j_lo), t;
    Fr::mul(t, c, fp);
    Fr::add(c, head, t);
  }
}
// Very long chains (x - zeta against 2^20 coefficients: 32 768 chunks): the chain is cut into G segments, a workgroup each.
// PHASE 0 composes every segment into one affine map (seg_map[r * G + g] = A | B); poly_div_seg_scan turns the G maps into the
// carry entering each segment; PHASE 1 replays the segments from those carries and writes the per-chunk carries.  Sequential
// depth 2 S + log2(lanes) per phase with S = chunks / (G * lanes) instead of chunks / lanes: the single workgroup above took
// 289 us per division (two per proof) on one CU of 256.
template <int PHASE>
__global__ void __launch_bounds__(1024) poly_div_binomial_carry_seg(size_t nq, size_t m, fr_t f, uint32_t K, size_t chunks_per_chain, uint32_t G,
                                                                    const fr_t* __restrict__ chunk_head, fr_t* __restrict__ seg_map,
                                                                    const fr_t* __restrict__ seg_carry, fr_t* __restrict__ carry) {
  const size_t r = blockIdx.x, g = blockIdx.y;
  if (r >= m || r >= nq) return;
  const size_t len = (nq - r + m - 1) / m;
  const size_t seg = (chunks_per_chain + G - 1) / G, s0 = g * seg, s1 = s0 + seg < chunks_per_chain ? s0 + seg : chunks_per_chain;
  const size_t S = (seg + blockDim.x - 1) / blockDim.x;
  const size_t ck0 = s0 + (size_t)threadIdx.x * S < s1 ? s0 + (size_t)threadIdx.x * S : s1, ck1 = ck0 + S < s1 ? ck0 + S : s1;
  const fr_t fK = fr_pow_u64(f, K);
  fr_t A = Fr::one(), B = Fr::zero();
  for (size_t ck = ck0; ck < ck1; ck++) {
    const size_t j_hi = len > ck * K ? len - ck * K : 0;
    if (j_hi == 0) break;
    const size_t j_lo = j_hi > K ? j_hi - K : 0;
    fr_t head = load_fr(&chunk_head[ck * m + r]), fp = (j_hi - j_lo == K) ? fK : fr_pow_u64(f, j_hi - j_lo);
    Fr::mul(A, A, fp);
    Fr::mul(B, B, fp);
    Fr::add(B, B, head);
  }
  fr_t* bufA = reinterpret_cast<fr_t*>(poly_lds_raw);
  fr_t* bufB = bufA + blockDim.x;
  bufA[threadIdx.x] = A;
  bufB[threadIdx.x] = B;
  __syncthreads();
  for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
    const bool has = threadIdx.x >= d;
    fr_t a1, b1;
    if (has) { a1 = bufA[threadIdx.x - d]; b1 = bufB[threadIdx.x - d]; }
    __syncthreads();
    if (has) {
      fr_t t;
      Fr::mul(t, A, b1);
      Fr::add(B, B, t);
      Fr::mul(A, A, a1);
      bufA[threadIdx.x] = A;
      bufB[threadIdx.x] = B;
    }
    __syncthreads();
  }
  if (PHASE == 0) {
    if (threadIdx.x == blockDim.x - 1) {
      store_fr(&seg_map[2 * (r * G + g)], A);
      store_fr(&seg_map[2 * (r * G + g) + 1], B);
    }
    return;
  }
  const fr_t cin = load_fr(&seg_carry[r * G + g]);                      // carry entering this segment
  fr_t c = cin;
  if (threadIdx.x) {                                                    // through the lanes before this one: A x + B
    fr_t t;
    Fr::mul(t, bufA[threadIdx.x - 1], cin);
    Fr::add(c, t, bufB[threadIdx.x - 1]);
  }
  for (size_t ck = ck0; ck < ck1; ck++) {
    store_fr(&carry[ck * m + r], c);
    const size_t j_hi = len > ck * K ? len - ck * K : 0;
    if (j_hi == 0) break;
    const size_t j_lo = j_hi > K ? j_hi - K : 0;
    fr_t head = load_fr(&chunk_head[ck * m + r]), fp = (j_hi - j_lo == K) ? fK : fr_pow_u64(f, j_hi - j_lo), t;
    Fr::mul(t, c, fp);
    Fr::add(c, head, t);
  }
}
// carries entering the G <= 64 segments of chain blockIdx.x: inclusive scan of the segments' affine maps (one lane each,
// Hillis-Steele in LDS: six levels), applied to a zero carry; segment g receives the B of the scan up to g - 1
__global__ void __launch_bounds__(64) poly_div_seg_scan(uint32_t G, const fr_t* __restrict__ seg_map, fr_t* __restrict__ seg_carry) {
  __shared__ fr_t bufA[64], bufB[64];
  const size_t r = blockIdx.x;
  const uint32_t g = threadIdx.x;
  fr_t A = Fr::one(), B = Fr::zero();
  if (g < G) {
    A = load_fr(&seg_map[2 * (r * G + g)]);
    B = load_fr(&seg_map[2 * (r * G + g) + 1]);
  }
  bufA[g] = A;
  bufB[g] = B;
  __syncthreads();
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const bool has = g >= d;
    fr_t a1, b1;
    if (has) { a1 = bufA[g - d]; b1 = bufB[g - d]; }
    __syncthreads();
    if (has) {
      fr_t t;
      Fr::mul(t, A, b1);
      Fr::add(B, B, t);
      Fr::mul(A, A, a1);
      bufA[g] = A;
      bufB[g] = B;
    }
    __syncthreads();
  }
  if (g < G) store_fr(&seg_carry[r * G + g], g ? bufB[g - 1] : Fr::zero());
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

// ---- multiplicative scans (prefix / suffix products) and the round-2 grand product (prover.rs:279-319) -----------
// Tile = 256 lanes x SCANM_PER_LANE elements.  reverse != 0 scans from the last element backwards (suffix products).
constexpr uint32_t SCANM_PER_LANE = 8, SCANM_TILE = 256 * SCANM_PER_LANE;

// exclusive product scan across the 256 lanes of a workgroup; returns the product of the lanes before this one
__device__ __forceinline__ fr_t block_exclusive_scan_mul(const fr_t& v, fr_t& block_total) {
  fr_t* buf = reinterpret_cast<fr_t*>(poly_lds_raw);
  buf[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t d = 1; d < blockDim.x; d <<= 1) {            // Hillis-Steele, inclusive
    fr_t mine = buf[threadIdx.x], other = Fr::one();
    const bool has = threadIdx.x >= d;
    if (has) other = buf[threadIdx.x - d];
    __syncthreads();
    if (has) {
      Fr::mul(mine, mine, other);
      buf[threadIdx.x] = mine;
    }
    __syncthreads();
  }
  block_total = buf[blockDim.x - 1];
  fr_t excl = threadIdx.x ? buf[threadIdx.x - 1] : Fr::one();
  __syncthreads();
  return excl;
}
__device__ __forceinline__ size_t scan_index(size_t pos, size_t n, int reverse) { return reverse ? n - 1 - pos : pos; }

__global__ void __launch_bounds__(256) fr_scan_tile_products(const fr_t* __restrict__ in, size_t n, int reverse, fr_t* __restrict__ tile_prod) {
  const size_t base = (size_t)blockIdx.x * SCANM_TILE + (size_t)threadIdx.x * SCANM_PER_LANE;
  fr_t p = Fr::one();
  for (uint32_t j = 0; j < SCANM_PER_LANE; j++)
    if (base + j < n) {
      fr_t v = load_fr(&in[scan_index(base + j, n, reverse)]);
      Fr::mul(p, p, v);
    }
  fr_t tot;
  (void)block_exclusive_scan_mul(p, tot);
  if (threadIdx.x == 0) store_fr(&tile_prod[blockIdx.x], tot);
}
// in-place exclusive scan of the tile products by one workgroup (n_tiles <= 256 * 64); total -> *total_out
__global__ void __launch_bounds__(256) fr_scan_tiles(fr_t* __restrict__ tile_prod, uint32_t n_tiles, fr_t* __restrict__ total_out) {
  const uint32_t per = (n_tiles + 255) / 256, lo = threadIdx.x * per;
  fr_t p = Fr::one();
  for (uint32_t j = 0; j < per; j++)
    if (lo + j < n_tiles) {
      fr_t v = load_fr(&tile_prod[lo + j]);
      Fr::mul(p, p, v);
    }
  fr_t tot, run = block_exclusive_scan_mul(p, tot);
  for (uint32_t j = 0; j < per; j++)
    if (lo + j < n_tiles) {
      fr_t v = load_fr(&tile_prod[lo + j]);
      store_fr(&tile_prod[lo + j], run);
      Fr::mul(run, run, v);
    }
  if (threadIdx.x == 0) store_fr(total_out, tot);
}
// out[idx] = product of the elements before (exclusive) or up to (inclusive) idx in scan order
__global__ void __launch_bounds__(256) fr_scan_apply(const fr_t* __restrict__ in, size_t n, int reverse, int inclusive,
                                                      const fr_t* __restrict__ tile_prefix, fr_t* __restrict__ out) {
  const size_t base = (size_t)blockIdx.x * SCANM_TILE + (size_t)threadIdx.x * SCANM_PER_LANE;
  fr_t v[SCANM_PER_LANE], p = Fr::one();
  for (uint32_t j = 0; j < SCANM_PER_LANE; j++) {
    v[j] = base + j < n ? load_fr(&in[scan_index(base + j, n, reverse)]) : Fr::one();
    Fr::mul(p, p, v[j]);
  }
  fr_t tot, run = block_exclusive_scan_mul(p, tot), tp = load_fr(&tile_prefix[blockIdx.x]);
  Fr::mul(run, run, tp);
  for (uint32_t j = 0; j < SCANM_PER_LANE; j++) {
    if (base + j >= n) break;
    fr_t incl;
    Fr::mul(incl, run, v[j]);
    store_fr(&out[scan_index(base + j, n, reverse)], inclusive ? incl : run);
    run = incl;
  }
}
// numerators and denominators of the permutation argument (prover.rs:286-317; Rlc = self + other*beta + gamma, utils.rs:161-169)
__global__ void __launch_bounds__(256) grand_product_terms(const fr_t* __restrict__ a, const fr_t* __restrict__ b, const fr_t* __restrict__ c,
                                                            const fr_t* __restrict__ s1, const fr_t* __restrict__ s2, const fr_t* __restrict__ s3,
                                                            const fr_t* __restrict__ roots, size_t n, fr_t beta, fr_t gamma, fr_t k1, fr_t k2,
                                                            fr_t* __restrict__ num, fr_t* __restrict__ den) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t w = load_fr(&roots[i]), wb, t, u, nu, de;
  fr_t ai = load_fr(&a[i]), bi = load_fr(&b[i]), ci = load_fr(&c[i]);
  Fr::mul(wb, w, beta);                 // beta * w^i
  Fr::add(t, ai, wb); Fr::add(nu, t, gamma);
  Fr::mul(u, wb, k1); Fr::add(t, bi, u); Fr::add(t, t, gamma); Fr::mul(nu, nu, t);
  Fr::mul(u, wb, k2); Fr::add(t, ci, u); Fr::add(t, t, gamma); Fr::mul(nu, nu, t);
  fr_t si = load_fr(&s1[i]);
  Fr::mul(u, si, beta); Fr::add(t, ai, u); Fr::add(de, t, gamma);
  si = load_fr(&s2[i]);
  Fr::mul(u, si, beta); Fr::add(t, bi, u); Fr::add(t, t, gamma); Fr::mul(de, de, t);
  si = load_fr(&s3[i]);
  Fr::mul(u, si, beta); Fr::add(t, ci, u); Fr::add(t, t, gamma); Fr::mul(de, de, t);
  store_fr(&num[i], nu);
  store_fr(&den[i], de);
}
// z_i = (prod_{j<i} num_j) * (prod_{j>=i} den_j) * (prod_all den)^-1
__global__ void __launch_bounds__(256) grand_product_combine(const fr_t* __restrict__ pn, const fr_t* __restrict__ sd, fr_t td_inv, size_t n,
                                                              fr_t* __restrict__ z) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t x = load_fr(&pn[i]), y = load_fr(&sd[i]);
  Fr::mul(x, x, y);
  Fr::mul(x, x, td_inv);
  store_fr(&z[i], x);
}

// ---- helpers of the device-resident polynomial pipeline -------------------------------------------------------
// out[0] = 1 + index of the last non-zero element (0 when all are zero); out[1] = number of non-zero elements in [lo, hi)
// grid-stride; one atomic pair per workgroup (both results land on the same two words, so per-wave atomics of a
// 2^22-element launch serialise into ~0.5 ms)
__global__ void __launch_bounds__(256) fr_nonzero_stats(const fr_t* __restrict__ a, size_t n, size_t lo, size_t hi, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long s_last[4], s_cnt[4];
  unsigned long long last = 0, cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    fr_t v = load_fr(&a[i]);
    if (!big_is_zero(v)) {
      last = i + 1;                                              // indices grow along the loop
      cnt += (i >= lo && i < hi) ? 1 : 0;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    unsigned long long l2 = __shfl_xor(last, d, 64), c2 = __shfl_xor(cnt, d, 64);
    last = l2 > last ? l2 : last;
    cnt += c2;
  }
  if ((threadIdx.x & 63) == 0) { s_last[threadIdx.x >> 6] = last; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { last = s_last[w] > last ? s_last[w] : last; cnt += s_cnt[w]; }
    if (last) atomicMax(&out[0], last);
    if (cnt) atomicAdd(&out[1], cnt);
  }
}
// ---- stable compaction of the non-zero elements (the reference's Div squeezes zero quotient coefficients out,
// polynomial.rs:371-376): count per tile, scan the tile counts, scatter.  Tile = 256 lanes x COMPACT_PER_LANE elements.
constexpr uint32_t COMPACT_PER_LANE = 8, COMPACT_TILE = 256 * COMPACT_PER_LANE;
__device__ __forceinline__ uint32_t compact_block_scan(uint32_t v, uint32_t* lds4, uint32_t& total) {       // exclusive scan over 256 lanes
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += t;
  }
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t x = lds4[w];
    if ((uint32_t)w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  total = tot;
  return base + incl - v;
}
__global__ void __launch_bounds__(256) fr_compact_count(const fr_t* __restrict__ a, size_t n, uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t lds4[4];
  const size_t base = (size_t)blockIdx.x * COMPACT_TILE + (size_t)threadIdx.x * COMPACT_PER_LANE;
  uint32_t c = 0;
  for (uint32_t j = 0; j < COMPACT_PER_LANE; j++)
    if (base + j < n && !big_is_zero(load_fr(&a[base + j]))) c++;
  uint32_t tot;
  (void)compact_block_scan(c, lds4, tot);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = tot;
}
// exclusive scan of the tile counts in place, one workgroup (any number of tiles); total -> *total_out
__global__ void __launch_bounds__(256) fr_compact_scan(uint32_t* __restrict__ tile_count, uint32_t n_tiles, unsigned long long* __restrict__ total_out) {
  __shared__ uint32_t lds4[4];
  uint32_t running = 0;
  for (uint32_t b = 0; b < n_tiles; b += 256) {
    const uint32_t i = b + threadIdx.x, v = i < n_tiles ? tile_count[i] : 0;
    uint32_t tot, ex = compact_block_scan(v, lds4, tot);
    if (i < n_tiles) tile_count[i] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) *total_out = running;
}
__global__ void __launch_bounds__(256) fr_compact_scatter(const fr_t* __restrict__ a, size_t n, const uint32_t* __restrict__ tile_offset,
                                                           fr_t* __restrict__ out) {
  __shared__ uint32_t lds4[4];
  const size_t base = (size_t)blockIdx.x * COMPACT_TILE + (size_t)threadIdx.x * COMPACT_PER_LANE;
  fr_t v[COMPACT_PER_LANE];
  uint32_t c = 0, keep = 0;
#pragma unroll
  for (uint32_t j = 0; j < COMPACT_PER_LANE; j++) {
    if (base + j < n) {
      v[j] = load_fr(&a[base + j]);
      if (!big_is_zero(v[j])) { keep |= 1u << j; c++; }
    }
  }
  uint32_t tot, pos = tile_offset[blockIdx.x] + compact_block_scan(c, lds4, tot);
#pragma unroll
  for (uint32_t j = 0; j < COMPACT_PER_LANE; j++)
    if (keep & (1u << j)) store_fr(&out[pos++], v[j]);
}

// out[i] = a[i] * w^i   (p(x) -> p(w x); prover.rs:661-674 monomial_z_to_z_omega)
__global__ void __launch_bounds__(256) fr_scale_powers(const fr_t* __restrict__ a, size_t n, fr_t w, fr_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t v = load_fr(&a[i]), p = fr_pow_u64(w, i);
  Fr::mul(v, v, p);
  store_fr(&out[i], v);
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
