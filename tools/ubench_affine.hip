// Batched-affine bucket accumulation, measured: what ONE round of pairwise affine additions costs on gfx950 against the
// mixed projective addition msm_accumulate runs today (tools/ubench_g1add.hip: 7.3e9 additions/s on registers, 6.6-7.0e9 in the pipeline).
//
// The round measured is the FIRST one of a 2^20-point MSM with 13 windows of 20 bits -- the most favourable one (largest batches, 6.8 M of
// the 13.1 M additions): every lane takes K pairs of table entries,
//   forward : d_j = x2 - x1 gathered from the fixed-base tables (128-byte slots), prefix products pre_j = d_0 ... d_j  -> scratch (HBM)
//   invert  : one field inversion per lane (Fermat here; every form is also timed with the inversion skipped: the rate a FREE inversion would give)
//   backward: 1 / d_j = inv * pre_{j-1}, inv *= d_j; lambda = (y2 - y1) / d_j; x3 = lambda^2 - x1 - x2; y3 = lambda (x1 - x3) - y1
// = 6 Montgomery products per addition (the complete mixed addition: 11 products + 8 reductions = 9.5 product-equivalents), at the
// price of gathering every operand twice and of 2 x 64 B of prefix traffic per pair.
// Also: the same loops on registers only (no memory), and a one-lane-per-pair form whose prefix products run as a tree over the 256
// lanes of a workgroup in LDS (operands stay in registers, every byte moves once).
// Build: hipcc --offload-arch=gfx950 -O3 -I baby_plonk_rust_amd/csrc -I include tools/ubench_affine.hip -o tools/ubench_affine
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "g1_28.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace bp;

using Lz = F28<MASK28 + 8, 6>;          // a lazy coordinate as it travels between rounds
struct alignas(64) pre_slot { uint4 q[4]; };

__device__ __forceinline__ F28n one28() {
  F28n r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = One28::limb(i);
  return r;
}
__device__ __forceinline__ F28n load_x28(const g1_affine28* __restrict__ p) {      // x only: 56 B of the slot's first 64
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v0 = q[0], v1 = q[1], v2 = q[2];
  uint2 v3 = *reinterpret_cast<const uint2*>(q + 3);
  F28n r;
  r.l[0] = v0.x; r.l[1] = v0.y; r.l[2] = v0.z; r.l[3] = v0.w; r.l[4] = v1.x; r.l[5] = v1.y; r.l[6] = v1.z; r.l[7] = v1.w;
  r.l[8] = v2.x; r.l[9] = v2.y; r.l[10] = v2.z; r.l[11] = v2.w; r.l[12] = v3.x; r.l[13] = v3.y;
  return r;
}
__device__ __forceinline__ g1_affine28 load_pt28(const g1_affine28* __restrict__ p) {
  g1_affine28 r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v[7];
#pragma unroll
  for (int j = 0; j < 7; j++) v[j] = q[j];
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < 7; j++) { w[4 * j] = v[j].x; w[4 * j + 1] = v[j].y; w[4 * j + 2] = v[j].z; w[4 * j + 3] = v[j].w; }
#pragma unroll
  for (int j = 0; j < N28; j++) { r.x.l[j] = w[j]; r.y.l[j] = w[N28 + j]; }
  return r;
}
__device__ __forceinline__ void store_pre(pre_slot* __restrict__ dst, const F28n& a) {
  dst->q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
  dst->q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
  dst->q[2] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
  dst->q[3] = make_uint4(a.l[12], a.l[13], 0, 0);
}
__device__ __forceinline__ F28n load_pre(const pre_slot* __restrict__ src) {
  uint4 a = src->q[0], b = src->q[1], c = src->q[2], d = src->q[3];
  F28n r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w; r.l[12] = d.x; r.l[13] = d.y;
  return r;
}
__device__ __forceinline__ void store_out(g1_affine28* __restrict__ dst, const F28n& x, const F28n& y) {
  uint32_t w[28];
#pragma unroll
  for (int j = 0; j < N28; j++) { w[j] = x.l[j]; w[N28 + j] = y.l[j]; }
  uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int j = 0; j < 7; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
// full sequential carry: limbs <= 2^28 - 1 (value unchanged, < 2^392)
template <uint64_t A, uint32_t VA>
__device__ __forceinline__ F28<MASK28, VA> carry28(const F28<A, VA>& a) {
  F28<MASK28, VA> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < N28 - 1; i++) { uint32_t v = a.l[i] + c; r.l[i] = v & MASK28; c = v >> 28; }
  r.l[N28 - 1] = a.l[N28 - 1] + c;
  return r;
}
template <class T>
__device__ __forceinline__ F28n as_n(const T& a) {       // stored form of the benchmark: carried limbs, value left lazy (< 8p fits 14 x 28 bits)
  auto c = carry28(a);
  F28n r;
#pragma unroll
  for (int i = 0; i < N28; i++) r.l[i] = c.l[i];
  return r;
}
// a^(p-2) in the Montgomery domain: 380 squarings + ~190 products
__device__ __noinline__ F28n fermat_inv28(const F28n& a) {
  F28n r = one28();
  for (int w = 11; w >= 0; w--) {
    const uint32_t e = FpParams::mod_minus_2(w);
    for (int b = (w == 11 ? 28 : 31); b >= 0; b--) {
      auto s = mul28(r, r);
      auto m = mul28(s, a);
      const bool bit = (e >> b) & 1;
#pragma unroll
      for (int i = 0; i < N28; i++) r.l[i] = bit ? m.l[i] : s.l[i];
    }
  }
  return r;
}

// one affine addition given 1 / (x2 - x1)
__device__ __forceinline__ void affine_finish(const g1_affine28& p1, const g1_affine28& p2, const F28n& dinv, F28n& x3o, F28n& y3o) {
  auto dy = sub28<2, 29>(p2.y, p1.y);
  auto lam = mul28(dy, dinv);
  auto l2 = mul28(lam, lam);
  auto x3 = norm28(sub28<4, 30>(l2, add28(p1.x, p2.x)));
  auto dx = sub28<8, 30>(p1.x, x3);
  auto y3 = norm28(sub28<2, 29>(mul28(lam, dx), p1.y));
  x3o = as_n(x3);
  y3o = as_n(y3);
}

// ------------------------------------------------------------------------------------------------ two passes through HBM
// pairs [t K, (t + 1) K) of lane t: entries 2 i, 2 i + 1 of `sorted`.  Prefix slot of (lane t, pair j): pre[j * lanes + t] (coalesced).
template <int INV>
__global__ void __launch_bounds__(256, 2)
k_two_pass(const g1_affine28* __restrict__ table, const uint32_t* __restrict__ sorted, uint32_t K, uint32_t lanes, pre_slot* __restrict__ pre,
           g1_affine28* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const uint2* ent = reinterpret_cast<const uint2*>(sorted) + (size_t)t * K;
  F28n acc = one28();
  uint2 e = ent[0];
  F28n x1 = load_x28(&table[e.x]), x2 = load_x28(&table[e.y]);
  uint2 en = K > 1 ? ent[1] : e;
  for (uint32_t j = 0; j < K; j++) {
    const F28n a = x1, b = x2;
    if (j + 1 < K) { x1 = load_x28(&table[en.x]); x2 = load_x28(&table[en.y]); }
    if (j + 2 < K) en = ent[j + 2];
    auto d = sub28<2, 29>(b, a);
    acc = as_n(mul28(acc, d));
    store_pre(&pre[(size_t)j * lanes + t], acc);
  }
  F28n inv = INV ? fermat_inv28(acc) : acc;
  e = ent[K - 1];
  g1_affine28 p1 = load_pt28(&table[e.x]), p2 = load_pt28(&table[e.y]);
  F28n pr = K > 1 ? load_pre(&pre[(size_t)(K - 2) * lanes + t]) : one28();
  en = K > 1 ? ent[K - 2] : e;
  for (uint32_t j = K; j-- > 0;) {
    const g1_affine28 a = p1, b = p2;
    const F28n pj = pr;
    if (j >= 1) { p1 = load_pt28(&table[en.x]); p2 = load_pt28(&table[en.y]); }
    if (j >= 2) { pr = load_pre(&pre[(size_t)(j - 2) * lanes + t]); en = ent[j - 2]; } else pr = one28();
    auto d = sub28<2, 29>(b.x, a.x);
    const F28n dinv = as_n(mul28(inv, pj));
    inv = as_n(mul28(inv, d));
    F28n x3, y3;
    affine_finish(a, b, dinv, x3, y3);
    store_out(&out[(size_t)t * K + j], x3, y3);
  }
}

// ------------------------------------------------------------------------------------------------ registers only
// the same arithmetic per pair (forward product, two backward products, lambda, square, y3) on loop-carried registers
__global__ void __launch_bounds__(256, 2) k_regs(uint32_t* out, uint32_t seed, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  g1_affine28 p1, p2;
  for (int i = 0; i < N28; i++) {
    p1.x.l[i] = (tid * 2654435761u + i * seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
    p1.y.l[i] = (tid * 40503u + i * 7919u + seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
    p2.x.l[i] = (tid * 69069u + i * 104729u + seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
    p2.y.l[i] = (tid * 1103515245u + i * 12345u + seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
  }
  F28n acc = one28(), inv = one28();
  for (int it = 0; it < iters; it++) {
    auto d = sub28<2, 29>(p2.x, p1.x);
    acc = as_n(mul28(acc, d));                         // forward
    const F28n dinv = as_n(mul28(inv, acc));           // backward
    inv = as_n(mul28(inv, d));
    F28n x3, y3;
    affine_finish(p1, p2, dinv, x3, y3);
    p1.x = x3; p1.y = y3;
    p2.x.l[3] ^= y3.l[0] & 0xffu;
  }
  uint32_t h = 0;
  for (int i = 0; i < N28; i++) h ^= p1.x.l[i] ^ p1.y.l[i] ^ acc.l[i] ^ inv.l[i];
  out[tid] = h;
}
__global__ void __launch_bounds__(256, 2) k_fermat(uint32_t* out, uint32_t seed, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  F28n a;
  for (int i = 0; i < N28; i++) a.l[i] = (tid * 2654435761u + i * seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
  for (int it = 0; it < iters; it++) a = fermat_inv28(a);
  uint32_t h = 0;
  for (int i = 0; i < N28; i++) h ^= a.l[i];
  out[tid] = h;
}

// ------------------------------------------------------------------------------------------------ one lane per pair, product tree in LDS
// Every byte moves once (2 x 112 B in, 112 B out); the prefix products of the workgroup's 256 denominators are a binary tree in LDS:
// up-sweep 255 products, one inversion of the root (lane 0), down-sweep 2 x 255 products.  Levels narrower than a wave leave lanes idle.
constexpr int TREE_L = 256;
template <int INV>
__global__ void __launch_bounds__(TREE_L, 2)
k_lds_tree(const g1_affine28* __restrict__ table, const uint32_t* __restrict__ sorted, uint32_t pairs, g1_affine28* __restrict__ out) {
  __shared__ uint32_t node[2 * TREE_L][N28 + 1];           // node[1] = root, node[TREE_L + i] = leaf i; +1 word: rows on different banks
  const uint32_t t = blockIdx.x * TREE_L + threadIdx.x, li = threadIdx.x;
  const bool live = t < pairs;
  const uint2 e = live ? reinterpret_cast<const uint2*>(sorted)[t] : make_uint2(0, 1);
  const g1_affine28 p1 = load_pt28(&table[e.x]), p2 = load_pt28(&table[e.y]);
  const F28n d = as_n(sub28<2, 29>(p2.x, p1.x));
#pragma unroll
  for (int i = 0; i < N28; i++) node[TREE_L + li][i] = d.l[i];
  __syncthreads();
  for (int w = TREE_L / 2; w >= 1; w >>= 1) {                // up-sweep: nodes [w, 2w)
    if (li < w) {
      F28n a, b;
#pragma unroll
      for (int i = 0; i < N28; i++) { a.l[i] = node[2 * (w + li)][i]; b.l[i] = node[2 * (w + li) + 1][i]; }
      const F28n r = as_n(mul28(a, b));
#pragma unroll
      for (int i = 0; i < N28; i++) node[w + li][i] = r.l[i];
    }
    __syncthreads();
  }
  if (li == 0) {
    F28n r;
#pragma unroll
    for (int i = 0; i < N28; i++) r.l[i] = node[1][i];
    const F28n v = INV ? fermat_inv28(r) : r;
#pragma unroll
    for (int i = 0; i < N28; i++) node[1][i] = v.l[i];
  }
  __syncthreads();
  for (int w = 1; w < TREE_L; w <<= 1) {                     // down-sweep: node n holds 1 / (product of its leaves); children swap factors
    F28n l, r;
    if (li < w) {
      F28n pinv, a, b;
#pragma unroll
      for (int i = 0; i < N28; i++) { pinv.l[i] = node[w + li][i]; a.l[i] = node[2 * (w + li)][i]; b.l[i] = node[2 * (w + li) + 1][i]; }
      l = as_n(mul28(pinv, b));
      r = as_n(mul28(pinv, a));
    }
    __syncthreads();
    if (li < w) {
#pragma unroll
      for (int i = 0; i < N28; i++) { node[2 * (w + li)][i] = l.l[i]; node[2 * (w + li) + 1][i] = r.l[i]; }
    }
    __syncthreads();
  }
  F28n dinv;
#pragma unroll
  for (int i = 0; i < N28; i++) dinv.l[i] = node[TREE_L + li][i];
  F28n x3, y3;
  affine_finish(p1, p2, dinv, x3, y3);
  if (live) store_out(&out[t], x3, y3);
}

// ------------------------------------------------------------------------------------------------ reference: one inversion per pair
__global__ void k_direct(const g1_affine28* __restrict__ table, const uint32_t* __restrict__ sorted, uint32_t pairs, g1_affine28* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pairs) return;
  const uint2 e = reinterpret_cast<const uint2*>(sorted)[t];
  const g1_affine28 p1 = load_pt28(&table[e.x]), p2 = load_pt28(&table[e.y]);
  const F28n dinv = fermat_inv28(as_n(sub28<2, 29>(p2.x, p1.x)));
  F28n x3, y3;
  affine_finish(p1, p2, dinv, x3, y3);
  store_out(&out[t], x3, y3);
}
__global__ void k_fill(g1_affine28* __restrict__ table, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = (uint32_t)i * 2654435761u + seed;
  uint32_t w[32];
  for (int j = 0; j < 32; j++) { s = s * 1664525u + 1013904223u; w[j] = (s >> 4) & MASK28; }
  w[13] &= 0xffu; w[27] &= 0xffu; w[28] = w[29] = w[30] = w[31] = 0;      // top limbs small: values < p
  uint4* q = reinterpret_cast<uint4*>(&table[i]);
  for (int j = 0; j < 8; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
__global__ void k_indices(uint32_t* __restrict__ sorted, size_t m, uint32_t slots, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint64_t s = (i + 1) * 0x9e3779b97f4a7c15ull + seed;
  s ^= s >> 29; s *= 0xbf58476d1ce4e5b9ull; s ^= s >> 32;
  sorted[i] = (uint32_t)(s % slots);
}
// canonical comparison of two result arrays: every limb after full reduction
__global__ void k_compare(const g1_affine28* __restrict__ a, const g1_affine28* __restrict__ b, uint32_t n, uint32_t* __restrict__ bad) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const g1_affine28 p = load_pt28(&a[t]), q = load_pt28(&b[t]);
  const fp_t px = fp_from_28(widen28<Lz>(p.x)), qx = fp_from_28(widen28<Lz>(q.x)), py = fp_from_28(widen28<Lz>(p.y)), qy = fp_from_28(widen28<Lz>(q.y));
  bool same = true;
  for (int i = 0; i < 12; i++) same = same && px.l[i] == qx.l[i] && py.l[i] == qy.l[i];
  if (!same) atomicAdd(bad, 1u);
}

template <class F>
static float time_best(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs, %.0f MHz\n", prop.name, cus, prop.clockRate / 1000.0);
  const uint32_t W = 13, logn = argc > 1 ? atoi(argv[1]) : 20;
  const bool quick = argc > 2;                                   // counters pass: one launch of each memory-bound form, nothing else
  const size_t slots = (size_t)W << logn, M = slots;            // one entry per (window, point), like the real sorted list
  const uint32_t pairs = (uint32_t)(M / 2);
  g1_affine28 *table, *out, *ref;
  uint32_t *sorted, *d_bad, *d_h;
  pre_slot* pre;
  CHECK(hipMalloc(&table, slots * sizeof(g1_affine28)));
  CHECK(hipMalloc(&out, (size_t)pairs * sizeof(g1_affine28)));
  CHECK(hipMalloc(&ref, (size_t)65536 * sizeof(g1_affine28)));
  CHECK(hipMalloc(&sorted, M * 4));
  CHECK(hipMalloc(&pre, (size_t)pairs * sizeof(pre_slot)));
  CHECK(hipMalloc(&d_bad, 4));
  CHECK(hipMalloc(&d_h, (size_t)cus * 4 * 256 * 4));
  hipLaunchKernelGGL(k_fill, dim3((slots + 255) / 256), dim3(256), 0, 0, table, slots, 99u);
  hipLaunchKernelGGL(k_indices, dim3((M + 255) / 256), dim3(256), 0, 0, sorted, M, (uint32_t)slots, 7u);
  CHECK(hipDeviceSynchronize());
  printf("table %.2f GB (%zu slots of 128 B), %u pairs\n", slots * 128.0 / 1e9, slots, pairs);

  if (quick) {
    for (uint32_t K : {52u, 104u}) {
      const uint32_t lanes = pairs / K;
      hipLaunchKernelGGL(k_two_pass<0>, dim3((lanes + 255) / 256), dim3(256), 0, 0, table, sorted, K, lanes, pre, out);
    }
    hipLaunchKernelGGL(k_lds_tree<0>, dim3((pairs + TREE_L - 1) / TREE_L), dim3(TREE_L), 0, 0, table, sorted, pairs, out);
    CHECK(hipDeviceSynchronize());
    return 0;
  }
  // ---- registers only
  for (int w = 1; w <= 2; w++) {
    const int blocks = cus * w, iters = 256;
    float ms = time_best([&] { hipLaunchKernelGGL(k_regs, dim3(blocks), dim3(256), 0, 0, d_h, 12345u, iters); }, 3);
    printf("registers only: affine pair (6 products, no inversion)  %d wave(s)/SIMD  %.3e additions/s\n", w, (double)blocks * 256 * iters / (ms * 1e-3));
  }
  {
    const int blocks = cus * 2;
    float ms = time_best([&] { hipLaunchKernelGGL(k_fermat, dim3(blocks), dim3(256), 0, 0, d_h, 12345u, 2); }, 3);
    printf("registers only: Fermat inversion  2 waves/SIMD  %.3e inversions/s  (%.1f us each per wave)\n", (double)blocks * 256 * 2 / (ms * 1e-3), ms * 1e3 / 2);
  }

  // ---- reference results for the first 65536 pairs
  hipLaunchKernelGGL(k_direct, dim3(65536 / 256), dim3(256), 0, 0, table, sorted, 65536u, ref);
  CHECK(hipDeviceSynchronize());
  auto check = [&](const char* what, uint32_t n) {
    CHECK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_compare, dim3((n + 255) / 256), dim3(256), 0, 0, out, ref, n, d_bad);
    uint32_t bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("    %s: %u of %u sampled results differ from one-inversion-per-pair\n", what, bad, n);
  };

  // ---- two passes through HBM
  const uint32_t Ks[] = {13, 26, 52, 104, 208};
  for (uint32_t K : Ks) {
    const uint32_t lanes = pairs / K;          // the remainder of the list is ignored: a rate, not a result
    const uint32_t grid = (lanes + 255) / 256;
    const double bytes = (double)lanes * K * (2 * 64.0 + 64 + 64 + 2 * 128.0 + 112);     // x gathers (one 64-B half line each), prefix out + in, full gathers, result
    float t0 = time_best([&] { hipLaunchKernelGGL(k_two_pass<0>, dim3(grid), dim3(256), 0, 0, table, sorted, K, lanes, pre, out); });
    float t1 = time_best([&] { hipLaunchKernelGGL(k_two_pass<1>, dim3(grid), dim3(256), 0, 0, table, sorted, K, lanes, pre, out); });
    printf("two passes, K = %3u pairs per lane (%6u lanes = %.2f waves/SIMD): free inversion %.3f ms = %.3e additions/s (%.2f TB/s useful) | Fermat per lane %.3f ms = %.3e additions/s\n",
           K, lanes, lanes / 64.0 / (cus * 4), t0, (double)lanes * K / (t0 * 1e-3), bytes / (t0 * 1e-3) / 1e12, t1, (double)lanes * K / (t1 * 1e-3));
    if (K == 52) check("two passes (Fermat)", 65536 / K * K);
  }
  // ---- one lane per pair, tree in LDS
  {
    const uint32_t grid = (pairs + TREE_L - 1) / TREE_L;
    float t0 = time_best([&] { hipLaunchKernelGGL(k_lds_tree<0>, dim3(grid), dim3(TREE_L), 0, 0, table, sorted, pairs, out); });
    float t1 = time_best([&] { hipLaunchKernelGGL(k_lds_tree<1>, dim3(grid), dim3(TREE_L), 0, 0, table, sorted, pairs, out); });
    printf("one lane per pair, LDS product tree over 256 lanes: free inversion %.3f ms = %.3e additions/s | Fermat by lane 0 %.3f ms = %.3e additions/s\n", t0,
           pairs / (t0 * 1e-3), t1, pairs / (t1 * 1e-3));
    check("LDS tree (Fermat)", 65536);
  }
  return 0;
}
