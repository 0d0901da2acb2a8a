// What one NTT butterfly costs on gfx950 when nothing but the VALU is involved (registers only, no LDS, no memory): the loop
// body is fr29_butterfly from csrc/fr29.hpp fed back into itself.  Prints clocks per butterfly per SIMD for 1, 2, 3 and 4 waves
// per SIMD, next to the static VALU count of the loop body (hipcc -S), to tell issue cost from LDS / barrier cost in the pass
// kernels (tools/ntt_occupancy_probe.sh measured 4.9 clk per VALU instruction there).
// Build: hipcc --offload-arch=gfx950 -O3 -I baby_plonk_rust_amd/csrc -I include tools/ubench_fr29.hip -o tools/ubench_fr29
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fr29.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 1024;
using namespace bp;

__global__ void __launch_bounds__(256) k_butterfly(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  fr29 u, v, w;
  for (int i = 0; i < N29; i++) {
    u.l[i] = (tid * 2654435761u + i * seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    v.l[i] = (tid * 40503u + i * 7919u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    w.l[i] = (tid * 69069u + i * 104729u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
  }
  for (int it = 0; it < ITERS; it++) fr29_butterfly(u, v, w);
  uint32_t x = 0;
  for (int i = 0; i < N29; i++) x ^= u.l[i] ^ v.l[i];
  out[tid] = x;
}
// two independent butterflies per iteration (what a radix-4 group offers the scheduler)
__global__ void __launch_bounds__(256) k_butterfly2(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  fr29 u, v, w, u2, v2;
  for (int i = 0; i < N29; i++) {
    u.l[i] = (tid * 2654435761u + i * seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    v.l[i] = (tid * 40503u + i * 7919u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    w.l[i] = (tid * 69069u + i * 104729u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    u2.l[i] = v.l[i] ^ 5u; v2.l[i] = u.l[i] ^ 9u;
  }
  for (int it = 0; it < ITERS / 2; it++) { fr29_butterfly(u, v, w); fr29_butterfly(u2, v2, w); }
  uint32_t x = 0;
  for (int i = 0; i < N29; i++) x ^= u.l[i] ^ v.l[i] ^ u2.l[i] ^ v2.l[i];
  out[tid] = x;
}

// ---- the twiddle product in Shoup's form, for the count and the clock only -- not used by the transforms.
// r = a w - floor(a w' / 2^261) q (mod 2^261) with w' = floor(w 2^261 / q) stored beside the twiddle: the high half of a w'
// (columns 9..16, plus 7 and 8 for the carry: what lies below changes the quotient by < 2^-20), the low halves of a w and of t q
// (q_0 = 1).  a: lazy limbs < 1.5 * 2^31, value < 2^258; result < 2q, normalised.  134 multiply-adds against fr29_mul's 153.
__device__ __forceinline__ fr29 fr29_mul_shoup(const fr29& a, const fr29& w, const fr29& wp) {
  uint32_t t[N29], lo1[N29], lo2[N29];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 7; k < 2 * N29 - 1; k++) {
#pragma unroll
    for (int i = (k < N29 ? 0 : k - N29 + 1); i <= (k < N29 ? k : N29 - 1); i++) acc += (uint64_t)a.l[i] * wp.l[k - i];
    if (k >= N29) t[k - N29] = (uint32_t)acc & MASK29;
    acc >>= 29;
  }
  t[N29 - 1] = (uint32_t)acc;
  acc = 0;
#pragma unroll
  for (int k = 0; k < N29; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * w.l[k - i];
    lo1[k] = (uint32_t)acc & MASK29;
    acc >>= 29;
  }
  acc = 0;
#pragma unroll
  for (int k = 0; k < N29; k++) {
    acc += t[k];                                           // t_k q_0, q_0 = 1
#pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)t[i] * Q29::mod(k - i);
    lo2[k] = (uint32_t)acc & MASK29;
    acc >>= 29;
  }
  fr29 r;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N29; i++) {
    const int32_t d = (int32_t)lo1[i] - (int32_t)lo2[i] + borrow;
    r.l[i] = (uint32_t)d & MASK29;                         // the top limb too: the difference is taken mod 2^261
    borrow = d >> 29;
  }
  fr29 m;                                                  // r in [0, 3q): below 2q it stays, else r - 2q
  const uint32_t b2 = fr29_sub_exact(m, r, [](int i) { return Q29::two_q(i); });
  fr29 o;
#pragma unroll
  for (int i = 0; i < N29; i++) o.l[i] = b2 ? r.l[i] : m.l[i];
  return o;
}
__device__ __forceinline__ void fr29_butterfly_shoup(fr29& u, fr29& v, const fr29& w, const fr29& wp) {
  fr29 s = fr29_add_lazy(u, v);
  v = fr29_mul_shoup(fr29_sub_lazy(u, v), w, wp);
  u = s;
}
__global__ void __launch_bounds__(256) k_butterfly_shoup(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  fr29 u, v, w, wp;
  for (int i = 0; i < N29; i++) {
    u.l[i] = (tid * 2654435761u + i * seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    v.l[i] = (tid * 40503u + i * 7919u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    w.l[i] = (tid * 69069u + i * 104729u + seed) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    wp.l[i] = (tid * 1103515245u + i * 12345u + seed) & (i == N29 - 1 ? 0x3ffffffu : MASK29);
  }
  for (int it = 0; it < ITERS; it++) fr29_butterfly_shoup(u, v, w, wp);
  uint32_t x = 0;
  for (int i = 0; i < N29; i++) x ^= u.l[i] ^ v.l[i];
  out[tid] = x;
}
// correctness of the form above on real constants: x (plain), x 2^261 mod q (Montgomery twiddle), floor(x 2^261 / q) (Shoup companion)
__global__ void k_shoup_check(uint32_t* bad) {
  const uint32_t xw[9] = {0x10abcdefu, 0x11a2b3c4u, 0xaf37bc4u, 0x8acf121u, 0x1cdef123u, 0xb3c4855u, 0x17bc48d1u, 0xf121579u, 0x123456u};
  const uint32_t xm[9] = {0xd71096au, 0x868bdc8u, 0x5d944ddu, 0x9410c51u, 0x14ecb550u, 0x1e14c2bau, 0x16a87bbcu, 0x11a49824u, 0x2d0c99u};
  const uint32_t xp[9] = {0x128ef696u, 0xc0ef6e7u, 0x1e217dfu, 0x891cda8u, 0x135fcf1cu, 0x75806eau, 0x9780693u, 0x92e277eu, 0x50667b9u};
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  fr29 u, v, w, wm, wp;
  for (int i = 0; i < N29; i++) {
    u.l[i] = (tid * 2654435761u + i * 977u) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    v.l[i] = (tid * 40503u + i * 7919u + 5u) & (i == N29 - 1 ? 0x3fffffu : MASK29);
    w.l[i] = xw[i]; wm.l[i] = xm[i]; wp.l[i] = xp[i];
  }
  const fr29 d = fr29_sub_lazy(u, v);
  const fr_t a = fr29_to_sat_canonical(fr29_mul(d, wm)), b = fr29_to_sat_canonical(fr29_mul_shoup(d, w, wp));
  bool same = true;
  for (int i = 0; i < 8; i++) same = same && a.l[i] == b.l[i];
  if (!same) atomicAdd(bad, 1u);
}

template <class K>
static void run(const char* name, K kernel, int waves_per_simd, uint32_t* d_out, int cus, double mhz) {
  const int blocks = cus * waves_per_simd;          // 256 lanes = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 777u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double clk = ms * 1e-3 * mhz * 1e6;                       // clocks of the run; every SIMD ran waves_per_simd waves x ITERS butterflies
  printf("%-14s %d wave(s)/SIMD  %8.3f ms  %7.0f clk per butterfly per SIMD\n", name, waves_per_simd, ms, clk / ((double)ITERS * waves_per_simd));
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("%s, %d CUs, %.0f MHz\n", p.name, cus, mhz);
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)cus * 4 * 256 * 4));
  for (int w = 1; w <= 4; w++) run("butterfly", k_butterfly, w, d_out, cus, mhz);
  for (int w = 1; w <= 4; w++) run("2 butterflies", k_butterfly2, w, d_out, cus, mhz);
  for (int w = 1; w <= 4; w++) run("Shoup form", k_butterfly_shoup, w, d_out, cus, mhz);
  uint32_t* d_bad;
  CHECK(hipMalloc(&d_bad, 4));
  CHECK(hipMemset(d_bad, 0, 4));
  hipLaunchKernelGGL(k_shoup_check, dim3(64), dim3(256), 0, 0, d_bad);
  uint32_t bad = 0;
  CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("Shoup form against fr29_mul on 16 384 operands (x = 0x1234..ef, its Montgomery twiddle and its Shoup companion): %u differ\n", bad);
  return 0;
}
