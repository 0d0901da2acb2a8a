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
  return 0;
}
