// What one bucket addition costs on gfx950 when nothing but the VALU is involved: g1_add_mixed28 (csrc/g1_28.hpp, the loop body
// of msm_accumulate) on a register-resident accumulator and a register-resident point whose limbs change every iteration.
// Prints clocks per addition per SIMD at one and two waves per SIMD (the kernel runs two), and additions per second for the
// whole chip: the ceiling `msm_accumulate` can reach without any memory access.
// Build: hipcc --offload-arch=gfx950 -O3 -I baby_plonk_rust_amd/csrc -I include tools/ubench_g1add.hip -o tools/ubench_g1add
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "g1_28.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 512;
using namespace bp;

__global__ void __launch_bounds__(256, 2) k_add(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  g1_proj28 acc = g1_identity28();
  F28n x, y;
  for (int i = 0; i < N28; i++) {
    x.l[i] = (tid * 2654435761u + i * seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
    y.l[i] = (tid * 40503u + i * 7919u + seed) & (i == N28 - 1 ? 0x1ffu : MASK28);
  }
  for (int it = 0; it < ITERS; it++) {
    g1_add_mixed28(acc, x, pt_y_signed(y, (it & 1) != 0));
    x.l[it % 13] ^= acc.x.l[0] & 0xffu;        // keeps the point a loop-carried value (not a valid curve point: only the cost matters)
  }
  uint32_t h = 0;
  for (int i = 0; i < N28; i++) h ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.z.l[i];
  out[tid] = h;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("%s, %d CUs, %.0f MHz\n", p.name, cus, mhz);
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)cus * 2 * 256 * 4));
  for (int w = 1; w <= 2; w++) {
    const int blocks = cus * w;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, d_out, 777u + rep);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double clk = best * 1e-3 * mhz * 1e6;
    const double adds = (double)blocks * 256 * ITERS;
    printf("g1_add_mixed28  %d wave(s)/SIMD  %8.3f ms  %7.0f clk per addition per SIMD  %.3e additions/s\n", w, best, clk / ((double)ITERS * w),
           adds / (best * 1e-3));
  }
  return 0;
}
