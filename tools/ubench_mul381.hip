// A/B of the two ways to form a 381-bit x 381-bit product on gfx950 (VERDICT r01 next #4b):
//   A  14 x 28-bit limbs, one v_mad_u64_u32 per limb product, 64-bit column sums (what fp28.hpp's mul28 does; product part only)
//   B  8 x 52-bit limbs held as doubles, two v_fma_f64 per limb product (high and low half by the 2^104 trick of Emmart et
//      al.), the halves added into 64-bit integer column sums
// Both loops feed their result back into an operand so nothing is hoisted; output = products per lane per microsecond and the
// static VALU count per product (from the assembly: hipcc -S).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mul381.hip -o tools/ubench_mul381
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048;

extern "C" __global__ void __launch_bounds__(256) k_mul28(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, M = (1u << 28) - 1;
  uint32_t a[14], b[14];
  for (int i = 0; i < 14; i++) { a[i] = (tid * 2654435761u + i * seed) & M; b[i] = (tid * 40503u + i * 7919u + seed) & M; }
  for (int it = 0; it < ITERS; it++) {
    uint32_t r[28];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
      for (int i = (k < 14 ? 0 : k - 13); i <= (k < 14 ? k : 13); i++) acc += (uint64_t)a[i] * b[k - i];
      r[k] = (uint32_t)acc & M;
      acc >>= 28;
    }
    r[27] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < 14; i++) a[i] = (r[i] ^ r[i + 14]) & M;          // next product depends on this one
  }
  uint32_t x = 0;
  for (int i = 0; i < 14; i++) x ^= a[i];
  out[tid] = x;
}

extern "C" __global__ void __launch_bounds__(256) k_mul52(uint32_t* out, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
  double a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = (double)((tid * 2654435761u + i * seed) & 0xfffffu) * 4294967296.0 + (double)(tid + i); b[i] = (double)((tid * 40503u + i * 7919u + seed) & 0xfffffu) * 4294967296.0 + (double)(seed + i); }
  for (int it = 0; it < ITERS; it++) {
    unsigned long long lo[16], hi[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lo[k] = 0; hi[k] = 0; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double h = __builtin_fma(a[i], b[j], c1);                   // high half of the 104-bit product, in units of 2^52
        const double l = __builtin_fma(a[i], b[j], c2 - h);               // low half
        hi[i + j] += (unsigned long long)__double_as_longlong(h);
        lo[i + j] += (unsigned long long)__double_as_longlong(l);
      }
    }
    // columns: value_k = lo[k] + hi[k - 1] (the per-column constants bits(c1), bits(2^52) are subtracted once per column);
    // back to 52-bit limbs as doubles for the next product
    unsigned long long carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      unsigned long long v = lo[k] + (k ? hi[k - 1] : 0ull) + carry;
      carry = v >> 52;
      const unsigned long long limb = (v ^ lo[k + 8] ^ hi[k + 7]) & ((1ull << 52) - 1);
      a[k] = (double)(long long)limb;
    }
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[tid] = (uint32_t)(long long)s;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 64 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-34s %6s %10s %22s\n", "381-bit product", "wv/CU", "ms", "products/us/lane x 1e3");
  struct { const char* name; void (*k)(uint32_t*, uint32_t); } ks[] = {{"A: 14x28-bit limbs, v_mad_u64_u32", k_mul28}, {"B: 8x52-bit limbs, v_fma_f64 pair", k_mul52}};
  for (auto& b : ks)
    for (int wpc : {8, 16}) {
      const int blocks = cus * wpc / 4;
      hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 777u + r);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("%-34s %6d %10.4f %22.3f\n", b.name, wpc, ms, 1e3 * ITERS / (ms * 1e3));
    }
  return 0;
}
