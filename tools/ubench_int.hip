// Instruction-rate microbenchmarks for the integer pipeline of gfx950 (MI355X).
// Answers SURVEY.md section 8(d)'s open question: the issue rate of v_mad_u64_u32 & friends,
// which bounds the Fp/Fr Montgomery multiplier (and therefore MSM and NTT).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o tools/ubench_int
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

// Each kernel runs ITERS*UNROLL copies of one instruction per lane on NCHAIN independent chains.
#define KHEAD(name) extern "C" __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#define LOOP(NCH, ...) for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int u = 0; u < UNROLL / NCH; u++) { __VA_ARGS__ } }

// --- v_mad_u64_u32, 4 independent chains
KHEAD(k_mad64_4)
  uint64_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t x = seed | 1, y = tid | 3;
  LOOP(4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");)
  out[tid] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
}
// --- v_mad_u64_u32, 1 dependent chain
KHEAD(k_mad64_1)
  uint64_t a0 = tid; uint32_t x = seed | 1, y = tid | 3;
  LOOP(1, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y) : "vcc");)
  out[tid] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32);
}
// --- mad + addc pair (the 96-bit accumulate step), 1 chain
KHEAD(k_macc_1)
  uint64_t a0 = tid; uint32_t h = 0; uint32_t x = seed | 1, y = tid | 3;
  LOOP(1, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(a0), "+v"(h) : "v"(x), "v"(y) : "vcc");)
  out[tid] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32) ^ h;
}
// --- mad + addc pair, 2 chains
KHEAD(k_macc_2)
  uint64_t a0 = tid; uint32_t h0 = 0; uint64_t a1 = tid * 3; uint32_t h1 = 0; uint32_t x = seed | 1, y = tid | 3;
  LOOP(2, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"
               : "+v"(a0), "+v"(h0), "+v"(a1), "+v"(h1) : "v"(x), "v"(y) : "vcc");)
  out[tid] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32) ^ h0 ^ (uint32_t)a1 ^ (uint32_t)(a1 >> 32) ^ h1;
}
// --- v_mul_lo_u32 / v_mul_hi_u32, 4 chains
KHEAD(k_mullo_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t y = seed | 3;
  LOOP(4, asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
KHEAD(k_mulhi_4)
  uint32_t a0 = ~tid, a1 = ~tid + 1, a2 = ~tid + 2, a3 = ~tid + 3; uint32_t y = ~seed | 3;
  LOOP(4, asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
// --- v_mad_u32_u24 (full-rate candidate), 4 chains
KHEAD(k_mad24_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t x = seed | 1, y = tid | 3;
  LOOP(4, asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n\tv_mad_u32_u24 %1, %4, %5, %1\n\tv_mad_u32_u24 %2, %4, %5, %2\n\tv_mad_u32_u24 %3, %4, %5, %3"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
// --- v_mul_hi_u32_u24, 4 chains
KHEAD(k_mulhi24_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t y = seed | 0xffff3;
  LOOP(4, asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\tv_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
// --- v_add_co_u32 + v_addc_co_u32 carry chain pieces, 4 chains of v_add_u32
KHEAD(k_add32_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t y = seed | 3;
  LOOP(4, asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
KHEAD(k_addc_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t y = seed | 0xfffffff3;
  LOOP(4, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) : "vcc");)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
// --- v_lshl_add_u64 (64-bit add), 4 chains
KHEAD(k_add64_4)
  uint64_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint64_t y = ((uint64_t)seed << 20) | 3;
  LOOP(4, asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));)
  out[tid] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3) >> 32);
}
// --- v_fma_f64, 4 chains
KHEAD(k_fma64_4)
  double a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; double x = 1.0000001, y = 1e-9 * seed;
  LOOP(4, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));)
  out[tid] = (uint32_t)(a0 + a1 + a2 + a3);
}
// --- v_fma_f32, 4 chains (full-rate yardstick)
KHEAD(k_fma32_4)
  float a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; float x = 1.0000001f, y = 1e-9f * seed;
  LOOP(4, asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));)
  out[tid] = (uint32_t)(a0 + a1 + a2 + a3);
}
// --- v_dot2_u32_u16 (2 x 16-bit MAC per lane per op), 4 chains
KHEAD(k_dot2_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t x = seed | 0x00030001, y = tid | 0x00050003;
  LOOP(4, asm volatile("v_dot2_u32_u16 %0, %4, %5, %0\n\tv_dot2_u32_u16 %1, %4, %5, %1\n\tv_dot2_u32_u16 %2, %4, %5, %2\n\tv_dot2_u32_u16 %3, %4, %5, %3"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
// --- v_dot4_u32_u8, 4 chains
KHEAD(k_dot4_4)
  uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3; uint32_t x = seed | 0x01030001, y = tid | 0x02050003;
  LOOP(4, asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n\tv_dot4_u32_u8 %1, %4, %5, %1\n\tv_dot4_u32_u8 %2, %4, %5, %2\n\tv_dot4_u32_u8 %3, %4, %5, %3"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));)
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Bench { const char* name; kern_t k; double ops_per_inst; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  uint32_t* out; 
  const int wpcs[] = {4, 8, 16};   // waves per CU (1, 2, 4 per SIMD)
  Bench benches[] = {
    {"v_fma_f32 x4chains", k_fma32_4, 1}, {"v_add_u32 x4", k_add32_4, 1}, {"v_add_co/addc x4", k_addc_4, 1},
    {"v_lshl_add_u64 x4", k_add64_4, 1}, {"v_fma_f64 x4", k_fma64_4, 1},
    {"v_mul_lo_u32 x4", k_mullo_4, 1}, {"v_mul_hi_u32 x4", k_mulhi_4, 1},
    {"v_mad_u64_u32 x4", k_mad64_4, 1}, {"v_mad_u64_u32 x1(dep)", k_mad64_1, 1},
    {"mad64+addc x1(dep)", k_macc_1, 2}, {"mad64+addc x2", k_macc_2, 2},
    {"v_mad_u32_u24 x4", k_mad24_4, 1}, {"v_mul_hi_u32_u24 x4", k_mulhi24_4, 1},
    {"v_dot2_u32_u16 x4", k_dot2_4, 1}, {"v_dot4_u32_u8 x4", k_dot4_4, 1},
  };
  CHECK(hipMalloc(&out, (size_t)cus * 16 * 64 * 4 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-26s %6s %12s %14s %16s\n", "instruction", "wv/CU", "ms", "Ginst/s(lane)", "lanes/clk/CU@2.4");
  for (auto& b : benches) {
    for (int wpc : wpcs) {
      int blocks = cus * wpc / 4;   // 256-thread blocks = 4 waves
      hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);   // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      const int reps = 5;
      for (int r = 0; r < reps; r++) hipLaunchKernelGGL(b.k, dim3(blocks), dim3(256), 0, 0, out, 12345u + r);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      double insts = (double)blocks * 256 * ITERS * UNROLL * b.ops_per_inst;   // lane-instructions
      double rate = insts / (ms * 1e-3);
      printf("%-26s %6d %12.4f %14.1f %16.2f\n", b.name, wpc, ms, rate / 1e9, rate / 2.4e9 / cus);
    }
  }
  CHECK(hipFree(out));
  return 0;
}
