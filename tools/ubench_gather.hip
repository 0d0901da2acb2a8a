// ubench_gather.hip -- what does FETCH_SIZE count for the gathers of msm_accumulate?  
// The guide (MI355X_MICROARCH.md, HBM) calibrates ONE case: a wide coalesced streaming read (16 B per lane in >= 128-B runs) shows up
// at exactly half its bytes.  msm_accumulate reads something else: every lane gathers its own 128-byte slot with 7 x dwordx4 (112 B of
// the line).  This program issues a known number of such gathers from a table far larger than L2 + Infinity Cache (default 4 GiB,
// every slot touched exactly once: index = an odd multiple of the gather's number mod the slot count), beside two controls -- the
// same bytes as a coalesced stream, and gathers of the full 128 bytes -- so that
//     factor = FETCH_SIZE (bytes) / (128 B x gathers)
// can be read off a counters-only pass:   rocprofv3 --pmc FETCH_SIZE -d out -- tools/ubench_gather [log2 slots] [log2 gathers]
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct alignas(128) Slot { uint4 q[8]; };

// WORDS x dwordx4 of every gathered slot, the access shape of load_affine28 (msm_kernels.hpp) for WORDS = 7
template <int WORDS>
__global__ void __launch_bounds__(256) gather(const Slot* __restrict__ table, uint64_t slot_mask, uint64_t stride, uint32_t per_lane, uint32_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < per_lane; i++) {
    const uint64_t g = (uint64_t)i * lanes + t;                 // number of this gather
    const uint4* q = reinterpret_cast<const uint4*>(&table[(g * stride) & slot_mask]);
    uint4 v[WORDS];
#pragma unroll
    for (int j = 0; j < WORDS; j++) v[j] = q[j];
#pragma unroll
    for (int j = 0; j < WORDS; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x12345678u) out[t & 1023] = acc;                 // keeps the loads alive
}
// the same bytes as a coalesced stream: consecutive lanes read consecutive 16-byte words (the guide's calibrated case)
__global__ void __launch_bounds__(256) stream16(const uint4* __restrict__ table, uint64_t words, uint32_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, lanes = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint64_t i = t; i < words; i += lanes) {
    const uint4 v = table[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[t & 1023] = acc;
}

int main(int argc, char** argv) {
  const int log_slots = argc > 1 ? atoi(argv[1]) : 25, log_g = argc > 2 ? atoi(argv[2]) : 24;
  const uint64_t slots = 1ull << log_slots, gathers = 1ull << log_g;
  if (log_g > log_slots) { fprintf(stderr, "gathers must not exceed slots (every slot at most once)\n"); return 1; }
  Slot* table;
  uint32_t* out;
  CK(hipMalloc((void**)&table, slots * sizeof(Slot)));
  CK(hipMalloc((void**)&out, 4096));
  CK(hipMemset(table, 0x5a, slots * sizeof(Slot)));
  CK(hipDeviceSynchronize());
  const uint32_t lanes = 131072, per_lane = (uint32_t)(gathers / lanes);
  const uint64_t stride = 0x9E3779B1ull | 1ull;                 // odd: g -> g * stride mod 2^k is a permutation of the slots
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather<7>, dim3(lanes / 256), dim3(256), 0, 0, table, slots - 1, stride, per_lane, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gather<7>  %llu gathers of 7 x 16 B from %llu slots of 128 B (%.1f GiB): %.3f ms, lines x 128 B = %llu bytes (%.0f GB/s)\n",
           (unsigned long long)gathers, (unsigned long long)slots, slots * 128.0 / (1 << 30), ms, (unsigned long long)(gathers * 128),
           gathers * 128.0 / ms / 1e6);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather<8>, dim3(lanes / 256), dim3(256), 0, 0, table, slots - 1, stride + 2, per_lane, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gather<8>  %llu gathers of 8 x 16 B: %.3f ms, %llu bytes (%.0f GB/s)\n", (unsigned long long)gathers, ms, (unsigned long long)(gathers * 128),
           gathers * 128.0 / ms / 1e6);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream16, dim3(lanes / 256), dim3(256), 0, 0, reinterpret_cast<const uint4*>(table), gathers * 8, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream16   %llu bytes as a coalesced 16-B-per-lane stream: %.3f ms (%.0f GB/s)\n", (unsigned long long)(gathers * 128), ms, gathers * 128.0 / ms / 1e6);
  }
  printf("EXPECT_BYTES %llu\n", (unsigned long long)(gathers * 128));
  return 0;
}
