// How fast can every CU stream the SAME weight panel out of L2 (gfx950)?  Decides the stage-2 tail kernel's shape:
// one workgroup per CU reads a 1 MiB bf16 panel (every workgroup the same bytes, as per-CU weight streaming would)
//   mode 0: global_load_dwordx4 straight to registers, each wave its own 16 KiB slices, U loads in flight
//   mode 1: LDS-DMA (global_load_lds_dwordx4) into a ring, all waves loading
// Reports GB/s per CU and chip-wide for 4 / 8 / 16 waves per workgroup and 1 or 2 workgroups per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o l2stream l2stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int U> __global__ __launch_bounds__(1024) void reg_stream(const uint4* __restrict__ w, size_t n16, int reps, float* out) {
  // the panel as n16 16-byte pieces; a wave takes pieces [k * 64 + lane] for k = wave, wave + nwaves, ...
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t wave_pieces = n16 / 64;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t k = wave; k + (size_t)(U - 1) * nw < wave_pieces; k += (size_t)U * nw) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = w[(k + (size_t)u * nw) * 64 + lane];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

__global__ __launch_bounds__(1024) void dma_stream(const unsigned char* __restrict__ w, size_t bytes, int reps, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // ring: nw waves x 4 KiB each x 2 halves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t pieces = bytes / 1024;   // 1 KiB per wave-instruction
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    int slot = 0;
    for (size_t k = wave; k < pieces; k += nw) {
      __builtin_amdgcn_global_load_lds((gptr_t)(w + k * 1024 + lane * 16),
                                       (lptr_t)(lds + (wave * 8 + (slot & 7)) * 1024), 16, 0, 0);
      ++slot;
      if ((slot & 7) == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // keep 4..8 pieces in flight per wave
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += reinterpret_cast<unsigned*>(lds)[threadIdx.x];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

int main() {
  const size_t bytes = 1 << 20;
  unsigned char* w;
  float* out;
  hipMalloc(&w, bytes);
  hipMalloc(&out, 4 << 20);
  hipMemset(w, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 64;
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int wgs = 256; wgs <= 512; wgs *= 2)
    for (int nw = 4; nw <= 16; nw *= 2) {
      if (wgs == 512 && nw == 16) continue;
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int t = 0; t < 3; ++t) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(reg_stream<4>, dim3(wgs), dim3(64 * nw), 0, 0, (const uint4*)w, bytes / 16, reps, out);
          else if (mode == 1) hipLaunchKernelGGL(reg_stream<8>, dim3(wgs), dim3(64 * nw), 0, 0, (const uint4*)w, bytes / 16, reps, out);
          else hipLaunchKernelGGL(dma_stream, dim3(wgs), dim3(64 * nw), nw * 8 * 1024, 0, w, bytes, reps, out);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        const double per_wg = (double)bytes * reps / (best * 1e-3) / 1e9;
        printf("%d WGs x %2d waves, %-22s: %.3f ms, %.1f GB/s per workgroup, %.2f TB/s chip-wide\n", wgs, nw,
               mode == 0 ? "dwordx4 to regs, 4 deep" : mode == 1 ? "dwordx4 to regs, 8 deep" : "LDS-DMA ring",
               best, per_wg, per_wg * wgs / 1e3);
        fflush(stdout);
      }
    }
  printf("done: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
