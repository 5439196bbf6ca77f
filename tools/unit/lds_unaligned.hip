// Does gfx950 serve a ds_read_b64 whose address is only 2-byte aligned, and at what price?  (A depthwise filter
// gradient on the 4x4x4 matrix pipe wants Toeplitz operands built from DATA: four consecutive x starting at any
// column of a 16-bit planar image.)  Prints mismatches per byte offset 0, 2, 4, 6 and cycles per read.
//   hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void probe(int* bad, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  for (int off = 0; off < 4; ++off) {
    // element index (2-byte units): 37 * lane + off -> byte address 74 lane + 2 off: every alignment class
    const int e0 = (37 * lane + off * 1) & 8191;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)(sm) + 2 * e0;
    uint2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    const unsigned short got[4] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16), (unsigned short)(v.y & 0xffff),
                                   (unsigned short)(v.y >> 16)};
    int nb = 0;
    for (int k = 0; k < 4; ++k) nb += got[k] != (unsigned short)(e0 + k);
    if (nb) atomicAdd(bad + ((2 * e0) & 7) / 2, nb);
  }
  // timing: aligned vs odd (2 mod 8) addresses, 64 dependent-free reads per rep
  for (int mode = 0; mode < 2; ++mode) {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)(sm) + 8 * (lane & 63) * 9 + (mode ? 2 : 0);
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        uint2 v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc += v.x ^ v.y;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
    if (acc == 0x12345678u) bad[7] = 1;
  }
}

int main() {
  int* bad;
  unsigned long long* cyc;
  hipMalloc(&bad, 8 * sizeof(int));
  hipMalloc(&cyc, 2 * sizeof(unsigned long long));
  hipMemset(bad, 0, 8 * sizeof(int));
  const int reps = 200;
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, bad, cyc, reps);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  int hb[8];
  unsigned long long hc[2];
  hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost);
  hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
  printf("ds_read_b64 mismatches by (byte address mod 8): 0:%d 2:%d 4:%d 6:%d\n", hb[0], hb[1], hb[2], hb[3]);
  printf("cycles per wave-level ds_read_b64 (4 waves on the CU): aligned %.1f, address = 2 mod 8 %.1f\n",
         (double)hc[0] / (reps * 16), (double)hc[1] / (reps * 16));
  return 0;
}
