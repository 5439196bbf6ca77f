// Probe for the 16-block 4x4x4 MFMA (gfx950) behind stage0b's depthwise-on-MFMA phase:
//   1. operand / result lane maps of v_mfma_f32_4x4x4_16b_f16 and the bf16_1k form, checked with exact
//      integers against the map assumed by the kernels:
//        block b = lane / 4;  A: lane (b, i = lane % 4) holds A_b[i][k = 0..3];
//        B: lane (b, j = lane % 4) holds B_b[k = 0..3][j];  D: lane (b, j) holds D_b[i = 0..3][j] in register i;
//   2. what a stream of these MFMAs leaves of the SIMD's VALU issue for the partner wave (two waves per SIMD):
//      waves 0-3 issue MFMAs, waves 4-7 v_fma_f32 chains; each alone, then together.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma4_probe mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void layout_kernel(float* out_f16, float* out_bf16) {
  const int lane = threadIdx.x, b = lane >> 2, r = lane & 3;
  f16x4 a, bb;
  s16x4 a2, b2;
  for (int k = 0; k < 4; ++k) {
    // A_b[i][k] = 1 + i + 4 k + (b & 3);   B_b[k][j] = 1 + 2 j + k + (b >> 2)   (small integers: exact)
    const float av = 1.f + r + 4 * k + (b & 3), bv = 1.f + 2 * r + k + (b >> 2);
    a[k] = (_Float16)av;
    bb[k] = (_Float16)bv;
    a2[k] = (short)(__builtin_bit_cast(unsigned, av) >> 16);
    b2[k] = (short)(__builtin_bit_cast(unsigned, bv) >> 16);
  }
  f32x4 z = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, z, 0, 0, 0);
  f32x4 e = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a2, b2, z, 0, 0, 0);
  for (int i = 0; i < 4; ++i) {
    out_f16[lane * 4 + i] = d[i];
    out_bf16[lane * 4 + i] = e[i];
  }
}

// MODE bit 0: waves 0-3 issue MFMAs; bit 1: waves 4-7 issue VALU; KIND 0: 4x4x4, 1: 16x16x32 bf16, 2: 32x32x16 bf16
template <int KIND> __global__ __launch_bounds__(512) void coissue_kernel(float* out, long long* cyc, int mode, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x4 ha = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f, (_Float16)1.f}, hb = ha;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (lane + i)); fb[i] = (__bf16)(0.02f * (lane - i)); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {0}, d1 = {0};
  float a0 = lane * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float fm = 0.999f;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    if (mode & 1) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c3, 0, 0, 0);
          } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c3, 0, 0, 0);
          } else {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d1, 0, 0, 0);
          }
        }
      }
    }
  } else if (mode & 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(fm));
        X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      }
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0.x + c1.y + c2.z + c3.w + d0[0] + d1[3];
  if (lane == 0 && blockIdx.x == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

template <int KIND> void coissue(float* out, long long* cyc, const char* name) {
  const int iters = 400;
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(coissue_kernel<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, mode, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("FAILED %s\n", hipGetErrorString(e)); return; }
    long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * 64, n_valu = (double)iters * 64;
    printf("%-10s mode %d (%s): MFMA wave %.2f cyc per MFMA; VALU wave %.2f cyc per v_fma\n", name, mode,
           mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both, one of each per SIMD",
           (mode & 1) ? (double)(h[1] - h[0]) / n_mfma : 0.0, (mode & 2) ? (double)(h[9] - h[8]) / n_valu : 0.0);
    fflush(stdout);
  }
}

int main() {
  float *o1, *o2, *out;
  long long* cyc;
  hipMalloc(&o1, 1024); hipMalloc(&o2, 1024); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, o1, o2);
  float h1[256], h2[256];
  hipMemcpy(h1, o1, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(h2, o2, 1024, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int b = lane >> 2, j = lane & 3;
    for (int i = 0; i < 4; ++i) {
      float ref = 0;
      for (int k = 0; k < 4; ++k) ref += (1.f + i + 4 * k + (b & 3)) * (1.f + 2 * j + k + (b >> 2));
      if (h1[lane * 4 + i] != ref) ++bad1;
      if (h2[lane * 4 + i] != ref) ++bad2;
    }
  }
  printf("layout check (assumed map): f16 mismatches %d / 256, bf16_1k mismatches %d / 256\n", bad1, bad2);
  if (bad1 || bad2) {
    printf("lane 0..7 f16 results:\n");
    for (int lane = 0; lane < 8; ++lane)
      printf("  lane %d: %g %g %g %g | bf16 %g %g %g %g\n", lane, h1[lane * 4], h1[lane * 4 + 1], h1[lane * 4 + 2],
             h1[lane * 4 + 3], h2[lane * 4], h2[lane * 4 + 1], h2[lane * 4 + 2], h2[lane * 4 + 3]);
  }
  fflush(stdout);
  coissue<0>(out, cyc, "4x4x4");
  coissue<1>(out, cyc, "16x16x32");
  coissue<2>(out, cyc, "32x32x16");
  printf("done: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
