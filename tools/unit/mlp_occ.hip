// Occupancy probe behind round 4's re-cut of the stage-0 / stage-1 MLP loops (gfx950): the fc1 -> GELU -> fc2 chunk loop
// of stage0b.hip / stage1b.hip with the filters resident in LDS (no DMA ring), at
//   NCB  column blocks (32 pixels each) per wave,
//   NW   waves per workgroup,
//   WPS  waves per SIMD the register budget allows (launch bounds),
//   BAR  one s_barrier per chunk (as the ring needs) or none.
// Same total work per launch for every variant (4096 column blocks x ITERS chunks); prints us per launch and cycles per
// (wave, chunk).  Build: hipcc -O3 --offload-arch=gfx950 -o mlp_occ mlp_occ.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ float relu_f(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float gelu3(float x) {
  const float a = __builtin_fabsf(x);
  float t = fmaf(a, -0.024772998623334343f, -0.49926576060257244f);
  t = fmaf(a, t, -1.1287482669759885f);
  t = fmaf(a, t, -1.0036805164077327f);
  return fmaf(-a, __builtin_amdgcn_exp2f(t), relu_f(x));
}

// C = 64 * KS1 / 4: KS1 = 4 -> stage 0 (64 channels, CT = 2), KS1 = 8 -> stage 1 (128 channels, CT = 4)
template <int NCB, int NW, int WPS, int KS1, bool BAR>
__global__ __launch_bounds__(NW * 64, WPS) void mlp_kernel(float* out, int iters) {
  constexpr int CT = KS1 / 2;
  constexpr int CHB = (KS1 + 2 * CT) * 1024;   // bytes of one chunk's fragments: W1 KS1 KiB + W2 2 CT KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 2 * CHB / 4; i += NW * 64)
    reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 x[NCB][CT];
  bf16x8 xf[NCB][KS1];
#pragma unroll
  for (int t = 0; t < NCB; ++t) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][ct][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[t][ks][e] = (__bf16)(0.01f * ((lane + e + ks + t) & 15) - 0.07f);
  }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const unsigned char* base = lds + (it & 1) * CHB + lane * 16;
    if (BAR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    bf16x8 a1[KS1], a2[CT][2];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) a1[ks] = *reinterpret_cast<const bf16x8*>(base + ks * 1024);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) a2[ct][s2] = *reinterpret_cast<const bf16x8*>(base + (KS1 + ct * 2 + s2) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 hacc[NCB];
#pragma unroll
    for (int t = 0; t < NCB; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[t][r] = 0.01f * r;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        hacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], xf[t][ks], hacc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NCB; ++t) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 hf;
#pragma unroll
        for (int r = 0; r < 8; ++r) hf[r] = (__bf16)gelu3(hacc[t][8 * s2 + r]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          x[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ct][s2], hf, x[t][ct], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NCB; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += x[t][ct][r];
  out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = s;
}


typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// packed-f16 GELU of two values: q(a) on v_pk_fma_f16, 2^q per half (v_exp_f16), max(x, 0) - a E packed
__device__ __forceinline__ f16x2 gelu_pk(float x0, float x1) {
  const f16x2 xh = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const f16x2 a = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, xh) & 0x7fff7fffu);
  const f16x2 c3 = {(_Float16)-0.024772998623334343f, (_Float16)-0.024772998623334343f};
  const f16x2 c2 = {(_Float16)-0.49926576060257244f, (_Float16)-0.49926576060257244f};
  const f16x2 c1 = {(_Float16)-1.1287482669759885f, (_Float16)-1.1287482669759885f};
  const f16x2 c0 = {(_Float16)-1.0036805164077327f, (_Float16)-1.0036805164077327f};
  f16x2 t = a * c3 + c2;
  t = a * t + c1;
  t = a * t + c0;
  const f16x2 e = __builtin_elementwise_exp2(t);   // hipcc: v_exp_f16 per half
  const f16x2 z = {(_Float16)0.f, (_Float16)0.f};
  const f16x2 r = __builtin_elementwise_max(xh, z);
  return r - a * e;
}

// The lean loop: one column block per wave; nothing but the accumulators lives across a chunk -- the fc1 B operand is
// re-read from LDS every chunk, filter fragments are read in groups of four right before their products
template <int NW, int WPS, int KS1, bool PK>
__global__ __launch_bounds__(NW * 64, WPS) void lean_kernel(float* out, int iters) {
  constexpr int CT = KS1 / 2;
  constexpr int CHB = (KS1 + 2 * CT) * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < (2 * CHB + KS1 * 1024 * NW) / 4; i += NW * 64)
    reinterpret_cast<unsigned*>(lds)[i] = 0x2c002c00u + (i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* xfb = lds + 2 * CHB + wave * KS1 * 1024 + lane * 16;
  f32x16 x[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[ct][r] = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const unsigned char* base = lds + (it & 1) * CHB + lane * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.01f * r;
#pragma unroll
    for (int g = 0; g < KS1 / 4; ++g) {
      bf16x8 a1[4], xf[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a1[k] = *reinterpret_cast<const bf16x8*>(base + (g * 4 + k) * 1024);
        xf[k] = *reinterpret_cast<const bf16x8*>(xfb + (g * 4 + k) * 1024);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (PK)
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[k]), __builtin_bit_cast(f16x8, xf[k]), hacc, 0, 0, 0);
        else
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[k], xf[k], hacc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 a2[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) a2[ct] = *reinterpret_cast<const bf16x8*>(base + (KS1 + ct * 2 + s2) * 1024);
      if (PK) {
        f16x8 hf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f16x2 g2 = gelu_pk(hacc[8 * s2 + 2 * r], hacc[8 * s2 + 2 * r + 1]);
          hf[2 * r] = g2[0];
          hf[2 * r + 1] = g2[1];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          x[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2[ct]), hf, x[ct], 0, 0, 0);
      } else {
        bf16x8 hf;
#pragma unroll
        for (int r = 0; r < 8; ++r) hf[r] = (__bf16)gelu3(hacc[8 * s2 + r]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          x[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ct], hf, x[ct], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[ct][r];
  out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = s;
}

// total_cb wave-tasks of `iters` chunks each
template <int NW, int WPS, int KS1, bool PK> void run_lean(const char* tag, float* out, int iters, int total_cb) {
  auto kern = lean_kernel<NW, WPS, KS1, PK>;
  constexpr int CT = KS1 / 2;
  const int ldsb = 2 * (KS1 + 2 * CT) * 1024 + KS1 * 1024 * NW;
  const int grid = total_cb / NW;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), ldsb, 0, out, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), ldsb, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  int nb = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, ldsb);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  const double us = ms * 1000.0 / reps;
  const double mfma_us = (double)iters * (KS1 + 2 * CT) * 32.0 * (total_cb / 1024.0) / 2400.0;
  printf("%-52s regs %3d  wg/CU %d  %8.1f us/launch  (MFMA floor %6.1f us: %.2f of peak)\n", tag, fa.numRegs, nb, us, mfma_us,
         mfma_us / us);
}

template <int NCB, int NW, int WPS, int KS1, bool BAR> void run(const char* tag, float* out, int iters) {
  auto kern = mlp_kernel<NCB, NW, WPS, KS1, BAR>;
  constexpr int CT = KS1 / 2;
  const int ldsb = 2 * (KS1 + 2 * CT) * 1024;
  const int total_cb = 4096;
  const int grid = total_cb / (NCB * NW);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), ldsb, 0, out, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), ldsb, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  int nb = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * 64, ldsb);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  const double us = ms * 1000.0 / reps;
  // MFMA floor: per column block and chunk KS1 + 2 CT products of 32 cycles; 4096 column blocks over 1024 SIMDs
  const double mfma_us = (double)iters * (KS1 + 2 * CT) * 32.0 * (total_cb / 1024.0) / 2400.0;
  printf("%-44s regs %3d  wg/CU %d  %8.1f us/launch  (MFMA floor %6.1f us: %.2f of peak)\n", tag, fa.numRegs, nb, us, mfma_us,
         mfma_us / us);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 256;
  float* out;
  hipMalloc(&out, 4096 * 64 * sizeof(float) * 2);
  printf("stage-0 shape (C = 64, 16 products per column block and chunk), %d chunks per column block\n", iters);
  run<2, 4, 2, 4, true>("2 cb/wave, 4 waves/WG, 2 waves/SIMD, barrier", out, iters);
  run<2, 4, 2, 4, false>("2 cb/wave, 4 waves/WG, 2 waves/SIMD", out, iters);
  run<1, 8, 4, 4, true>("1 cb/wave, 8 waves/WG, 4 waves/SIMD, barrier", out, iters);
  run<1, 8, 4, 4, false>("1 cb/wave, 8 waves/WG, 4 waves/SIMD", out, iters);
  run<1, 4, 4, 4, true>("1 cb/wave, 4 waves/WG, 4 waves/SIMD, barrier", out, iters);
  run<1, 4, 3, 4, true>("1 cb/wave, 4 waves/WG, 3 waves/SIMD, barrier", out, iters);
  run<1, 4, 2, 4, true>("1 cb/wave, 4 waves/WG, 2 waves/SIMD, barrier", out, iters);
  run<2, 4, 1, 4, true>("2 cb/wave, 4 waves/WG, 1 wave/SIMD, barrier", out, iters);
  printf("stage-1 shape (C = 128, 16 products per column block and chunk)\n");
  run<1, 4, 2, 8, true>("1 cb/wave, 4 waves/WG, 2 waves/SIMD, barrier", out, iters);
  run<1, 4, 2, 8, false>("1 cb/wave, 4 waves/WG, 2 waves/SIMD", out, iters);
  run<1, 4, 3, 8, true>("1 cb/wave, 4 waves/WG, 3 waves/SIMD, barrier", out, iters);
  run<1, 4, 1, 8, true>("1 cb/wave, 4 waves/WG, 1 wave/SIMD, barrier", out, iters);
  run<2, 4, 1, 8, true>("2 cb/wave, 4 waves/WG, 1 wave/SIMD, barrier", out, iters);
  printf("lean loop (B operand of fc1 re-read per chunk, fragments in groups of four), same total work\n");
  run_lean<4, 2, 4, false>("C=64  bf16, 4 waves/WG, 2/SIMD", out, iters, 4096);
  run_lean<8, 4, 4, false>("C=64  bf16, 8 waves/WG, 4/SIMD", out, iters, 4096);
  run_lean<8, 4, 4, true>("C=64  f16 packed GELU, 8 waves/WG, 4/SIMD", out, iters, 4096);
  run_lean<4, 2, 4, true>("C=64  f16 packed GELU, 4 waves/WG, 2/SIMD", out, iters, 4096);
  run_lean<4, 2, 8, false>("C=128 bf16, 4 waves/WG, 2/SIMD", out, iters, 4096);
  run_lean<4, 3, 8, false>("C=128 bf16, 4 waves/WG, 3/SIMD", out, iters, 4096);
  run_lean<8, 4, 8, false>("C=128 bf16, 8 waves/WG, 4/SIMD, half the chunks each", out, iters / 2, 8192);
  run_lean<4, 4, 8, false>("C=128 bf16, 4 waves/WG, 4/SIMD, half the chunks each", out, iters / 2, 8192);
  run_lean<8, 4, 8, true>("C=128 f16 packed GELU, 8 waves/WG, 4/SIMD, half each", out, iters / 2, 8192);
  run_lean<4, 2, 8, true>("C=128 f16 packed GELU, 4 waves/WG, 2/SIMD", out, iters, 4096);
  return 0;
}
