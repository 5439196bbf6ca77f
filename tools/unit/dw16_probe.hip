// Depthwise 7x7 of a 15x15x64 map on the matrix pipe, two formulations, in isolation (gfx950):
//   A  the shipped one (stage0b.hip): 16-block 4x4x4 MFMAs, Toeplitz taps of 16 channels in 21 register fragments,
//      19 row steps x 4 ds_read_b64, 280 products per wave and block;
//   B  one channel per 16x16x16 MFMA: A = the channel's 16x16 Toeplitz matrix of tap row ky (a packed 512-byte fragment
//      per (channel, ky), streamed from L2: 229 KB per block), B = 4 consecutive x of row y + ky - 3 (one ds_read_b64 from
//      the same planar image), 7 products per channel, 112 per wave and block; channel sums stay in-lane.
// Same launch shape as stage0b (256 threads, 2 workgroups per CU); prints cycles per (wave, block).
//   hipcc -O3 --offload-arch=gfx950 -o dw16_probe dw16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int PL_XQ = 22 * 8, PL_CH = 800, PLB = 64 * PL_CH;

__global__ __launch_bounds__(256, 2) void dw_a(const uint2* __restrict__ taps, float* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pl[];
  for (int i = threadIdx.x; i < PLB / 4; i += 256) reinterpret_cast<unsigned*>(pl)[i] = 0x3c003c00u + (i & 31);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dj = lane & 3, dch = wave * 16 + (lane >> 2);
  s16x4 tw[21];
  for (int r = 0; r < 21; ++r) tw[r] = __builtin_bit_cast(s16x4, taps[r * 256 + wave * 64 + lane]);
  const unsigned char* lb = pl + dch * PL_CH + dj * 8;
  float tot = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    f32x4 acc[4][4];
#pragma unroll
    for (int yb = 0; yb < 4; ++yb)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) acc[yb][xb] = f32x4{0.1f, 0.1f, 0.1f, 0.1f};
#pragma unroll
    for (int s = 0; s < 19; ++s) {
      s16x4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(lb + q * PL_XQ + s * 8));
#pragma unroll
      for (int yb = 0; yb < 4; ++yb) {
        const int ky = s - 4 * yb;
        if (ky < 0 || ky > 6) continue;
#pragma unroll
        for (int rbi = 0; rbi < 3; ++rbi)
#pragma unroll
          for (int xb = 0; xb < 4; ++xb) {
            const int q = xb + rbi - 1;
            if (q < 0 || q > 3) continue;
            acc[yb][xb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(tw[ky * 3 + rbi], bq[q], acc[yb][xb], 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int yb = 0; yb < 4; ++yb)
#pragma unroll
      for (int xb = 0; xb < 4; ++xb) tot += acc[yb][xb][0] + acc[yb][xb][3];
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = tot;
}

// DEPTH channels' fragments (7 each) requested ahead of their use
template <int DEPTH>
__global__ __launch_bounds__(256, 2) void dw_b(const uint2* __restrict__ frags, float* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pl[];
  for (int i = threadIdx.x; i < PLB / 4; i += 256) reinterpret_cast<unsigned*>(pl)[i] = 0x3c003c00u + (i & 31);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;                 // B / D column = output row y, k group = x quad
  const uint2* fsrc = frags + (size_t)(wave * 16) * 7 * 64 + lane;   // [channel][ky][lane]
  float tot = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    f32x4 acc[16];
    s16x4 ring[DEPTH][7];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) ring[d][ky] = __builtin_bit_cast(s16x4, fsrc[(d * 7 + ky) * 64]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned char* lb = pl + (wave * 16 + i) * PL_CH + g * PL_XQ + n * 8;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const s16x4 b = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(lb + ky * 8));
        a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ring[i % DEPTH][ky], b, a, 0, 0, 0);
      }
      acc[i] = a;
      if (i + DEPTH < 16) {
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) ring[i % DEPTH][ky] = __builtin_bit_cast(s16x4, fsrc[((i + DEPTH) * 7 + ky) * 64]);
      }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    tot += s[0] + s[1] + s[2] + s[3];
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = tot;
}

template <typename K> void run(const char* tag, K kern, const uint2* p, float* out, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PLB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(512), dim3(256), PLB, 0, p, out, reps);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(512), dim3(256), PLB, 0, p, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %8.1f us per launch = %7.2f us per block (512 workgroups, 2 per CU, %d blocks each)\n", tag, ms * 200.0,
         ms * 200.0 / reps, reps);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 64;
  uint2* p;
  float* out;
  hipMalloc(&p, 64 * 7 * 64 * 8);
  hipMemset(p, 0x3c, 64 * 7 * 64 * 8);
  hipMalloc(&out, 512 * 256 * 4);
  run("A: 4x4x4 x 280, taps in registers", dw_a, p, out, reps);
  run("B: 16x16x16 x 112, fragments 2 channels ahead", dw_b<2>, p, out, reps);
  run("B: 16x16x16 x 112, fragments 4 channels ahead", dw_b<4>, p, out, reps);
  run("B: 16x16x16 x 112, fragments 6 channels ahead", dw_b<6>, p, out, reps);
  return 0;
}
