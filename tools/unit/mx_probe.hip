// Probe of the block-scaled fp8 MFMAs on gfx950: operand lane maps and the scale operand (exact small-integer data).
//   hipcc --offload-arch=gfx950 -O2 tools/unit/mx_probe.hip -o tools/unit/mx_probe && tools/unit/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned char to_fp8(float v) { return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff); }

// A [M][K], B [K][N] as floats (small integers); lane maps under test:
//   16x16x128: lane l: A row l & 15, B col l & 15, k = 32 (l >> 4) + j;   32x32x64: row/col l & 31, k = 32 (l >> 5) + j
__global__ void probe16(const float* A, const float* B, float* D, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; ++j) {
    ab[j] = to_fp8(A[(l & 15) * 128 + 32 * (l >> 4) + j]);
    bb[j] = to_fp8(B[(32 * (l >> 4) + j) * 16 + (l & 15)]);
  }
  v8i av, bv;
  for (int w = 0; w < 8; ++w) {
    av[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    bv[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];   // row = 4 (l >> 4) + r, col = l & 15
}
__global__ void probe32(const float* A, const float* B, float* D, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; ++j) {
    ab[j] = to_fp8(A[(l & 31) * 64 + 32 * (l >> 5) + j]);
    bb[j] = to_fp8(B[(32 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  v8i av, bv;
  for (int w = 0; w < 8; ++w) {
    av[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    bv[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

template <int M, int K> int check(const char* name, void (*kern)(const float*, const float*, float*, const int*, const int*),
                                  int scale_lane_a, int scale_lane_b) {
  const int N = M;
  float *A, *B, *D;
  int *sa, *sb;
  hipMallocManaged(&A, M * K * 4); hipMallocManaged(&B, K * N * 4); hipMallocManaged(&D, M * N * 4);
  hipMallocManaged(&sa, 64 * 4); hipMallocManaged(&sb, 64 * 4);
  srand(1);
  for (int i = 0; i < M * K; ++i) A[i] = (float)(rand() % 5 - 2);
  for (int i = 0; i < K * N; ++i) B[i] = (float)(rand() % 7 - 3);
  for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }      // E8M0 127 = 2^0
  if (scale_lane_a >= 0) sa[scale_lane_a] = 128;                    // 2^1 in the low byte of ONE lane's scale register
  if (scale_lane_b >= 0) sb[scale_lane_b] = 129;                    // 2^2
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, A, B, D, sa, sb);
  hipDeviceSynchronize();
  // hypothesis: lane l's scale applies to ITS 32-value block: A row (l % M), k-block (l / M); B col (l % M), k-block (l / M)
  int bad = 0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double ref = 0;
      for (int kb = 0; kb < K / 32; ++kb) {
        double s = 0;
        for (int k = 32 * kb; k < 32 * kb + 32; ++k) s += A[i * K + k] * B[k * N + j];
        double f = 1;
        if (scale_lane_a >= 0 && scale_lane_a % M == i && scale_lane_a / M == kb) f *= 2;
        if (scale_lane_b >= 0 && scale_lane_b % M == j && scale_lane_b / M == kb) f *= 4;
        ref += f * s;
      }
      if (fabs(ref - D[i * N + j]) > 1e-3) {
        if (bad < 5) printf("  %s: D[%d][%d] = %g, expected %g\n", name, i, j, D[i * N + j], ref);
        ++bad;
      }
    }
  printf("%s scale lanes (%d, %d): %s (%d mismatches)\n", name, scale_lane_a, scale_lane_b, bad ? "MISMATCH" : "ok", bad);
  return bad;
}

int main() {
  int bad = 0;
  bad += check<16, 128>("16x16x128", probe16, -1, -1);
  bad += check<16, 128>("16x16x128", probe16, 21, -1);
  bad += check<16, 128>("16x16x128", probe16, -1, 37);
  bad += check<16, 128>("16x16x128", probe16, 5, 60);
  bad += check<32, 64>("32x32x64", probe32, -1, -1);
  bad += check<32, 64>("32x32x64", probe32, 40, -1);
  bad += check<32, 64>("32x32x64", probe32, -1, 9);
  bad += check<32, 64>("32x32x64", probe32, 33, 63);
  return bad != 0;
}
