// Issue-rate probes (gfx950) behind DESIGN.md's stage-0 re-cut: packed 16-bit VALU, dot2, conversions,
// aligned / misaligned ds_read_b128, small MFMA shapes.  Cycles per wave-instruction at 1 and 2 waves per
// SIMD, every CU busy.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

enum { M_FMA32 = 0, M_PKFMA16, M_PKMUL16, M_PKMAX16, M_PKADD16, M_CVTPKRTZ, M_CVTPKBF16, M_CVTF32F16,
       M_DOT2F16, M_DOT2BF16, M_FMAMIX, M_FMAMIXLO, M_EXP32, M_RCP32, M_EXP16, M_RCP16, M_MED3, M_PERM,
       M_LDS128A, M_LDS128U2, M_LDS128U4, M_LDS128U8, M_LDSU16, M_LDSW16, M_LDS64A, M_LDS64U2,
       M_MFMA16, M_MFMA16_LDS, M_MFMA16_LDSU, M_MFMA4, M_MFMA32, M_MFMA16_VALU4, M_N };
const char* const NAMES[M_N] = {
    "v_fma_f32", "v_pk_fma_f16", "v_pk_mul_f16", "v_pk_max_f16", "v_pk_add_f16", "v_cvt_pkrtz_f16_f32",
    "v_cvt_pk_bf16_f32", "v_cvt_f32_f16", "v_dot2_f32_f16", "v_dot2_f32_bf16", "v_fma_mix_f32",
    "v_fma_mixlo_f16", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_med3_f32", "v_perm_b32",
    "ds_read_b128 aligned", "ds_read_b128 +2B", "ds_read_b128 +4B", "ds_read_b128 +8B", "ds_read_u16_d16",
    "ds_write_b16", "ds_read_b64 aligned", "ds_read_b64 +2B",
    "mfma16x16x32bf16", "mfma16 + 2 ds_read_b128 aligned", "mfma16 + 1 aligned 1 (+2B) read",
    "mfma4x4x4f16 (16 blocks)", "mfma32x32x16bf16", "mfma16 + 4 v_pk_fma_f16"};

template <int MODE> __global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned u0 = 0x3c003800u + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  const unsigned um = 0x3bff3bfeu;
  const float fm = 0.999f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane LDS byte address: 16-B pieces, conflict-free pattern for b128 (lane-linear), + misalignment
  unsigned la = wave * 2048 + lane * 16;
  if (MODE == M_LDS128U2 || MODE == M_LDS64U2) la += 2;
  if (MODE == M_LDS128U4) la += 4;
  if (MODE == M_LDS128U8) la += 8;
  if (MODE == M_LDS64A || MODE == M_LDS64U2) la = wave * 2048 + lane * 8 + (MODE == M_LDS64U2 ? 2 : 0);
  unsigned lb = la + 1024 + (MODE == M_MFMA16_LDSU ? 2 : 0);
  f32x4 l0 = {0, 0, 0, 0}, l1 = l0, l2 = l0, l3 = l0;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 d0 = {0}, d1 = {0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (lane + i)); fb[i] = (__bf16)(0.02f * (lane - i)); }
  f16x4 ha = {(_Float16)0.1f, (_Float16)0.2f, (_Float16)0.3f, (_Float16)0.4f}, hb = ha;

  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == M_FMA32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(fm));
        REP8(X)
#undef X
      } else if (MODE == M_PKFMA16) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(u##i) : "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_PKMUL16) {
#define X(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u##i) : "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_PKMAX16) {
#define X(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u##i) : "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_PKADD16) {
#define X(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u##i) : "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_CVTPKRTZ) {
#define X(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u##i) : "v"(a##i), "v"(fm));
        REP8(X)
#undef X
      } else if (MODE == M_CVTPKBF16) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u##i) : "v"(a##i), "v"(fm));
        REP8(X)
#undef X
      } else if (MODE == M_CVTF32F16) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a##i) : "v"(u##i));
        REP8(X)
#undef X
      } else if (MODE == M_DOT2F16) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a##i) : "v"(u##i), "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_DOT2BF16) {
#define X(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a##i) : "v"(u##i), "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_FMAMIX) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a##i) : "v"(u##i), "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_FMAMIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %1 op_sel_hi:[0,1,0]" : "+v"(u##i) : "v"(a##i), "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_EXP32) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (MODE == M_RCP32) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (MODE == M_EXP16) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(u##i));
        REP8(X)
#undef X
      } else if (MODE == M_RCP16) {
#define X(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(u##i));
        REP8(X)
#undef X
      } else if (MODE == M_MED3) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(fm));
        REP8(X)
#undef X
      } else if (MODE == M_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u##i) : "v"(um));
        REP8(X)
#undef X
      } else if (MODE == M_LDS128A || MODE == M_LDS128U2 || MODE == M_LDS128U4 || MODE == M_LDS128U8) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"
                     "ds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288\n\t"
                     "ds_read_b128 %0, %4 offset:16384\n\tds_read_b128 %1, %4 offset:20480\n\t"
                     "ds_read_b128 %2, %4 offset:24576\n\tds_read_b128 %3, %4 offset:28672\n\t"
                     "s_waitcnt lgkmcnt(4)"
                     : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(la) : "memory");
      } else if (MODE == M_LDS64A || MODE == M_LDS64U2) {
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:4096\n\t"
                     "ds_read_b64 %2, %4 offset:8192\n\tds_read_b64 %3, %4 offset:12288\n\t"
                     "ds_read_b64 %0, %4 offset:16384\n\tds_read_b64 %1, %4 offset:20480\n\t"
                     "ds_read_b64 %2, %4 offset:24576\n\tds_read_b64 %3, %4 offset:28672\n\t"
                     "s_waitcnt lgkmcnt(4)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(la) : "memory");
      } else if (MODE == M_LDSU16) {
#define X(i) asm volatile("ds_read_u16_d16 %0, %1 offset:" #i "*130" : "+v"(u##i) : "v"(la) : "memory");
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if (MODE == M_LDSW16) {
#define X(i) asm volatile("ds_write_b16 %1, %0 offset:" #i "*130" : : "v"(u##i), "v"(la) : "memory");
        REP8(X)
#undef X
      } else if (MODE == M_MFMA16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c3, 0, 0, 0);
        }
      } else if (MODE == M_MFMA16_LDS || MODE == M_MFMA16_LDSU) {
#define X(q)                                                                                                    \
  {                                                                                                             \
    f32x4 ra, rb;                                                                                               \
    asm volatile("ds_read_b128 %0, %2 offset:" #q "*4096\n\tds_read_b128 %1, %3 offset:" #q "*4096\n\ts_waitcnt lgkmcnt(0)" \
                 : "=&v"(ra), "=&v"(rb) : "v"(la), "v"(lb) : "memory");                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra), __builtin_bit_cast(bf16x8, rb), c0, 0, 0, 0); \
  }
        REP8(X)
#undef X
      } else if (MODE == M_MFMA4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, c3, 0, 0, 0);
        }
      } else if (MODE == M_MFMA32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, d1, 0, 0, 0);
        }
      } else if (MODE == M_MFMA16_VALU4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u0), "+v"(u1) : "v"(um));
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u2), "+v"(u3) : "v"(um));
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0);
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u4), "+v"(u5) : "v"(um));
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u6), "+v"(u7) : "v"(um));
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c2, 0, 0, 0);
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u0), "+v"(u1) : "v"(um));
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u2), "+v"(u3) : "v"(um));
          c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c3, 0, 0, 0);
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u4), "+v"(u5) : "v"(um));
          asm volatile("v_pk_fma_f16 %0, %0, %2, %2\n\tv_pk_fma_f16 %1, %1, %2, %2" : "+v"(u6), "+v"(u7) : "v"(um));
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  s += (float)(q0 + q1 + q2 + q3) + l0.x + l1.y + l2.z + l3.w + c0.x + c1.y + c2.z + c3.w + d0[0] + d1[5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

template <int MODE> void run(float* out, long long* cyc, int wps, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 65536, 0, out, cyc, iters);
}
template <int MODE> void all(float* out, long long* cyc, int lo, int hi) {
  if constexpr (MODE < M_N) {
    const int iters = 200;
    for (int wps = 1; wps <= 2 && MODE >= lo && MODE <= hi; ++wps) {
      printf("%-36s %d waves/SIMD: ", NAMES[MODE], wps);
      fflush(stdout);
      run<MODE>(out, cyc, wps, iters);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); fflush(stdout); return; }
      long long h[32];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      long long tmin = h[0], tmax = h[1];
      for (int w = 0; w < 4 * wps; ++w) { if (h[2 * w] < tmin) tmin = h[2 * w]; if (h[2 * w + 1] > tmax) tmax = h[2 * w + 1]; }
      const double n = (double)iters * 8 * 8;
      printf("wave 0 %.2f cyc per own instr; SIMD %.2f cyc per instr\n", (double)(h[1] - h[0]) / n,
             (double)(tmax - tmin) / n / wps);
      fflush(stdout);
    }
    all<MODE + 1>(out, cyc, lo, hi);
  }
}
int main(int argc, char** argv) {
  const int lo = argc > 1 ? atoi(argv[1]) : 0, hi = argc > 2 ? atoi(argv[2]) : M_N - 1;
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096);
  all<0>(out, cyc, lo, hi);
  hipError_t e = hipDeviceSynchronize();
  printf("done: %s\n", hipGetErrorString(e));
  return 0;
}
