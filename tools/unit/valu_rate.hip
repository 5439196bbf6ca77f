// VALU issue-rate probe (gfx950): cycles per wave-instruction for v_fma_f32, v_pk_fma_f32,
// v_pk_mul_f32, v_exp_f32, v_rcp_f32, v_add_f32_dpp; 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE> __global__ void k(float* out, long long* cyc, int iters) {
  float a[8]; f32x2 p[8]; h2 hh[8]; _Float16 hs[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 0.5f}; hh[i] = h2{(_Float16)a[i], (_Float16)(a[i] * 0.5f)}; hs[i] = (_Float16)a[i]; }
  const h2 hm = {(_Float16)0.999f, (_Float16)0.998f};
  const float m = 0.999f; const f32x2 pm = {0.999f, 0.998f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = fmaf(a[i], m, 0.001f);
        if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pm, (f32x2)(0.001f));
        if (MODE == 2) p[i] = p[i] * pm;
        if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);
        if (MODE == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
        if (MODE == 5) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, true));
        if (MODE == 6) a[i] = __builtin_amdgcn_fmed3f(a[i], -5.f, 5.f);
        if (MODE == 7) hh[i] = __builtin_elementwise_fma(hh[i], hm, hm);
        if (MODE == 8) hs[i] = __builtin_amdgcn_rcph(hs[i]);
        if (MODE == 9) hs[i] = __builtin_exp2f16(hs[i]);
        if (MODE == 10) { auto t = __builtin_amdgcn_cvt_pkrtz(a[i], a[(i + 1) & 7]); hh[i] = h2{(_Float16)t[0], (_Float16)t[1]}; }
        if (MODE == 11) a[i] = (float)(__bf16)a[i];
      }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)hh[i][0] + (float)hh[i][1] + (float)hs[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp", "v_med3_f32", "v_pk_fma_f16", "v_rcp_f16", "v_exp_f16", "v_cvt_pkrtz_f16_f32", "cvt f32->bf16->f32"};
  const int iters = 200;
  for (int wps = 1; wps <= 2; wps *= 2)
    for (int mode = 0; mode < 12; ++mode) {
      dim3 g(256), b(256 * wps);
      switch (mode) {
        case 0: k<0><<<g, b>>>(out, cyc, iters); break; case 1: k<1><<<g, b>>>(out, cyc, iters); break;
        case 2: k<2><<<g, b>>>(out, cyc, iters); break; case 3: k<3><<<g, b>>>(out, cyc, iters); break;
        case 4: k<4><<<g, b>>>(out, cyc, iters); break; case 5: k<5><<<g, b>>>(out, cyc, iters); break;
        case 6: k<6><<<g, b>>>(out, cyc, iters); break; case 7: k<7><<<g, b>>>(out, cyc, iters); break;
        case 8: k<8><<<g, b>>>(out, cyc, iters); break; case 9: k<9><<<g, b>>>(out, cyc, iters); break;
        case 10: k<10><<<g, b>>>(out, cyc, iters); break; case 11: k<11><<<g, b>>>(out, cyc, iters); break;
      }
      long long h[64]; hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
      long long tmin = h[0], tmax = h[1], own = h[1] - h[0];
      for (int w = 0; w < 4 * wps; ++w) { if (h[2 * w] < tmin) tmin = h[2 * w]; if (h[2 * w + 1] > tmax) tmax = h[2 * w + 1]; }
      printf("%d waves/SIMD  %-14s wave 0: %.2f cycles per own instruction; all waves of the CU: %.2f cycles per instruction per SIMD\n",
             wps, names[mode], (double)own / (iters * REP), (double)(tmax - tmin) / (iters * REP) / wps);
    }
  return 0;
}
