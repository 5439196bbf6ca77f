// Stem of the per-op schedule on the matrix pipe (gfx950, 16-bit modes): Conv2d(3, C0, k4, s4) + bias + LayerNorm(C0),
// timm ConvNeXt `stem.0` / `stem.1` reached from /root/reference/btsbot/architectures.py:108,132 -- the same computation
// as convnext.hip's stem_kernel (fp32 FMAs, one output pixel per lane: 3840 FMAs per lane for C0 = 80, 75-86 us per 1024
// alerts) with the 48-tap products as three 32x32x16 MFMA k-steps per 32-pixel column block, the way stage0b.hip's stem
// phase does it.  Who runs it: convnext_nano (its widths do not fit the stage-0 megakernel), the training forward and the
// NO_STAGE0 schedule, in the bf16 / f16 modes (the fp32 mode keeps stem_kernel).
//   * one alert per 256-thread workgroup; wave w owns pixels 64 w .. 64 w + 63 (two column blocks), 225 of 256 live;
//   * A = the fp32 filter [C0][48] converted in registers (k = ci * 16 + ky * 4 + kx): nothing packed is read, so the
//     training step's operand re-pack keeps overlapping the stem; CT = ceil(C0 / 32) row tiles (80 -> 3, rows >= 80 zero);
//   * B = 4 x 4 input pixels per (lane, input channel): two unaligned 16-byte loads, converted in registers;
//   * LayerNorm over the lane's 16 CT values + its partner lane's (lane ^ 32), fp32; NHWC fp32 rows out
//     (+ optionally the pre-LayerNorm rows the training backward keeps).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int HW = 15, P = 225;
constexpr float LN_EPS = 1e-6f;
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

template <typename T> struct SM;
template <> struct SM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct SM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T, int C0>
__global__ __launch_bounds__(256) void stem16_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, float* __restrict__ out,
                                                     float* __restrict__ pre_out, int B) {
  using frag = typename SM<T>::frag;
  constexpr int CT = (C0 + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int alert = blockIdx.x;
  const float* src = img + (size_t)alert * 3 * 63 * 63;
  // filter fragments: row = output channel ct * 32 + lr, k = ci * 16 + 8 h + e
  frag af[3][CT];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int row = ct * 32 + lr;
      const bool ok = row < C0;
      const float* wr = w + (size_t)(ok ? row : 0) * 48 + ci * 16 + h * 8;
      const float4 w0 = *reinterpret_cast<const float4*>(wr);
      const float4 w1 = *reinterpret_cast<const float4*>(wr + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) af[ci][ct][e] = (T)(ok ? wv[e] : 0.f);
    }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pix = wave * 64 + t * 32 + lr;
    const bool live = pix < P;
    const int pc = live ? pix : 0;
    const int py = pc / HW, px = pc - py * HW;
    f32x16 x[CT];
    // accumulator row (r & 3) + 8 (r >> 2) + 4 h of tile ct = channel ct * 32 + that row
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = ct * 32 + 8 * qd + 4 * h;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C0) bv = *reinterpret_cast<const float4*>(bias + c);
        x[ct][4 * qd + 0] = bv.x;
        x[ct][4 * qd + 1] = bv.y;
        x[ct][4 * qd + 2] = bv.z;
        x[ct][4 * qd + 3] = bv.w;
      }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float* r0 = src + (ci * 63 + 4 * py + 2 * h) * 63 + 4 * px;
      const f4u v0 = *reinterpret_cast<const f4u*>(r0);
      const f4u v1 = *reinterpret_cast<const f4u*>(r0 + 63);
      frag bf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf[e] = (T)v0.v[e];
        bf[4 + e] = (T)v1.v[e];
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) x[ct] = SM<T>::run(af[ci][ct], bf, x[ct]);
    }
    float* po = pre_out != nullptr ? pre_out + ((size_t)alert * P + pc) * C0 : nullptr;
    if (po != nullptr && live) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = ct * 32 + 8 * qd + 4 * h;
          if (c < C0)
            *reinterpret_cast<float4*>(po + c) =
                make_float4(x[ct][4 * qd], x[ct][4 * qd + 1], x[ct][4 * qd + 2], x[ct][4 * qd + 3]);
        }
    }
    // ---- LayerNorm over the pixel's C0 channels (this lane's + lane ^ 32's), two-pass
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        if (ct * 32 + 8 * qd + 4 * h < C0)
          s += x[ct][4 * qd] + x[ct][4 * qd + 1] + x[ct][4 * qd + 2] + x[ct][4 * qd + 3];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C0);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        if (ct * 32 + 8 * qd + 4 * h < C0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = x[ct][4 * qd + e] - mean;
            q += d * d;
          }
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / C0) + LN_EPS);
    if (live) {
      float* o = out + ((size_t)alert * P + pix) * C0;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = ct * 32 + 8 * qd + 4 * h;
          if (c < C0) {
            const float4 wv = *reinterpret_cast<const float4*>(lnw + c);
            const float4 bv = *reinterpret_cast<const float4*>(lnb + c);
            *reinterpret_cast<float4*>(o + c) =
                make_float4((x[ct][4 * qd + 0] - mean) * rstd * wv.x + bv.x, (x[ct][4 * qd + 1] - mean) * rstd * wv.y + bv.y,
                            (x[ct][4 * qd + 2] - mean) * rstd * wv.z + bv.z, (x[ct][4 * qd + 3] - mean) * rstd * wv.w + bv.w);
          }
        }
    }
  }
}

template <typename T>
int launch_stem16_t(const float* img, const float* w, const float* bias, const float* lnw, const float* lnb, float* out,
                    int B, int C0, hipStream_t st, float* pre_out) {
  if (C0 == 64)
    hipLaunchKernelGGL((stem16_kernel<T, 64>), dim3(B), dim3(256), 0, st, img, w, bias, lnw, lnb, out, pre_out, B);
  else
    hipLaunchKernelGGL((stem16_kernel<T, 80>), dim3(B), dim3(256), 0, st, img, w, bias, lnw, lnb, out, pre_out, B);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool stem16_supported(int prec, int C0) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (C0 == 64 || C0 == 80);
}

// w: the fp32 filter [C0][48] (k = ci * 16 + ky * 4 + kx: the master layout [C0][3][4][4])
int launch_stem16(int prec, const float* img, const float* w, const float* bias, const float* lnw, const float* lnb,
                  float* out, int B, int C0, hipStream_t st, float* pre_out) {
  if (B <= 0) return BTSBOT_OK;
  if (!stem16_supported(prec, C0)) {
    btsbot_set_error("stem16: unsupported (prec %d, C0 %d)", prec, C0);
    return BTSBOT_ERR_INVALID_ARG;
  }
  return prec == BTSBOT_BF16 ? launch_stem16_t<bf16_t>(img, w, bias, lnw, lnb, out, B, C0, st, pre_out)
                             : launch_stem16_t<f16_t>(img, w, bias, lnw, lnb, out, B, C0, st, pre_out);
}
