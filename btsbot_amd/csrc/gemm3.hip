// LDS-free streaming GEMM for short reductions (gfx950, 16-bit modes):
//
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ),   K in {64, 128},  N % 32 == 0
//
// The MaxViT 1x1 convolutions and linears of the wide stages (conv1, qkv, proj, shortcut at C = 64 / 128,
// timm MbConvBlock / AttentionCl reached from /root/reference/btsbot/architectures.py:51,97) have millions
// of rows and one to four k-steps: they are streaming kernels, and the LDS-tiled GEMMs spend their time
// in the load -> barrier -> MFMA -> stage -> store chain of a single k-tile.  Here nothing is staged:
//   * a wave keeps its slice of the filter (up to 16 x 16 output channels) in registers as MFMA A
//     fragments, loaded once;
//   * the B fragment of a 16-row tile is one 16-byte piece per lane straight from the activation rows;
//     the next tile's pieces are requested before the current tile's MFMAs (register double buffer);
//   * v_mfma_f32_16x16x32 "transposed" (filter rows = A): a lane ends with output channels of ONE row, and
//     the filter rows are assigned to MFMA rows so that a lane's two tiles of a pair hold 8 CONSECUTIVE
//     channels (tile 2j row 4g+r = channel 32j+8g+r, tile 2j+1 = 32j+8g+4+r): 16-byte stores.
// No LDS, no barrier; waves are independent.  Epilogues as in gemm.hip: SILU / BIAS_T / GELU (typed out),
// RESID (f32 out = resid + gamma (acc + bias)), BIAS (f32 out).
#include "common.h"

namespace {

template <typename T> struct M3;
template <> struct M3<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct M3<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// NT = 16-channel tiles per wave slice (even), KS = K / 32
template <typename T, int KS, int NT, int EPI>
__global__ __launch_bounds__(256) void gemm3_kernel(const T* __restrict__ X, const T* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ gamma, const float* resid,
                                                    void* out, int M, int N, int tiles_per_wave) {
  using frag = typename M3<T>::frag;
  constexpr int K = KS * 32;
  constexpr bool TOUT = EPI == EPI_SILU || EPI == EPI_BIAS_T || EPI == EPI_GELU;
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * NT * 16;                         // this workgroup's channel slice
  const int nt_act = min(NT, (N - n0) / 16);                   // active tiles (even: N % 32 == 0)
  // filter fragments: MFMA row l15 = 4g'+r' of tile t <-> channel n0 + 32(t/2) + 8g' + 4(t&1) + r'
  frag wf[NT][KS];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = min(n0 + 32 * (t >> 1) + 8 * (l15 >> 2) + 4 * (t & 1) + (l15 & 3), N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[t][ks] = *reinterpret_cast<const frag*>(W + (size_t)ch * K + ks * 32 + g * 8);
  }
  const int ntile = (M + 15) / 16;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  int t0 = wid * tiles_per_wave;
  const int t1 = min(t0 + tiles_per_wave, ntile);
  if (t0 >= t1) return;
  frag xf[KS], xn[KS];
  auto fetch = [&](int tile, frag* dst) {
    const long row = min((long)tile * 16 + l15, (long)M - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      dst[ks] = *reinterpret_cast<const frag*>(X + row * K + ks * 32 + g * 8);
  };
  fetch(t0, xf);
  for (int tile = t0; tile < t1; ++tile) {
    if (tile + 1 < t1) fetch(tile + 1, xn);
    const long m = (long)tile * 16 + l15;
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
      if (2 * j >= nt_act) break;                              // wave-uniform
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0 = M3<T>::run(wf[2 * j][ks], xf[ks], a0);
        a1 = M3<T>::run(wf[2 * j + 1][ks], xf[ks], a1);
      }
      if (m >= M) continue;
      const int n = n0 + 32 * j + 8 * g;                       // the lane's 8 consecutive channels
      float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      if (EPI != EPI_PLAIN) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + n);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      const size_t o = (size_t)m * N + n;
      if (TOUT) {
        typedef T __attribute__((ext_vector_type(8))) T8;
        T8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          r[e] = (T)(EPI == EPI_SILU ? silu_fast(v[e]) : EPI == EPI_GELU ? gelu_fast(v[e]) : v[e]);
        *reinterpret_cast<T8*>(reinterpret_cast<T*>(out) + o) = r;
      } else {
        if (EPI == EPI_RESID) {
          const float4 g0 = *reinterpret_cast<const float4*>(gamma + n);
          const float4 g1 = *reinterpret_cast<const float4*>(gamma + n + 4);
          const float4 r0 = *reinterpret_cast<const float4*>(resid + o);
          const float4 r1 = *reinterpret_cast<const float4*>(resid + o + 4);
          v[0] = r0.x + g0.x * v[0]; v[1] = r0.y + g0.y * v[1];
          v[2] = r0.z + g0.z * v[2]; v[3] = r0.w + g0.w * v[3];
          v[4] = r1.x + g1.x * v[4]; v[5] = r1.y + g1.y * v[5];
          v[6] = r1.z + g1.z * v[6]; v[7] = r1.w + g1.w * v[7];
        }
        float* op = reinterpret_cast<float*>(out) + o;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = xn[ks];
  }
}

template <typename T, int KS, int NT, int EPI>
int launch3(const void* X, const void* W, const float* bias, const float* gamma, const float* resid,
            void* out, int M, int N, hipStream_t st) {
  const int ntile = (M + 15) / 16;
  // a few thousand waves: enough to fill 256 CUs x 8 wave slots several times, few enough that the
  // filter slice (up to 32 KB per wave, from L2) is amortised over >= 8 row tiles
  int tpw = ntile / 8192;
  tpw = tpw < 8 ? 8 : (tpw > 64 ? 64 : tpw);
  const int waves = (ntile + tpw - 1) / tpw;
  const dim3 grid((waves + 3) / 4, (N + NT * 16 - 1) / (NT * 16));
  hipLaunchKernelGGL((gemm3_kernel<T, KS, NT, EPI>), grid, dim3(256), 0, st,
                     reinterpret_cast<const T*>(X), reinterpret_cast<const T*>(W), bias, gamma, resid,
                     out, M, N, tpw);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T, int EPI>
int launch3_k(const void* X, const void* W, const float* bias, const float* gamma, const float* resid,
              void* out, int M, int N, int K, hipStream_t st) {
  if (K == 64) return launch3<T, 2, 16, EPI>(X, W, bias, gamma, resid, out, M, N, st);
  return launch3<T, 4, 8, EPI>(X, W, bias, gamma, resid, out, M, N, st);
}

template <typename T>
int launch3_epi(int epi, const void* X, const void* W, const float* bias, const float* gamma,
                const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  switch (epi) {
    case EPI_SILU: return launch3_k<T, EPI_SILU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS_T: return launch3_k<T, EPI_BIAS_T>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_GELU: return launch3_k<T, EPI_GELU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_RESID: return launch3_k<T, EPI_RESID>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS: return launch3_k<T, EPI_BIAS>(X, W, bias, gamma, resid, out, M, N, K, st);
  }
  btsbot_set_error("launch_gemm3: bad epilogue %d", epi);
  return BTSBOT_ERR_INVALID_ARG;
}

}  // namespace

bool gemm3_supported(int prec, int epi, int M, int N, int K) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (K == 64 || K == 128) && N % 32 == 0 && N >= 32 &&
         M >= 1 &&
         (epi == EPI_SILU || epi == EPI_BIAS_T || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_BIAS);
}

int launch_gemm3(int prec, int epi, const void* X, const void* W, const float* bias, const float* gamma,
                 const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  if (!gemm3_supported(prec, epi, M, N, K)) {
    btsbot_set_error("launch_gemm3: unsupported (prec %d, epi %d, M %d, N %d, K %d)", prec, epi, M, N, K);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16)
    return launch3_epi<bf16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
  return launch3_epi<f16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
}
