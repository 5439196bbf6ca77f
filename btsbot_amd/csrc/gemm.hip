// MFMA GEMM with fused ConvNeXt epilogues (gfx950).
//
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] )
//
// X = activations (rows = pixels of NHWC maps), W = a 1x1 / 2x2-patch filter bank stored exactly as
// PyTorch stores it ([N][K], K contiguous), so both operands are read K-contiguous.  This is the
// pointwise-conv pair of every ConvNeXt block (timm ConvNeXtBlock.mlp.fc1 / .fc2 reached from
// /root/reference/btsbot/architectures.py:108,132) and the downsample conv.
//
// Tiling: 256 threads = 4 waves; workgroup tile TM x TN, K tile = 128 bytes.  Operands are staged
// global -> registers -> LDS (144-byte padded rows) with the next tile's global loads issued before
// the current tile's MFMAs (T14 split).  The MFMA runs "transposed": the filter rows are the A
// operand and the activation rows the B operand, so each lane ends up with 4 CONSECUTIVE output
// channels of one pixel (C/D layout: col = lane&15 -> pixel, row = 4*(lane>>4)+r -> channel) and
// the epilogue writes 8/16-byte vectors.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LDS_STRIDE = 144;  // bytes per staged row: 128 data + 16 pad

template <typename T> struct Mma;
template <> struct Mma<float> {
  using frag = float;
  static constexpr int KSTEP_BYTES = 16;  // 4 floats per v_mfma_f32_16x16x4_f32
  static constexpr int LANE_BYTES = 4;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  using frag = bf16x8;
  static constexpr int KSTEP_BYTES = 64;  // 32 bf16 per v_mfma_f32_16x16x32_bf16
  static constexpr int LANE_BYTES = 16;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  using frag = f16x8;
  static constexpr int KSTEP_BYTES = 64;
  static constexpr int LANE_BYTES = 16;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T, int TM, int TN, int WM, int WN, int EPI, bool GATED = false>
__global__ __launch_bounds__(256) void gemm_kernel(const T* __restrict__ X,
                                                   const T* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ gamma,
                                                   const float* resid, void* out, int M, int N,
                                                   int K, const float* __restrict__ gate = nullptr,
                                                   int rows_per_alert = 1) {
  using MM = Mma<T>;
  using frag = typename MM::frag;
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 128 / (int)sizeof(T);   // elements per K tile
  constexpr int WTM = TM / WM, WTN = TN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int XCH = TM * 8 / 256, WCH = TN * 8 / 256;
  constexpr int KSTEPS = 128 / MM::KSTEP_BYTES;
  static_assert(WM * WN == 4 && XCH >= 1 && WCH >= 1 && MI >= 1 && NI >= 1, "tile/wave layout");

  __shared__ __attribute__((aligned(16))) unsigned char smem[(TM + TN) * LDS_STRIDE];
  unsigned char* Xs = smem;
  unsigned char* Ws = smem + TM * LDS_STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;

  uint4 xr[XCH], wr[WCH];
  const float* grow[XCH];   // GATED: this thread's rows' gate vectors
  if (GATED) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int gm = min(m0 + ((tid + i * 256) >> 3), M - 1);
      grow[i] = gate + (size_t)(gm / rows_per_alert) * K;
    }
  }
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = c & 7;
      const int gm = m0 + row, gk = k0 + kc * EPC;
      xr[i] = (gm < M && gk < K) ? *reinterpret_cast<const uint4*>(X + (size_t)gm * K + gk)
                                 : make_uint4(0, 0, 0, 0);
      if (GATED && gk < K) {   // A operand times the per-alert squeeze-excite gate
        T* e = reinterpret_cast<T*>(&xr[i]);
#pragma unroll
        for (int q = 0; q < EPC; ++q) e[q] = (T)((float)e[q] * grow[i][gk + q]);
      }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = c & 7;
      const int gn = n0 + row, gk = k0 + kc * EPC;
      wr[i] = (gn < N && gk < K) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * K + gk)
                                 : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(Xs + row * LDS_STRIDE + kc * 16) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(Ws + row * LDS_STRIDE + kc * 16) = wr[i];
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  const int lrow = lane & 15, lk = (lane >> 4) * MM::LANE_BYTES;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      frag bfr[MI], afr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        bfr[mi] = *reinterpret_cast<const frag*>(Xs + (wm * WTM + mi * 16 + lrow) * LDS_STRIDE +
                                                 ks * MM::KSTEP_BYTES + lk);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        afr[ni] = *reinterpret_cast<const frag*>(Ws + (wn * WTN + ni * 16 + lrow) * LDS_STRIDE +
                                                 ks * MM::KSTEP_BYTES + lk);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MM::run(afr[ni], bfr[mi], acc[ni][mi]);
    }
    __syncthreads();
  }

  // epilogue: lane owns channels n..n+3 of pixel m
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * WTN + ni * 16 + (lane >> 4) * 4;
    if (n >= N) continue;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI != EPI_DGELU && EPI != EPI_PLAIN && !GATED) bv = *reinterpret_cast<const float4*>(bias + n);
    float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == EPI_RESID && !GATED) gv = *reinterpret_cast<const float4*>(gamma + n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * WTM + mi * 16 + lrow;
      if (m >= M) continue;
      const f32x4 a = acc[ni][mi];
      const size_t o = (size_t)m * N + n;
      if (EPI == EPI_GELU_SAVE) {
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 pre, v;
        pre[0] = (T)(a[0] + bv.x);
        pre[1] = (T)(a[1] + bv.y);
        pre[2] = (T)(a[2] + bv.z);
        pre[3] = (T)(a[3] + bv.w);
        // the saved pre-activation is what the backward differentiates: take GELU of the ROUNDED value
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)gelu_for<T>((float)pre[e]);
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(const_cast<float*>(resid)) + o) = pre;
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(out) + o) = v;
      } else if (EPI == EPI_DGELU) {
        typedef T __attribute__((ext_vector_type(4))) T4;
        const T4 pre = *reinterpret_cast<const T4*>(reinterpret_cast<const T*>(resid) + o);
        T4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)(a[e] * gelu_grad_for<T>((float)pre[e]));
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(out) + o) = v;
      } else if (EPI == EPI_GELU) {
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 v;
        v[0] = (T)gelu_for<T>(a[0] + bv.x);
        v[1] = (T)gelu_for<T>(a[1] + bv.y);
        v[2] = (T)gelu_for<T>(a[2] + bv.z);
        v[3] = (T)gelu_for<T>(a[3] + bv.w);
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(out) + o) = v;
      } else if (EPI == EPI_SILU || EPI == EPI_BIAS_T) {
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 v;
        const float p[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)(EPI == EPI_SILU ? silu_for<T>(p[e]) : p[e]);
        *reinterpret_cast<T4*>(reinterpret_cast<T*>(out) + o) = v;
      } else if (EPI == EPI_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(resid + o);
        float4 v;
        v.x = r.x + gv.x * (a[0] + bv.x);
        v.y = r.y + gv.y * (a[1] + bv.y);
        v.z = r.z + gv.z * (a[2] + bv.z);
        v.w = r.w + gv.w * (a[3] + bv.w);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = v;
      } else {
        float4 v = make_float4(a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = v;
      }
    }
  }
}

template <typename T, int TM, int TN, int WM, int WN, int EPI>
int launch_tile(const T* x, const T* w, const float* bias, const float* gamma, const float* resid,
                void* out, int M, int N, int K, hipStream_t st) {
  dim3 grid((M + TM - 1) / TM, (N + TN - 1) / TN);
  hipLaunchKernelGGL((gemm_kernel<T, TM, TN, WM, WN, EPI>), grid, dim3(256), 0, st, x, w, bias,
                     gamma, resid, out, M, N, K);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// Tile choice: the largest tile that still gives the 256 CUs a few workgroups each.  The skinny
// late-stage GEMMs (M = 1024..9216) would run on 32..144 CUs with 128x128 tiles and are
// latency-bound there; 64x64 tiles trade LDS traffic per MFMA for 4x the workgroups.
template <typename T, int EPI>
int launch_typed(const void* X, const void* W, const float* bias, const float* gamma,
                 const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  const T* x = reinterpret_cast<const T*>(X);
  const T* w = reinterpret_cast<const T*>(W);
  const long wg128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long wg12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
  if (N >= 128 && wg128 >= 512)
    return launch_tile<T, 128, 128, 2, 2, EPI>(x, w, bias, gamma, resid, out, M, N, K, st);
  if (wg12864 >= 512 || N < 64)
    return launch_tile<T, 128, 64, 4, 1, EPI>(x, w, bias, gamma, resid, out, M, N, K, st);
  return launch_tile<T, 64, 64, 2, 2, EPI>(x, w, bias, gamma, resid, out, M, N, K, st);
}

template <typename T>
int launch_epi(int epi, const void* X, const void* W, const float* bias, const float* gamma,
               const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  switch (epi) {
    case EPI_GELU: return launch_typed<T, EPI_GELU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_RESID: return launch_typed<T, EPI_RESID>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS: return launch_typed<T, EPI_BIAS>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_GELU_SAVE:
      return launch_typed<T, EPI_GELU_SAVE>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_DGELU: return launch_typed<T, EPI_DGELU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_PLAIN: return launch_typed<T, EPI_PLAIN>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_SILU: return launch_typed<T, EPI_SILU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS_T: return launch_typed<T, EPI_BIAS_T>(X, W, bias, gamma, resid, out, M, N, K, st);
  }
  btsbot_set_error("launch_gemm: bad epilogue %d", epi);
  return BTSBOT_ERR_INVALID_ARG;
}

template <typename T>
int launch_gated_typed(const void* X, const float* gate, int rpa, const void* W, const float* resid,
                       float* out, int M, int N, int K, hipStream_t st) {
  const T* x = reinterpret_cast<const T*>(X);
  const T* w = reinterpret_cast<const T*>(W);
  const long wg12864 = (long)((M + 127) / 128) * ((N + 63) / 64);
  if (wg12864 >= 512) {
    dim3 grid((M + 127) / 128, (N + 63) / 64);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 4, 1, EPI_RESID, true>), grid, dim3(256), 0, st, x, w,
                       nullptr, nullptr, resid, out, M, N, K, gate, rpa);
  } else {
    dim3 grid((M + 63) / 64, (N + 63) / 64);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, EPI_RESID, true>), grid, dim3(256), 0, st, x, w,
                       nullptr, nullptr, resid, out, M, N, K, gate, rpa);
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

int launch_gemm_gated(int prec, const void* X, const float* gate, int rows_per_alert, const void* W,
                      const float* resid, float* out, int M, int N, int K, hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  const int epc = prec == BTSBOT_F32 ? 4 : 8;
  if (K % epc != 0 || N % 4 != 0 || rows_per_alert < 1) {
    btsbot_set_error("launch_gemm_gated: K=%d must be a multiple of %d and N=%d of 4", K, epc, N);
    return BTSBOT_ERR_INVALID_ARG;
  }
  switch (prec) {
    case BTSBOT_F32: return launch_gated_typed<float>(X, gate, rows_per_alert, W, resid, out, M, N, K, st);
    case BTSBOT_BF16: return launch_gated_typed<bf16_t>(X, gate, rows_per_alert, W, resid, out, M, N, K, st);
    case BTSBOT_F16: return launch_gated_typed<f16_t>(X, gate, rows_per_alert, W, resid, out, M, N, K, st);
  }
  btsbot_set_error("launch_gemm_gated: bad precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}

int launch_gemm(int prec, int epi, const void* X, const void* W, const float* bias,
                const float* gamma, const float* resid, void* out, int M, int N, int K,
                hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  const int epc = prec == BTSBOT_F32 ? 4 : 8;
  if (K % epc != 0 || N % 4 != 0) {
    btsbot_set_error("launch_gemm: K=%d must be a multiple of %d and N=%d of 4", K, epc, N);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (M / 128 + 1 > 0x7fffffff / 1 || (N + 63) / 64 > 65535) {
    btsbot_set_error("launch_gemm: grid too large (M=%d N=%d)", M, N);
    return BTSBOT_ERR_INVALID_ARG;
  }
  static const bool v1_only = [] {
    const char* e = getenv("BTSBOT_AMD_GEMM_V1");   // A/B switch for timing
    return e != nullptr && e[0] == '1';
  }();
  static const bool train_v1 = [] {
    const char* e = getenv("BTSBOT_AMD_TRAIN_GEMM_V1");   // A/B: training epilogues on the register-staged kernel
    return e != nullptr && e[0] == '1';
  }();
  const bool train_epi = epi == EPI_GELU_SAVE || epi == EPI_DGELU || epi == EPI_PLAIN;
  if (!v1_only && !(train_epi && train_v1) && gemm2_supported(prec, M, N, K))
    return launch_gemm2(prec, epi, X, W, bias, gamma, resid, out, M, N, K, st);
  switch (prec) {
    case BTSBOT_F32: return launch_epi<float>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
    case BTSBOT_BF16: return launch_epi<bf16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
    case BTSBOT_F16: return launch_epi<f16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
  }
  btsbot_set_error("launch_gemm: bad precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
