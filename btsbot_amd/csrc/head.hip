// Fused classifier heads (fp32): optional head LayerNorm on the 1x1 image feature, metadata branch
// (BatchNorm1d folded to scale/shift -> Linear -> act -> Linear [-> act]), concat (image first, then
// metadata), fusion MLP, logits + sigmoid scores -- one kernel, nothing but the logits leaves the CU.
//
// Reference wiring: /root/reference/btsbot/architectures.py:146-171 (mm_ConvNeXt, GELU),
// :109-122 (ConvNeXt head), :282-293 (um_nn, ReLU), :299-313,358-372 (frozen_fusion, ReLU, metadata
// branch without its trailing activation); sigmoid: inference_example.py:91.
// Also the small parameter (re)packing kernels.
#include "common.h"

namespace {

constexpr int HG = 8;     // alerts per workgroup
constexpr int HNT = 512;   // threads (1024 would cap VGPRs at 128 and spill)
constexpr float HN_EPS = 1e-6f;

// Activations of the workgroup's HG alerts live in LDS k-major: v[k][g] (8 alerts = two float4),
// so one thread reads all alerts of a k with two ds_read_b128 broadcasts.
// out[n][g] = act(bias[n] + sum_k in[k][g] * wt[k][n]); wt is K-major so consecutive threads (n)
// read consecutive addresses.  The HNT threads cover min(N,HNT) neurons x KS slices of K
// (KS = HNT / N).  The head is a chain of small dependent layers, i.e. latency-bound: every thread
// keeps TWO groups of 16 weight loads in flight (a slice of <= 32 k issues all its loads before
// the first FMA), partial sums meet in LDS (`part`), biases are fetched before the K loop.
__device__ __forceinline__ void dense(const float* in, int K, const float* __restrict__ wt,
                                      const float* __restrict__ bias, int N, int act, float* outp,
                                      float* part) {
  constexpr int GK = 16;
  int ks = 1;
  while (ks * 2 * N <= HNT) ks *= 2;
  const int kchunk = (K + ks - 1) / ks;
  for (int n0 = 0; n0 < N; n0 += HNT) {              // N > HNT: several passes (ks == 1)
    const int n = n0 + (threadIdx.x % (N < HNT ? N : HNT));
    const int slice = N < HNT ? threadIdx.x / N : 0;
    const bool live = n < N && slice < ks;
    // bias of the output this thread finalises after the K loop
    const int fin = threadIdx.x;                      // ks > 1: element index into [N][HG]
    const float bfin = ks == 1 ? (live ? bias[n] : 0.f) : (fin < HG * N ? bias[fin / HG] : 0.f);
    float acc[HG];
#pragma unroll
    for (int g = 0; g < HG; ++g) acc[g] = 0.f;
    if (live) {
      const int k0 = slice * kchunk, k1 = min(K, k0 + kchunk);
      const int ngrp = k1 > k0 ? (k1 - k0 + GK - 1) / GK : 0;
      float wa[GK], wb[GK];
      if (ngrp > 0) {
#pragma unroll
        for (int u = 0; u < GK; ++u) wa[u] = k0 + u < k1 ? wt[(size_t)(k0 + u) * N + n] : 0.f;
      }
      if (ngrp > 1) {
#pragma unroll
        for (int u = 0; u < GK; ++u)
          wb[u] = k0 + GK + u < k1 ? wt[(size_t)(k0 + GK + u) * N + n] : 0.f;
      }
      for (int gi = 0; gi < ngrp; ++gi) {
        const int kb = k0 + gi * GK;
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int kk = min(kb + u, k1 - 1);         // padded taps carry weight 0
          const float4 a0 = *reinterpret_cast<const float4*>(in + kk * HG);
          const float4 a1 = *reinterpret_cast<const float4*>(in + kk * HG + 4);
          const float w = wa[u];
          acc[0] = fmaf(a0.x, w, acc[0]); acc[1] = fmaf(a0.y, w, acc[1]);
          acc[2] = fmaf(a0.z, w, acc[2]); acc[3] = fmaf(a0.w, w, acc[3]);
          acc[4] = fmaf(a1.x, w, acc[4]); acc[5] = fmaf(a1.y, w, acc[5]);
          acc[6] = fmaf(a1.z, w, acc[6]); acc[7] = fmaf(a1.w, w, acc[7]);
        }
#pragma unroll
        for (int u = 0; u < GK; ++u) wa[u] = wb[u];
        if (gi + 2 < ngrp) {
          const int kn = k0 + (gi + 2) * GK;
#pragma unroll
          for (int u = 0; u < GK; ++u) wb[u] = kn + u < k1 ? wt[(size_t)(kn + u) * N + n] : 0.f;
        }
      }
    }
    if (ks == 1) {
      if (live) {
#pragma unroll
        for (int g = 0; g < HG; ++g) outp[n * HG + g] = apply_act(acc[g] + bfin, act);
      }
    } else {
      if (live) {
#pragma unroll
        for (int g = 0; g < HG; ++g) part[(slice * N + n) * HG + g] = acc[g];
      }
      __syncthreads();
      for (int i = fin; i < HG * N; i += HNT) {
        float t = i == fin ? bfin : bias[i / HG];
        for (int s2 = 0; s2 < ks; ++s2) t += part[s2 * N * HG + i];
        outp[i] = apply_act(t, act);
      }
    }
  }
}

__global__ __launch_bounds__(HNT) void head_kernel(HeadArgs a) {
  static_assert(HG == 8, "dense() reads the 8 alerts of a k as two float4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int zd = a.dims[0];
  int maxw = a.f1 > a.n_meta ? a.f1 : a.n_meta;
  for (int i = 1; i <= a.n_layers; ++i) maxw = a.dims[i] > maxw ? a.dims[i] : maxw;
  float* z = smem;               // [zd][HG]
  float* t0 = z + HG * zd;       // [maxw][HG]
  float* t1 = t0 + HG * maxw;    // [maxw][HG]
  float* part = t1 + HG * maxw;  // [KS][N][HG], KS * N <= HNT
  const int b0 = blockIdx.x * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- image feature (+ head LayerNorm) -> z[0:feat_dim][g]
  if (a.feat_dim > 0) {
    for (int g = wave; g < HG; g += HNT / 64) {
      const int b = b0 + g;
      const float* src = a.feat + (size_t)(b < a.B ? b : a.B - 1) * a.feat_dim;
      if (a.hn_w != nullptr) {
        float v[12];                                   // feat_dim <= 768: one load per value
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int c = lane + 64 * i;
          v[i] = c < a.feat_dim ? src[c] : 0.f;
          sum += v[i];
        }
        const float mean = wave_sum(sum) / a.feat_dim;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const float d = lane + 64 * i < a.feat_dim ? v[i] - mean : 0.f;
          sq += d * d;
        }
        const float rstd = rsqrtf(wave_sum(sq) / a.feat_dim + HN_EPS);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int c = lane + 64 * i;
          if (c < a.feat_dim) z[c * HG + g] = (v[i] - mean) * rstd * a.hn_w[c] + a.hn_b[c];
        }
      } else {
        for (int c = lane; c < a.feat_dim; c += 64) z[c * HG + g] = src[c];
      }
    }
  }
  // ---- metadata branch -> z[feat_dim : feat_dim + f2][g]
  if (a.n_meta > 0) {
    for (int i = tid; i < HG * a.n_meta; i += HNT) {
      const int g = i / a.n_meta, j = i - g * a.n_meta;
      const int b = b0 + g;
      const float v = a.meta[(size_t)(b < a.B ? b : a.B - 1) * a.n_meta + j];
      t0[j * HG + g] = fmaf(v, a.bn_scale[j], a.bn_shift[j]);
    }
    __syncthreads();
    dense(t0, a.n_meta, a.m1_wt, a.m1_b, a.f1, a.meta_act, t1, part);
    __syncthreads();
    dense(t1, a.f1, a.m2_wt, a.m2_b, a.f2, a.meta_trailing_act ? a.meta_act : ACT_NONE,
          z + a.feat_dim * HG, part);
  }
  __syncthreads();
  // ---- fusion MLP
  const float* in = z;
  float* bufs[2] = {t0, t1};
  for (int i = 0; i < a.n_layers; ++i) {
    float* o = bufs[i & 1];
    dense(in, a.dims[i], a.wt[i], a.b[i], a.dims[i + 1],
          i + 1 < a.n_layers ? a.comb_act : ACT_NONE, o, part);
    __syncthreads();
    in = o;
  }
  if (tid < HG && b0 + tid < a.B) {
    const float zz = in[tid];
    a.logits[b0 + tid] = zz;
    if (a.scores != nullptr) a.scores[b0 + tid] = 1.0f / (1.0f + expf(-zz));
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ s, T* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    d[i] = (T)s[i];
}

__global__ void transpose_kernel(const float* __restrict__ s, float* __restrict__ d, int R, int Cc) {
  const int64_t n = (int64_t)R * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);  // d[c][r]
    d[i] = s[(int64_t)r * Cc + c];
  }
}

template <typename T>
__global__ void pack_down_kernel(const float* __restrict__ s, T* __restrict__ d, int Cout, int Cin) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    // d[co][q][ci] <- s[co][ci][q],  q = ky*2 + kx
    const int ci = (int)(i % Cin);
    const int q = (int)((i / Cin) & 3);
    const int co = (int)(i / (4 * (int64_t)Cin));
    d[i] = (T)s[((int64_t)co * Cin + ci) * 4 + q];
  }
}

__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv,
                               float* scale, float* shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = w[i] / sqrtf(rv[i] + 1e-5f);
    scale[i] = s;
    shift[i] = b[i] - rm[i] * s;
  }
}

inline int nblocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int launch_head(const HeadArgs& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  int maxw = a.f1 > a.n_meta ? a.f1 : a.n_meta;
  for (int i = 1; i <= a.n_layers; ++i) maxw = a.dims[i] > maxw ? a.dims[i] : maxw;
  const size_t lds = (size_t)HG * (a.dims[0] + 2 * maxw + HNT) * sizeof(float);
  if (a.feat_dim > 768) {
    btsbot_set_error("head: feature width %d above 768", a.feat_dim);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (lds > 64 * 1024) {
    btsbot_set_error("head: layer widths too large for one workgroup (%zu bytes of LDS)", lds);
    return BTSBOT_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(head_kernel, dim3((a.B + HG - 1) / HG), dim3(HNT), lds, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_cast(int prec, const float* src, void* dst, int64_t n, hipStream_t st) {
  if (n <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
      return BTSBOT_OK;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<bf16_t*>(dst), n);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(cast_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<f16_t*>(dst), n);
      break;
    default:
      btsbot_set_error("cast: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_transpose_f32(const float* src, float* dst, int R, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(nblocks((int64_t)R * Cc)), dim3(256), 0, st, src, dst,
                     R, Cc);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_pack_down(int prec, const float* src, void* dst, int Cout, int Cin, hipStream_t st) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(pack_down_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<float*>(dst), Cout, Cin);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(pack_down_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<bf16_t*>(dst), Cout, Cin);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(pack_down_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<f16_t*>(dst), Cout, Cin);
      break;
    default:
      btsbot_set_error("pack_down: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                   float* shift, int n, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, b, rm, rv, scale,
                     shift, n);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
