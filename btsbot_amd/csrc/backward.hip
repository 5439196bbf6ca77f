// Backward kernels of the ConvNeXt image branch (gfx950).  Replaces the autograd graph that
// loss.backward() walks at /root/reference/btsbot/train.py:526 for timm's ConvNeXt blocks:
//
//   y = x + gamma * fc2(gelu(fc1(LN(dwconv(x)))))
//
// Given dy: G = dy^T h (wgrad GEMM) -> dW2 = gamma (.) G, dgamma = <W2, G> + b2 * colsum(dy);
// dh = (gamma (.) dy) W2, da = dh * gelu'(a) (dgrad GEMM epilogue); dW1 = da^T xn; dxn = da W1;
// LayerNorm backward on the recomputed depthwise output; depthwise dgrad = the same depthwise
// kernel with the 7x7 filter flipped; depthwise wgrad = per-tap correlation reduced over pixels
// and alerts.  Reductions over the batch use fp32 atomics into the (pre-zeroed) gradient arena.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr float LN_EPS = 1e-6f;

template <typename T> struct Mw;
template <> struct Mw<float> {
  using frag = float;
  static constexpr int KR = 4;   // reduction depth per MFMA
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Mw<bf16_t> {
  using frag = bf16x8;
  static constexpr int KR = 32;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mw<f16_t> {
  using frag = f16x8;
  static constexpr int KR = 32;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------
// wgrad: out[n][k] += sum_m D[m][n] * A[m][k]   (D: [M][N], A: [M][K], both row-major, type T)
// Workgroup = 64x64 output tile x one slice of M; 4 waves in 2x2, each a 32x32 sub-tile.
// Both operands are "reduction-major" in memory, so fragments are gathered from row-major LDS
// tiles with strided 16-bit reads (lane (i, g) takes rows 8g..8g+7 of column i).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ D,
                                                    const T* __restrict__ A,
                                                    float* __restrict__ out, int M, int N, int K,
                                                    int ldo, int mslice) {
  using MM = Mw<T>;
  using frag = typename MM::frag;
  constexpr int TM = 64;                 // rows of the reduction dimension per LDS tile
  constexpr int PITCH = 64 + 8;          // elements per LDS row (padded)
  __shared__ T Ds[TM * PITCH];
  __shared__ T As[TM * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mbeg = blockIdx.z * mslice, mend = min(M, mbeg + mslice);
  const int li = lane & 15, lg = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int m0 = mbeg; m0 < mend; m0 += TM) {
    // stage [64 m][64 n] of D and [64 m][64 k] of A (zero-filled at the edges)
    for (int i = tid; i < TM * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      const int m = m0 + r;
      Ds[r * PITCH + c] = (m < mend && n0 + c < N) ? D[(size_t)m * N + n0 + c] : (T)0.f;
      As[r * PITCH + c] = (m < mend && k0 + c < K) ? A[(size_t)m * K + k0 + c] : (T)0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ms = 0; ms < TM; ms += MM::KR) {
      frag af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int cn = wn * 32 + t * 16 + li, ck = wk * 32 + t * 16 + li;
        if constexpr (MM::KR == 32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            af[t][e] = Ds[(ms + 8 * lg + e) * PITCH + cn];
            bf[t][e] = As[(ms + 8 * lg + e) * PITCH + ck];
          }
        } else {
          af[t] = Ds[(ms + lg) * PITCH + cn];
          bf[t] = As[(ms + lg) * PITCH + ck];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = MM::run(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }
  // C/D layout: col = lane&15 -> k, row = 4*(lane>>4)+r -> n
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + i * 16 + lg * 4 + r;
        const int k = k0 + wk * 32 + j * 16 + li;
        if (n < N && k < K) atomicAdd(out + (size_t)n * ldo + k, acc[i][j][r]);
      }
}

// The same product for fp32 operands at the matrix pipe's fp32 rate (v_mfma_f32_32x32x2_f32: 256 FLOP per clock and CU).
// The kernel above stages 64 x 64 tiles element by element under a bounds test and ran the fp32 mode's 28 filter-gradient
// GEMMs per training step at 24 TFLOP/s -- 5.7 ms of a 10.5 ms step, on the side stream that bounds it.  Here: TN x TK
// output tile (64 or 128 each: one wave per 64 x 64), both operands in 16-row stages through LDS as 16-byte pieces, the
// next stage's loads in flight under the current stage's products (registers -> the other LDS buffer, one barrier per
// stage).  Both operands are reduction-major in memory, which is what the 32x32x2 fragments want: lane l reads element
// [m + (l >> 5)][base + (l & 31)] -- 32 consecutive floats per half wave.  N % TN == K % TK == 0 (the launcher picks).
typedef __attribute__((ext_vector_type(16))) float f32x16w;
template <int TN, int TK>
__global__ __launch_bounds__(64 * (TN / 64) * (TK / 64)) void wgrad_f32_kernel(const float* __restrict__ D,
                                                                                const float* __restrict__ A,
                                                                                float* __restrict__ out, int M, int N, int K,
                                                                                int ldo, int mslice, float* __restrict__ cs) {
  // cs != nullptr: cs[n] += sum_m D[m][n] as well (the bias gradient beside the filter gradient): the workgroups of the
  // first k-tile column add up the D rows they stage anyway -- was a launch of its own reading D once more (fp32 mode:
  // 32 launches, 1.9 ms of kernel time per training step)
  constexpr int WK = TK / 64, NTH = 64 * (TN / 64) * WK, TM = 16;
  constexpr int DV = TM * TN / 4, AV = TM * TK / 4;                 // 16-byte pieces per stage
  constexpr int DPT = (DV + NTH - 1) / NTH, APT = (AV + NTH - 1) / NTH;
  static_assert(DV % NTH == 0 && AV % NTH == 0, "whole pieces per thread");
  __shared__ __attribute__((aligned(16))) float Ds[2][TM * TN];
  __shared__ __attribute__((aligned(16))) float As[2][TM * TK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave - wn * WK;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
  const int mbeg = blockIdx.z * mslice, mend = min(M, mbeg + mslice);
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16w acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 dr[DPT], ar[APT];
  const bool do_cs = cs != nullptr && blockIdx.y == 0;   // (uniform)
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);         // this thread's columns 4 (tid % (TN / 4)) .. + 3, its rows
  static_assert(NTH % (TN / 4) == 0, "a thread's pieces share their columns");
  auto fetch = [&](int m0) {
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
      const int i = tid + u * NTH, r = i / (TN / 4), c = i - r * (TN / 4);
      dr[u] = m0 + r < mend ? *reinterpret_cast<const float4*>(D + (size_t)(m0 + r) * N + n0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < APT; ++u) {
      const int i = tid + u * NTH, r = i / (TK / 4), c = i - r * (TK / 4);
      ar[u] = m0 + r < mend ? *reinterpret_cast<const float4*>(A + (size_t)(m0 + r) * K + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
      reinterpret_cast<float4*>(Ds[buf])[tid + u * NTH] = dr[u];
      if (do_cs) {
        csum.x += dr[u].x;
        csum.y += dr[u].y;
        csum.z += dr[u].z;
        csum.w += dr[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < APT; ++u) reinterpret_cast<float4*>(As[buf])[tid + u * NTH] = ar[u];
  };
  if (mbeg < mend) {
    fetch(mbeg);
    park(0);
  }
  __syncthreads();
  int buf = 0;
  for (int m0 = mbeg; m0 < mend; m0 += TM) {
    const bool more = m0 + TM < mend;   // (uniform)
    if (more) fetch(m0 + TM);
    const float* ds = Ds[buf] + wn * 64 + l31;
    const float* as = As[buf] + wk * 64 + l31;
#pragma unroll
    for (int ms = 0; ms < TM; ms += 2) {
      float af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = ds[(ms + lh) * TN + 32 * t];
        bf[t] = as[(ms + lh) * TK + 32 * t];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) park(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // C/D layout of 32x32: column (lane & 31) -> k, register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = k0 + wk * 64 + 32 * j + l31;
        atomicAdd(out + (size_t)n * ldo + k, acc[i][j][r]);
      }
  if (do_cs) {   // the row groups' column sums meet in LDS (the operand stages are dead: the loop ended on a barrier)
    constexpr int CG = TN / 4, RG = NTH / CG;
    float4* red = reinterpret_cast<float4*>(Ds[0]);
    static_assert(RG * CG * 16 <= (int)sizeof(Ds), "the partial sums fit the D stages");
    red[tid] = csum;   // [row group = tid / CG][column group = tid % CG]
    __syncthreads();
    if (tid < CG) {
      float4 t = red[tid];
#pragma unroll
      for (int g = 1; g < RG; ++g) {
        const float4 e = red[g * CG + tid];
        t.x += e.x;
        t.y += e.y;
        t.z += e.z;
        t.w += e.w;
      }
      float* d = cs + n0 + 4 * tid;
      atomicAdd(d, t.x);
      atomicAdd(d + 1, t.y);
      atomicAdd(d + 2, t.z);
      atomicAdd(d + 3, t.w);
    }
  }
}

// out[n] += sum_m in[m][n].  Block = 64 columns x 4 row groups over one slice of M; the row groups
// meet in LDS so that every column costs ONE atomic per block (same-address atomics serialise).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ in, float* out, int M,
                                                     int N, int mslice, float* part) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  const int mbeg = blockIdx.y * mslice, mend = min(M, mbeg + mslice);
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int m = mbeg + rg;
    for (; m + 4 < mend; m += 8) {
      s0 += (float)in[(size_t)m * N + n];
      s1 += (float)in[(size_t)(m + 4) * N + n];
    }
    if (m < mend) s0 += (float)in[(size_t)m * N + n];
  }
  sh[rg][cl] = s0 + s1;
  __syncthreads();
  if (rg == 0 && n < N) det_add(out + n, sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl], part, (size_t)blockIdx.y * N + n);
}

// out[m][c] = (T)(in[m][c] * scale[c])   (scale may be null)
template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ scale,
                                                         T* __restrict__ out, long n, int C) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (T)(in[i] * (scale != nullptr ? scale[i % C] : 1.f));
}

// out[r][c] = (T)(in[r][c] * rowscale[r])
template <typename T>
__global__ __launch_bounds__(256) void rowscale_cast_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ rowscale,
                                                            T* __restrict__ out, long n, int cols) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (T)(in[i] * rowscale[i / cols]);
}

// fc2 weight/scale gradients from G = dy^T h:  dW2[c][k] = gamma[c] G[c][k];
// dgamma[c] = sum_k W2[c][k] G[c][k] + b2[c] S[c];  db2[c] = gamma[c] S[c]    (S = colsum(dy))
// G and S are accumulators of the filter-gradient GEMM in front: every element is read exactly once here and left
// zero for the next block's GEMM (was: a memset launch per block)
__global__ __launch_bounds__(256) void fc2_grads_kernel(float* __restrict__ G,
                                                        float* __restrict__ S,
                                                        const float* __restrict__ w2,
                                                        const float* __restrict__ b2,
                                                        const float* __restrict__ gamma,
                                                        float* dW2, float* db2, float* dgamma,
                                                        int C, int H) {
  const int c = blockIdx.x;
  const float g = gamma[c];
  float part = 0.f;
  for (int k = threadIdx.x; k < H; k += 256) {
    const float gv = G[(size_t)c * H + k];
    G[(size_t)c * H + k] = 0.f;
    dW2[(size_t)c * H + k] = g * gv;
    part += w2[(size_t)c * H + k] * gv;
  }
  part = wave_sum(part);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sc = S[c];
    S[c] = 0.f;
    dgamma[c] = sh[0] + sh[1] + sh[2] + sh[3] + b2[c] * sc;
    db2[c] = g * sc;
  }
}

// Source row of the LayerNorm backward's incoming gradient.  Dense: dxn + r * C.  After a 2x2 / stride-2 downsample
// (patch_hw = side of the downsample's INPUT map): the gradient arrives as patches [B * HO * HO][4 C] (k = q * C + c,
// q = 2 (y & 1) + (x & 1)); input pixel r = (b, y, x) reads its slice of the patch that covers it, pixels outside
// every patch (the odd last row / column) get zero.  (Was: unpatch_kernel scattering into a dense map first.)
__device__ __forceinline__ const float* ln_bwd_src(const float* dxn, long r, int C, int patch_hw, bool* zero) {
  *zero = false;
  if (patch_hw == 0) return dxn + r * C;
  const int P = patch_hw * patch_hw, HO = patch_hw / 2;
  const long b = r / P;
  const int p = (int)(r - b * P), iy = p / patch_hw, ix = p - iy * patch_hw;
  if (iy >= 2 * HO || ix >= 2 * HO) {
    *zero = true;
    return dxn;
  }
  return dxn + ((b * HO + (iy >> 1)) * HO + (ix >> 1)) * 4 * (long)C + ((iy & 1) * 2 + (ix & 1)) * C;
}

// LayerNorm backward over the C channels of each row.  One wave per row (grid-stride).
//   xhat = (d - mean) * rstd;  t = dxn * g;  dd = rstd * (t - mean(t) - xhat * mean(t * xhat))
//   dg += dxn * xhat;  dbeta += dxn
template <int CPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ d,
                                                     const float* __restrict__ dxn,
                                                     const float* __restrict__ g, float* dd,
                                                     float* dg, float* dbeta, long rows, int C,
                                                     void* __restrict__ out16, int prec16, int patch_hw,
                                                     float* part) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float gl[CPT], adg[CPT], adb[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = lane + 64 * i;
    gl[i] = c < C ? g[c] : 0.f;
    adg[i] = adb[i] = 0.f;
  }
  // dd may be dxn (in place): a load cannot move above an earlier store, so the NEXT row's pieces are requested
  // explicitly before this row's stores (they arrive under the four wave reductions)
  float nv[CPT], ndx[CPT];
  auto fetch = [&](long r) {
    bool zero;
    const float* src = ln_bwd_src(dxn, r, C, patch_hw, &zero);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = lane + 64 * i;
      nv[i] = c < C ? d[r * C + c] : 0.f;
      const float g = c < C ? src[c] : 0.f;
      ndx[i] = zero ? 0.f : g;
    }
  };
  if (w0 < rows) fetch(w0);
  for (long r = w0; r < rows; r += nw) {
    float v[CPT], t[CPT], dxr[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      v[i] = nv[i];
      dxr[i] = ndx[i];
      s += v[i];
    }
    if (r + nw < rows) fetch(r + nw);
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < C ? v[i] - mean : 0.f;
      q += v[i] * v[i];
    }
    const float rstd = rsqrtf(wave_sum(q) / C + LN_EPS);
    float st = 0.f, stx = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const float dx = dxr[i];
      v[i] *= rstd;                     // xhat
      t[i] = dx * gl[i];
      st += t[i];
      stx += t[i] * v[i];
      adg[i] += dx * v[i];
      adb[i] += dx;
    }
    st = wave_sum(st) / C;
    stx = wave_sum(stx) / C;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        const float o = rstd * (t[i] - st - v[i] * stx);
        dd[r * C + c] = o;
        if (out16 != nullptr) {   // the same values as the next 16-bit GEMM's operand (was a cast launch)
          if (prec16 == BTSBOT_BF16) reinterpret_cast<bf16_t*>(out16)[r * C + c] = (bf16_t)o;
          else reinterpret_cast<f16_t*>(out16)[r * C + c] = (f16_t)o;
        }
      }
    }
  }
  // the block's 4 waves meet in LDS: one atomic per channel per block
  __shared__ float sh[2][4][CPT * 64];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    sh[0][wv][lane + 64 * i] = adg[i];
    sh[1][wv][lane + 64 * i] = adb[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    det_add(dg + c, sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c], part, (size_t)blockIdx.x * 2 * C + c);
    det_add(dbeta + c, sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c], part, (size_t)blockIdx.x * 2 * C + C + c);
  }
}

// 1x1 maps (stage 3): the block's LayerNorm backward, depthwise filter gradient and depthwise input gradient in one
// launch.  On a 1x1 map the 7x7 depthwise convolution is its centre tap, so everything is per (alert, channel):
//   dd = LN'(d) . dxn;   dy += dd * w[24][c];   dW[c][24] += dd * x_in;   db[c] += dd;   dg, dbeta as ln_bwd_kernel
// One wave per alert (grid-stride); the workgroup's four waves meet in LDS, one atomic per channel per workgroup and
// gradient.  (Was ln_bwd_kernel + dw_wgrad_kernel<1> + its column sum + dw_plain_kernel<1>: 48 us per block.)
template <int CPT>
__global__ __launch_bounds__(256) void ln_dw1_bwd_kernel(const float* __restrict__ d, const float* __restrict__ dxn,
                                                         const float* __restrict__ g, const float* __restrict__ xin,
                                                         const float* __restrict__ wc, float* dy,
                                                         void* __restrict__ out16, int prec16, float* dg, float* dbeta,
                                                         float* dw, float* dbias, long rows, int C, float* part) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float gl[CPT], wl[CPT], adg[CPT], adb[CPT], adw[CPT], adc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = lane + 64 * i;
    gl[i] = c < C ? g[c] : 0.f;
    wl[i] = c < C ? wc[c] : 0.f;
    adg[i] = adb[i] = adw[i] = adc[i] = 0.f;
  }
  // Two rows per trip (r and r + nw): their 8 x CPT loads are in flight together and their wave sums interleave -- one
  // row per trip was a chain of dependent memory round trips and lane reductions (8 rows per wave: 23-27 us per launch
  // for a 2 MB tensor)
  for (long r = w0; r < rows; r += 2 * nw) {
    const bool two = r + nw < rows;          // wave-uniform
    const long rr[2] = {r, two ? r + nw : r};
    float v[2][CPT], t[2][CPT], dxr[2][CPT], xr[2][CPT], yr[2][CPT];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = c < C ? d[rr[u] * C + c] : 0.f;
        dxr[u][i] = c < C && (u == 0 || two) ? dxn[rr[u] * C + c] : 0.f;   // (an absent second row: dxn = 0, so dd = 0)
        xr[u][i] = c < C ? xin[rr[u] * C + c] : 0.f;
        yr[u][i] = c < C ? dy[rr[u] * C + c] : 0.f;
        s[u] += v[u][i];
      }
    float mean[2], q[2] = {0.f, 0.f}, rstd[2], st[2] = {0.f, 0.f}, stx[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) mean[u] = wave_sum(s[u]) / C;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = lane + 64 * i;
        v[u][i] = c < C ? v[u][i] - mean[u] : 0.f;
        q[u] += v[u][i] * v[u][i];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) rstd[u] = rsqrtf(wave_sum(q[u]) / C + LN_EPS);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        v[u][i] *= rstd[u];                     // xhat
        t[u][i] = dxr[u][i] * gl[i];
        st[u] += t[u][i];
        stx[u] += t[u][i] * v[u][i];
        adg[i] += dxr[u][i] * v[u][i];
        adb[i] += dxr[u][i];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = wave_sum(st[u]) / C;
      stx[u] = wave_sum(stx[u]) / C;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
          const float dd = rstd[u] * (t[u][i] - st[u] - v[u][i] * stx[u]);
          adw[i] += dd * xr[u][i];
          adc[i] += dd;
          const float o = yr[u][i] + dd * wl[i];
          dy[rr[u] * C + c] = o;
          if (out16 != nullptr) {
            if (prec16 == BTSBOT_BF16) reinterpret_cast<bf16_t*>(out16)[rr[u] * C + c] = (bf16_t)o;
            else reinterpret_cast<f16_t*>(out16)[rr[u] * C + c] = (f16_t)o;
          }
        }
      }
    }
  }
  __shared__ float sh[4][4][CPT * 64];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    sh[0][wv][lane + 64 * i] = adg[i];
    sh[1][wv][lane + 64 * i] = adb[i];
    sh[2][wv][lane + 64 * i] = adw[i];
    sh[3][wv][lane + 64 * i] = adc[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const size_t row = (size_t)blockIdx.x * 4 * C;
    det_add(dg + c, sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c], part, row + c);
    det_add(dbeta + c, sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c], part, row + C + c);
    det_add(dw + (size_t)c * 49 + 24, sh[2][0][c] + sh[2][1][c] + sh[2][2][c] + sh[2][3][c], part, row + 2 * C + c);
    det_add(dbias + c, sh[3][0][c] + sh[3][1][c] + sh[3][2][c] + sh[3][3][c], part, row + 3 * C + c);
  }
}

// Depthwise 7x7 p3 without LayerNorm, fp32 in / fp32 out, one workgroup per alert, thread =
// channel (grid-stride over channels), whole map in LDS.  out = conv(x, w) [+ bias] [+ addend].
// With the taps flipped (w'[t] = w[48 - t]) this is the gradient w.r.t. the conv input.
template <int HW>
__global__ __launch_bounds__(256) void dw_plain_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w, int flip,
                                                       const float* __restrict__ bias,
                                                       const float* addend, float* out, int C,
                                                       void* __restrict__ out16, int prec16) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [HW*HW][C] map | [49][C] taps
  constexpr int P = HW * HW;
  float* ws = xs + P * C;
  const size_t base = (size_t)blockIdx.x * P * C;
  // taps in LDS (already flipped if asked) and a rolled ky loop: with the 49 taps in registers and the ky
  // loop unrolled the 15x15 instance needed 256 VGPRs + scratch and ran one wave per SIMD (131 us per launch)
  constexpr bool TAPS_LDS = HW >= 15;
  constexpr bool TAPS_GLOBAL = HW == 7;   // 7x7 maps: rolled ky loop, the row's 7 taps re-read from L1 per ky
  constexpr int KY_UNROLL = (TAPS_LDS || TAPS_GLOBAL) ? 1 : 7;   // small maps: the taps stay in registers (staging them costs more
                                        // than the whole convolution there)
  if (TAPS_LDS)
    for (int i = threadIdx.x; i < 49 * C; i += 256) {
      const int t = i / C, cc = i - t * C;
      ws[i] = w[(flip ? 48 - t : t) * C + cc];
    }
  {
    // 16-byte pieces, eight in flight per thread before the first LDS store (the scalar copy loop waited
    // for every 4-byte load: 56 exposed HBM latencies per workgroup at 15x15x64)
    const float4* src = reinterpret_cast<const float4*>(x + base);
    float4* dst = reinterpret_cast<float4*>(xs);
    const int n4 = P * C / 4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 2048) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[min(i0 + k * 256, n4 - 1)];   // (clamped: no conditional definition)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k * 256 < n4) dst[i0 + k * 256] = v[k];
    }
  }
  __syncthreads();
  // thread = (channel, row group): narrow maps (C < 256) spread their rows over the idle waves
  const int G = C < 256 ? 256 / C : 1;
  const int rg = C < 256 ? threadIdx.x / C : 0;
  const bool idle = C < 256 && (int)threadIdx.x >= G * C;   // C does not divide 256 (nano)
  for (int c = idle ? C : (C < 256 ? threadIdx.x % C : threadIdx.x); c < C; c += 256) {
    const float b = bias != nullptr ? bias[c] : 0.f;
    float wv[(TAPS_LDS || TAPS_GLOBAL) ? 1 : 49];
    if (!TAPS_LDS && !TAPS_GLOBAL) {
#pragma unroll
      for (int t = 0; t < 49; ++t) wv[t] = w[(flip ? 48 - t : t) * C + c];
    }
    for (int y = rg; y < HW; y += G) {
      float acc[HW];
      // the row's addend pieces are requested before the taps run: `out` may be `addend` (dx = dy + conv), so inside
      // the store loop below every load had to wait for the previous store -- HW exposed HBM round trips per row
#pragma unroll
      for (int xx = 0; xx < HW; ++xx)
        acc[xx] = b + (addend != nullptr ? addend[base + (size_t)(y * HW + xx) * C + c] : 0.f);
#pragma unroll KY_UNROLL
      for (int ky = 0; ky < 7; ++ky) {
        const int iy = y + ky - 3;
        if (iy < 0 || iy >= HW) continue;
        float in[HW], wk[7];
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) in[xx] = xs[(iy * HW + xx) * C + c];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
          wk[kx] = TAPS_LDS      ? ws[(ky * 7 + kx) * C + c]
                   : TAPS_GLOBAL ? w[(flip ? 48 - (ky * 7 + kx) : ky * 7 + kx) * C + c]
                                 : wv[(TAPS_LDS || TAPS_GLOBAL) ? 0 : ky * 7 + kx];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const int ix = xx + kx - 3;
            if (ix >= 0 && ix < HW) acc[xx] = fmaf(in[ix], wk[kx], acc[xx]);
          }
      }
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) {
        const size_t o = base + (size_t)(y * HW + xx) * C + c;
        out[o] = acc[xx];
        if (out16 != nullptr) {   // the same values as the next block's 16-bit GEMM operand (was a cast launch)
          if (prec16 == BTSBOT_BF16) reinterpret_cast<bf16_t*>(out16)[o] = (bf16_t)acc[xx];
          else reinterpret_cast<f16_t*>(out16)[o] = (f16_t)acc[xx];
        }
      }
    }
  }
}

// Depthwise filter gradient: sum_{alerts,pixels} dd[p][c] * x[p + delta_t][c] per tap, and sum dd
// for the bias.  Workgroup = GA alerts (sequentially), thread = (channel, row group); every row
// group writes ONE partial row [C*49 | C] which launch_dw_wgrad then column-sums into the gradient arena (the filter
// and its bias are adjacent there) -- no same-address atomic storm.
template <int HW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ dd, float* dw,
                                                       float* dbias, int B, int C, int ga) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // xs [P][C], ds [P][C]
  constexpr int P = HW * HW;
  float* xs = sm;
  float* ds = sm + P * C;
  const int a0 = blockIdx.x * ga, a1 = min(B, a0 + ga);
  // thread = (channel, row group), as in dw_plain_kernel; every row group owns a partial row
  const int G = C < 256 ? 256 / C : 1;
  const int rg = C < 256 ? threadIdx.x / C : 0;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const bool idle = C < 256 && (int)threadIdx.x >= G * C;   // C does not divide 256 (nano)
    const int c = idle ? C : (C < 256 ? threadIdx.x % C : c0 + threadIdx.x);
    float acc[49], ab = 0.f;
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = 0.f;
    for (int a = a0; a < a1; ++a) {
      __syncthreads();
      const size_t base = (size_t)a * P * C;
      {
        const float4* sx = reinterpret_cast<const float4*>(x + base);
        const float4* sd = reinterpret_cast<const float4*>(dd + base);
        float4* dx4 = reinterpret_cast<float4*>(xs);
        float4* dd4 = reinterpret_cast<float4*>(ds);
        const int n4 = P * C / 4;
        // 16 pieces in flight per thread (two trips for a 15x15x64 map; with 4 in flight the 7 trips were 7 exposed
        // HBM latencies per alert, most of this kernel's time: one workgroup per CU, nothing else to switch to)
        constexpr int NB = 8;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * NB) {
          float4 a[NB], b[NB];
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const int ii = min(i0 + k * 256, n4 - 1);
            a[k] = sx[ii];
            b[k] = sd[ii];
          }
#pragma unroll
          for (int k = 0; k < NB; ++k)
            if (i0 + k * 256 < n4) {
              dx4[i0 + k * 256] = a[k];
              dd4[i0 + k * 256] = b[k];
            }
        }
      }
      __syncthreads();
      if (c < C) {
        for (int y = rg; y < HW; y += G) {
          float g[HW];
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            g[xx] = ds[(y * HW + xx) * C + c];
            ab += g[xx];
          }
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= HW) continue;
            float in[HW];
#pragma unroll
            for (int xx = 0; xx < HW; ++xx) in[xx] = xs[(iy * HW + xx) * C + c];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int xx = 0; xx < HW; ++xx) {
                const int ix = xx + kx - 3;
                if (ix >= 0 && ix < HW) acc[ky * 7 + kx] = fmaf(g[xx], in[ix], acc[ky * 7 + kx]);
              }
          }
        }
      }
    }
    if (c < C) {   // this workgroup's partial row: [C][49] filter taps, then [C] biases
      float* row = dw + ((size_t)blockIdx.x * G + rg) * 50 * C;
#pragma unroll
      for (int t = 0; t < 49; ++t) row[(size_t)c * 49 + t] = acc[t];
      row[(size_t)49 * C + c] = ab;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img,
                                                          T* __restrict__ patches, int B) {
  const long n = (long)B * 225 * 48;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int k = (int)(i % 48);
    const long r = i / 48;
    const int p = (int)(r % 225), b = (int)(r / 225);
    const int ci = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
    const int py = p / 15, px = p - py * 15;
    patches[i] = (T)img[((size_t)b * 3 + ci) * 3969 + (4 * py + ky) * 63 + 4 * px + kx];
  }
}

inline int gridn(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

template <typename T>
int wgrad_t(const void* D, const void* A, float* out, int M, int N, int K, int ldo,
            hipStream_t st) {
  // slices of the reduction: enough workgroups to fill the chip, at least 512 rows each
  const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
  int nsl = (1024 + tiles - 1) / tiles;
  if (nsl > (M + 511) / 512) nsl = (M + 511) / 512;
  if (nsl < 1) nsl = 1;
  int mslice = ((M + nsl - 1) / nsl + 63) / 64 * 64;
  nsl = (M + mslice - 1) / mslice;
  dim3 grid((N + 63) / 64, (K + 63) / 64, nsl);
  hipLaunchKernelGGL(wgrad_kernel<T>, grid, dim3(256), 0, st, reinterpret_cast<const T*>(D),
                     reinterpret_cast<const T*>(A), out, M, N, K, ldo, mslice);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <int TN, int TK>
int wgrad_f32_launch(const float* D, const float* A, float* out, int M, int N, int K, int ldo, hipStream_t st, float* cs) {
  // slices of the reduction: ~768 workgroups (two or three per CU), at least 256 rows each
  const int tiles = (N / TN) * (K / TK);
  int nsl = (768 + tiles - 1) / tiles;
  if (nsl > (M + 255) / 256) nsl = (M + 255) / 256;
  if (nsl < 1) nsl = 1;
  const int mslice = ((M + nsl - 1) / nsl + 15) / 16 * 16;
  nsl = (M + mslice - 1) / mslice;
  hipLaunchKernelGGL((wgrad_f32_kernel<TN, TK>), dim3(N / TN, K / TK, nsl), dim3(64 * (TN / 64) * (TK / 64)), 0, st, D, A, out, M, N,
                     K, ldo, mslice, cs);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// fp32: the matrix-pipe kernel where the shapes allow it; *cs_done says whether the column sums went along
int wgrad_f32(const void* Dv, const void* Av, float* out, int M, int N, int K, int ldo, hipStream_t st, float* cs, bool* cs_done) {
  const float* D = reinterpret_cast<const float*>(Dv);
  const float* A = reinterpret_cast<const float*>(Av);
  if (cs_done != nullptr) *cs_done = false;
  static const bool old_form = [] {   // BTSBOT_AMD_WGRAD_F32_OLD=1: the 64 x 64 kernel for every shape (A/B timing, parity)
    const char* e = getenv("BTSBOT_AMD_WGRAD_F32_OLD");
    return e != nullptr && e[0] == '1';
  }();
  const bool aligned = ((reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(A)) & 15) == 0;
  if (!old_form && aligned && N % 64 == 0 && K % 64 == 0) {
    // (deterministic mode: the column sums keep their own launch with its fixed-order reduction)
    float* csk = cs != nullptr && cs_done != nullptr && det_alloc(0) == nullptr ? cs : nullptr;
    if (csk != nullptr) *cs_done = true;
    const bool n128 = N % 128 == 0, k128 = K % 128 == 0;
    if (n128 && k128) return wgrad_f32_launch<128, 128>(D, A, out, M, N, K, ldo, st, csk);
    if (n128) return wgrad_f32_launch<128, 64>(D, A, out, M, N, K, ldo, st, csk);
    if (k128) return wgrad_f32_launch<64, 128>(D, A, out, M, N, K, ldo, st, csk);
    return wgrad_f32_launch<64, 64>(D, A, out, M, N, K, ldo, st, csk);
  }
  const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
  int nsl = (1024 + tiles - 1) / tiles;
  if (nsl > (M + 511) / 512) nsl = (M + 511) / 512;
  if (nsl < 1) nsl = 1;
  int mslice = ((M + nsl - 1) / nsl + 63) / 64 * 64;
  nsl = (M + mslice - 1) / mslice;
  dim3 grid((N + 63) / 64, (K + 63) / 64, nsl);
  hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), 0, st, D, A, out, M, N, K, ldo, mslice);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
template <>
int wgrad_t<float>(const void* D, const void* A, float* out, int M, int N, int K, int ldo, hipStream_t st) {
  return wgrad_f32(D, A, out, M, N, K, ldo, st, nullptr, nullptr);
}

template <typename T>
int colsum_t(const void* in, float* out, int M, int N, hipStream_t st) {
  const int nb = (N + 63) / 64;
  int nsl = (768 + nb - 1) / nb;                 // ~768 blocks in flight
  if (nsl > 64) nsl = 64;                        // ... but at most 64 atomics per column
  if (nsl > (M + 63) / 64) nsl = (M + 63) / 64;
  if (nsl < 1) nsl = 1;
  const int mslice = (M + nsl - 1) / nsl;
  dim3 grid(nb, (M + mslice - 1) / mslice);
  float* part = det_alloc((size_t)grid.y * N);
  hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, st, reinterpret_cast<const T*>(in), out,
                     M, N, mslice, part);
  LAUNCH_CHECK();
  if (part != nullptr) {
    const DetOut o{out, 1};
    return launch_det_reduce(part, (int)grid.y, N, 1, &o, st);
  }
  return BTSBOT_OK;
}

}  // namespace

#define BY_PREC(prec, CALL_F32, CALL_BF16, CALL_F16)            \
  switch (prec) {                                               \
    case BTSBOT_F32: return CALL_F32;                           \
    case BTSBOT_BF16: return CALL_BF16;                         \
    case BTSBOT_F16: return CALL_F16;                           \
  }                                                             \
  btsbot_set_error("backward: bad precision %d", prec);         \
  return BTSBOT_ERR_INVALID_ARG;

// fp32 operands: filter gradient and (cs != nullptr) the column sums of D, in one launch where the matrix-pipe kernel applies
int launch_wgrad_cs_f32(const float* D, const float* A, float* out, float* cs, int M, int N, int K, int ldo, hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  bool done = false;
  const int rc = wgrad_f32(D, A, out, M, N, K, ldo, st, cs, &done);
  if (rc != BTSBOT_OK || cs == nullptr || done) return rc;
  return launch_colsum(BTSBOT_F32, D, cs, M, N, st);
}

int launch_wgrad(int prec, const void* D, const void* A, float* out, int M, int N, int K, int ldo,
                 hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  BY_PREC(prec, wgrad_t<float>(D, A, out, M, N, K, ldo, st),
          wgrad_t<bf16_t>(D, A, out, M, N, K, ldo, st), wgrad_t<f16_t>(D, A, out, M, N, K, ldo, st))
}

// ---- deterministic mode: the scratch of the running backward call (thread-local: launchers run on the caller's thread)
namespace {
thread_local float* g_det_base = nullptr;
thread_local size_t g_det_cap = 0, g_det_used = 0;
thread_local bool g_det_short = false;   // a request did not fit the scratch: that reduction fell back to atomics

// 32 columns x 8 row groups per workgroup: group q adds rows q, q + 8, ... in order, the groups meet in LDS in order
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ part, int nrows, int C, int W,
                                                         DetOut o0, DetOut o1, DetOut o2, DetOut o3) {
  __shared__ float sh[8][32];
  const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (col < W)
    for (int r = q; r < nrows; r += 8) s += part[(size_t)r * W + col];
  sh[q][cl] = s;
  __syncthreads();
  if (q == 0 && col < W) {
    float t = sh[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += sh[i][cl];
    const int o = col / C, c = col - o * C;
    const DetOut d = o == 0 ? o0 : o == 1 ? o1 : o == 2 ? o2 : o3;
    d.ptr[(size_t)c * d.stride] += t;
  }
}
}  // namespace

void det_begin(float* base, size_t floats) {
  g_det_base = base;
  g_det_cap = base != nullptr ? floats : 0;
  g_det_used = 0;
}
void det_end() { det_begin(nullptr, 0); }
bool det_fell_short() {   // since the last call: did any reduction of this thread's backward not fit the scratch?
  const bool v = g_det_short;
  g_det_short = false;
  return v;
}
float* det_alloc(size_t floats) {
  if (g_det_base == nullptr) return nullptr;
  if (g_det_used + floats > g_det_cap) {
    g_det_short = true;
    return nullptr;
  }
  float* p = g_det_base + g_det_used;
  g_det_used += (floats + 63) / 64 * 64;
  return p;
}
int launch_det_reduce(const float* part, int nrows, int C, int nout, const DetOut* outs, hipStream_t st) {
  if (nout < 1 || nout > 4) {
    btsbot_set_error("det_reduce: %d outputs", nout);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int W = nout * C;
  DetOut o[4] = {outs[0], outs[0], outs[0], outs[0]};
  for (int i = 0; i < nout; ++i) o[i] = outs[i];
  hipLaunchKernelGGL(det_reduce_kernel, dim3((W + 31) / 32), dim3(256), 0, st, part, nrows, C, W, o[0], o[1], o[2], o[3]);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_colsum(int prec, const void* in, float* out, int M, int N, hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  BY_PREC(prec, colsum_t<float>(in, out, M, N, st), colsum_t<bf16_t>(in, out, M, N, st),
          colsum_t<f16_t>(in, out, M, N, st))
}

int launch_scale_cast(int prec, const float* in, const float* scale, void* out, long n, int C,
                      hipStream_t st) {
  if (n <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(scale_cast_kernel<float>, dim3(gridn(n)), dim3(256), 0, st, in, scale,
                         reinterpret_cast<float*>(out), n, C);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, dim3(gridn(n)), dim3(256), 0, st, in, scale,
                         reinterpret_cast<bf16_t*>(out), n, C);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(scale_cast_kernel<f16_t>, dim3(gridn(n)), dim3(256), 0, st, in, scale,
                         reinterpret_cast<f16_t*>(out), n, C);
      break;
    default:
      btsbot_set_error("scale_cast: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

__global__ __launch_bounds__(256) void rowscale_cast_lo_kernel(const float* __restrict__ in, const float* __restrict__ rowscale,
                                                              f16_t* __restrict__ out, long n, int cols) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = rowscale != nullptr ? in[i] * rowscale[i / cols] : in[i];
  const f16_t hi = (f16_t)v;
  out[i] = (f16_t)(v - (float)hi);
}

int launch_rowscale_cast_lo(const float* in, const float* rowscale, void* out, int rows, int cols, hipStream_t st) {
  const long n = (long)rows * cols;
  if (n <= 0) return BTSBOT_OK;
  hipLaunchKernelGGL(rowscale_cast_lo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, rowscale,
                     reinterpret_cast<f16_t*>(out), n, cols);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_rowscale_cast(int prec, const float* in, const float* rowscale, void* out, int rows,
                         int cols, hipStream_t st) {
  const long n = (long)rows * cols;
  if (n <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(rowscale_cast_kernel<float>, dim3(gridn(n)), dim3(256), 0, st, in, rowscale,
                         reinterpret_cast<float*>(out), n, cols);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(rowscale_cast_kernel<bf16_t>, dim3(gridn(n)), dim3(256), 0, st, in, rowscale,
                         reinterpret_cast<bf16_t*>(out), n, cols);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(rowscale_cast_kernel<f16_t>, dim3(gridn(n)), dim3(256), 0, st, in, rowscale,
                         reinterpret_cast<f16_t*>(out), n, cols);
      break;
    default:
      btsbot_set_error("rowscale_cast: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_fc2_grads(float* G, float* S, const float* w2, const float* b2,
                     const float* gamma, float* dW2, float* db2, float* dgamma, int C, int H,
                     hipStream_t st) {
  hipLaunchKernelGGL(fc2_grads_kernel, dim3(C), dim3(256), 0, st, G, S, w2, b2, gamma, dW2, db2,
                     dgamma, C, H);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// Narrow maps (C = 4 * LPR, LPR = 16 or 32 lanes per pixel row): every lane owns 4 consecutive
// channels (float4 loads), a wave normalises 64 / LPR rows at once and runs two such groups per
// iteration, so that 4 x 16-byte loads per lane are in flight instead of one dword.
template <int LPR>
__global__ __launch_bounds__(256) void ln_bwd_narrow_kernel(const float* __restrict__ d,
                                                            const float* __restrict__ dxn,
                                                            const float* __restrict__ g, float* dd,
                                                            float* dg, float* dbeta, long rows,
                                                            void* __restrict__ out16, int prec16, int patch_hw,
                                                            float* part) {
  constexpr int C = 4 * LPR, R = 64 / LPR;       // rows per wave pass
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const float4 gl = *reinterpret_cast<const float4*>(g + 4 * l);
  float adg[4] = {0.f, 0.f, 0.f, 0.f}, adb[4] = {0.f, 0.f, 0.f, 0.f};
  auto gsum = [](float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
  const long stride = (long)gridDim.x * 4 * R * 2;
  // (dd may be dxn: the next pass's pieces are requested before this pass's stores, as in ln_bwd_kernel)
  float4 nv[2], ndx[2];
  auto fetch = [&](long r0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long r = r0 + u * R + sub;
      const long rr = r < rows ? r : 0;
      nv[u] = *reinterpret_cast<const float4*>(d + rr * C + 4 * l);
      bool zero;
      const float* src = ln_bwd_src(dxn, rr, C, patch_hw, &zero);
      const float4 g4 = *reinterpret_cast<const float4*>(src + 4 * l);
      ndx[u] = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : g4;
    }
  };
  const long rfirst = ((long)blockIdx.x * 4 + wv) * R * 2;
  if (rfirst < rows) fetch(rfirst);
  for (long r0 = rfirst; r0 < rows; r0 += stride) {
    float4 v[2], dx[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ok[u] = r0 + u * R + sub < rows;
      v[u] = nv[u];
      dx[u] = ndx[u];
    }
    if (r0 + stride < rows) fetch(r0 + stride);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long r = r0 + u * R + sub;
      float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      const float dxa[4] = {dx[u].x, dx[u].y, dx[u].z, dx[u].w};
      const float ga[4] = {gl.x, gl.y, gl.z, gl.w};
      const float mean = gsum((x[0] + x[1]) + (x[2] + x[3])) * (1.f / C);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] -= mean;
        q += x[i] * x[i];
      }
      const float rstd = rsqrtf(gsum(q) * (1.f / C) + LN_EPS);
      float t[4], st = 0.f, stx = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] *= rstd;
        t[i] = dxa[i] * ga[i];
        st += t[i];
        stx += t[i] * x[i];
        if (ok[u]) {
          adg[i] += dxa[i] * x[i];
          adb[i] += dxa[i];
        }
      }
      st = gsum(st) * (1.f / C);
      stx = gsum(stx) * (1.f / C);
      if (ok[u]) {
        const float4 o = make_float4(rstd * (t[0] - st - x[0] * stx), rstd * (t[1] - st - x[1] * stx),
                                     rstd * (t[2] - st - x[2] * stx), rstd * (t[3] - st - x[3] * stx));
        *reinterpret_cast<float4*>(dd + r * C + 4 * l) = o;
        if (out16 != nullptr) {   // the same values as the next 16-bit GEMM's operand (was a cast launch)
          if (prec16 == BTSBOT_BF16) {
            typedef bf16_t __attribute__((ext_vector_type(4))) b4;
            *reinterpret_cast<b4*>(reinterpret_cast<bf16_t*>(out16) + r * C + 4 * l) =
                b4{(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
          } else {
            typedef f16_t __attribute__((ext_vector_type(4))) h4;
            *reinterpret_cast<h4*>(reinterpret_cast<f16_t*>(out16) + r * C + 4 * l) =
                h4{(f16_t)o.x, (f16_t)o.y, (f16_t)o.z, (f16_t)o.w};
          }
        }
      }
    }
  }
  // row groups of the wave and the 4 waves meet in LDS: one atomic per channel per block
  __shared__ float sh[2][4 * R][C];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sh[0][wv * R + sub][4 * l + i] = adg[i];
    sh[1][wv * R + sub][4 * l + i] = adb[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 4 * R; ++j) {
      a += sh[0][j][c];
      b += sh[1][j][c];
    }
    det_add(dg + c, a, part, (size_t)blockIdx.x * 2 * C + c);
    det_add(dbeta + c, b, part, (size_t)blockIdx.x * 2 * C + C + c);
  }
}

int launch_ln_bwd(const float* d, const float* dxn, const float* g, float* dd, float* dg,
                  float* dbeta, long rows, int C, hipStream_t st, void* out16, int prec16, int patch_hw) {
  if (rows <= 0) return BTSBOT_OK;
  static const long cap = [] {
    const char* e = getenv("BTSBOT_AMD_LNBWD_BLOCKS");   // tuning knob: workgroups (= same-address atomics per channel;
    const long v = e ? atol(e) : 0;                      //  measured per step: 256 4.64 ms, 384/512 4.58, 1024 4.70, 2048 5.02)
    return v >= 1 ? v : 512;
  }();
  long blocks = (rows + 3) / 4;
  if (blocks > cap) blocks = cap;     // <= cap same-address atomics per channel
  if (C == 64 || C == 128) {
    const int rpb = 4 * (C == 64 ? 4 : 2) * 2;   // rows per block pass
    long nb = (rows + rpb - 1) / rpb;
    if (nb > cap) nb = cap;
    float* part = det_alloc((size_t)nb * 2 * C);
    if (C == 64)
      hipLaunchKernelGGL((ln_bwd_narrow_kernel<16>), dim3((unsigned)nb), dim3(256), 0, st, d, dxn, g,
                         dd, dg, dbeta, rows, out16, prec16, patch_hw, part);
    else
      hipLaunchKernelGGL((ln_bwd_narrow_kernel<32>), dim3((unsigned)nb), dim3(256), 0, st, d, dxn, g,
                         dd, dg, dbeta, rows, out16, prec16, patch_hw, part);
    LAUNCH_CHECK();
    if (part != nullptr) {
      const DetOut o[2] = {{dg, 1}, {dbeta, 1}};
      return launch_det_reduce(part, (int)nb, C, 2, o, st);
    }
    return BTSBOT_OK;
  }
  const int cpt = (C + 63) / 64;
  float* part = det_alloc((size_t)blocks * 2 * C);
#define LNB(CPT)                                                                                \
  hipLaunchKernelGGL((ln_bwd_kernel<CPT>), dim3((unsigned)blocks), dim3(256), 0, st, d, dxn, g, \
                     dd, dg, dbeta, rows, C, out16, prec16, patch_hw, part)
  if (cpt <= 1) LNB(1);
  else if (cpt <= 2) LNB(2);
  else if (cpt <= 4) LNB(4);
  else if (cpt <= 5) LNB(5);
  else if (cpt <= 8) LNB(8);
  else if (cpt <= 10) LNB(10);
  else {
    btsbot_set_error("ln_bwd: C=%d too wide", C);
    return BTSBOT_ERR_INVALID_ARG;
  }
#undef LNB
  LAUNCH_CHECK();
  if (part != nullptr) {
    const DetOut o[2] = {{dg, 1}, {dbeta, 1}};
    return launch_det_reduce(part, (int)blocks, C, 2, o, st);
  }
  return BTSBOT_OK;
}

// w: taps [49][C] (the centre row is read); dw: filter gradient [C][49], dbias [C] (master-arena layout)
int launch_ln_dw1_bwd(const float* d, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                      void* out16, int prec16, float* dg, float* dbeta, float* dw, float* dbias, long rows, int C,
                      hipStream_t st) {
  if (rows <= 0) return BTSBOT_OK;
  long blocks = (rows + 3) / 4;
  if (blocks > 32) blocks = 32;   // 4 x <= 32 same-address atomics per channel: with 256 workgroups the step was 0.04 ms slower
  const float* wc = w + (size_t)24 * C;
  const int cpt = (C + 63) / 64;
  float* part = det_alloc((size_t)blocks * 4 * C);
#define LD1(CPT)                                                                                       \
  hipLaunchKernelGGL((ln_dw1_bwd_kernel<CPT>), dim3((unsigned)blocks), dim3(256), 0, st, d, dxn, g, xin, wc, dy, \
                     out16, prec16, dg, dbeta, dw, dbias, rows, C, part)
  if (cpt <= 8) LD1(8);
  else if (cpt <= 10) LD1(10);
  else {
    btsbot_set_error("ln_dw1_bwd: C=%d too wide", C);
    return BTSBOT_ERR_INVALID_ARG;
  }
#undef LD1
  LAUNCH_CHECK();
  if (part != nullptr) {
    const DetOut o[4] = {{dg, 1}, {dbeta, 1}, {dw + 24, 49}, {dbias, 1}};
    return launch_det_reduce(part, (int)blocks, C, 4, o, st);
  }
  return BTSBOT_OK;
}

int launch_dw_plain(const float* x, const float* w, int flip, const float* bias,
                    const float* addend, float* out, int B, int HW, int C, hipStream_t st,
                    void* out16, int prec16) {
  if (B <= 0) return BTSBOT_OK;
  const size_t lds = ((size_t)HW * HW + 49) * C * sizeof(float);
#define DWP(H)                                                                                 \
  {                                                                                            \
    static size_t attr = 0;                                                                    \
    if (lds > attr) {                                                                          \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dw_plain_kernel<H>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      attr = lds;                                                                              \
    }                                                                                          \
    hipLaunchKernelGGL((dw_plain_kernel<H>), dim3(B), dim3(256), lds, st, x, w, flip, bias,    \
                       addend, out, C, out16, prec16);                                         \
  }
  if (HW == 15) DWP(15)
  else if (HW == 7) DWP(7)
  else if (HW == 3) DWP(3)
  else if (HW == 1) DWP(1)
  else {
    btsbot_set_error("dw_plain: no kernel for HW=%d", HW);
    return BTSBOT_ERR_INVALID_ARG;
  }
#undef DWP
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_dw_wgrad(const float* x, const float* dd, float* dw, float* dbias, float* partials,
                    int B, int HW, int C, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (dbias != dw + (size_t)49 * C) {
    btsbot_set_error("dw_wgrad: filter and bias gradients must be adjacent in the arena");
    return BTSBOT_ERR_INVALID_ARG;
  }
  const size_t lds = 2 * (size_t)HW * HW * C * sizeof(float);
  int ga = (B + 255) / 256;            // <= 256 workgroups => <= 256 atomics per filter tap
  if (ga < 1) ga = 1;
  const int grid = (B + ga - 1) / ga;
#define DWW(H)                                                                                 \
  {                                                                                            \
    static size_t attr = 0;                                                                    \
    if (lds > attr) {                                                                          \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dw_wgrad_kernel<H>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      attr = lds;                                                                              \
    }                                                                                          \
    hipLaunchKernelGGL((dw_wgrad_kernel<H>), dim3(grid), dim3(256), lds, st, x, dd, partials,  \
                       dbias, B, C, ga);                                                       \
  }
  if (HW == 15) DWW(15)
  else if (HW == 7) DWW(7)
  else if (HW == 3) DWW(3)
  else if (HW == 1) DWW(1)
  else {
    btsbot_set_error("dw_wgrad: no kernel for HW=%d", HW);
    return BTSBOT_ERR_INVALID_ARG;
  }
#undef DWW
  LAUNCH_CHECK();
  return colsum_t<float>(partials, dw, grid * (C < 256 ? 256 / C : 1), 50 * C, st);
}

int launch_stem_im2col(int prec, const float* img, void* patches, int B, hipStream_t st) {
  const long n = (long)B * 225 * 48;
  if (n <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(gridn(n)), dim3(256), 0, st, img,
                         reinterpret_cast<float*>(patches), B);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(gridn(n)), dim3(256), 0, st, img,
                         reinterpret_cast<bf16_t*>(patches), B);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(stem_im2col_kernel<f16_t>, dim3(gridn(n)), dim3(256), 0, st, img,
                         reinterpret_cast<f16_t*>(patches), B);
      break;
    default:
      btsbot_set_error("stem_im2col: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
