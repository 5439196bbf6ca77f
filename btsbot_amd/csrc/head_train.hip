#include <stdlib.h>
// Training-mode heads (fp32): forward with BatchNorm1d batch statistics + dropout, and the backward
// pass of the fusion head and the metadata branch.
//
// Reference: the nn.Sequential heads of /root/reference/btsbot/architectures.py:146-164 (mm_ConvNeXt),
// :282-290 (um_nn), :358-365 (frozen_fusion) under model.train(), and loss.backward() at
// /root/reference/btsbot/train.py:525-526.  For frozen_fusion only `combined_head` has
// requires_grad (train.py:224-232) -- but its frozen branches still run in train mode, so the
// metadata BatchNorm uses batch statistics and keeps updating its running stats; reproduced here.
//
// Sizes are tiny (B x {25,128,128,640,128,32,1}); kernels are plain fp32 tiles, one launch per
// layer, deterministic summation order (no atomics).
#include <algorithm>

#include "ctx.h"

#define TRY_RET(call)               \
  do {                              \
    int _s = (call);                \
    if (_s != BTSBOT_OK) return _s; \
  } while (0)

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f, HN_EPS = 1e-6f;

__device__ __forceinline__ float act_fwd(float x, int act) { return apply_act(x, act); }
__device__ __forceinline__ float act_bwd(float x, int act) {
  if (act == ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == ACT_GELU) {  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
  }
  return 1.f;
}

// out_pre[m][n] = bias[n] + sum_k in[m*ldi + k] * wt[k*N + n];  out_act = dropout(act(pre))
__global__ __launch_bounds__(256) void lin_fwd_kernel(const float* __restrict__ in, int ldi,
                                                      const float* __restrict__ wt,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ pre, float* outa, int ldo,
                                                      int M, int N, int K, int act,
                                                      const uint8_t* __restrict__ mask,
                                                      float keep_scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int m = idx / N, n = idx - m * N;
  const float* ip = in + (size_t)m * ldi;
  const float* wp = wt + n;
  float a0 = bias[n], a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four chains, two groups per trip: 16 loads in flight
  int k = 0;
#pragma unroll 2
  for (; k + 3 < K; k += 4) {
    a0 = fmaf(ip[k], wp[(size_t)k * N], a0);
    a1 = fmaf(ip[k + 1], wp[(size_t)(k + 1) * N], a1);
    a2 = fmaf(ip[k + 2], wp[(size_t)(k + 2) * N], a2);
    a3 = fmaf(ip[k + 3], wp[(size_t)(k + 3) * N], a3);
  }
  for (; k < K; ++k) a0 = fmaf(ip[k], wp[(size_t)k * N], a0);
  float acc = (a0 + a1) + (a2 + a3);
  if (pre != nullptr) pre[idx] = acc;
  float v = act_fwd(acc, act);
  if (mask != nullptr) v = mask[idx] ? v * keep_scale : 0.f;
  outa[(size_t)m * ldo + n] = v;
}

// outa[m*ldo + n] = dropout(act(pre[m][n]))      (behind the MFMA GEMM that wrote `pre` for a wide layer)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ pre, float* __restrict__ outa, int ldo,
                                                      int M, int N, int act, const uint8_t* __restrict__ mask,
                                                      float keep_scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int m = idx / N, n = idx - m * N;
  float v = act_fwd(pre[idx], act);
  if (mask != nullptr) v = mask[idx] ? v * keep_scale : 0.f;
  outa[(size_t)m * ldo + n] = v;
}

// A fusion layer wide enough for the matrix pipe (the 1024 x 640 x 128 first layer: 42 us forward, 28 us input gradient
// as one thread per output walking K) goes through gemm.hip's exact-fp32 MFMA GEMM: both operands are K-contiguous as
// stored ([N][K] filter for the forward, the packed transpose [K][N] for the input gradient).
static bool wide_layer(int M, int N, int K, int ldi) {
  static const bool off = [] {
    const char* e = getenv("BTSBOT_AMD_HEAD_NO_GEMM");   // 1: every head layer on the per-output kernels (A/B)
    return e != nullptr && e[0] == '1';
  }();
  return !off && ldi == K && K % 4 == 0 && N % 4 == 0 && (long)M * N * K >= (1L << 24);
}

// dpre[m][n] = dout[m*ldd + n] * act'(pre[m][n]) * mask * keep_scale     (in place allowed)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dout, int ldd,
                                                      const float* __restrict__ pre, float* dpre,
                                                      int M, int N, int act,
                                                      const uint8_t* __restrict__ mask,
                                                      float keep_scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * N) return;
  const int m = idx / N, n = idx - m * N;
  float d = dout[(size_t)m * ldd + n];
  if (mask != nullptr) d = mask[idx] ? d * keep_scale : 0.f;
  dpre[idx] = d * act_bwd(pre[idx], act);
}

// din[m*ldi + k] = sum_n dpre[m][n] * w[n][k]      (w in its natural [N][K] layout)
__global__ __launch_bounds__(256) void lin_bwd_in_kernel(const float* __restrict__ dpre,
                                                         const float* __restrict__ w, float* din,
                                                         int ldi, int M, int N, int K) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * K) return;
  const int m = idx / K, k = idx - m * K;
  const float* dp = dpre + (size_t)m * N;
  const float* wp = w + k;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four chains, two groups per trip (one chain, one load pair in
  int n = 0;                                       //  flight per trip: 64 us for the 1024 x 640 x 128 fusion layer)
#pragma unroll 2
  for (; n + 3 < N; n += 4) {
    a0 = fmaf(dp[n], wp[(size_t)n * K], a0);
    a1 = fmaf(dp[n + 1], wp[(size_t)(n + 1) * K], a1);
    a2 = fmaf(dp[n + 2], wp[(size_t)(n + 2) * K], a2);
    a3 = fmaf(dp[n + 3], wp[(size_t)(n + 3) * K], a3);
  }
  for (; n < N; ++n) a0 = fmaf(dp[n], wp[(size_t)n * K], a0);
  din[(size_t)m * ldi + k] = (a0 + a1) + (a2 + a3);
}

// dw[n][k] = sum_m dpre[m][n] * in[m*ldi + k];  db[n] = sum_m dpre[m][n]
// Workgroup = OUTS consecutive outputs x SL interleaved slices of m (OUTS * SL = 256); the slices meet in LDS in a
// fixed order, so the result is deterministic (no atomics).  16 x 16 for wide layers; 4 x 64 for the first metadata
// layer (128 x 26 outputs: with 16 slices its 208 workgroups walked 64 rows each behind exposed load latencies).
template <int OUTS, int SL>
__global__ __launch_bounds__(256) void lin_bwd_w_kernel(const float* __restrict__ dpre,
                                                        const float* __restrict__ in, int ldi,
                                                        float* __restrict__ dw,
                                                        float* __restrict__ db, int M, int N,
                                                        int K) {
  static_assert(OUTS * SL == 256, "one thread per (output, slice)");
  __shared__ float sh[SL][OUTS + 1];
  const int o = threadIdx.x % OUTS, g = threadIdx.x / OUTS;
  const int idx = blockIdx.x * OUTS + o;
  const bool live = idx < N * (K + 1);
  const int n = live ? idx / (K + 1) : 0, k = live ? idx - n * (K + 1) : 0;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (live) {
    const float* dp = dpre + n;
    if (k < K) {
      const float* ip = in + k;
      int m = g;
      for (; m + 3 * SL < M; m += 4 * SL) {
        a0 = fmaf(dp[(size_t)m * N], ip[(size_t)m * ldi], a0);
        a1 = fmaf(dp[(size_t)(m + SL) * N], ip[(size_t)(m + SL) * ldi], a1);
        a2 = fmaf(dp[(size_t)(m + 2 * SL) * N], ip[(size_t)(m + 2 * SL) * ldi], a2);
        a3 = fmaf(dp[(size_t)(m + 3 * SL) * N], ip[(size_t)(m + 3 * SL) * ldi], a3);
      }
      for (; m < M; m += SL) a0 = fmaf(dp[(size_t)m * N], ip[(size_t)m * ldi], a0);
    } else {
      for (int m = g; m < M; m += SL) a0 += dp[(size_t)m * N];
    }
  }
  sh[g][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < OUTS && live) {
    float s = 0.f;
#pragma unroll
    for (int gg = 0; gg < SL; ++gg) s += sh[gg][o];
    if (k < K) dw[(size_t)n * K + k] = s;
    else db[n] = s;
  }
}

// The same for K % 4 == 0 (every layer but the first metadata one): a lane owns FOUR consecutive k of one n (one
// 16-byte load of `in` per row instead of four dwords, the dpre value shared by all four products), workgroup =
// 64 consecutive k of one n x 16 interleaved slices of m, which meet in LDS in a fixed order as above.
__global__ __launch_bounds__(256) void lin_bwd_w4_kernel(const float* __restrict__ dpre,
                                                         const float* __restrict__ in, int ldi,
                                                         float* __restrict__ dw,
                                                         float* __restrict__ db, int M, int N, int K) {
  __shared__ float4 sh[16][17];
  __shared__ float shb[16];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int kb = (K + 63) / 64;
  const int n = blockIdx.x / kb, k0 = (blockIdx.x - n * kb) * 64 + 4 * o;
  const bool live = k0 < K;
  const bool bias = blockIdx.x - n * kb == 0 && o == 0;   // this lane also sums dpre[:, n]
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  float b0 = 0.f;
  const float* dp = dpre + n;
  const float* ip = in + (live ? k0 : 0);
  int m = g;
  for (; m + 16 < M; m += 32) {
    const float d0 = dp[(size_t)m * N], d1 = dp[(size_t)(m + 16) * N];
    const float4 v0 = *reinterpret_cast<const float4*>(ip + (size_t)m * ldi);
    const float4 v1 = *reinterpret_cast<const float4*>(ip + (size_t)(m + 16) * ldi);
    a0.x = fmaf(d0, v0.x, a0.x); a0.y = fmaf(d0, v0.y, a0.y); a0.z = fmaf(d0, v0.z, a0.z); a0.w = fmaf(d0, v0.w, a0.w);
    a1.x = fmaf(d1, v1.x, a1.x); a1.y = fmaf(d1, v1.y, a1.y); a1.z = fmaf(d1, v1.z, a1.z); a1.w = fmaf(d1, v1.w, a1.w);
    b0 += d0 + d1;
  }
  for (; m < M; m += 16) {
    const float d0 = dp[(size_t)m * N];
    const float4 v0 = *reinterpret_cast<const float4*>(ip + (size_t)m * ldi);
    a0.x = fmaf(d0, v0.x, a0.x); a0.y = fmaf(d0, v0.y, a0.y); a0.z = fmaf(d0, v0.z, a0.z); a0.w = fmaf(d0, v0.w, a0.w);
    b0 += d0;
  }
  sh[g][o] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  if (o == 0) shb[g] = b0;
  __syncthreads();
  if (threadIdx.x < 16) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float sb = 0.f;
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) {
      const float4 v = sh[gg][o];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      sb += shb[gg];
    }
    if (live) *reinterpret_cast<float4*>(dw + (size_t)n * K + k0) = s;
    if (bias) db[n] = sb;
  }
}

static int launch_lin_bwd_w(const float* dpre, const float* in, int ldi, float* dw, float* db, int M, int N, int K,
                            hipStream_t st) {
  const bool vec = K % 4 == 0 && ldi % 4 == 0 && (((uintptr_t)in | (uintptr_t)dw) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(lin_bwd_w4_kernel, dim3(N * ((K + 63) / 64)), dim3(256), 0, st, dpre, in, ldi, dw, db, M, N, K);
  else if ((long)N * (K + 1) <= 8192)
    hipLaunchKernelGGL((lin_bwd_w_kernel<4, 64>), dim3(((long)N * (K + 1) + 3) / 4), dim3(256), 0, st, dpre, in, ldi,
                       dw, db, M, N, K);
  else
    hipLaunchKernelGGL((lin_bwd_w_kernel<16, 16>), dim3(((long)N * (K + 1) + 15) / 16), dim3(256), 0, st, dpre, in,
                       ldi, dw, db, M, N, K);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// BatchNorm1d, training: one workgroup per feature j.  Biased variance normalises, unbiased
// variance goes into running_var (torch semantics), momentum 0.1.
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(const float* __restrict__ x, int M,
                                                           int n, const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           float* run_mean, float* run_var,
                                                           float* __restrict__ xhat,
                                                           float* __restrict__ out,
                                                           float* __restrict__ rstd_out) {
  __shared__ float sh[8];
  const int j = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int m = tid; m < M; m += 256) s += x[(size_t)m * n + j];
  s = wave_sum(s);
  if ((tid & 63) == 0) sh[tid >> 6] = s;
  __syncthreads();
  const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / M;
  __syncthreads();
  float q = 0.f;
  for (int m = tid; m < M; m += 256) {
    const float d = x[(size_t)m * n + j] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) sh[4 + (tid >> 6)] = q;
  __syncthreads();
  const float ss = sh[4] + sh[5] + sh[6] + sh[7];
  const float var = ss / M;
  const float rstd = rsqrtf(var + BN_EPS);
  for (int m = tid; m < M; m += 256) {
    const float xh = (x[(size_t)m * n + j] - mean) * rstd;
    xhat[(size_t)m * n + j] = xh;
    out[(size_t)m * n + j] = xh * w[j] + b[j];
  }
  if (tid == 0) {
    rstd_out[j] = rstd;
    if (run_mean != nullptr) {
      run_mean[j] = (1.f - BN_MOM) * run_mean[j] + BN_MOM * mean;
      run_var[j] = (1.f - BN_MOM) * run_var[j] + BN_MOM * (M > 1 ? ss / (M - 1) : var);
    }
  }
}

// dx = (w * rstd / M) * (M*dy - sum(dy) - xhat * sum(dy*xhat));  dw = sum(dy*xhat);  db = sum(dy)
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(const float* __restrict__ dy, int M,
                                                           int n, const float* __restrict__ w,
                                                           const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd,
                                                           float* __restrict__ dw,
                                                           float* __restrict__ db) {
  __shared__ float sh[8];
  const int j = blockIdx.x, tid = threadIdx.x;
  float s = 0.f, q = 0.f;
  for (int m = tid; m < M; m += 256) {
    const float d = dy[(size_t)m * n + j];
    s += d;
    q += d * xhat[(size_t)m * n + j];
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((tid & 63) == 0) {
    sh[tid >> 6] = s;
    sh[4 + (tid >> 6)] = q;
  }
  __syncthreads();
  if (tid == 0) {
    db[j] = sh[0] + sh[1] + sh[2] + sh[3];
    dw[j] = sh[4] + sh[5] + sh[6] + sh[7];
  }
  (void)w;
  (void)rstd;   // dx of the metadata INPUT is never needed (the inputs are data)
}

// rows of feat -> (optional LayerNorm) -> z[:, 0:F]
__global__ __launch_bounds__(256) void feat_rows_kernel(const float* __restrict__ feat, int F,
                                                        const float* __restrict__ hw,
                                                        const float* __restrict__ hb, float* z,
                                                        int ldz, int M) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* src = feat + (size_t)m * F;
  float mean = 0.f, rstd = 1.f;
  if (hw != nullptr) {
    float s = 0.f;
    for (int c = lane; c < F; c += 64) s += src[c];
    mean = wave_sum(s) / F;
    float q = 0.f;
    for (int c = lane; c < F; c += 64) {
      const float d = src[c] - mean;
      q += d * d;
    }
    rstd = rsqrtf(wave_sum(q) / F + HN_EPS);
  }
  for (int c = lane; c < F; c += 64)
    z[(size_t)m * ldz + c] = hw != nullptr ? (src[c] - mean) * rstd * hw[c] + hb[c] : src[c];
}

__global__ void logits_kernel(const float* __restrict__ in, float* logits, float* scores, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float z = in[i];
  logits[i] = z;
  if (scores != nullptr) scores[i] = 1.0f / (1.0f + expf(-z));
}

// dst[m][0:cols] = src[m][0:cols]  (rows ldd / lds floats apart).  A kernel, not hipMemcpy2DAsync: the runtime's
// rectangular device-to-device copy held the calling thread until the stream had drained -- once per step the host lost
// its whole lead over the GPU, and the backward's first ~25 launches then arrived one host round trip apart.
__global__ __launch_bounds__(256) void copy_cols_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src,
                                                        int lds, int cols, int M) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * cols) return;
  const int m = idx / cols, c = idx - m * cols;
  dst[(size_t)m * ldd + c] = src[(size_t)m * lds + c];
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// ---------------------------------------------------------------------------------------
// cache layout (floats) for a batch of M alerts
// ---------------------------------------------------------------------------------------
// Row width of the three backward scratch buffers: the widest tensor any of them ever holds -- d(z) of the
// concatenated fusion input, a fusion layer's input gradient, or the metadata branch's d(a2) / d(h1) / d(x).
static size_t train_dwidth(const btsbot_ctx* h) {
  const btsbot_config& c = h->cfg;
  size_t w = 64;
  for (int i = 0; i <= h->n_comb; ++i) w = std::max(w, (size_t)h->comb_dims[i]);
  if (h->has_image) w = std::max(w, (size_t)c.dims[3]);
  if (h->has_meta) w = std::max(std::max(w, (size_t)c.n_meta), std::max((size_t)c.meta_fc1, (size_t)c.meta_fc2));
  return (w + 3) / 4 * 4;
}

size_t train_cache_floats(const btsbot_ctx* h, int M) {
  const btsbot_config& c = h->cfg;
  const size_t F = h->has_image ? c.dims[3] : 0;
  size_t n = 0;
  n += (size_t)M * F;                                   // raw backbone features
  if (h->has_meta) n += (size_t)M * (2 * c.n_meta + 2 * c.meta_fc1 + c.meta_fc2) + c.n_meta;
  n += (size_t)M * h->comb_dims[0];                     // z
  for (int i = 0; i < h->n_comb; ++i) n += 2 * (size_t)M * h->comb_dims[i + 1];   // pre + act
  n += 5 * (size_t)M * train_dwidth(h);                 // backward scratch (d-buffers)
  return n + 1024;
}

struct TrainPtrs {
  float *feat, *xhat, *x1, *bn_rstd, *a1, *h1, *a2, *z, *pre[3], *actv[3], *dbuf[5];
};

static TrainPtrs carve(const btsbot_ctx* h, float* base, int M) {
  const btsbot_config& c = h->cfg;
  const size_t F = h->has_image ? c.dims[3] : 0;
  TrainPtrs p;
  float* cur = base;
  auto take = [&](size_t n) {
    float* r = cur;
    cur += (n + 3) / 4 * 4;
    return r;
  };
  p.feat = take((size_t)M * F);
  p.xhat = p.x1 = p.bn_rstd = p.a1 = p.h1 = p.a2 = nullptr;
  if (h->has_meta) {
    p.xhat = take((size_t)M * c.n_meta);
    p.x1 = take((size_t)M * c.n_meta);
    p.bn_rstd = take(c.n_meta);
    p.a1 = take((size_t)M * c.meta_fc1);
    p.h1 = take((size_t)M * c.meta_fc1);
    p.a2 = take((size_t)M * c.meta_fc2);
  }
  p.z = take((size_t)M * h->comb_dims[0]);
  for (int i = 0; i < 3; ++i) p.pre[i] = p.actv[i] = nullptr;
  for (int i = 0; i < h->n_comb; ++i) {
    p.pre[i] = take((size_t)M * h->comb_dims[i + 1]);
    p.actv[i] = take((size_t)M * h->comb_dims[i + 1]);
  }
  const size_t dw = train_dwidth(h);
  for (int i = 0; i < 5; ++i) p.dbuf[i] = take((size_t)M * dw);
  return p;
}

// The metadata branch's training forward (BatchNorm1d batch statistics -> fc1 -> act -> dropout -> fc2 (-> act)) into the
// cache and columns F.. of the concatenated fusion input z.  It reads nothing of the image branch: btsbot_forward_train
// queues it on the side stream, beside the backbone, and head_train_forward(meta_done = true) picks up behind a join.
int head_train_meta_forward(btsbot_ctx* h, float* cache, const float* meta, int M, const uint8_t* meta_mask, float* master,
                            hipStream_t st) {
  const btsbot_config& c = h->cfg;
  const float* m = h->mirror;
  TrainPtrs p = carve(h, cache, M);
  const int F = h->has_image ? c.dims[3] : 0;
  const int zd = h->comb_dims[0];
  hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(c.n_meta), dim3(256), 0, st, meta, M, c.n_meta,
                     m + h->bn_w, m + h->bn_b, master ? master + h->bn_rm : nullptr,
                     master ? master + h->bn_rv : nullptr, p.xhat, p.x1, p.bn_rstd);
  LAUNCH_CHECK();
  const float ks1 = c.meta_dropout < 1.f ? 1.f / (1.f - c.meta_dropout) : 0.f;
  hipLaunchKernelGGL(lin_fwd_kernel, g1((long)M * c.meta_fc1), dim3(256), 0, st, p.x1, c.n_meta,
                     reinterpret_cast<const float*>(h->extra + h->p_m1), m + h->m1_b, p.a1, p.h1,
                     c.meta_fc1, M, c.meta_fc1, c.n_meta, h->act,
                     c.meta_dropout > 0.f ? meta_mask : nullptr, ks1);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(lin_fwd_kernel, g1((long)M * c.meta_fc2), dim3(256), 0, st, p.h1,
                     c.meta_fc1, reinterpret_cast<const float*>(h->extra + h->p_m2), m + h->m2_b,
                     p.a2, p.z + F, zd, M, c.meta_fc2, c.meta_fc1,
                     h->meta_trailing_act ? h->act : ACT_NONE, nullptr, 1.f);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// Forward of the heads in training mode.  feat: [M][F] backbone features (already in the cache).
int head_train_forward(btsbot_ctx* h, float* cache, const float* meta, float* logits,
                       float* scores, int M, const uint8_t* meta_mask, const uint8_t* comb_mask,
                       float* master, hipStream_t st, bool meta_done) {
  const btsbot_config& c = h->cfg;
  const float* m = h->mirror;
  TrainPtrs p = carve(h, cache, M);
  const int F = h->has_image ? c.dims[3] : 0;
  const int zd = h->comb_dims[0];
  if (h->has_image) {
    hipLaunchKernelGGL(feat_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, p.feat, F,
                       h->hn_w >= 0 ? m + h->hn_w : nullptr, h->hn_b >= 0 ? m + h->hn_b : nullptr,
                       p.z, zd, M);
    LAUNCH_CHECK();
  }
  if (h->has_meta && !meta_done) TRY_RET(head_train_meta_forward(h, cache, meta, M, meta_mask, master, st));
  const float* in = p.z;
  int ldi = zd;
  const float ksc = c.comb_dropout < 1.f ? 1.f / (1.f - c.comb_dropout) : 0.f;
  for (int i = 0; i < h->n_comb; ++i) {
    const int N = h->comb_dims[i + 1], K = h->comb_dims[i];
    const bool last = i + 1 == h->n_comb;
    // Dropout sits after the activation of the layer BEFORE the final Linear (architectures.py:162)
    const bool drop = !last && i + 2 == h->n_comb && c.comb_dropout > 0.f;
    if (wide_layer(M, N, K, ldi)) {
      TRY_RET(launch_gemm(BTSBOT_F32, EPI_BIAS, in, m + h->comb_w[i], m + h->comb_b[i], nullptr, nullptr, p.pre[i], M, N,
                          K, st));
      hipLaunchKernelGGL(act_fwd_kernel, g1((long)M * N), dim3(256), 0, st, p.pre[i], p.actv[i], N, M, N,
                         last ? ACT_NONE : h->act, drop ? comb_mask : nullptr, ksc);
    } else {
      hipLaunchKernelGGL(lin_fwd_kernel, g1((long)M * N), dim3(256), 0, st, in, ldi,
                         reinterpret_cast<const float*>(h->extra + h->p_comb[i]), m + h->comb_b[i],
                         p.pre[i], p.actv[i], N, M, N, K, last ? ACT_NONE : h->act,
                         drop ? comb_mask : nullptr, ksc);
    }
    LAUNCH_CHECK();
    in = p.actv[i];
    ldi = N;
  }
  hipLaunchKernelGGL(logits_kernel, g1(M), dim3(256), 0, st, in, logits, scores, M);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// Backward of the heads.  Writes d(loss)/d(param) for the fusion head (always) and the metadata
// branch (need_meta) into `grads` (master-arena layout).  The chain the image branch waits on (d(z) through the
// fusion layers, the head LayerNorm) runs on `st`; the fusion head's weight gradients and the whole metadata-branch
// backward read what that chain leaves behind (one gradient buffer per layer) and run on `sd` behind side_fork().
int head_train_backward(btsbot_ctx* h, float* cache, const float* dlogits, float* grads, int M,
                        int need_meta, int need_image, float** dfeat_out,
                        const uint8_t* meta_mask, const uint8_t* comb_mask, hipStream_t st) {
  const btsbot_config& c = h->cfg;
  const float* m = h->mirror;
  TrainPtrs p = carve(h, cache, M);
  const int F = h->has_image ? c.dims[3] : 0;
  const int zd = h->comb_dims[0];
  const float ksc = c.comb_dropout < 1.f ? 1.f / (1.f - c.comb_dropout) : 0.f;
  // fusion head, last layer first; dout[i] = gradient w.r.t. layer i's (pre-activation) output
  const float* dout[3] = {nullptr, nullptr, nullptr};
  const float* dz = nullptr;      // d(loss)/d(z), z = [image feature | metadata feature]
  dout[h->n_comb - 1] = dlogits;
  const bool want_dz = (need_meta && h->has_meta) || need_image;
  for (int i = h->n_comb - 1; i >= 0; --i) {
    const int N = h->comb_dims[i + 1], K = h->comb_dims[i];
    if (i == 0 && !want_dz) break;
    float* din = p.dbuf[i];
    if (wide_layer(M, K, N, N)) {   // din[m][k] = sum_n dout[m][n] Wt[k][n]
      TRY_RET(launch_gemm(BTSBOT_F32, EPI_PLAIN, dout[i], h->extra + h->p_comb[i], nullptr, nullptr, nullptr, din, M, K,
                          N, st));
    } else {
      hipLaunchKernelGGL(lin_bwd_in_kernel, g1((long)M * K), dim3(256), 0, st, dout[i], m + h->comb_w[i],
                         din, K, M, N, K);
    }
    LAUNCH_CHECK();
    if (i > 0) {   // through dropout + activation of layer i-1
      const bool drop = i + 1 == h->n_comb && c.comb_dropout > 0.f;
      hipLaunchKernelGGL(act_bwd_kernel, g1((long)M * K), dim3(256), 0, st, din, K, p.pre[i - 1],
                         din, M, K, h->act, drop ? comb_mask : nullptr, ksc);
      LAUNCH_CHECK();
      dout[i - 1] = din;
    } else {
      dz = din;
    }
  }
  float* dfeat = p.dbuf[3];
  if (need_image && h->has_image)
    // d(z)[:, 0:F] is the gradient of the (head-normalised) image feature; pulled out of the concat layout
    // (the metadata backward below recycles d(z))
  {
    hipLaunchKernelGGL(copy_cols_kernel, g1((long)M * F), dim3(256), 0, st, dfeat, F, dz, zd, F, M);
    LAUNCH_CHECK();
  }
  hipStream_t sd = st;
  TRY_RET(side_fork(h, st, &sd));
  if (need_image && h->has_image) {
    if (h->hn_w >= 0)   // undo the head LayerNorm
      TRY_RET(launch_ln_bwd(p.feat, dfeat, m + h->hn_w, dfeat, grads + h->hn_w, grads + h->hn_b, M,
                            F, st));
    *dfeat_out = dfeat;
  }
  for (int i = h->n_comb - 1; i >= 0; --i) {
    const int N = h->comb_dims[i + 1], K = h->comb_dims[i];
    const float* in = i == 0 ? p.z : p.actv[i - 1];
    const int ldi = i == 0 ? zd : K;
    TRY_RET(launch_lin_bwd_w(dout[i], in, ldi, grads + h->comb_w[i], grads + h->comb_b[i], M, N, K, sd));
  }
  if (need_meta && h->has_meta) {
    // the metadata features are columns F .. F+f2 of d(z) [M][zd]
    const float ks1 = c.meta_dropout < 1.f ? 1.f / (1.f - c.meta_dropout) : 0.f;
    float* da2 = p.dbuf[4];
    hipLaunchKernelGGL(act_bwd_kernel, g1((long)M * c.meta_fc2), dim3(256), 0, sd, dz + F, zd,
                       p.a2, da2, M, c.meta_fc2, h->meta_trailing_act ? h->act : ACT_NONE, nullptr,
                       1.f);
    LAUNCH_CHECK();
    TRY_RET(launch_lin_bwd_w(da2, p.h1, c.meta_fc1, grads + h->m2_w, grads + h->m2_b, M, c.meta_fc2, c.meta_fc1, sd));
    float* dh1 = const_cast<float*>(dz);     // d(z) is dead once da2 exists
    hipLaunchKernelGGL(lin_bwd_in_kernel, g1((long)M * c.meta_fc1), dim3(256), 0, sd, da2,
                       m + h->m2_w, dh1, c.meta_fc1, M, c.meta_fc2, c.meta_fc1);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(act_bwd_kernel, g1((long)M * c.meta_fc1), dim3(256), 0, sd, dh1, c.meta_fc1,
                       p.a1, dh1, M, c.meta_fc1, h->act, c.meta_dropout > 0.f ? meta_mask : nullptr,
                       ks1);
    LAUNCH_CHECK();
    TRY_RET(launch_lin_bwd_w(dh1, p.x1, c.n_meta, grads + h->m1_w, grads + h->m1_b, M, c.meta_fc1, c.n_meta, sd));
    hipLaunchKernelGGL(lin_bwd_in_kernel, g1((long)M * c.n_meta), dim3(256), 0, sd, dh1,
                       m + h->m1_w, da2, c.n_meta, M, c.meta_fc1, c.n_meta);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(c.n_meta), dim3(256), 0, sd, da2, M, c.n_meta,
                       m + h->bn_w, p.xhat, p.bn_rstd, grads + h->bn_w, grads + h->bn_b);
    LAUNCH_CHECK();
  }
  return BTSBOT_OK;
}

float* train_cache_feat(btsbot_ctx* h, float* cache, int M) { return carve(h, cache, M).feat; }
