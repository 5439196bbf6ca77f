// Training of the MaxViT image branch (timm maxvit_tiny_rw_224 as /root/reference/btsbot/architectures.py:25-101
// builds it; the reference fine-tunes it end to end: train.py:218-236, 510-527): the training-mode forward --
// BatchNorm2d on BATCH statistics, running statistics updated in the master arena -- and the backward of every
// layer: stem convolutions, MBConv (pre-norm, 1x1 expand, depthwise 3x3, squeeze-excite, 1x1 project, pooled /
// projected shortcut), window and grid attention with the relative-position bias, the MLPs, the final LayerNorm2d
// and the global pool.  The algorithm is restated (with autograd) in oracle/maxvit_oracle.py, branch_training=True.
//
// An fp32 engine of its own, whatever the handle's operand mode: activations are fp32 NHWC rows [B*H*H][C]; every
// 1x1 convolution / Linear runs on the exact-fp32 MFMA GEMM (gemm.hip), its input gradient on the same GEMM against
// the transposed filter, its filter gradient on backward.hip's split-K GEMM; what has no GEMM shape (BatchNorm
// statistics, depthwise 3x3, squeeze-excite, attention inside a 49-token partition) is a small kernel here.  It is
// correctness-first -- one launch per layer, every intermediate through HBM (~110 MB of cache per alert) -- and
// says so in DESIGN.md: configs[2], the training benchmark, is ConvNeXt.
#include <string.h>

#include <vector>

#include "ctx.h"
#include "maxvit.h"
#include "maxvit_tables.h"

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f;

#define VTRY(call)                  \
  do {                              \
    int _s = (call);                \
    if (_s != BTSBOT_OK) return _s; \
  } while (0)

inline unsigned nblk(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// The reductions below share one thread layout: a workgroup owns 64 channels, lane q = threadIdx.x & 15 four of them (one
// 16-byte load), row lane rl = threadIdx.x >> 4 every 16th row of its share, four rows in flight per trip (one 4-byte
// load per lane and trip left these kernels at 1.5-1.8 TB/s).
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
// sums over the 16 row lanes of a workgroup, then one atomic per channel: dst[c] += total (n4 accumulators of 4 channels)
template <int N>
__device__ __forceinline__ void quad_reduce_atomic(const float4 (&acc)[N], float* const (&dst)[N], int c0, int C, float* sh) {
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
#pragma unroll
  for (int t = 0; t < N; ++t) {
    __syncthreads();
    *reinterpret_cast<float4*>(sh + (rl * 16 + q) * 4) = acc[t];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += sh[r * 64 + threadIdx.x];
      atomicAdd(dst[t] + c0 + threadIdx.x, a);
    }
  }
}
// ---- per-channel sums over the rows of x [M][C]: sums[c] += sum (x - shift[c]), sums[C + c] += sum (x - shift[c])^2
__global__ __launch_bounds__(256) void col_moments_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                          float* __restrict__ sums, long M, int C) {
  __shared__ float sh[16 * 64];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.x * 64, c = c0 + 4 * q;
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (c < C) {
    const float4 sf = shift != nullptr ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long step = (long)gridDim.y * 16;
    for (long r = (long)blockIdx.y * 16 + rl; r < M; r += 4 * step) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + min(r + u * step, M - 1) * C + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * step < M) {
          const float4 d = f4_sub(v[u], sf);
          acc[0] = f4_add(acc[0], d);
          acc[1] = f4_fma(d, d, acc[1]);
        }
    }
  }
  float* const dst[2] = {sums, sums + C};
  quad_reduce_atomic<2>(acc, dst, c0, C, sh);
}
// pass 1 -> mean;  pass 2 (sums taken about the mean) -> rstd, and the running statistics in the master arena
__global__ void bn_mean_kernel(const float* sums, float* stat, long M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) stat[c] = sums[c] / (float)M;
}
__global__ void bn_finish_kernel(const float* sums, float* stat, float* run_mean, float* run_var, long M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mean = stat[c] + sums[c] / (float)M;    // (the centred sum is ~0: it only polishes the mean)
  const float d = sums[c] / (float)M;
  const float var = fmaxf(sums[C + c] / (float)M - d * d, 0.f);
  stat[c] = mean;
  stat[C + c] = rsqrtf(var + BN_EPS);
  if (run_mean != nullptr) {   // torch: momentum 0.1, the running variance takes the unbiased estimate
    run_mean[c] = (1.0f - BN_MOM) * run_mean[c] + BN_MOM * mean;
    run_var[c] = (1.0f - BN_MOM) * run_var[c] + BN_MOM * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
  }
}
// y = act(xhat * w + b), xhat = (x - mean) * rstd;  act 0: none, 1: SiLU.  n4 = M * C / 4 (four channels per thread)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stat, const float* __restrict__ w,
                                const float* __restrict__ b, float* __restrict__ y, long n4, int C, int act) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const float4 mean = *reinterpret_cast<const float4*>(stat + c), rstd = *reinterpret_cast<const float4*>(stat + C + c);
  const float4 z = f4_fma(f4_mul(f4_sub(v, mean), rstd), *reinterpret_cast<const float4*>(w + c),
                          *reinterpret_cast<const float4*>(b + c));
  reinterpret_cast<float4*>(y)[i] =
      act ? make_float4(z.x * sigmoid_f(z.x), z.y * sigmoid_f(z.y), z.z * sigmoid_f(z.z), z.w * sigmoid_f(z.w)) : z;
}
// backward sums: sums[c] += sum dz, sums[C + c] += sum dz * xhat, dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ stat, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ sums, long M,
                                                          int C, int act) {
  __shared__ float sh[16 * 64];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.x * 64, c = c0 + 4 * q;
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (c < C) {
    const float4 mean = *reinterpret_cast<const float4*>(stat + c), rstd = *reinterpret_cast<const float4*>(stat + C + c);
    const float4 wc = *reinterpret_cast<const float4*>(w + c), bc = *reinterpret_cast<const float4*>(b + c);
    const long step = (long)gridDim.y * 16;
    for (long r = (long)blockIdx.y * 16 + rl; r < M; r += 2 * step) {
      float4 xv[2], dv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long rr = min(r + u * step, M - 1);
        xv[u] = *reinterpret_cast<const float4*>(x + rr * C + c);
        dv[u] = *reinterpret_cast<const float4*>(dy + rr * C + c);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (r + u * step < M) {
          const float4 xh = f4_mul(f4_sub(xv[u], mean), rstd);
          float4 dz = dv[u];
          if (act) {
            const float4 z = f4_fma(xh, wc, bc);
            dz = f4_mul(dz, make_float4(silu_grad(z.x), silu_grad(z.y), silu_grad(z.z), silu_grad(z.w)));
          }
          acc[0] = f4_add(acc[0], dz);
          acc[1] = f4_fma(dz, xh, acc[1]);
        }
    }
  }
  float* const dst[2] = {sums, sums + C};
  quad_reduce_atomic<2>(acc, dst, c0, C, sh);
}
// dx (+)= w rstd (dz - sum dz / M - xhat sum(dz xhat) / M);  block 0 also adds the parameter gradients.  Four channels per thread
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stat,
                                    const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ sums,
                                    float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, long M, int C,
                                    int act, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      atomicAdd(dw + c, sums[C + c]);
      atomicAdd(db + c, sums[c]);
    }
  }
  if (i >= M * C / 4) return;
  const int c = (int)((i * 4) % C);
  const float4 rstd = *reinterpret_cast<const float4*>(stat + C + c);
  const float4 xh = f4_mul(f4_sub(reinterpret_cast<const float4*>(x)[i], *reinterpret_cast<const float4*>(stat + c)), rstd);
  const float4 wc = *reinterpret_cast<const float4*>(w + c);
  float4 dz = reinterpret_cast<const float4*>(dy)[i];
  if (act) {
    const float4 z = f4_fma(xh, wc, *reinterpret_cast<const float4*>(b + c));
    dz = f4_mul(dz, make_float4(silu_grad(z.x), silu_grad(z.y), silu_grad(z.z), silu_grad(z.w)));
  }
  const float inv = 1.0f / (float)M;
  const float4 s0 = *reinterpret_cast<const float4*>(sums + c), s1 = *reinterpret_cast<const float4*>(sums + C + c);
  float4 g;
  g.x = wc.x * rstd.x * (dz.x - s0.x * inv - xh.x * s1.x * inv);
  g.y = wc.y * rstd.y * (dz.y - s0.y * inv - xh.y * s1.y * inv);
  g.z = wc.z * rstd.z * (dz.z - s0.z * inv - xh.z * s1.z * inv);
  g.w = wc.w * rstd.w * (dz.w - s0.w * inv - xh.w * s1.w * inv);
  float4* o = reinterpret_cast<float4*>(dx) + i;
  *o = accumulate ? f4_add(*o, g) : g;
}

// ---- depthwise 3x3, padding 1, stride s: in [B][H][H][C] -> out [B][Ho][Ho][C]; taps w9 [9][C].  A thread owns four
// channels of a pixel; its nine taps are requested as one batch (addresses clamped into the map, values zeroed by a
// select: a load under a bounds branch is waited for on the spot)
__global__ void dw3_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w9, const float* __restrict__ bias,
                               float* __restrict__ out, int B, int H, int C, int s) {
  const int Ho = H / s, C4 = C / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * Ho * Ho * C4) return;
  const int c = (int)(i % C4) * 4;
  long p = i / C4;
  const int ox = (int)(p % Ho);
  p /= Ho;
  const int oy = (int)(p % Ho), b = (int)(p / Ho);
  float4 v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = min(max(oy * s + t / 3 - 1, 0), H - 1), ix = min(max(ox * s + t % 3 - 1, 0), H - 1);
    v[t] = *reinterpret_cast<const float4*>(in + (((long)b * H + iy) * H + ix) * C + c);
  }
  float4 a = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    asm volatile("" : "+v"(v[t].x), "+v"(v[t].y), "+v"(v[t].z), "+v"(v[t].w));   // (keeps the load unconditional)
    const int iy = oy * s + t / 3 - 1, ix = ox * s + t % 3 - 1;
    if (!(iy >= 0 && iy < H && ix >= 0 && ix < H)) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    a = f4_fma(*reinterpret_cast<const float4*>(w9 + t * C + c), v[t], a);
  }
  reinterpret_cast<float4*>(out)[i] = a;
}
__global__ void dw3_bwd_in_kernel(const float* __restrict__ dout, const float* __restrict__ w9, float* __restrict__ din,
                                  int B, int H, int C, int s) {
  const int Ho = H / s, C4 = C / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * H * C4) return;
  const int c = (int)(i % C4) * 4;
  long p = i / C4;
  const int ix = (int)(p % H);
  p /= H;
  const int iy = (int)(p % H), b = (int)(p / H);
  float4 v[9];
  bool ok[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ty = iy + 1 - t / 3, tx = ix + 1 - t % 3;
    ok[t] = ty >= 0 && ty % s == 0 && ty / s < Ho && tx >= 0 && tx % s == 0 && tx / s < Ho;
    const int oy = min(max(ty, 0) / s, Ho - 1), ox = min(max(tx, 0) / s, Ho - 1);
    v[t] = *reinterpret_cast<const float4*>(dout + (((long)b * Ho + oy) * Ho + ox) * C + c);
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    asm volatile("" : "+v"(v[t].x), "+v"(v[t].y), "+v"(v[t].z), "+v"(v[t].w));
    if (!ok[t]) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    a = f4_fma(*reinterpret_cast<const float4*>(w9 + t * C + c), v[t], a);
  }
  reinterpret_cast<float4*>(din)[i] = a;
}
// dw9[t][c] += sum dout * in(shifted by tap t), dbias[c] += sum dout   (the reductions' thread layout; a pixel's nine taps in
// one batch)
__global__ __launch_bounds__(256) void dw3_bwd_w_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                        float* __restrict__ dw9, float* __restrict__ dbias, int B, int H,
                                                        int C, int s) {
  __shared__ float sh[16 * 64];
  const int Ho = H / s;
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.x * 64, c = c0 + 4 * q;
  float4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const long npix = (long)B * Ho * Ho;
    for (long p = (long)blockIdx.y * 16 + rl; p < npix; p += (long)gridDim.y * 16) {
      const int ox = (int)(p % Ho), oy = (int)((p / Ho) % Ho), b = (int)(p / ((long)Ho * Ho));
      const float4 d = *reinterpret_cast<const float4*>(dout + p * C + c);
      float4 v[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = min(max(oy * s + t / 3 - 1, 0), H - 1), ix = min(max(ox * s + t % 3 - 1, 0), H - 1);
        v[t] = *reinterpret_cast<const float4*>(in + (((long)b * H + iy) * H + ix) * C + c);
      }
      acc[9] = f4_add(acc[9], d);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        asm volatile("" : "+v"(v[t].x), "+v"(v[t].y), "+v"(v[t].z), "+v"(v[t].w));
        const int iy = oy * s + t / 3 - 1, ix = ox * s + t % 3 - 1;
        if (!(iy >= 0 && iy < H && ix >= 0 && ix < H)) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc[t] = f4_fma(d, v[t], acc[t]);
      }
    }
  }
  float* dst[10];
#pragma unroll
  for (int t = 0; t < 9; ++t) dst[t] = dw9 + t * C;
  dst[9] = dbias;
  float* const (&dref)[10] = reinterpret_cast<float* const (&)[10]>(dst);
  quad_reduce_atomic<10>(acc, dref, c0, C, sh);
}

// ---- per-alert column sums: out[b][c] = scale * sum_p a[b][p][c] * (m ? m[b][p][c] : 1)   (the reductions' thread layout)
__global__ __launch_bounds__(256) void alert_colsum_kernel(const float* __restrict__ a, const float* __restrict__ m,
                                                           float* __restrict__ out, int P, int C, float scale) {
  __shared__ float sh[16 * 64];
  const int q = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.x * 64, c = c0 + 4 * q, b = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (int p = rl; p < P; p += 64) {
      float4 av[4], mv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = ((long)b * P + min(p + 16 * u, P - 1)) * C + c;
        av[u] = *reinterpret_cast<const float4*>(a + i);
        mv[u] = m != nullptr ? *reinterpret_cast<const float4*>(m + i) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + 16 * u < P) s = f4_fma(av[u], mv[u], s);
    }
  *reinterpret_cast<float4*>(sh + (rl * 16 + q) * 4) = s;
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sh[r * 64 + threadIdx.x];
    out[(long)b * C + c0 + threadIdx.x] = scale * t;
  }
}
// small dense layers over the batch (squeeze-excite: B rows): y[b][o] = act(bias[o] + sum_i x[b][i] w[o][i]); act 0 none,
// 1 SiLU, 2 sigmoid; pre (optional) keeps the pre-activation
__global__ void lin_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ pre, float* __restrict__ y, int B, int I, int O, int act) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * O) return;
  const int o = i % O, b = i / O;
  // (four chains: one chain over up to 2048 inputs was 58 us of exposed load + FMA latency on a 64-row batch)
  float a0 = bias[o], a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float* xp = x + (long)b * I;
  const float* wp = w + (long)o * I;
  int k = 0;
#pragma unroll 2
  for (; k + 3 < I; k += 4) {
    a0 = fmaf(xp[k], wp[k], a0);
    a1 = fmaf(xp[k + 1], wp[k + 1], a1);
    a2 = fmaf(xp[k + 2], wp[k + 2], a2);
    a3 = fmaf(xp[k + 3], wp[k + 3], a3);
  }
  for (; k < I; ++k) a0 = fmaf(xp[k], wp[k], a0);
  const float a = (a0 + a1) + (a2 + a3);
  if (pre != nullptr) pre[i] = a;
  y[i] = act == 1 ? a * sigmoid_f(a) : act == 2 ? sigmoid_f(a) : a;
}
// dpre[b][o] = dy[b][o] * act'(pre)   (act 1: SiLU from the pre-activation; 2: sigmoid from its OUTPUT in pre)
__global__ void act_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre,
                                     int n, int act) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = pre[i];
  dpre[i] = dy[i] * (act == 1 ? silu_grad(v) : v * (1.0f - v));
}
__global__ void lin_bwd_in_small_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ dx,
                                        int B, int I, int O) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * I) return;
  const int k = i % I, b = i / I;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four chains (as lin_fwd_small_kernel: 137 us with one)
  const float* dp = dpre + (long)b * O;
  const float* wp = w + k;
  int o = 0;
#pragma unroll 2
  for (; o + 3 < O; o += 4) {
    a0 = fmaf(dp[o], wp[(long)o * I], a0);
    a1 = fmaf(dp[o + 1], wp[(long)(o + 1) * I], a1);
    a2 = fmaf(dp[o + 2], wp[(long)(o + 2) * I], a2);
    a3 = fmaf(dp[o + 3], wp[(long)(o + 3) * I], a3);
  }
  for (; o < O; ++o) a0 = fmaf(dp[o], wp[(long)o * I], a0);
  dx[i] = (a0 + a1) + (a2 + a3);
}
__global__ void lin_bwd_w_small_kernel(const float* __restrict__ dpre, const float* __restrict__ x, float* __restrict__ dw,
                                       float* __restrict__ db, int B, int I, int O) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= O * I) return;
  const int k = i % I, o = i / I;
  float a = 0.f, s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dpre[(long)b * O + o];
    a = fmaf(d, x[(long)b * I + k], a);
    s += d;
  }
  dw[i] += a;
  if (k == 0) db[o] += s;
}
// y[b][p][c] = a[b][p][c] * g[b][c]  (four channels per thread; y may be a)
__global__ void gate_mul_kernel(const float* a, const float* __restrict__ g, float* y, int P, int C, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const long b = (i * 4) / ((long)P * C);
  reinterpret_cast<float4*>(y)[i] = f4_mul(reinterpret_cast<const float4*>(a)[i], *reinterpret_cast<const float4*>(g + b * C + c));
}
// d[b][p][c] += v[b][c] * scale
__global__ void bcast_add_kernel(float* __restrict__ d, const float* __restrict__ v, int P, int C, long n, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long b = i / ((long)P * C);
  d[i] += v[b * C + c] * scale;
}
__global__ void bcast_set_kernel(float* __restrict__ d, const float* __restrict__ v, int P, int C, long n, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long b = i / ((long)P * C);
  d[i] = v[b * C + c] * scale;
}
// dx[b][y][x][c] (+)= 0.25 g[b][y / 2][x / 2][c]
__global__ void avgpool2_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int B, int H, int C, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * H * C) return;
  const int c = (int)(i % C);
  long p = i / C;
  const int x = (int)(p % H);
  p /= H;
  const int y = (int)(p % H), b = (int)(p / H), Ho = H / 2;
  const float v = 0.25f * g[(((long)b * Ho + y / 2) * Ho + x / 2) * C + c];
  dx[i] = accumulate ? dx[i] + v : v;
}
// (elementwise kernels: four values per thread; every count here is a multiple of 4)
__global__ void add_kernel(float* __restrict__ a, const float* __restrict__ b, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<float4*>(a)[i] = f4_add(reinterpret_cast<float4*>(a)[i], reinterpret_cast<const float4*>(b)[i]);
}
__global__ void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(pre)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
}
__global__ void gelu_bwd_kernel(const float* __restrict__ pre, float* __restrict__ d, long n4) {   // d *= gelu'(pre)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(pre)[i];
  float4* o = reinterpret_cast<float4*>(d) + i;
  *o = f4_mul(*o, make_float4(gelu_grad(v.x), gelu_grad(v.y), gelu_grad(v.z), gelu_grad(v.w)));
}
// col2im of a 3x3 s1 p1 convolution: din[b][y][x][c] = sum over taps dcol[b][y - ky + 1][x - kx + 1][(ky*3+kx)*C + c]
__global__ void col2im3_kernel(const float* __restrict__ dcol, float* __restrict__ din, int B, int H, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H * H * C) return;
  const int c = (int)(i % C);
  long p = i / C;
  const int x = (int)(p % H);
  p /= H;
  const int y = (int)(p % H), b = (int)(p / H);
  float a = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int oy = y - ky + 1;
    if (oy < 0 || oy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ox = x - kx + 1;
      if (ox < 0 || ox >= H) continue;
      a += dcol[(((long)b * H + oy) * H + ox) * (9 * C) + (ky * 3 + kx) * C + c];
    }
  }
  din[i] = a;
}
// packed convolution gradient [O][(ky*3+kx)*C + c] (row pitch ldp) -> master layout [O][C][3][3], accumulated
__global__ void unpack_conv3_grad_kernel(const float* __restrict__ gp, float* __restrict__ g, int O, int C, int ldp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= O * C * 9) return;
  const int t = i % 9, c = (i / 9) % C, o = i / (9 * C);
  g[i] += gp[(long)o * ldp + t * C + c];
}
// tap-major depthwise gradient [9][C] -> master [C][1][3][3], accumulated
__global__ void unpack_dw_grad_kernel(const float* __restrict__ g9, float* __restrict__ g, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * C) return;
  const int t = i % 9, c = i / 9;
  g[i] += g9[t * C + c];
}
// relative-position bias gradient [heads][49 key][49 query] -> table [169][heads], accumulated
__global__ void relbias_grad_kernel(const float* __restrict__ dbias, float* __restrict__ dtable, int heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // [169][heads]
  if (i >= 169 * heads) return;
  const int hd = i % heads, idx = i / heads;
  const int dy = idx / 13 - 6, dx = idx % 13 - 6;
  float s = 0.f;
  for (int kj = 0; kj < 49; ++kj) {
    const int qy = kj / 7 + dy, qx = kj % 7 + dx;
    if (qy < 0 || qy >= 7 || qx < 0 || qx >= 7) continue;
    s += dbias[((long)hd * 49 + kj) * 49 + qy * 7 + qx];
  }
  dtable[i] += s;
}

// ---- attention backward: one wave per (alert, partition, head); the forward's conventions (mv_attn_kernel,
// maxvit_ops.hip): q scaled by 32^-0.5, s[query][key j] = q . k_j + bias_t[head][j][query], softmax over the keys.
// Phase 1, lane = query t: P[t][:], dP[t][j] = dO_t . v_j, dS[t][j] = P (dP - sum_j P dP), dq_t = SC sum_j dS k_j,
// dbias[head][j][t] += dS.  Phase 2, lane = key j: dk_j = sum_t dS[t][j] (SC q_t), dv_j = sum_t P[t][j] dO_t.
// A workgroup walks `units` (alert, partition) pairs of ONE head and keeps the bias gradient of its lane's query in
// registers (49 values), added to the table once at the end: one atomic per (block, j, t) instead of one per (unit, j, t)
// -- 8192 units of stage 0 hammering the same 2401 addresses per head was most of this kernel.  The lane's own q and dO
// rows stay in registers; k, v (and q, dO for phase 2) are read from LDS as 16-byte broadcasts.
__global__ __launch_bounds__(64) void mv_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t,
                                                         const float* __restrict__ dout, float* __restrict__ dqkv,
                                                         float* __restrict__ dbias, int H, int C, int grid_mode, int units) {
  constexpr int RP = 36;               // row pitch in floats (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) float ks[49][RP];
  __shared__ __attribute__((aligned(16))) float vs[49][RP];
  __shared__ __attribute__((aligned(16))) float qs[49][RP];     // SC * q
  __shared__ __attribute__((aligned(16))) float dos[49][RP];
  __shared__ float ps[49][50];     // [query][key]
  __shared__ float dss[49][50];
  const int heads = C / 32, G = H / 7, nW = G * G;
  const int head = blockIdx.x % heads, ustep = gridDim.x / heads;
  const int t = threadIdx.x;
  const bool active = t < 49;
  const int ty = t / 7, tx = t % 7;
  constexpr float SC = 0.17677669529663687f;
  // (the products run as PACKED fp32 FMAs -- v_pk_fma_f32, two per instruction: this kernel is bound by the ~8 k FMA
  //  instructions a (partition, head) costs on one wave)
  typedef float f2 __attribute__((ext_vector_type(2)));
  float db[49];
#pragma unroll
  for (int j = 0; j < 49; ++j) db[j] = 0.f;
  for (int u = blockIdx.x / heads; u < units; u += ustep) {
    const int w = u % nW;
    const long b = u / nW;
    const int wy = w / G, wx = w % G;
    const int py = grid_mode ? ty * G + wy : wy * 7 + ty;
    const int px = grid_mode ? tx * G + wx : wx * 7 + tx;
    const long row = active ? (b * H + py) * H + px : 0;
    f2 qr[16], dor[16];            // this lane's SC * q row and dO row
    __syncthreads();               // the previous unit's phase 2 has read the images
    if (active) {
      const float4* base = reinterpret_cast<const float4*>(qkv + row * 3 * C + head * 96);
      const float4* dob = reinterpret_cast<const float4*>(dout + row * C + head * 32);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float4 q4 = base[d], o4 = dob[d];
        const float4 qs4 = make_float4(q4.x * SC, q4.y * SC, q4.z * SC, q4.w * SC);
        qr[2 * d] = f2{qs4.x, qs4.y};
        qr[2 * d + 1] = f2{qs4.z, qs4.w};
        dor[2 * d] = f2{o4.x, o4.y};
        dor[2 * d + 1] = f2{o4.z, o4.w};
        *reinterpret_cast<float4*>(&qs[t][4 * d]) = qs4;
        *reinterpret_cast<float4*>(&ks[t][4 * d]) = base[8 + d];
        *reinterpret_cast<float4*>(&vs[t][4 * d]) = base[16 + d];
        *reinterpret_cast<float4*>(&dos[t][4 * d]) = o4;
      }
    }
    __syncthreads();
    if (active) {
      const float* bt = bias_t + (size_t)head * 2401 + t;
      float mx = -3.0e38f;
      for (int j = 0; j < 49; ++j) {
        f2 a2 = f2{0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 16; ++d) a2 = qr[d] * *reinterpret_cast<const f2*>(&ks[j][2 * d]) + a2;
        const float a = a2.x + a2.y + bt[j * 49];
        ps[t][j] = a;
        mx = fmaxf(mx, a);
      }
      float sum = 0.f;
      for (int j = 0; j < 49; ++j) {
        const float e = __expf(ps[t][j] - mx);
        ps[t][j] = e;
        sum += e;
      }
      const float inv = 1.0f / sum;
      float dot = 0.f;
      for (int j = 0; j < 49; ++j) {
        f2 d2 = f2{0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 16; ++d) d2 = dor[d] * *reinterpret_cast<const f2*>(&vs[j][2 * d]) + d2;
        const float dp = d2.x + d2.y;
        const float p = ps[t][j] * inv;
        ps[t][j] = p;
        dss[t][j] = dp;
        dot = fmaf(p, dp, dot);
      }
      f2 dq[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) dq[d] = f2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 49; ++j) {
        const float ds = ps[t][j] * (dss[t][j] - dot);
        dss[t][j] = ds;
        db[j] += ds;
        const f2 ds2 = f2{ds, ds};
#pragma unroll
        for (int d = 0; d < 16; ++d) dq[d] = ds2 * *reinterpret_cast<const f2*>(&ks[j][2 * d]) + dq[d];
      }
      float4* dst = reinterpret_cast<float4*>(dqkv + row * 3 * C + head * 96);
#pragma unroll
      for (int d = 0; d < 8; ++d)
        dst[d] = make_float4(dq[2 * d].x * SC, dq[2 * d].y * SC, dq[2 * d + 1].x * SC, dq[2 * d + 1].y * SC);
    }
    __syncthreads();
    if (active) {   // lane = key t
      f2 dk[16], dv[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) dk[d] = dv[d] = f2{0.f, 0.f};
      for (int q = 0; q < 49; ++q) {
        const float ds = dss[q][t], p = ps[q][t];
        const f2 ds2 = f2{ds, ds}, p2 = f2{p, p};
#pragma unroll
        for (int d = 0; d < 16; ++d) {
          dk[d] = ds2 * *reinterpret_cast<const f2*>(&qs[q][2 * d]) + dk[d];
          dv[d] = p2 * *reinterpret_cast<const f2*>(&dos[q][2 * d]) + dv[d];
        }
      }
      float4* dst = reinterpret_cast<float4*>(dqkv + row * 3 * C + head * 96);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        dst[8 + d] = make_float4(dk[2 * d].x, dk[2 * d].y, dk[2 * d + 1].x, dk[2 * d + 1].y);
        dst[16 + d] = make_float4(dv[2 * d].x, dv[2 * d].y, dv[2 * d + 1].x, dv[2 * d + 1].y);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 49; ++j) atomicAdd(dbias + ((size_t)head * 49 + j) * 49 + t, db[j]);
  }
}

// ---- the engine's cache: every activation the backward needs, carved from one allocation -------------------------
constexpr size_t MVT_WPART_FLOATS = (size_t)16 << 20;
constexpr int MVT_SUM_SLOTS = 96;   // BatchNorm reduction slots per pass: 2 per layer forward (34 layers), 1 backward

struct AttnAct {
  float *n1, *qkv, *o, *y1, *n2, *f1, *gl, *y2;
};
struct BlkAct {
  float *xin, *pool_in, *a0, *c1, *a1, *d2, *a2, *sepool, *rpre, *r, *g, *y;
  float *st_pre, *st1, *st2;   // BatchNorm batch statistics: mean [C] | rstd [C]
  AttnAct at[2];
};
struct MvtCache {
  float *col1, *y1, *a1s, *st_stem, *x0, *w1p, *w2p;
  std::vector<BlkAct> blk;
  float *xfin;                     // final LayerNorm input = last block's output
  // backward scratch
  float *dA, *dB, *dC, *wt, *sums, *dsmall, *dbias, *g9, *gconv, *xn_final;
  // 16-bit operand modes: the GEMMs' operands are cast per call into these (fp32 everywhere else)
  void *x16, *d16, *w16;
  // ... except the filters, cast once per forward as one pass over the whole parameter mirror (filter W at wall16 + (W - mirror)),
  // and the GEMM inputs of the forward, whose 16-bit copies stay (xkeep, handed out in call order) for the filter
  // gradients of the backward: 230 of the step's 612 cast launches gone
  void *wall16, *xkeep;
  size_t xkeep_floats;
  float* wpart;                    // partial tiles of the 16-bit filter-gradient GEMM (two-pass reduction)
  size_t total;
};

MvtCache carve(const MaxVit* mv, unsigned char* base, int B, int64_t nparams = 0) {
  MvtCache k;
  size_t cur = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + cur) : nullptr;
    cur += (floats * 4 + 255) / 256 * 256;
    return p;
  };
  const size_t n = (size_t)B, M0 = n * 12544;
  k.col1 = take(M0 * 32);
  k.y1 = take(M0 * 32);
  k.a1s = take(M0 * 32);
  k.st_stem = take(64);
  k.x0 = take(M0 * 64);
  k.w1p = take(32 * 32);
  k.w2p = take(64 * 288);
  float* x = k.x0;
  for (const MvBlock& b : mv->blocks) {
    BlkAct a;
    const size_t Min = n * b.hin * b.hin, Mo = n * b.hout * b.hout;
    a.xin = x;
    a.pool_in = b.stride == 2 ? take(Mo * b.cin) : nullptr;
    a.a0 = take(Min * b.cin);
    a.c1 = take(Min * b.mid);
    a.a1 = take(Min * b.mid);
    a.d2 = take(Mo * b.mid);
    a.a2 = take(Mo * b.mid);
    a.sepool = take(n * b.mid);
    a.rpre = take(n * b.rd);
    a.r = take(n * b.rd);
    a.g = take(n * b.mid);
    a.y = take(Mo * b.c);
    a.st_pre = take(2 * b.cin);
    a.st1 = take(2 * b.mid);
    a.st2 = take(2 * b.mid);
    float* yin = a.y;
    for (int g = 0; g < 2; ++g) {
      AttnAct& t = a.at[g];
      t.n1 = take(Mo * b.c);
      t.qkv = take(Mo * 3 * b.c);
      t.o = take(Mo * b.c);
      t.y1 = take(Mo * b.c);
      t.n2 = take(Mo * b.c);
      t.f1 = take(Mo * 4 * b.c);
      t.gl = take(Mo * 4 * b.c);
      t.y2 = take(Mo * b.c);
      yin = t.y2;
    }
    x = yin;
    k.blk.push_back(a);
  }
  k.xfin = x;
  k.dA = take(M0 * 256);       // the largest gradient maps: [12544][256] per alert (block 0's expanded map)
  k.dB = take(M0 * 256);
  k.dC = take(M0 * 288);       // stem: gradient of the im2col matrix [12544][288]; also forward's im2col scratch
  k.wt = take(2048 * 512);
  k.sums = take((size_t)MVT_SUM_SLOTS * 4096);   // per-BatchNorm slots of a zeroed arena (sum | sum of squares, <= 2048 channels)
  k.dsmall = take(n * 2048 * 4);
  k.dbias = take(32 * 2401);   // gradient of the bias [heads <= 16][49][49] | its forward image
  k.g9 = take(10 * 2048);
  k.gconv = take(64 * 288);
  k.xn_final = take(n * 49 * 512);
  k.x16 = take(M0 * 288 / 2);
  k.d16 = take(M0 * 288 / 2);
  k.w16 = take(2048 * 512 / 2);
  k.wall16 = take((size_t)nparams / 2 + 64);
  {
    // 16-bit copies of every forward GEMM input, in call order (two per float)
    size_t e = M0 * 32 + M0 * 288;
    for (const MvBlock& b : mv->blocks) {
      const size_t Min = n * b.hin * b.hin, Mo = n * b.hout * b.hout;
      e += (b.stride == 2 && b.sc_w >= 0 ? Mo * b.cin + 128 : 0) + Min * b.cin + Mo * b.mid + 2 * (3 * Mo * b.c + Mo * 4 * b.c) + 12 * 128;
    }
    k.xkeep_floats = e / 2 + 256;
    k.xkeep = take(k.xkeep_floats);
  }
  k.wpart = take(MVT_WPART_FLOATS);
  k.total = cur;
  return k;
}

}  // namespace

size_t maxvit_train_cache_bytes(const btsbot_ctx* h, int B) { return carve(h->mv, nullptr, B, h->total_floats).total; }

// BatchNorm2d, training mode: batch statistics of x [M][C] into stat (mean | rstd), running statistics updated in the
// master arena, y = act(...)
// (the reductions add into slots of an arena zeroed ONCE per pass -- MvtSums -- instead of behind a memset of their own:
//  141 fills per step were 0.66 ms)
struct MvtSums {
  float* base;
  int next;
  float* take() { return next < MVT_SUM_SLOTS ? base + (size_t)(next++) * 4096 : nullptr; }
};
static int bn_train(const btsbot_ctx* h, const float* x, const BnPk& bn, float* stat, MvtSums& arena, float* y, long M, int C,
                    int act, float* master, hipStream_t st) {
  const float* m = h->mirror;
  const dim3 grid((C + 63) / 64, (unsigned)(M / 256 > 256 ? 256 : (M / 256 > 0 ? M / 256 : 1)));
  float* sums0 = arena.take();
  float* sums = arena.take();
  if (sums == nullptr) {
    btsbot_set_error("maxvit_train: out of BatchNorm reduction slots");
    return BTSBOT_ERR_STATE;
  }
  hipLaunchKernelGGL(col_moments_kernel, grid, dim3(256), 0, st, x, (const float*)nullptr, sums0, M, C);
  hipLaunchKernelGGL(bn_mean_kernel, dim3(nblk(C)), dim3(256), 0, st, sums0, stat, M, C);
  hipLaunchKernelGGL(col_moments_kernel, grid, dim3(256), 0, st, x, (const float*)stat, sums, M, C);
  hipLaunchKernelGGL(bn_finish_kernel, dim3(nblk(C)), dim3(256), 0, st, sums, stat, master ? master + bn.rm : nullptr,
                     master ? master + bn.rv : nullptr, M, C);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(M * C / 4)), dim3(256), 0, st, x, stat, m + bn.w, m + bn.b, y, M * C / 4, C, act);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
// ... and its backward: dy = gradient w.r.t. the layer's output (behind the activation), dx (+)= gradient w.r.t. x
static int bn_train_bwd(const btsbot_ctx* h, const float* x, const float* dy, const BnPk& bn, const float* stat, MvtSums& arena,
                        float* dx, float* grads, long M, int C, int act, int accumulate, hipStream_t st) {
  const float* m = h->mirror;
  const dim3 grid((C + 63) / 64, (unsigned)(M / 256 > 256 ? 256 : (M / 256 > 0 ? M / 256 : 1)));
  float* sums = arena.take();
  if (sums == nullptr) {
    btsbot_set_error("maxvit_train: out of BatchNorm reduction slots");
    return BTSBOT_ERR_STATE;
  }
  hipLaunchKernelGGL(bn_bwd_sums_kernel, grid, dim3(256), 0, st, x, dy, stat, m + bn.w, m + bn.b, sums, M, C, act);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(M * C / 4)), dim3(256), 0, st, x, dy, stat, m + bn.w, m + bn.b,
                     (const float*)sums, dx, grads + bn.w, grads + bn.b, M, C, act, accumulate);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int maxvit_train_forward(btsbot_ctx* h, const float* img, int B, float* master, hipStream_t st, float** feat_out) {
  MaxVit* mv = h->mv;
  const float* m = h->mirror;
  MvtCache k = carve(mv, h->bbcache, B, h->total_floats);
  MvtSums sums_arena{k.sums, 0};
  HIP_TRY(hipMemsetAsync(k.sums, 0, (size_t)MVT_SUM_SLOTS * 4096 * 4, st));
  auto F = [&](size_t off) { return reinterpret_cast<float*>(h->extra + off); };
  const float* zero = F(mv->p_zero);
  const float* one = F(mv->p_one);
  // 16-bit operand modes: the 1x1 convolutions / Linear layers run the LDS-DMA GEMMs on operands cast per call (fp32
  // accumulation, fp32 results; the split mode keeps cfg.precision = fp32 and with it the exact GEMMs)
  const int prec = h->cfg.precision;
  const bool lowp = prec == BTSBOT_BF16 || prec == BTSBOT_F16;
  // (16-bit modes) every filter of the parameter mirror in the operand type, one launch; the inputs' copies are kept
  if (lowp) VTRY(launch_cast(prec, m, k.wall16, h->total_floats, st));
  mv->xkept.clear();
  size_t xcur = 0;   // bytes handed out of xkeep
  auto w16_of = [&](const float* W, int64_t n, const void** out) -> int {
    if (W >= m && W + n <= m + h->total_floats) {
      *out = reinterpret_cast<const unsigned char*>(k.wall16) + (size_t)(W - m) * 2;
      return BTSBOT_OK;
    }
    VTRY(launch_cast(prec, W, k.w16, n, st));   // (the stem's packed filters live outside the mirror)
    *out = k.w16;
    return BTSBOT_OK;
  };
  auto x16_of = [&](const float* X, int64_t n, const void** out) -> int {
    if (X == k.dA || X == k.dB || X == k.dC) {   // scratch maps (rewritten before the backward): cast per call, not kept
      VTRY(launch_cast(prec, X, k.x16, n, st));
      *out = k.x16;
      return BTSBOT_OK;
    }
    const size_t bytes = ((size_t)n * 2 + 255) / 256 * 256;
    if (xcur + bytes > k.xkeep_floats * 4) {
      btsbot_set_error("maxvit_train_forward: the kept-operand arena is too small");
      return BTSBOT_ERR_STATE;
    }
    void* dst = reinterpret_cast<unsigned char*>(k.xkeep) + xcur;
    xcur += bytes;
    VTRY(launch_cast(prec, X, dst, n, st));
    mv->xkept.push_back({X, dst});
    *out = dst;
    return BTSBOT_OK;
  };
  auto gemm = [&](const float* X, const float* W, const float* bias, float* out, long M, int N, int K) -> int {
    if (!lowp) return launch_gemm(BTSBOT_F32, EPI_BIAS, X, W, bias ? bias : zero, nullptr, nullptr, out, (int)M, N, K, st);
    const void *x16, *w16;
    VTRY(x16_of(X, (int64_t)M * K, &x16));
    VTRY(w16_of(W, (int64_t)N * K, &w16));
    return launch_gemm(prec, EPI_BIAS, x16, w16, bias ? bias : zero, nullptr, nullptr, out, (int)M, N, K, st);
  };
  auto gemm_resid = [&](const float* X, const float* W, const float* bias, const float* resid, float* out, long M, int N,
                        int K) -> int {
    if (!lowp) return launch_gemm(BTSBOT_F32, EPI_RESID, X, W, bias ? bias : zero, one, resid, out, (int)M, N, K, st);
    const void *x16, *w16;
    VTRY(x16_of(X, (int64_t)M * K, &x16));
    VTRY(w16_of(W, (int64_t)N * K, &w16));
    return launch_gemm(prec, EPI_RESID, x16, w16, bias ? bias : zero, one, resid, out, (int)M, N, K, st);
  };
  const long M0 = (long)B * 12544;
  // ---- stem: resize + conv 3x3 s2 (im2col GEMM) -> BN + SiLU -> conv 3x3 s1
  VTRY(launch_mv_pack_stem1(BTSBOT_F32, m + mv->stem1_w, one, k.w1p, st));
  VTRY(launch_mv_pack_conv3(BTSBOT_F32, m + mv->stem2_w, k.w2p, 64, 32, st));
  VTRY(launch_mv_resize_im2col(BTSBOT_F32, img, k.col1, B, st));
  VTRY(gemm(k.col1, k.w1p, nullptr, k.y1, M0, 32, 32));
  VTRY(bn_train(h, k.y1, mv->stem_bn, k.st_stem, sums_arena, k.a1s, M0, 32, 1, master, st));
  VTRY(launch_mv_im2col3(BTSBOT_F32, k.a1s, k.dC, B, 112, 32, st));
  VTRY(gemm(k.dC, k.w2p, nullptr, k.x0, M0, 64, 288));
  if (h->debug && h->taps[0]) HIP_TRY(hipMemcpyAsync(h->taps[0], k.x0, (size_t)M0 * 64 * 4, hipMemcpyDeviceToDevice, st));
  int stage = 0, jblk = 0;
  constexpr int DEPTHS[4] = {2, 2, 5, 2};
  for (size_t bi = 0; bi < mv->blocks.size(); ++bi) {
    const MvBlock& b = mv->blocks[bi];
    BlkAct& a = k.blk[bi];
    const long Min = (long)B * b.hin * b.hin, Mo = (long)B * b.hout * b.hout;
    const int hw2 = b.hout * b.hout;
    // shortcut: x | avgpool2(x) | avgpool2(x) Wsc^T -- goes straight into y, the projection adds to it
    const float* sc = a.xin;
    if (b.stride == 2) {
      VTRY(launch_mv_avgpool2(BTSBOT_F32, a.xin, a.pool_in, 0, B, b.hin, b.cin, st));
      sc = a.pool_in;
      if (b.sc_w >= 0) {
        VTRY(gemm(a.pool_in, m + b.sc_w, nullptr, a.y, Mo, b.c, b.cin));
        sc = a.y;
      }
    }
    VTRY(bn_train(h, a.xin, b.pre, a.st_pre, sums_arena, a.a0, Min, b.cin, 0, master, st));
    VTRY(gemm(a.a0, m + b.c1_w, m + b.c1_b, a.c1, Min, b.mid, b.cin));
    VTRY(bn_train(h, a.c1, b.n1, a.st1, sums_arena, a.a1, Min, b.mid, 1, master, st));
    VTRY(launch_mv_pack_dw(m + b.c2_w, one, k.g9, b.mid, st));
    hipLaunchKernelGGL(dw3_fwd_kernel, dim3(nblk(Mo * b.mid / 4)), dim3(256), 0, st, a.a1, k.g9, m + b.c2_b, a.d2, B, b.hin,
                       b.mid, b.stride);
    VTRY(bn_train(h, a.d2, b.n2, a.st2, sums_arena, a.a2, Mo, b.mid, 1, master, st));
    // squeeze-excite
    hipLaunchKernelGGL(alert_colsum_kernel, dim3((b.mid + 63) / 64, B), dim3(256), 0, st, a.a2, (const float*)nullptr,
                       a.sepool, hw2, b.mid, 1.0f / (float)hw2);
    hipLaunchKernelGGL(lin_fwd_small_kernel, dim3(nblk((long)B * b.rd)), dim3(256), 0, st, a.sepool, m + b.se1_w,
                       m + b.se1_b, a.rpre, a.r, B, b.mid, b.rd, 1);
    hipLaunchKernelGGL(lin_fwd_small_kernel, dim3(nblk((long)B * b.mid)), dim3(256), 0, st, a.r, m + b.se2_w, m + b.se2_b,
                       (float*)nullptr, a.g, B, b.rd, b.mid, 2);
    hipLaunchKernelGGL(gate_mul_kernel, dim3(nblk(Mo * b.mid / 4)), dim3(256), 0, st, a.a2, a.g, k.dA, hw2, b.mid, Mo * b.mid / 4);
    LAUNCH_CHECK();
    VTRY(gemm_resid(k.dA, m + b.c3_w, nullptr, sc, a.y, Mo, b.c, b.mid));
    const float* yin = a.y;
    for (int g = 0; g < 2; ++g) {
      const AttnPk& p = b.attn[g];
      AttnAct& t = a.at[g];
      const int c = b.c;
      VTRY(launch_mv_ln(BTSBOT_F32, yin, m + p.n1w, m + p.n1b, t.n1, Mo, c, st));
      VTRY(gemm(t.n1, m + p.qkv_w, m + p.qkv_b, t.qkv, Mo, 3 * c, c));
      VTRY(launch_mv_pack_relbias(m + p.rel, k.dbias, c / 32, st));
      VTRY(launch_mv_attn(BTSBOT_F32, t.qkv, k.dbias, t.o, B, b.hout, c, g, st));
      VTRY(gemm_resid(t.o, m + p.proj_w, m + p.proj_b, yin, t.y1, Mo, c, c));
      VTRY(launch_mv_ln(BTSBOT_F32, t.y1, m + p.n2w, m + p.n2b, t.n2, Mo, c, st));
      VTRY(gemm(t.n2, m + p.fc1_w, m + p.fc1_b, t.f1, Mo, 4 * c, c));
      hipLaunchKernelGGL(gelu_fwd_kernel, dim3(nblk(Mo * c)), dim3(256), 0, st, t.f1, t.gl, Mo * c);
      LAUNCH_CHECK();
      VTRY(gemm_resid(t.gl, m + p.fc2_w, m + p.fc2_b, t.y1, t.y2, Mo, c, 4 * c));
      yin = t.y2;
    }
    ++jblk;
    if (jblk == DEPTHS[stage]) {
      if (h->debug && h->taps[stage + 1])
        HIP_TRY(hipMemcpyAsync(h->taps[stage + 1], yin, (size_t)Mo * b.c * 4, hipMemcpyDeviceToDevice, st));
      ++stage;
      jblk = 0;
    }
  }
  float* feat = k.dsmall;   // [B][512] (copied out by the caller before the backward reuses the scratch)
  VTRY(launch_mv_final(k.xfin, m + mv->norm_w, m + mv->norm_b, feat, B, 49, 512, st));
  *feat_out = feat;
  return BTSBOT_OK;
}

int maxvit_train_backward(btsbot_ctx* h, const float* img, const float* dfeat, float* grads, int B, hipStream_t st) {
  MaxVit* mv = h->mv;
  const float* m = h->mirror;
  MvtCache k = carve(mv, h->bbcache, B, h->total_floats);
  MvtSums sums_arena{k.sums, 0};
  HIP_TRY(hipMemsetAsync(k.sums, 0, (size_t)MVT_SUM_SLOTS * 4096 * 4, st));
  auto F = [&](size_t off) { return reinterpret_cast<float*>(h->extra + off); };
  const float* zero = F(mv->p_zero);
  const int prec = h->cfg.precision;
  const bool lowp = prec == BTSBOT_BF16 || prec == BTSBOT_F16;   // (16-bit GEMM operands: see maxvit_train_forward)
  // dX [M][K] = dY [M][N] . W [N][K]  (the GEMM against the transposed filter)
  // (a filter gradient directly in front of the input gradient of the same layer has already cast dY)
  const float* d16_src = nullptr;
  int64_t d16_n = 0;
  auto dgrad = [&](const float* dY, const float* W, float* dX, long M, int N, int K) -> int {
    if (lowp) {
      if (!(d16_src == dY && d16_n == (int64_t)M * N)) VTRY(launch_cast(prec, dY, k.d16, (int64_t)M * N, st));
      d16_src = nullptr;
      VTRY(launch_transpose_cast(prec, W, nullptr, k.w16, N, K, st));
      return launch_gemm(prec, EPI_BIAS, k.d16, k.w16, zero, nullptr, nullptr, dX, (int)M, K, N, st);
    }
    VTRY(launch_transpose_f32(W, k.wt, N, K, st));
    return launch_gemm(BTSBOT_F32, EPI_BIAS, dY, k.wt, zero, nullptr, nullptr, dX, (int)M, K, N, st);
  };
  // dW [N][K] += dY^T X, db [N] += column sums of dY
  auto wgrad = [&](const float* dY, const float* X, float* dW, float* db, long M, int N, int K) -> int {
    d16_src = nullptr;
    if (lowp && N % 8 == 0 && K % 8 == 0) {
      VTRY(launch_cast(prec, dY, k.d16, (int64_t)M * N, st));
      d16_src = dY;
      d16_n = (int64_t)M * N;
      const void* x16 = nullptr;
      for (const auto& e : mv->xkept)   // the forward's copy of this input, if it made one (a buffer is cast where it is
        if (e.first == X) x16 = e.second;   // written last: scratch maps reused by the forward appear once per use -- the last wins)
      if (x16 == nullptr) {
        VTRY(launch_cast(prec, X, k.x16, (int64_t)M * K, st));
        x16 = k.x16;
      }
      return launch_wgrad16(prec, k.d16, x16, dW, db, (int)M, N, K, K, st, k.wpart, MVT_WPART_FLOATS);
    }
    return launch_wgrad_cs_f32(dY, X, dW, db, (int)M, N, K, K, st);
  };
  // ---- final LayerNorm2d + global average pool: d(xn)[b][p][c] = dfeat[b][c] / 49
  float* dy = k.dA;         // gradient w.r.t. the current map [rows][C]
  float* dt = k.dB;         // second map
  {
    const long n = (long)B * 49 * 512;
    hipLaunchKernelGGL(bcast_set_kernel, dim3(nblk(n)), dim3(256), 0, st, k.xn_final, dfeat, 49, 512, n, 1.0f / 49.0f);
    LAUNCH_CHECK();
    VTRY(launch_ln_bwd(k.xfin, k.xn_final, m + mv->norm_w, dy, grads + mv->norm_w, grads + mv->norm_b, (long)B * 49, 512, st));
  }
  for (int bi = (int)mv->blocks.size() - 1; bi >= 0; --bi) {
    const MvBlock& b = mv->blocks[bi];
    BlkAct& a = k.blk[bi];
    const long Min = (long)B * b.hin * b.hin, Mo = (long)B * b.hout * b.hout;
    const int hw2 = b.hout * b.hout, c = b.c;
    // ---- the two partition-attention layers, grid first (it ran last); dy = d(loss)/d(y2)
    for (int g = 1; g >= 0; --g) {
      const AttnPk& p = b.attn[g];
      AttnAct& t = a.at[g];
      const float* yin = g == 0 ? a.y : a.at[0].y2;
      // y2 = y1 + gl W2^T + b2
      VTRY(wgrad(dy, t.gl, grads + p.fc2_w, grads + p.fc2_b, Mo, c, 4 * c));
      VTRY(dgrad(dy, m + p.fc2_w, dt, Mo, c, 4 * c));                       // d(gl) [Mo][4c]
      hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblk(Mo * c)), dim3(256), 0, st, t.f1, dt, Mo * c);
      LAUNCH_CHECK();
      VTRY(wgrad(dt, t.n2, grads + p.fc1_w, grads + p.fc1_b, Mo, 4 * c, c));
      VTRY(dgrad(dt, m + p.fc1_w, k.dC, Mo, 4 * c, c));                     // d(n2) [Mo][c]
      VTRY(launch_ln_bwd(t.y1, k.dC, m + p.n2w, dt, grads + p.n2w, grads + p.n2b, Mo, c, st));
      hipLaunchKernelGGL(add_kernel, dim3(nblk(Mo * c / 4)), dim3(256), 0, st, dy, dt, Mo * c / 4);   // dy = d(y1)
      LAUNCH_CHECK();
      // y1 = yin + o Wp^T + bp
      VTRY(wgrad(dy, t.o, grads + p.proj_w, grads + p.proj_b, Mo, c, c));
      VTRY(dgrad(dy, m + p.proj_w, dt, Mo, c, c));                          // d(o) [Mo][c]
      const int heads = c / 32;
      HIP_TRY(hipMemsetAsync(k.dbias, 0, (size_t)heads * 2401 * 4, st));
      VTRY(launch_mv_pack_relbias(m + p.rel, k.dbias + 16 * 2401, heads, st));   // (bias_t in the slot's second half)
      {
        // three workgroups fit a CU (LDS): 768 x 4 of them walk the units, each for one head
        const long units = (long)B * (b.hout / 7) * (b.hout / 7);
        const long per_head = units < 3072 / heads ? units : 3072 / heads;
        hipLaunchKernelGGL(mv_attn_bwd_kernel, dim3((unsigned)(per_head * heads)), dim3(64), 0, st, t.qkv, k.dbias + 16 * 2401,
                           dt, k.dC, k.dbias, b.hout, c, g, (int)units);
        hipLaunchKernelGGL(relbias_grad_kernel, dim3(nblk(169 * heads)), dim3(256), 0, st, k.dbias, grads + p.rel, heads);
        LAUNCH_CHECK();
      }
      VTRY(wgrad(k.dC, t.n1, grads + p.qkv_w, grads + p.qkv_b, Mo, 3 * c, c));
      VTRY(dgrad(k.dC, m + p.qkv_w, dt, Mo, 3 * c, c));                     // d(n1) [Mo][c]
      VTRY(launch_ln_bwd(yin, dt, m + p.n1w, k.dC, grads + p.n1w, grads + p.n1b, Mo, c, st));
      hipLaunchKernelGGL(add_kernel, dim3(nblk(Mo * c / 4)), dim3(256), 0, st, dy, k.dC, Mo * c / 4);   // dy = d(yin)
      LAUNCH_CHECK();
    }
    // ---- MBConv: y = sc + (a2 * g) W3^T;  dy = d(loss)/d(y) [Mo][c]
    float* dx = dt;           // gradient w.r.t. the block's input [Min][cin], accumulated from three paths
    // a3 = a2 * g (recomputed), d(a3) = dy W3
    hipLaunchKernelGGL(gate_mul_kernel, dim3(nblk(Mo * b.mid / 4)), dim3(256), 0, st, a.a2, a.g, k.dC, hw2, b.mid, Mo * b.mid / 4);
    LAUNCH_CHECK();
    VTRY(wgrad(dy, k.dC, grads + b.c3_w, nullptr, Mo, c, b.mid));
    VTRY(dgrad(dy, m + b.c3_w, k.dC, Mo, c, b.mid));                        // d(a3) [Mo][mid]
    // squeeze-excite: dg[b][c] = sum_p d(a3) a2; d(a2) = d(a3) g (+ the pooled path below)
    float* dgate = k.dsmall;                    // [B][mid]
    float* dr = k.dsmall + (size_t)B * 2048;    // [B][rd] and scratch
    float* dpool = k.dsmall + (size_t)B * 2048 * 2;
    float* dgpre = k.dsmall + (size_t)B * 2048 * 3;
    hipLaunchKernelGGL(alert_colsum_kernel, dim3((b.mid + 63) / 64, B), dim3(256), 0, st, (const float*)k.dC,
                       (const float*)a.a2, dgate, hw2, b.mid, 1.0f);
    hipLaunchKernelGGL(gate_mul_kernel, dim3(nblk(Mo * b.mid / 4)), dim3(256), 0, st, (const float*)k.dC, a.g, k.dC, hw2, b.mid,
                       Mo * b.mid / 4);                                           // d(a2) = d(a3) * g, in place
    hipLaunchKernelGGL(act_bwd_small_kernel, dim3(nblk((long)B * b.mid)), dim3(256), 0, st, (const float*)dgate,
                       (const float*)a.g, dgpre, B * b.mid, 2);
    hipLaunchKernelGGL(lin_bwd_w_small_kernel, dim3(nblk((long)b.mid * b.rd)), dim3(256), 0, st, (const float*)dgpre,
                       (const float*)a.r, grads + b.se2_w, grads + b.se2_b, B, b.rd, b.mid);
    hipLaunchKernelGGL(lin_bwd_in_small_kernel, dim3(nblk((long)B * b.rd)), dim3(256), 0, st, (const float*)dgpre,
                       m + b.se2_w, dr, B, b.rd, b.mid);
    hipLaunchKernelGGL(act_bwd_small_kernel, dim3(nblk((long)B * b.rd)), dim3(256), 0, st, (const float*)dr,
                       (const float*)a.rpre, dr, B * b.rd, 1);
    hipLaunchKernelGGL(lin_bwd_w_small_kernel, dim3(nblk((long)b.rd * b.mid)), dim3(256), 0, st, (const float*)dr,
                       (const float*)a.sepool, grads + b.se1_w, grads + b.se1_b, B, b.mid, b.rd);
    hipLaunchKernelGGL(lin_bwd_in_small_kernel, dim3(nblk((long)B * b.mid)), dim3(256), 0, st, (const float*)dr, m + b.se1_w,
                       dpool, B, b.mid, b.rd);
    hipLaunchKernelGGL(bcast_add_kernel, dim3(nblk(Mo * b.mid)), dim3(256), 0, st, k.dC, (const float*)dpool, hw2, b.mid,
                       Mo * b.mid, 1.0f / (float)hw2);
    LAUNCH_CHECK();
    // BN2 + SiLU: d(d2) in place of d(a2)
    VTRY(bn_train_bwd(h, a.d2, k.dC, b.n2, a.st2, sums_arena, k.dC, grads, Mo, b.mid, 1, 0, st));
    // depthwise 3x3: filter / bias gradients (tap-major, then into the master layout), input gradient d(a1) [Min][mid]
    HIP_TRY(hipMemsetAsync(k.g9, 0, (size_t)10 * b.mid * 4, st));
    {
      const long npix = Mo;
      const dim3 grid((b.mid + 63) / 64, (unsigned)(npix / 64 > 256 ? 256 : (npix / 64 > 0 ? npix / 64 : 1)));
      hipLaunchKernelGGL(dw3_bwd_w_kernel, grid, dim3(256), 0, st, (const float*)a.a1, (const float*)k.dC, k.g9,
                         grads + b.c2_b, B, b.hin, b.mid, b.stride);
      hipLaunchKernelGGL(unpack_dw_grad_kernel, dim3(nblk(9L * b.mid)), dim3(256), 0, st, (const float*)k.g9, grads + b.c2_w,
                         b.mid);
      VTRY(launch_mv_pack_dw(m + b.c2_w, F(mv->p_one), k.g9, b.mid, st));
      hipLaunchKernelGGL(dw3_bwd_in_kernel, dim3(nblk(Min * b.mid / 4)), dim3(256), 0, st, (const float*)k.dC, (const float*)k.g9,
                         dx, B, b.hin, b.mid, b.stride);
      LAUNCH_CHECK();
    }
    // BN1 + SiLU: d(c1) in place; conv1 1x1
    VTRY(bn_train_bwd(h, a.c1, dx, b.n1, a.st1, sums_arena, dx, grads, Min, b.mid, 1, 0, st));
    VTRY(wgrad(dx, a.a0, grads + b.c1_w, grads + b.c1_b, Min, b.mid, b.cin));
    VTRY(dgrad(dx, m + b.c1_w, k.dC, Min, b.mid, b.cin));                   // d(a0) [Min][cin]
    // pre-norm BatchNorm (no activation): d(xin) = its input gradient ...
    VTRY(bn_train_bwd(h, a.xin, k.dC, b.pre, a.st_pre, sums_arena, dx, grads, Min, b.cin, 0, 0, st));
    // ... plus the shortcut's: dy through identity | avgpool2 | avgpool2 . Wsc
    if (b.stride == 1) {
      hipLaunchKernelGGL(add_kernel, dim3(nblk(Min * b.cin / 4)), dim3(256), 0, st, dx, (const float*)dy, Min * b.cin / 4);
    } else if (b.sc_w >= 0) {
      VTRY(wgrad(dy, a.pool_in, grads + b.sc_w, nullptr, Mo, c, b.cin));
      VTRY(dgrad(dy, m + b.sc_w, k.dC, Mo, c, b.cin));
      hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(nblk(Min * b.cin)), dim3(256), 0, st, (const float*)k.dC, dx, B, b.hin,
                         b.cin, 1);
    } else {
      hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(nblk(Min * b.cin)), dim3(256), 0, st, (const float*)dy, dx, B, b.hin, b.cin,
                         1);
    }
    LAUNCH_CHECK();
    float* tswap = dy;       // dx becomes the next (earlier) block's dy
    dy = dx;
    dt = tswap;
  }
  // ---- stem: x0 = im2col3(a1s) W2^T;  a1s = silu(BN(y1));  y1 = col1 W1^T
  {
    const long M0 = (long)B * 12544;
    VTRY(launch_mv_im2col3(BTSBOT_F32, k.a1s, k.dC, B, 112, 32, st));
    HIP_TRY(hipMemsetAsync(k.gconv, 0, (size_t)64 * 288 * 4, st));
    VTRY(wgrad(dy, k.dC, k.gconv, nullptr, M0, 64, 288));
    hipLaunchKernelGGL(unpack_conv3_grad_kernel, dim3(nblk(64 * 32 * 9)), dim3(256), 0, st, (const float*)k.gconv,
                       grads + mv->stem2_w, 64, 32, 288);
    LAUNCH_CHECK();
    VTRY(dgrad(dy, k.w2p, k.dC, M0, 64, 288));                              // d(col2) [M0][288]
    hipLaunchKernelGGL(col2im3_kernel, dim3(nblk(M0 * 32)), dim3(256), 0, st, (const float*)k.dC, dt, B, 112, 32);
    LAUNCH_CHECK();
    VTRY(bn_train_bwd(h, k.y1, dt, mv->stem_bn, k.st_stem, sums_arena, dt, grads, M0, 32, 1, 0, st));
    HIP_TRY(hipMemsetAsync(k.gconv, 0, (size_t)32 * 32 * 4, st));
    VTRY(wgrad(dt, k.col1, k.gconv, nullptr, M0, 32, 32));
    hipLaunchKernelGGL(unpack_conv3_grad_kernel, dim3(nblk(32 * 3 * 9)), dim3(256), 0, st, (const float*)k.gconv,
                       grads + mv->stem1_w, 32, 3, 32);
    LAUNCH_CHECK();
  }
  for (int i = 0; i < h->n_buckets; ++i) HIP_TRY(hipEventRecord(h->bucket_ev[i], st));
  return BTSBOT_OK;
}
