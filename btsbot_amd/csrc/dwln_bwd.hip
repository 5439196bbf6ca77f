// Backward of a ConvNeXt block's front half -- LayerNorm and the 7x7 depthwise convolution -- as ONE kernel
// (training step; replaces what autograd does for timm's ConvNeXtBlock.norm / .conv_dw between
// /root/reference/btsbot/train.py:510 and :526):
//
//     dd  = LN'(d) . dxn                       d = dwconv(x_in) + bias, kept by the forward;  dxn = d(loss)/d(LN out)
//     dg += dxn * xhat,  dbeta += dxn          (LayerNorm weight / bias gradients, per channel)
//     dW[c][t] += sum_p dd[p][c] x_in[p + delta_t][c],   db[c] += sum_p dd[p][c]      (depthwise filter / bias)
//     dy  = dy + conv(dd, flipped taps)        (gradient w.r.t. the block input, joins the residual path)
//
// Unfused (ln_bwd_kernel -> dw_wgrad_kernel -> dw_plain_kernel, backward.hip) dd makes a round trip through HBM and is
// read twice, and stages 2-3 pay three launch latencies for 9 MB tensors.  Here dd exists only in LDS:
// a workgroup takes `ga` alerts one after the other (3x3 maps: four per pass); per alert it stages x_in, runs the LayerNorm backward over the
// map's pixels (C/4 lanes per pixel, float4 pieces, result straight into LDS), then every thread = (channel, row group)
// runs both convolutions off the two LDS maps.  Filter-gradient taps stay in registers across the workgroup's alerts
// and leave, with the LayerNorm parameter gradients, as one partial row per workgroup which the caller column-sums
// into the arena (<= 64 atomics per column there instead of one per workgroup here).  fp32 throughout (every mode).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr float LN_EPS = 1e-6f;

template <int HW, int C, int NT, int NA>
__global__ __launch_bounds__(NT) void dwln_bwd_kernel(const float* __restrict__ d, const float* __restrict__ dxn,
                                                      const float* __restrict__ g, const float* __restrict__ xin,
                                                      const float* __restrict__ w, float* dy,
                                                      void* __restrict__ out16, int prec16,
                                                      float* __restrict__ partials, int B, int ga, int nplanes,
                                                      size_t pstride) {
  // The kernel moves 325 MB per launch at 15x15x64 (d, dxn, x_in, dy in; dy out twice: 81 us at 4 TB/s of its 98-113).
  // (Measured in round 5 and removed in round 6: recomputing d from x_in -- the third convolution cost more than the 59 MB
  //  saved, + 10 us per step -- and bf16 addend planes of dxn -- 10 us per step for 0.12 of the trajectory test's band.)
  // dxn may arrive as `nplanes` addends, pstride floats apart (mlp_bwd_kernel's hidden slices each write their own)
  extern __shared__ __attribute__((aligned(16))) float sm[];   // xs [NA P][C] | ds [NA P][C] | flipped taps [49][C]
  constexpr int P = HW * HW, PA = NA * P;   // NA alerts share a pass (3x3 maps: their latencies are paid once)
  constexpr int LPR = C / 4, R = 64 / LPR, NW = NT / 64;   // lanes per pixel row, rows per wave pass, waves
  constexpr int G = NT / C;                                // row groups of the convolution phase
  static_assert(LPR <= 64 && NT % C == 0 && 2 * NW * R * C <= 2 * PA * C, "geometry");
  float* xs = sm;
  float* ds = sm + PA * C;
  float* ws = ds + PA * C;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int c = tid % C, rg = tid / C;
  for (int i = tid; i < 49 * C; i += NT) {
    const int t = i / C, cc = i - t * C;
    ws[i] = w[(48 - t) * C + cc];
  }
  const float4 gl = *reinterpret_cast<const float4*>(g + 4 * l);
  const float ga4[4] = {gl.x, gl.y, gl.z, gl.w};
  float adg[4] = {0.f, 0.f, 0.f, 0.f}, adb[4] = {0.f, 0.f, 0.f, 0.f};
  float acc[49], ab = 0.f;
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = 0.f;
  auto gsum = [](float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
  const int a0 = blockIdx.x * ga, a1 = min(B, a0 + ga);
  for (int a = a0; a < a1; a += NA) {
    const int na = min(NA, a1 - a);          // alerts of this pass (consecutive alerts = consecutive rows)
    const int pa = na * P;
    const size_t base = (size_t)a * P * C;
    __syncthreads();   // the previous alert's convolutions have read both maps (first trip: the taps are staged)
    {
      // x_in -> LDS: eight 16-byte pieces in flight per thread before the first LDS store
      const float4* src = reinterpret_cast<const float4*>(xin + base);
      float4* dst = reinterpret_cast<float4*>(xs);
      constexpr int NB = PA * C / 4 >= 8 * NT ? 8 : (PA * C / 4 + NT - 1) / NT;
      const int n4 = pa * C / 4;
      for (int i0 = tid; i0 < n4; i0 += NT * NB) {
        float4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = src[min(i0 + k * NT, n4 - 1)];
#pragma unroll
        for (int k = 0; k < NB; ++k)
          if (i0 + k * NT < n4) dst[i0 + k * NT] = v[k];
      }
    }
    const float* dsrc = d + base;   // pixel rows of the kept depthwise output
    // LayerNorm backward, two row groups per wave pass (4 x 16-byte loads in flight per lane)
    for (int r0 = wv * R * 2; r0 < pa; r0 += NW * R * 2) {
      float4 v[2], dx[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * R + sub;
        ok[u] = r < pa;
        const int rr = ok[u] ? r : 0;
        v[u] = *reinterpret_cast<const float4*>(dsrc + (size_t)rr * C + 4 * l);
        dx[u] = *reinterpret_cast<const float4*>(dxn + base + (size_t)rr * C + 4 * l);
        for (int pl = 1; pl < nplanes; ++pl) {
          const float4 e = *reinterpret_cast<const float4*>(dxn + pl * pstride + base + (size_t)rr * C + 4 * l);
          dx[u].x += e.x;
          dx[u].y += e.y;
          dx[u].z += e.z;
          dx[u].w += e.w;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * R + sub;
        float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const float dxa[4] = {dx[u].x, dx[u].y, dx[u].z, dx[u].w};
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
          t[i] = dxa[i] * ga4[i];
          st += t[i];
          stx += t[i] * x[i];
          if (ok[u]) {
            adg[i] += dxa[i] * x[i];
            adb[i] += dxa[i];
          }
        }
        st = gsum(st) * (1.f / C);
        stx = gsum(stx) * (1.f / C);
        if (ok[u])
          *reinterpret_cast<float4*>(ds + r * C + 4 * l) =
              make_float4(rstd * (t[0] - st - x[0] * stx), rstd * (t[1] - st - x[1] * stx),
                          rstd * (t[2] - st - x[2] * stx), rstd * (t[3] - st - x[3] * stx));
      }
    }
    __syncthreads();
    // both convolutions, thread = (channel, row group)
    for (int q = rg; q < na * HW; q += G) {
      const int al = q / HW, y = q - al * HW;       // (alert of the pass, map row)
      const float* xsa = xs + al * P * C;
      const float* dsa = ds + al * P * C;
      const size_t abase = base + (size_t)al * P * C;
      float gy[HW], ax[HW];
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) ax[xx] = dy[abase + (size_t)(y * HW + xx) * C + c];   // requested first
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) {
        gy[xx] = dsa[(y * HW + xx) * C + c];
        ab += gy[xx];
      }
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int iy = y + ky - 3;
        if (iy < 0 || iy >= HW) continue;
        float in[HW];
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) in[xx] = xsa[(iy * HW + xx) * C + c];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const int ix = xx + kx - 3;
            if (ix >= 0 && ix < HW) acc[ky * 7 + kx] = fmaf(gy[xx], in[ix], acc[ky * 7 + kx]);
          }
        float wk[7];
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) in[xx] = dsa[(iy * HW + xx) * C + c];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) wk[kx] = ws[(ky * 7 + kx) * C + c];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const int ix = xx + kx - 3;
            if (ix >= 0 && ix < HW) ax[xx] = fmaf(in[ix], wk[kx], ax[xx]);
          }
      }
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) {
        const size_t o = abase + (size_t)(y * HW + xx) * C + c;
        dy[o] = ax[xx];
        if (out16 != nullptr) {   // the same values as the next 16-bit GEMM's operand
          if (prec16 == BTSBOT_BF16) reinterpret_cast<bf16_t*>(out16)[o] = (bf16_t)ax[xx];
          else reinterpret_cast<f16_t*>(out16)[o] = (f16_t)ax[xx];
        }
      }
    }
  }
  // the row groups' filter-gradient taps meet in LDS (the maps are dead) by halving: the upper half of the groups
  // parks its taps ([G/2][50][C] floats at most: fits the maps' footprint), the lower half adds them; the workgroup
  // leaves ONE partial row: [C][49] filter taps, then [C] biases
#pragma unroll
  for (int half = G / 2; half >= 1; half >>= 1) {
    __syncthreads();
    if (rg >= half && rg < 2 * half) {
      float* red = sm + (size_t)(rg - half) * 50 * C;
#pragma unroll
      for (int t = 0; t < 49; ++t) red[t * C + c] = acc[t];
      red[49 * C + c] = ab;
    }
    __syncthreads();
    if (rg < half) {
      const float* red = sm + (size_t)rg * 50 * C;
#pragma unroll
      for (int t = 0; t < 49; ++t) acc[t] += red[t * C + c];
      ab += red[49 * C + c];
    }
  }
  float* row = partials + (size_t)blockIdx.x * 52 * C;
  if (rg == 0) {
#pragma unroll
    for (int t = 0; t < 49; ++t) row[(size_t)c * 49 + t] = acc[t];
    row[(size_t)49 * C + c] = ab;
  }
  // LayerNorm parameter gradients: the wave's row groups and the waves meet in LDS, then join the partial row
  // (no same-address atomics here: 512 workgroups x 2 C of them cost ~8 us per launch in ln_bwd_kernel)
  __syncthreads();
  float* sh = sm;   // [2][NW * R][C]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sh[(wv * R + sub) * C + 4 * l + i] = adg[i];
    sh[(NW * R + wv * R + sub) * C + 4 * l + i] = adb[i];
  }
  __syncthreads();
  for (int cc = tid; cc < C; cc += NT) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int j = 0; j < NW * R; ++j) {
      sa += sh[j * C + cc];
      sb += sh[(NW * R + j) * C + cc];
    }
    row[(size_t)50 * C + cc] = sa;
    row[(size_t)51 * C + cc] = sb;
  }
}

// ---- 3x3 maps (stage 2: 256 channels): thread = channel, an alert's nine pixels of dxn / x_in / dy in registers.
// The depthwise output d is RECOMPUTED from x_in (81 FMAs) instead of read; both backward convolutions are 81 register
// FMAs per alert; the LayerNorm sums over the 256 channels are wave sums that meet through 16-byte LDS rows.  One alert
// per pass and <= 168 registers: three workgroups per CU cover each other's load latencies and barriers (two alerts per
// pass took 306 registers, one workgroup per CU: 43 us).  The workgroup leaves a COMPACT partial row, tap-major
// [25 live taps | bias | LayerNorm weight | LayerNorm bias][256] = 28 KB of coalesced stores (the general kernel's row is
// the arena's [C][49] layout: 52 KB per workgroup in 4-byte pieces 196 bytes apart, 27 MB per launch for 9.4 MB tensors);
// dw3_rows_kernel adds the rows into the arena.  The general kernel above spends a 3x3 map's pass in its phase latencies
// and took 31-36 us per launch (six launches per step in the chain) + 18-23 us for the column sum of its rows.
constexpr int DW3_ROW = 28;   // floats per channel of a compact partial row
__global__ __launch_bounds__(256, 3) void dw3ln_bwd_kernel(const float* __restrict__ dwb, const float* __restrict__ dxn,
                                                           const float* __restrict__ g, const float* __restrict__ xin,
                                                           const float* __restrict__ w, float* dy,
                                                           void* __restrict__ out16, int prec16,
                                                           float* __restrict__ partials, int B, int ga, int nplanes,
                                                           size_t pstride) {
  constexpr int C = 256;
  __shared__ __attribute__((aligned(16))) float red[4][9][4];   // [set][pixel][wave]
  const int c = threadIdx.x, lane = c & 63, wv = c >> 6;
  float wt[25];   // central 5x5 of the 7x7 taps (the others never meet a 3x3 map)
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) wt[ky * 5 + kx] = w[((ky + 1) * 7 + kx + 1) * C + c];
  const float gc = g[c], bias = dwb[c];
  float acc[25], ab = 0.f, adg = 0.f, adb = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = 0.f;
  // every lane ends with the four waves' total of pixel p's value, set s
  auto put = [&](int s, int p, float v) {
    const float t = wave_sum(v);
    if (lane == 0) red[s][p][wv] = t;
  };
  auto get = [&](int s, int p) {
    const float4 r = *reinterpret_cast<const float4*>(&red[s][p][0]);
    return (r.x + r.y) + (r.z + r.w);
  };
  const int a0 = blockIdx.x * ga, a1 = min(B, a0 + ga);
  for (int a = a0; a < a1; ++a) {
    const size_t base = (size_t)a * 9 * C + c;
    float xv[9], gx[9], yv[9], dv[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      xv[p] = xin[base + p * C];
      gx[p] = dxn[base + p * C];
      yv[p] = dy[base + p * C];
    }
    for (int pl = 1; pl < nplanes; ++pl)
#pragma unroll
      for (int p = 0; p < 9; ++p) gx[p] += dxn[pl * pstride + base + p * C];
    // d[o] = bias + sum_i w[iy - oy + 2][ix - ox + 2] x[i]   (the forward's dw3_ln_kernel, same order of taps)
#pragma unroll
    for (int o = 0; o < 9; ++o) dv[o] = bias;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx)
#pragma unroll
        for (int o = 0; o < 9; ++o) {
          const int iy = o / 3 + ky - 2, ix = o % 3 + kx - 2;
          if (iy >= 0 && iy < 3 && ix >= 0 && ix < 3) dv[o] = fmaf(xv[iy * 3 + ix], wt[ky * 5 + kx], dv[o]);
        }
#pragma unroll
    for (int p = 0; p < 9; ++p) put(0, p, dv[p]);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      dv[p] -= get(0, p) * (1.f / C);
      put(1, p, dv[p] * dv[p]);
    }
    __syncthreads();
    float rstd[9], t1[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      rstd[p] = rsqrtf(get(1, p) * (1.f / C) + LN_EPS);
      dv[p] *= rstd[p];                 // xhat
      t1[p] = gx[p] * gc;
      adg += gx[p] * dv[p];
      adb += gx[p];
      put(2, p, t1[p]);
      put(3, p, t1[p] * dv[p]);
    }
    __syncthreads();
    float dd[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      const float st = get(2, p) * (1.f / C), stx = get(3, p) * (1.f / C);
      dd[p] = rstd[p] * (t1[p] - st - dv[p] * stx);
      ab += dd[p];
    }
    // dx[i] += w[..] dd[o],  dW[..] += dd[o] x[i]
#pragma unroll
    for (int o = 0; o < 9; ++o)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int t = (i / 3 - o / 3 + 2) * 5 + (i % 3 - o % 3 + 2);
        yv[i] = fmaf(wt[t], dd[o], yv[i]);
        acc[t] = fmaf(dd[o], xv[i], acc[t]);
      }
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      dy[base + p * C] = yv[p];
      if (out16 != nullptr) {
        if (prec16 == BTSBOT_BF16) reinterpret_cast<bf16_t*>(out16)[base + p * C] = (bf16_t)yv[p];
        else reinterpret_cast<f16_t*>(out16)[base + p * C] = (f16_t)yv[p];
      }
    }
  }
  float* row = partials + (size_t)blockIdx.x * DW3_ROW * C + c;
#pragma unroll
  for (int t = 0; t < 25; ++t) row[t * C] = acc[t];
  row[25 * C] = ab;
  row[26 * C] = adg;
  row[27 * C] = adb;
}

// arena[...] += sum over the compact rows.  out = the gradient of conv_dw.weight; conv_dw.bias, norm.weight and
// norm.bias follow it (the arena's order: [C][49] | [C] | [C] | [C]).  Workgroup = 64 columns x 4 row groups over one
// slice of the rows; the slices meet in the arena through one atomic per column and slice.
__global__ __launch_bounds__(256) void dw3_rows_kernel(const float* __restrict__ rows, float* out, int nrows, int rslice) {
  constexpr int C = 256, N = DW3_ROW * C;
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  const int rbeg = blockIdx.y * rslice, rend = min(nrows, rbeg + rslice);
  float s0 = 0.f, s1 = 0.f;
  int r = rbeg + rg;
  for (; r + 4 < rend; r += 8) {
    s0 += rows[(size_t)r * N + n];
    s1 += rows[(size_t)(r + 4) * N + n];
  }
  if (r < rend) s0 += rows[(size_t)r * N + n];
  sh[rg][cl] = s0 + s1;
  __syncthreads();
  if (rg != 0) return;
  const float v = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
  const int t = n / C, c = n - t * C;
  const int o = t < 25 ? c * 49 + (t / 5 + 1) * 7 + t % 5 + 1 : (49 + t - 25) * C + c;
  if (gridDim.y == 1) out[o] += v;   // (one slice: a fixed order of additions, the deterministic mode)
  else atomicAdd(out + o, v);
}

// workgroups of the 3x3 kernel: at most 512 (up to three fit a CU), each a whole number of alerts
int dw3_ga(int B) {
  static const int wgs = [] {
    const char* e = getenv("BTSBOT_AMD_DW3_WGS");   // tuning knob
    const int v = e ? atoi(e) : 0;
    return v >= 1 ? v : 512;
  }();
  const int per = (B + wgs - 1) / wgs;
  return per < 1 ? 1 : per;
}
int dw3_grid(int B) { return (B + dw3_ga(B) - 1) / dw3_ga(B); }
bool dw3_old() {
  static const bool v = [] {
    const char* e = getenv("BTSBOT_AMD_DW3_OLD");   // 1: the general kernel on the 3x3 maps too (A/B timing)
    return e != nullptr && e[0] == '1';
  }();
  return v;
}

template <int HW, int C, int NT, int NA> struct DwlnCfg {
  static constexpr size_t lds = ((size_t)2 * NA * HW * HW + 49) * C * sizeof(float);
  // one workgroup per CU where the maps take 128 KB (15x15x64), two where they take <= 80 KB
  static constexpr int WGS = lds > 80 * 1024 ? 256 : 512;
  static int ga(int B) { return B < WGS ? 1 : (B + WGS - 1) / WGS; }
  static int grid(int B) { return (B + ga(B) - 1) / ga(B); }
};

template <int HW, int C, int NT, int NA>
int dwln_launch(const float* d, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                void* out16, int prec16, float* partials, int B, hipStream_t st, int nplanes, size_t pstride) {
  using K = DwlnCfg<HW, C, NT, NA>;
  constexpr int G = NT / C;
  static_assert((size_t)(G / 2) * 50 * C * sizeof(float) <= K::lds, "closing reduction fits the maps' footprint");
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dwln_bwd_kernel<HW, C, NT, NA>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::lds));
    attr.done();
  }
  hipLaunchKernelGGL((dwln_bwd_kernel<HW, C, NT, NA>), dim3(K::grid(B)), dim3(NT), K::lds, st, d, dxn, g, xin, w, dy,
                     out16, prec16, partials, B, K::ga(B), nplanes, pstride);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

// the (map, width) pairs of convnext_pico's stages 0-2; everything else keeps the three-kernel form
bool dwln_bwd_supported(int HW, int C) { return (HW == 15 && C == 64) || (HW == 7 && C == 128) || (HW == 3 && C == 256); }

// partial rows one launch writes for a batch of B alerts (each 52 * C floats)
int dwln_bwd_rows(int HW, int C, int B) {
  if (B <= 0) return 0;
  if (HW == 15 && C == 64) return DwlnCfg<15, 64, 512, 1>::grid(B);
  if (HW == 7 && C == 128) return DwlnCfg<7, 128, 512, 1>::grid(B);
  if (HW == 3 && C == 256) {   // (room for either kernel's rows: BTSBOT_AMD_DW3_OLD=1 runs the general one; the deterministic mode keeps the 3x3 kernel and adds its rows in one slice)
    const int r0 = DwlnCfg<3, 256, 512, 4>::grid(B), r1 = dw3_grid(B);
    return r0 > r1 ? r0 : r1;
  }
  return 0;
}

// w: taps [49][C].  partials: dwln_bwd_rows() x 52 * C floats; row = [C][49] depthwise filter | [C] depthwise bias |
// [C] LayerNorm weight | [C] LayerNorm bias gradients of one workgroup -- the master arena's layout of those four
// tensors, so the caller finishes with ONE column sum of the rows into the arena (launch_colsum, any stream).
int launch_dwln_bwd(const float* d, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                    void* out16, int prec16, float* partials, int B, int HW, int C, hipStream_t st, int nplanes,
                    size_t pstride) {
  if (B <= 0) return BTSBOT_OK;
  if (d == nullptr) {
    btsbot_set_error("dwln_bwd: the kept depthwise output is missing");
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (HW == 15 && C == 64) return dwln_launch<15, 64, 512, 1>(d, dxn, g, xin, w, dy, out16, prec16, partials, B, st, nplanes, pstride);
  if (HW == 7 && C == 128) return dwln_launch<7, 128, 512, 1>(d, dxn, g, xin, w, dy, out16, prec16, partials, B, st, nplanes, pstride);
  if (HW == 3 && C == 256) return dwln_launch<3, 256, 512, 4>(d, dxn, g, xin, w, dy, out16, prec16, partials, B, st, nplanes, pstride);
  btsbot_set_error("dwln_bwd: no kernel for a %dx%d map of %d channels", HW, HW, C);
  return BTSBOT_ERR_INVALID_ARG;
}

// The 3x3 maps' own kernel (256 channels) unless BTSBOT_AMD_DW3_OLD=1.  launch_dw3ln_bwd recomputes the depthwise output from x_in and the depthwise bias `dwb`;
// launch_dw3_rows (any stream that has seen the kernel) adds its dw3_rows(B) compact partial rows into the arena at
// out = gradient of conv_dw.weight, with conv_dw.bias | norm.weight | norm.bias behind it.
bool dw3_bwd_active(int HW, int C) { return HW == 3 && C == 256 && !dw3_old(); }
int dw3_rows(int B) { return dw3_grid(B); }
int launch_dw3ln_bwd(const float* dwb, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                     void* out16, int prec16, float* partials, int B, hipStream_t st, int nplanes, size_t pstride) {
  if (B <= 0) return BTSBOT_OK;
  hipLaunchKernelGGL(dw3ln_bwd_kernel, dim3(dw3_grid(B)), dim3(256), 0, st, dwb, dxn, g, xin, w, dy, out16, prec16, partials,
                     B, dw3_ga(B), nplanes, pstride);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
int launch_dw3_rows(const float* partials, float* out, int nrows, hipStream_t st) {
  if (nrows <= 0) return BTSBOT_OK;
  // (deterministic mode: one slice, i.e. every column is added by one thread group in a fixed order)
  const int nsl = det_alloc(0) != nullptr ? 1 : nrows >= 256 ? 8 : nrows >= 32 ? 4 : 1, rslice = (nrows + nsl - 1) / nsl;
  hipLaunchKernelGGL(dw3_rows_kernel, dim3(DW3_ROW * 256 / 64, (nrows + rslice - 1) / rslice), dim3(256), 0, st, partials, out,
                     nrows, rslice);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
