// ConvNeXt non-GEMM kernels for 63x63 ZTF triplets on gfx950: patch stem + LayerNorm,
// depthwise 7x7 + LayerNorm, downsample LayerNorm + 2x2 patch gather.
//
// Reference: the timm ConvNeXt graph that /root/reference/btsbot/architectures.py:108,132 builds
// (stem = Conv2d(3,C0,4,4)+LayerNorm2d; block = conv_dw 7x7 p3 groups=C -> LayerNorm2d(eps 1e-6)
// -> ...; downsample = LayerNorm2d + Conv2d(k2,s2)).  At 63x63 the maps are 15x15 -> 7x7 -> 3x3
// -> 1x1, so a whole alert's map fits in one CU's LDS and every kernel here works on whole maps.
//
// Layout: activations are NHWC ("pixel rows", channel contiguous), residual stream fp32.
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-6f;

// ---------------------------------------------------------------------------------------
// "transposing" wave reduction: each lane brings NV partial values; two swap-and-add steps
// (v_permlane32_swap, v_permlane16_swap: no selects, no LDS) halve the value count twice while
// summing across lane bits 5 and 4, then NV/4 values are butterflied inside each 16-lane row.
// Afterwards v[j], j < NV/4, holds the 64-lane total of value index (lane>>4)*(NV/4) + j.
// (A select-based variant makes hipcc index the value array dynamically: 16-way cndmask chains.)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float swap_add32(float a, float b) {
  // a' = {a.lo, b.lo}, b' = {a.hi, b.hi}; a'+b': lanes 0-31 total of a, lanes 32-63 total of b
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  // odd 16-lane rows of a swap with even rows of b; a'+b': even rows total of a, odd rows of b
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int NV> __device__ __forceinline__ void treduce(float (&v)[NV]) {
  static_assert(NV == 4 || NV == 8 || NV == 16, "treduce");
#pragma unroll
  for (int i = 0; i < NV / 2; ++i) v[i] = swap_add32(v[i], v[i + NV / 2]);
#pragma unroll
  for (int i = 0; i < NV / 4; ++i) v[i] = swap_add16(v[i], v[i + NV / 4]);
#pragma unroll
  for (int i = 0; i < NV / 4; ++i) v[i] = group16_sum(v[i]);
}

template <int C> struct Chunking {
  // lanes of a wave that own a channel (LPC) and waves per pixel row (NCH); LPC * NCH == C
  static constexpr int LPC = (C % 64 == 0) ? 64 : 40;
  static constexpr int NCH = C / LPC;
  static_assert(LPC * NCH == C, "unsupported channel count");
};

// ---------------------------------------------------------------------------------------
// depthwise 7x7 (pad 3) + bias + LayerNorm(C).  One workgroup = G whole maps staged in LDS.
// A wave owns one (map row, 64-channel chunk): lane = channel, the row's HW outputs live in
// registers, filter taps in 49 registers; out-of-map taps are skipped at compile time (x) or
// wave-uniformly (y).  LayerNorm statistics: transposing wave reduction (+ LDS across chunks),
// two-pass variance.
// ---------------------------------------------------------------------------------------
template <int C, int HW, int G, int NW, typename T>
__global__ __launch_bounds__(NW * 64) void dwconv_ln_kernel(
    const float* __restrict__ x, const float* __restrict__ wdw, const float* __restrict__ bdw,
    const float* __restrict__ lnw, const float* __restrict__ lnb, T* __restrict__ xn, int B,
    float* __restrict__ dsave) {
  constexpr int LPC = Chunking<C>::LPC, NCH = Chunking<C>::NCH;
  constexpr int P = HW * HW;
  constexpr int NV = HW <= 3 ? 4 : (HW <= 7 ? 8 : 16);
  constexpr int ITEMS = G * HW * NCH;
  constexpr int ROUNDS = (ITEMS + NW - 1) / NW;
  static_assert(NW % NCH == 0, "waves per workgroup must be a multiple of chunks per row");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                   // [G][P][C]
  float* red = smem + G * P * C;      // [2][NW][NV]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = wave % NCH;
  const bool act = lane < LPC;
  const int c = chunk * LPC + (act ? lane : 0);
  const int a0 = blockIdx.x * G;
  const int nal = min(G, B - a0);

  {  // stage the maps (coalesced 16-byte loads)
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)a0 * P * C);
    const int n4 = nal * P * C / 4;
    for (int i = tid; i < G * P * C / 4; i += NW * 64)
      reinterpret_cast<float4*>(xs)[i] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float w[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) w[t] = wdw[t * C + c];
  const float bias = bdw[c], g = lnw[c], bb = lnb[c];
  __syncthreads();

#pragma unroll 1
  for (int r = 0; r < ROUNDS; ++r) {
    const int item = r * NW + wave;
    const bool valid = item < ITEMS;
    const int slot = item / NCH;
    const int ga = slot / HW, y = slot - ga * HW;

    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = (i < HW) ? bias : 0.f;
    if (valid) {
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int iy = y + ky - 3;
        if (iy < 0 || iy >= HW) continue;
        const float* row = xs + ((ga * HW + iy) * HW) * C + c;
        float in[HW];
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) in[xx] = row[xx * C];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const int ix = xx + kx - 3;
            if (ix >= 0 && ix < HW) acc[xx] = fmaf(in[ix], w[ky * 7 + kx], acc[xx]);
          }
        }
      }
    }
    // ---- LayerNorm over C for each of the row's HW pixels
    float mean[NV], rstd[NV];
    float s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = (act && valid) ? acc[i] : 0.f;
    // Row totals go through LDS (also when one wave covers the row).
    constexpr int NR = NV / 4;
    treduce<NV>(s);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < NR; ++j) red[wave * NV + (lane >> 4) * NR + j] = s[j];
    }
    __syncthreads();
    const int w0 = wave - chunk;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) t += red[(w0 + j) * NV + i];
      mean[i] = t * (1.0f / C);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float d = acc[i] - mean[i];
      s[i] = (act && valid) ? d * d : 0.f;
    }
    treduce<NV>(s);
    float* red2 = red + NW * NV;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < NR; ++j) red2[wave * NV + (lane >> 4) * NR + j] = s[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) t += red2[(w0 + j) * NV + i];
      rstd[i] = rsqrtf(t * (1.0f / C) + LN_EPS);
    }
    if (valid && act && ga < nal) {
      T* dst = xn + ((size_t)(a0 + ga) * P + y * HW) * C + c;
#pragma unroll
      for (int xx = 0; xx < HW; ++xx)
        dst[xx * C] = (T)((acc[xx] - mean[xx]) * rstd[xx] * g + bb);
      if (dsave != nullptr) {   // training: the pre-LayerNorm map, so the backward need not recompute it
        float* dd = dsave + ((size_t)(a0 + ga) * P + y * HW) * C + c;
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) dd[xx * C] = acc[xx];
      }
    }
  }
}

// 3x3 maps (stage 2): every output sees at most the central 5x5 taps and all 9 inputs.  One
// workgroup per alert, lane = channel, all 9 outputs of a channel in one thread; LayerNorm needs the
// C-wide sums of 9 pixels: wave butterflies, then one LDS exchange across the C/64 waves.
template <int C, typename T>
__global__ __launch_bounds__(C) void dw3_ln_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ wdw,
                                                   const float* __restrict__ bdw,
                                                   const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb,
                                                   T* __restrict__ xn, int B,
                                                   float* __restrict__ dsave) {
  constexpr int NW = C / 64;
  __shared__ float red[2][NW][9];
  const int c = threadIdx.x, wave = c >> 6;
  const int a = blockIdx.x;
  const float* src = x + (size_t)a * 9 * C + c;
  float in[9];
#pragma unroll
  for (int p = 0; p < 9; ++p) in[p] = src[p * C];
  float acc[9];
  const float bias = bdw[c];
#pragma unroll
  for (int p = 0; p < 9; ++p) acc[p] = bias;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const float w = wdw[((dy + 3) * 7 + dx + 3) * C + c];
#pragma unroll
      for (int oy = 0; oy < 3; ++oy)
#pragma unroll
        for (int ox = 0; ox < 3; ++ox) {
          const int iy = oy + dy, ix = ox + dx;
          if (iy >= 0 && iy < 3 && ix >= 0 && ix < 3)
            acc[oy * 3 + ox] = fmaf(in[iy * 3 + ix], w, acc[oy * 3 + ox]);
        }
    }
  float mean[9], rstd[9];
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    const float s = wave_sum(acc[p]);
    if ((c & 63) == 0) red[0][wave][p] = s;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    float t = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) t += red[0][w2][p];
    mean[p] = t * (1.0f / C);
    const float d = acc[p] - mean[p];
    const float q = wave_sum(d * d);
    if ((c & 63) == 0) red[1][wave][p] = q;
  }
  __syncthreads();
  const float g = lnw[c], bb = lnb[c];
  T* dst = xn + (size_t)a * 9 * C + c;
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    float t = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) t += red[1][w2][p];
    rstd[p] = rsqrtf(t * (1.0f / C) + LN_EPS);
    dst[p * C] = (T)((acc[p] - mean[p]) * rstd[p] * g + bb);
    if (dsave != nullptr) dsave[(size_t)a * 9 * C + c + p * C] = acc[p];
  }
}

// 1x1 map (last stage): the 7x7 filter only ever sees its centre tap.  One wave per alert.
template <int CPT, typename T>
__global__ __launch_bounds__(256) void dw1_ln_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ wdw,
                                                     const float* __restrict__ bdw,
                                                     const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb,
                                                     T* __restrict__ xn, int B, int C,
                                                     float* __restrict__ dsave) {
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= B) return;
  float v[CPT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? fmaf(x[(size_t)a * C + c], wdw[24 * C + c], bdw[c]) : 0.f;
    sum += v[i];
  }
  const float mean = wave_sum(sum) / C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    sq += d * d;
  }
  const float rstd = rsqrtf(wave_sum(sq) / C + LN_EPS);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = lane + 64 * i;
    if (c < C) xn[(size_t)a * C + c] = (T)((v[i] - mean) * rstd * lnw[c] + lnb[c]);
    if (c < C && dsave != nullptr) dsave[(size_t)a * C + c] = v[i];
  }
}

// ---------------------------------------------------------------------------------------
// downsample prologue: LayerNorm(Cin) per input pixel, scattered into 2x2/s2 patch rows.
// 16 lanes per input pixel (4 pixels per wave), CPL = Cin/16 contiguous channels per lane; the odd
// last row/column of the map is never read (floor division).
// ---------------------------------------------------------------------------------------
template <int CPL, typename T>
__global__ __launch_bounds__(256) void ln_patch_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb,
                                                       T* __restrict__ patches, int B, int HW) {
  constexpr int Cin = CPL * 16;
  const int sub = threadIdx.x & 15;
  const int HO = HW / 2;
  const long total = (long)B * HO * HO * 4;
  long id = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = id < total;
  if (!live) id = total - 1;
  const int q = (int)(id & 3);  // ky*2 + kx
  long rest = id >> 2;
  const int ox = (int)(rest % HO);
  rest /= HO;
  const int oy = (int)(rest % HO);
  const int b = (int)(rest / HO);
  const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
  // a lane's channels: M4 consecutive ones from sub * M4 on (16-byte loads) and, where CPL is no multiple of 4
  // (convnext_nano: 80 / 160 channels), R more behind the 16 lanes' main part
  constexpr int M4 = CPL / 4 * 4, R = CPL - M4;
  const float* src = x + (((size_t)b * HW + iy) * HW + ix) * Cin;
  float v[CPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < M4; i += 4) {
    const float4 t = *reinterpret_cast<const float4*>(src + sub * M4 + i);
    v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) v[M4 + i] = src[16 * M4 + sub * R + i];
#pragma unroll
  for (int i = 0; i < CPL; ++i) sum += v[i];
  const float mean = group16_sum(sum) * (1.0f / Cin);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const float d = v[i] - mean;
    sq += d * d;
  }
  const float rstd = rsqrtf(group16_sum(sq) * (1.0f / Cin) + LN_EPS);
  if (!live) return;
  T* dst = patches + ((((size_t)b * HO + oy) * HO + ox) * 4 + q) * Cin;
  typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < M4; i += 4) {
    const float4 w4 = *reinterpret_cast<const float4*>(lnw + sub * M4 + i), b4 = *reinterpret_cast<const float4*>(lnb + sub * M4 + i);
    T4 o;
    o[0] = (T)((v[i] - mean) * rstd * w4.x + b4.x);
    o[1] = (T)((v[i + 1] - mean) * rstd * w4.y + b4.y);
    o[2] = (T)((v[i + 2] - mean) * rstd * w4.z + b4.z);
    o[3] = (T)((v[i + 3] - mean) * rstd * w4.w + b4.w);
    *reinterpret_cast<T4*>(dst + sub * M4 + i) = o;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int c = 16 * M4 + sub * R + i;
    dst[c] = (T)((v[M4 + i] - mean) * rstd * lnw[c] + lnb[c]);
  }
}

// ---------------------------------------------------------------------------------------
// stem: Conv2d(3, C0, k4, s4) + bias + LayerNorm(C0).  One workgroup per alert, lane = output
// pixel (225 of 256 lanes): the 48 patch values stay in registers, filter taps arrive as
// wave-uniform scalar loads, all C0 outputs of the pixel stay in registers so LayerNorm needs no
// cross-lane traffic.  The normalised row is transposed through padded LDS so the NHWC store is
// coalesced.  The last 3 rows/columns of the 63x63 cutout are never read (floor(63/4) = 15).
// ---------------------------------------------------------------------------------------
template <int C0>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img,
                                                   const float* __restrict__ w,  // [C0][48]
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb,
                                                   float* __restrict__ out, int B,
                                                   float* __restrict__ pre_out) {
  constexpr int LDW = C0 + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [225][C0+1]
  const int a = blockIdx.x;
  const int p = threadIdx.x;
  const bool live = p < 225;
  const int py = live ? p / 15 : 0, px = live ? p % 15 : 0;
  const float* src = img + (size_t)a * 3 * 63 * 63;
  float patch[48];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
        patch[(ci * 4 + ky) * 4 + kx] = src[(ci * 63 + 4 * py + ky) * 63 + 4 * px + kx];
  float o[C0];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < C0; ++c) {
    float acc = bias[c];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc = fmaf(patch[k], w[c * 48 + k], acc);
    o[c] = acc;
    sum += acc;
  }
  const float mean = sum * (1.0f / C0);
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < C0; ++c) {
    const float d = o[c] - mean;
    sq += d * d;
  }
  const float rstd = rsqrtf(sq * (1.0f / C0) + LN_EPS);
  if (live) {
#pragma unroll
    for (int c = 0; c < C0; ++c) smem[p * LDW + c] = (o[c] - mean) * rstd * lnw[c] + lnb[c];
  }
  __syncthreads();
  float* dst = out + (size_t)a * 225 * C0;
  for (int i = threadIdx.x; i < 225 * C0; i += 256) {
    const int pp = i / C0, c = i - pp * C0;
    dst[i] = smem[pp * LDW + c];
  }
  if (pre_out != nullptr) {   // training: the convolution's output before the LayerNorm, kept for its backward
    __syncthreads();
    if (live) {
#pragma unroll
      for (int c = 0; c < C0; ++c) smem[p * LDW + c] = o[c];
    }
    __syncthreads();
    float* dp = pre_out + (size_t)a * 225 * C0;
    for (int i = threadIdx.x; i < 225 * C0; i += 256) {
      const int pp = i / C0, c = i - pp * C0;
      dp[i] = smem[pp * LDW + c];
    }
  }
}

template <int C, int HW, int G, int NW, typename T>
int launch_dw_cfg(const float* x, const float* wdw, const float* bdw, const float* lnw,
                  const float* lnb, void* xn, int B, hipStream_t st, float* dsave) {
  constexpr int NV = HW <= 3 ? 4 : (HW <= 7 ? 8 : 16);
  const size_t lds = ((size_t)G * HW * HW * C + 2 * NW * NV) * sizeof(float);
  auto kern = dwconv_ln_kernel<C, HW, G, NW, T>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3((B + G - 1) / G), dim3(NW * 64), lds, st, x, wdw, bdw, lnw, lnb,
                     reinterpret_cast<T*>(xn), B, dsave);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T>
int launch_dw_typed(const float* x, const float* wdw, const float* bdw, const float* lnw,
                    const float* lnb, void* xn, int B, int HW, int C, hipStream_t st, float* dsave) {
#define DW_CASE(CC, HH, GG, WW) \
  if (C == CC && HW == HH) return launch_dw_cfg<CC, HH, GG, WW, T>(x, wdw, bdw, lnw, lnb, xn, B, st, dsave)
  DW_CASE(64, 15, 1, 8);
  DW_CASE(128, 7, 2, 8);
  if (HW == 3 && (C == 256 || C == 320)) {
    T* o = reinterpret_cast<T*>(xn);
    if (C == 256)
      hipLaunchKernelGGL((dw3_ln_kernel<256, T>), dim3(B), dim3(256), 0, st, x, wdw, bdw, lnw, lnb, o, B,
                         dsave);
    else
      hipLaunchKernelGGL((dw3_ln_kernel<320, T>), dim3(B), dim3(320), 0, st, x, wdw, bdw, lnw, lnb, o, B,
                         dsave);
    LAUNCH_CHECK();
    return BTSBOT_OK;
  }
  DW_CASE(80, 15, 1, 8);
  DW_CASE(160, 7, 2, 8);
#undef DW_CASE
  if (HW == 1) {
    const int cpt = (C + 63) / 64;
    dim3 grid((B + 3) / 4), blk(256);
    T* o = reinterpret_cast<T*>(xn);
    if (cpt <= 8)
      hipLaunchKernelGGL((dw1_ln_kernel<8, T>), grid, blk, 0, st, x, wdw, bdw, lnw, lnb, o, B, C, dsave);
    else if (cpt <= 10)
      hipLaunchKernelGGL((dw1_ln_kernel<10, T>), grid, blk, 0, st, x, wdw, bdw, lnw, lnb, o, B, C, dsave);
    else {
      btsbot_set_error("dwconv_ln: C=%d too wide for the 1x1 kernel", C);
      return BTSBOT_ERR_INVALID_ARG;
    }
    LAUNCH_CHECK();
    return BTSBOT_OK;
  }
  btsbot_set_error("dwconv_ln: no kernel for C=%d HW=%d", C, HW);
  return BTSBOT_ERR_INVALID_ARG;
}

template <typename T>
int launch_lnp_typed(const float* x, const float* lnw, const float* lnb, void* patches, int B,
                     int HW, int Cin, hipStream_t st) {
  const int HO = HW / 2;
  const long pix = (long)B * HO * HO * 4;
  dim3 grid((unsigned)((pix + 15) / 16)), blk(256);
  T* o = reinterpret_cast<T*>(patches);
#define LNP_CASE(CPL)                                                                          \
  if (Cin == CPL * 16) {                                                                       \
    hipLaunchKernelGGL((ln_patch_kernel<CPL, T>), grid, blk, 0, st, x, lnw, lnb, o, B, HW);    \
    LAUNCH_CHECK();                                                                            \
    return BTSBOT_OK;                                                                          \
  }
  LNP_CASE(4) LNP_CASE(8) LNP_CASE(16) LNP_CASE(5) LNP_CASE(10) LNP_CASE(20)
#undef LNP_CASE
  btsbot_set_error("ln_patch: no kernel for Cin=%d", Cin);
  return BTSBOT_ERR_INVALID_ARG;
}

}  // namespace

int launch_dwconv_ln(int prec, const float* x, const float* wdw, const float* bdw,
                     const float* lnw, const float* lnb, void* xn, int B, int HW, int C,
                     hipStream_t st, float* dsave) {
  if (B <= 0) return BTSBOT_OK;
  // 15x15 maps of the inference forward, 16-bit modes: on the matrix pipe (the training forward keeps the per-tap kernel:
  // its backward differentiates the convolution of the fp32 map.  Routed there too it held the oracle bounds of
  // tests/test_gpu_train.py and was worth 17 us of the 2.75 ms step: not taken)
  if (HW == 15 && dsave == nullptr && dw15_supported(prec, C))
    return launch_dw15_ln(prec, x, wdw, bdw, lnw, lnb, xn, B, C, st);
  switch (prec) {
    case BTSBOT_F32: return launch_dw_typed<float>(x, wdw, bdw, lnw, lnb, xn, B, HW, C, st, dsave);
    case BTSBOT_BF16: return launch_dw_typed<bf16_t>(x, wdw, bdw, lnw, lnb, xn, B, HW, C, st, dsave);
    case BTSBOT_F16: return launch_dw_typed<f16_t>(x, wdw, bdw, lnw, lnb, xn, B, HW, C, st, dsave);
  }
  btsbot_set_error("dwconv_ln: bad precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}

int launch_ln_patch(int prec, const float* x, const float* lnw, const float* lnb, void* patches,
                    int B, int HW, int Cin, hipStream_t st) {
  if (B <= 0 || HW < 2) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32: return launch_lnp_typed<float>(x, lnw, lnb, patches, B, HW, Cin, st);
    case BTSBOT_BF16: return launch_lnp_typed<bf16_t>(x, lnw, lnb, patches, B, HW, Cin, st);
    case BTSBOT_F16: return launch_lnp_typed<f16_t>(x, lnw, lnb, patches, B, HW, Cin, st);
  }
  btsbot_set_error("ln_patch: bad precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}

int launch_stem(const float* img, const float* w, const float* bias, const float* lnw,
                const float* lnb, float* out, int B, int C0, hipStream_t st, float* pre_out) {
  if (B <= 0) return BTSBOT_OK;
  const size_t lds = (size_t)225 * (C0 + 1) * sizeof(float);
  if (C0 == 64) {
    hipLaunchKernelGGL((stem_kernel<64>), dim3(B), dim3(256), lds, st, img, w, bias, lnw, lnb, out,
                       B, pre_out);
  } else if (C0 == 80) {
    static DevOnce attr_set;
    if (attr_set.need()) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_kernel<80>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.done();
    }
    hipLaunchKernelGGL((stem_kernel<80>), dim3(B), dim3(256), lds, st, img, w, bias, lnw, lnb, out,
                       B, pre_out);
  } else {
    btsbot_set_error("stem: no kernel for C0=%d", C0);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
