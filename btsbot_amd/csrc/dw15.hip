// Depthwise 7x7 (pad 3) + bias + LayerNorm(C) of the per-op schedule's 15x15 maps on the matrix pipe (gfx950, 16-bit
// modes): timm ConvNeXtBlock.conv_dw / .norm of stage 0, reached from /root/reference/btsbot/architectures.py:108,132.
// The same computation as convnext.hip's dwconv_ln_kernel (one FMA per tap and output: 735 FMAs + 105 LDS reads per
// (row, channel), 28.5 k issue slots per alert at C = 80 with 40 of a wave's 64 lanes live -- 65 us per 1024 alerts,
// three times its HBM time) in stage0b.hip's formulation:
//   * the map in LDS as a planar 16-bit image [channel][x quad][row -3..18][4 x] (8-byte entries, zero padded);
//   * per channel (= one block of the 16-block 4x4x4 MFMA) and 4x4 output tile
//         D[i][j] = out[4 yb + j][4 xb + i] = sum_{ky, rb, k} W[ky][4 rb + k - i + 3] in[4 yb + j + ky - 3][4 (xb + rb) + k]
//     A = Toeplitz taps (21 register fragments per lane, built here from the fp32 tap-major filter: nothing packed),
//     B = 4 consecutive x of 4 consecutive rows (one ds_read_b64): 19 row steps, 280 products per wave (16 channels);
//   * LayerNorm: transposing lane reduction over the wave's 16 channels, the C / 16 waves meet in LDS (single pass);
//   * the normalised rows leave through a [pixel][channel] LDS image (overlays the planar one) as 16-byte pieces.
// Workgroup = C / 16 waves, one alert at a time (two workgroups per CU).
// The map enters the products rounded to the operand type (as in the fused stage-0 kernel); sums and LayerNorm are fp32.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <typename T> struct D4;
template <> struct D4<bf16_t> {
  static __device__ __forceinline__ f32x4 run(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <> struct D4<f16_t> {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 run(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0);
  }
};

constexpr int HW = 15, P = 225;
constexpr int PL_ROWS = 22, PL_XQ = PL_ROWS * 8, PL_CH = 800;   // (stage0b.hip: the channel stride is 4 mod 32 entries)
constexpr float LN_EPS = 1e-6f;

__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Transposing sum over the 16 blocks (lane bits 2..5) of 64 values per lane (SQ: of their squares);
// out[0..3] = the 16-lane totals of values 16 (lane >> 4) + 4 ((lane >> 2) & 3) + 0..3   (stage0b.hip)
template <bool SQ> __device__ __forceinline__ void block_reduce64(const float (&v)[64], int lane, float (&out)[4]) {
  float w[32];
#pragma unroll
  for (int n = 0; n < 32; ++n)
    w[n] = SQ ? swap_add32(v[n] * v[n], v[n + 32] * v[n + 32]) : swap_add32(v[n], v[n + 32]);
#pragma unroll
  for (int n = 0; n < 16; ++n) w[n] = swap_add16(w[n], w[n + 16]);
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const float own = b3 ? w[n + 8] : w[n], send = b3 ? w[n] : w[n + 8];
    w[n] = own + dpp_mov<0x128>(send);                       // row_ror:8 = lane ^ 8
  }
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float own = b2 ? w[n + 4] : w[n], send = b2 ? w[n] : w[n + 4];
    const float lo = dpp_mov<0x124>(send), hi = dpp_mov<0x12C>(send);   // row_ror:4 / :12 = lane - 4 / lane + 4
    out[n] = own + (b2 ? lo : hi);
  }
}

template <int C> struct L15 {
  static constexpr int NW = C / 16, NT = NW * 64;    // waves / threads that work on one alert
  // C = 80: two five-wave teams (two alerts) per workgroup, one workgroup per CU -- two five-wave workgroups are only
  // placed together on a CU whose SIMDs all have room for two more waves (55-70 us per 1024 alerts that way)
  static constexpr int TEAMS = NW == 5 ? 2 : 1;
  static constexpr int PLB = C * PL_CH;                 // planar image
  static constexpr int PITCH = 2 * C + 16;              // [pixel][channel] image: bytes per pixel row
  static constexpr int OFF_PART = PLB;                  // [2][NW][256] floats
  static constexpr int OFF_ST = OFF_PART + 2 * NW * 256 * 4;   // [2][256] floats: rstd, -mean * rstd per pixel slot
  static constexpr int BYTES = OFF_ST + 2 * 256 * 4;
  static_assert(256 * PITCH <= PLB, "the [pixel][channel] image overlays the planar one");
  static_assert(BYTES <= 80 * 1024, "two alerts per CU");
};

template <typename T, int C>
__global__ __launch_bounds__(L15<C>::NT * L15<C>::TEAMS, L15<C>::TEAMS == 2 ? 3 : 2) void dw15_ln_kernel(const float* __restrict__ x, const float* __restrict__ wdw,
                                                             const float* __restrict__ bdw,
                                                             const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, T* __restrict__ xn, int B) {
  using L = L15<C>;
  constexpr int NW = L::NW, NT = L::NT, PITCH = L::PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int team = L::TEAMS == 2 ? (int)threadIdx.x / NT : 0;      // (wave-uniform: NT is a whole number of waves)
  unsigned char* smem = smem_all + team * L::BYTES;
  unsigned char* pl = smem;
  unsigned char* map = smem;                                       // overlay, never live together
  float* part = reinterpret_cast<float*>(smem + L::OFF_PART);
  float* st = reinterpret_cast<float*>(smem + L::OFF_ST);
  const int tid = (int)threadIdx.x - team * NT, lane = tid & 63, wave = tid >> 6;
  // depthwise roles: lane = (block b = channel 16 wave + b, row offset j)
  const int dj = lane & 3, dch = wave * 16 + (lane >> 2);
  const int dyb = lane >> 4, dxb = (lane >> 2) & 3;                // where this lane's LayerNorm sums end up
  // Toeplitz taps of this lane's channel: tw[ky * 3 + rb + 1][k] = W[ky][4 rb + k - dj + 3]  (0 outside the 7 taps)
  s16x4 tw[21];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int rbi = 0; rbi < 3; ++rbi) {
      T q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kx = 4 * (rbi - 1) + k - dj + 3;
        const bool ok = kx >= 0 && kx < 7;
        const float w = wdw[(ky * 7 + (ok ? kx : 0)) * C + dch];
        q[k] = (T)(ok ? w : 0.f);
      }
      tw[ky * 3 + rbi] = s16x4{__builtin_bit_cast(short, q[0]), __builtin_bit_cast(short, q[1]),
                               __builtin_bit_cast(short, q[2]), __builtin_bit_cast(short, q[3])};
    }
  const float dwbias = bdw[dch], lng = lnw[dch], lnb2 = lnb[dch];

  // C = 64 (four waves): a workgroup takes alerts blockIdx.x, + grid, ... and builds its taps once (32 us per 1024
  // alerts against 36 with one alert per workgroup); C = 80: one alert per team
  constexpr bool LOOP = C == 64;
  int a = blockIdx.x * L::TEAMS + team;
  const bool live = a < B;          // (an odd batch: the last workgroup's second team walks the barriers on alert B - 1
  if (!live) a = B - 1;             //  and writes nothing)
  do {
    // ---- planar image: zero (pad rows / columns; the previous alert's overlay), then this alert's map
    for (int i = tid; i < L::PLB / 16; i += NT) reinterpret_cast<uint4*>(pl)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    {
      // item = (row y, x quad, channel quad): 4 pixels x 4 channels in, four 8-byte entries (4 x of one channel) out
      constexpr int CQ = C / 4, ITEMS = HW * 4 * CQ;
      const float* src = x + (size_t)a * P * C;
      for (int it = tid; it < ITEMS; it += NT) {
        const int c4 = it % CQ, rq = it / CQ;
        const int xq = rq & 3, y = rq >> 2;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = 4 * xq + k;
          v[k] = xx < HW ? *reinterpret_cast<const float4*>(src + ((size_t)(y * HW + xx)) * C + 4 * c4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        unsigned char* dst = pl + (size_t)(4 * c4) * PL_CH + xq * PL_XQ + (y + 3) * 8;
        const float e[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x},
                               {v[0].y, v[1].y, v[2].y, v[3].y},
                               {v[0].z, v[1].z, v[2].z, v[3].z},
                               {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          T q[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = (T)e[cc][k];
          *reinterpret_cast<s16x4*>(dst + cc * PL_CH) =
              s16x4{__builtin_bit_cast(short, q[0]), __builtin_bit_cast(short, q[1]), __builtin_bit_cast(short, q[2]),
                    __builtin_bit_cast(short, q[3])};
        }
      }
    }
    __syncthreads();

    // ---- depthwise 7x7: a row step s = 4 yb + ky serves every (yb, ky) pair with that sum
    float v[64];
    {
      f32x4 acc[4][4];
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb) acc[yb][xb] = f32x4{dwbias, dwbias, dwbias, dwbias};
      const unsigned char* lb = pl + dch * PL_CH + dj * 8;
      s16x4 bq[2][4];
      auto read_step = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bq[buf][q] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(lb + q * PL_XQ + s * 8));
      };
      read_step(0, 0);
#pragma unroll
      for (int s = 0; s < 19; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bq[s & 1][q]));   // wait for this step's reads here
        if (s + 1 < 19) read_step(s + 1, (s + 1) & 1);
#pragma unroll
        for (int yb = 0; yb < 4; ++yb) {
          const int ky = s - 4 * yb;
          if (ky < 0 || ky > 6) continue;
#pragma unroll
          for (int rbi = 0; rbi < 3; ++rbi)
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) {
              const int q = xb + rbi - 1;
              if (q < 0 || q > 3) continue;
              acc[yb][xb] = D4<T>::run(tw[ky * 3 + rbi], bq[s & 1][q], acc[yb][xb]);
            }
        }
      }
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[yb * 16 + xb * 4 + i] = acc[yb][xb][i];
    }
    // ---- LayerNorm over the C channels of a pixel: this wave's 16 blocks (transposing lane reduction), then the
    //      waves through LDS; single-pass variance
    {
      float s1[4], s2[4];
      block_reduce64<false>(v, lane, s1);
      block_reduce64<true>(v, lane, s2);
      const int slot = (4 * dyb + dj) * 16 + 4 * dxb;
      *reinterpret_cast<float4*>(part + wave * 256 + slot) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(part + (NW + wave) * 256 + slot) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    __syncthreads();   // partial sums complete; nobody reads the planar image any more
    for (int t = tid; t < 256; t += NT) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        t1 += part[w * 256 + t];
        t2 += part[(NW + w) * 256 + t];
      }
      const float mean = t1 * (1.0f / C);
      const float rstd = rsqrtf(fmaxf(t2 * (1.0f / C) - mean * mean, 0.0f) + LN_EPS);
      st[t] = rstd;
      st[256 + t] = -mean * rstd;
    }
    __syncthreads();
    {
      T* mo = reinterpret_cast<T*>(map) + dch;
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb) {
          const int slot = (4 * yb + dj) * 16 + 4 * xb;
          const float4 r4 = *reinterpret_cast<const float4*>(st + slot);
          const float4 m4 = *reinterpret_cast<const float4*>(st + 256 + slot);
          const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
          if (yb < 3 || dj < 3) {   // row 15 is padding
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (xb == 3 && i == 3) continue;   // column 15 is padding
              const float y = fmaf(fmaf(v[yb * 16 + xb * 4 + i], rr[i], mm[i]), lng, lnb2);
              mo[((4 * yb + dj) * HW + 4 * xb + i) * (PITCH / 2)] = (T)y;
            }
          }
        }
    }
    __syncthreads();   // the [pixel][channel] image is complete
    {
      constexpr int PPR = C * 2 / 16;   // 16-byte pieces per pixel row
      unsigned char* dst = reinterpret_cast<unsigned char*>(xn) + (size_t)a * P * C * 2;
      for (int i = tid; i < (live ? P * PPR : 0); i += NT) {
        const int p = i / PPR, c = i - p * PPR;
        *reinterpret_cast<uint4*>(dst + (size_t)p * C * 2 + 16 * c) = *reinterpret_cast<const uint4*>(map + p * PITCH + 16 * c);
      }
    }
    __syncthreads();   // the image has left before the next alert's zero fill
  } while (LOOP && (a += gridDim.x) < B);
}

template <typename T, int C>
int launch_dw15_t(const float* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb, void* xn, int B,
                  hipStream_t st) {
  auto kern = dw15_ln_kernel<T, C>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(L15<C>::BYTES * L15<C>::TEAMS)));
    attr_set.done();
  }
  constexpr int TEAMS = L15<C>::TEAMS;
  const int grid = TEAMS == 2 ? (B + 1) / 2 : (B > 512 ? 512 : B);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(L15<C>::NT * TEAMS), L15<C>::BYTES * TEAMS, st, x, wdw, bdw, lnw, lnb,
                     reinterpret_cast<T*>(xn), B);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool dw15_supported(int prec, int C) {
  static const bool off = [] {
    const char* e = getenv("BTSBOT_AMD_NO_DW15");   // 1: the 15x15 depthwise + LayerNorm stays on the per-tap kernel (A/B)
    return e != nullptr && e[0] == '1';
  }();
  return !off && (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (C == 64 || C == 80);
}

// wdw: the depthwise filter tap-major [49][C] fp32; x: [B][225][C] fp32; xn: [B][225][C] in the operand type
int launch_dw15_ln(int prec, const float* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb,
                   void* xn, int B, int C, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (!(prec == BTSBOT_BF16 || prec == BTSBOT_F16) || !(C == 64 || C == 80)) {
    btsbot_set_error("dw15_ln: unsupported (prec %d, C %d)", prec, C);
    return BTSBOT_ERR_INVALID_ARG;
  }
#define DW15(TT, CC) return launch_dw15_t<TT, CC>(x, wdw, bdw, lnw, lnb, xn, B, st)
  if (prec == BTSBOT_BF16) {
    if (C == 64) DW15(bf16_t, 64);
    DW15(bf16_t, 80);
  }
  if (C == 64) DW15(f16_t, 64);
  DW15(f16_t, 80);
#undef DW15
}
