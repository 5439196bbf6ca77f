// Stage-0 megakernel (gfx950), one launch for
//
//   stem (conv 4x4 s4 + LN)  ->  2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]
//                            ->  downsample (LN + conv 2x2 s2)  ->  [49][128] f32
//
// (timm ConvNeXt stem / stages[0] / stages[1].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132) -- cut so that TWO workgroups share a CU:
// a single 512-thread workgroup per CU (the first layout of this kernel, rounds 1-2, no longer in the tree) runs its
// phases in lockstep (depthwise = VALU + LDS, MLP = MFMA + VALU, parameter fetches = latency) and nothing overlaps; two independent
// 256-thread workgroups drift apart and fill each other's stalls, at the same 2 waves per SIMD
// that the f32 VALU needs for its full rate (tools/unit/valu_rate.hip).
//
// What had to shrink to fit 80 KB of LDS and 4 waves per workgroup:
//   * ONE 16-bit map image ([256 px][64 ch], 144-byte rows).  The depthwise phase keeps its LN
//     outputs in registers until every wave has finished reading the image, then overwrites it;
//     the MLP reads that as its MFMA B operand and writes the new x back at its end.
//   * the pointwise filters stream through a 4-slot ring of 8 KB chunks (32 hidden units: W1 rows
//     + W2 columns) by LDS-DMA straight from the plain row-major 16-bit filters -- the per-lane
//     source address does the re-arrangement (XOR swizzles for conflict-free ds_read_b128, and the
//     W1 rows of a chunk in bit-2/bit-3-swapped order so that the fc1 accumulator of a lane IS the
//     fc2 B operand in plain k order);
//   * layer scale is folded into the fc2 filter (gamma * W2, packed once), so fc2 accumulates
//     straight into the fp32 residual registers: no second accumulator tile;
//   * a wave owns 64 pixels (2 column blocks of the 32x32 MFMA): residual = 64 registers.
#include <type_traits>

#include "common.h"
#include "stage0.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct SBM;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <> struct SBM<bf16_t> {
  using frag = bf16x8;
  using frag4 = s16x4;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // 16 independent 4x4x4 products (one per channel): lane = (block, row of A / column of B and D)
  static __device__ __forceinline__ f32x4 run4(frag4 a, frag4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <> struct SBM<f16_t> {
  using frag = f16x8;
  using frag4 = f16x4;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 run4(frag4 a, frag4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 64, HW = 15, P = 225, CT = 2, HID = 256;
constexpr int PITCH = 2 * C + 16;                 // [pixel][channel] image: 144 bytes per pixel row
static_assert(256 * PITCH <= C * 800, "the [pixel][channel] image overlays the planar one");
// planar image for the depthwise phase: [channel][x quad 0..3][row -3..18][4 x] in the operand type.
// 8-byte entries (4 consecutive x of one row); 22 rows of a quad are contiguous, so the four lanes j of a
// 4x4x4 MFMA block read 4 consecutive entries; the channel stride (100 entries) is 4 mod 32, which spreads
// the 8 blocks x 4 rows of a 32-lane half over all 32 eight-byte bank slots: ds_read_b64 without conflicts.
constexpr int PL_ROWS = 22, PL_XQ = PL_ROWS * 8, PL_CH = 800;
constexpr int PLB = C * PL_CH;                    // 51200 (the [pixel][channel] image overlays it)
constexpr int CHUNKB = 8192, NCH = HID / 32, NSLOT = 3;
constexpr int RINGB = NSLOT * CHUNKB;             // 24576
// per-block parameter image in HBM (launch_pack_s0par):
//   Toeplitz taps, operand type: [r = ky * 3 + (rb + 1)][channel][i][k] = W[channel][ky][4 rb + k - i + 3] (0 outside
//   the 7 taps): the A operand of the 4x4x4 MFMA that maps input columns 4 (xb + rb) + k to outputs 4 xb + i;
//   then fp32: dw bias [64] | LN weight [64] | LN bias [64] | fc1 bias [256] | gamma * fc2 bias [64]
constexpr int TW_R = 21, TW_BYTES = TW_R * C * 4 * 4 * 2;   // 43008
constexpr int PF_DWB = 0, PF_LNW = C, PF_LNB = 2 * C, PF_B1 = 3 * C, PF_B2 = PF_B1 + HID, PF_FLOATS = 512;
static_assert(PF_B2 + C == PF_FLOATS, "parameter image layout");
constexpr int PARB = TW_BYTES + PF_FLOATS * 4;    // 45056
// split mode (BTSBOT_F16X2): the taps' f16 remainders follow their heads, the fp32 part comes last
constexpr int PARB_X2 = 2 * TW_BYTES + PF_FLOATS * 4;
// LDS layout.  Split mode (X2): a second planar image behind the first (the f16 remainders of the map the depthwise phase
// reads) and chunks of twice the size (the filters' remainders behind their heads) -- 154,880 bytes, one workgroup per CU.
template <bool X2> struct S0L {
  static constexpr int PLANES = X2 ? 2 : 1;
  static constexpr int CHB = X2 ? 2 * CHUNKB : CHUNKB;     // bytes of a ring slot
  static constexpr int OFF_RING = PLANES * PLB;
  static constexpr int OFF_B1 = OFF_RING + NSLOT * CHB;    // 256 + 64 floats: this block's fc1 bias, gamma*b2
  static constexpr int OFF_ST = OFF_B1 + (HID + C) * 4;    // LayerNorm (rstd, -mean * rstd) per padded pixel slot: 2 x 256 floats
  static constexpr int LDS_BYTES = OFF_ST + 2 * 256 * 4;   // 79104 (two workgroups per CU) / 154880
};
static_assert(S0L<false>::LDS_BYTES <= 81920, "two workgroups per CU");
static_assert(S0L<true>::LDS_BYTES <= 160 * 1024, "one workgroup per CU");
constexpr float LN_EPS = 1e-6f;

#define SB_STAMP(i)                                                                \
  do {                                                                             \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {             \
      a.stamps[i] = clock64();                                                     \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* keep the counted waits of the DMA ring exact */ \
    }                                                                              \
    if (a.wgt != nullptr && threadIdx.x == 0 && ((i) == 0 || (i) == 13))           \
      a.wgt[2 * blockIdx.x + ((i) == 13)] = wall_clock64();                        \
  } while (0)

struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int swz4(int row) {   // F[(row >> 2) & 3], F = {0,3,2,1}
  return (4 - ((row >> 2) & 3)) & 3;
}
__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// LayerNorm over the 64 channels of this lane's pixel (x[2][16] here + the partner lane ^ 32)
__device__ __forceinline__ void ln_regs(const f32x16 (&x)[CT], const float* __restrict__ w,
                                        const float* __restrict__ b, int h, f32x16 (&y)[CT]) {
  float s = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[ct][r];
  s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = x[ct][r] - mean;
      q += d * d;
    }
  q += __shfl_xor(q, 32, 64);
  const float rstd = rsqrtf(q * (1.0f / C) + LN_EPS);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int c = ct * 32 + 8 * qd + 4 * h;
      const float4 wv = *reinterpret_cast<const float4*>(w + c);
      const float4 bv = *reinterpret_cast<const float4*>(b + c);
      y[ct][4 * qd + 0] = (x[ct][4 * qd + 0] - mean) * rstd * wv.x + bv.x;
      y[ct][4 * qd + 1] = (x[ct][4 * qd + 1] - mean) * rstd * wv.y + bv.y;
      y[ct][4 * qd + 2] = (x[ct][4 * qd + 2] - mean) * rstd * wv.z + bv.z;
      y[ct][4 * qd + 3] = (x[ct][4 * qd + 3] - mean) * rstd * wv.w + bv.w;
    }
}

// LOPLANE > 0: also the values' f16 remainders, LOPLANE bytes behind (split mode)
template <typename T, int LOPLANE = 0>
__device__ __forceinline__ void regs_to_map(const f32x16 (&x)[CT], unsigned char* map, int p, int h) {
  typedef T T4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      T4 v, w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (T)x[ct][4 * qd + e];
        if (LOPLANE > 0) w[e] = (T)(x[ct][4 * qd + e] - (float)v[e]);
      }
      *reinterpret_cast<T4*>(map + p * PITCH + (ct * 32 + 8 * qd + 4 * h) * 2) = v;
      if (LOPLANE > 0) *reinterpret_cast<T4*>(map + LOPLANE + p * PITCH + (ct * 32 + 8 * qd + 4 * h) * 2) = w;
    }
}

// this lane's pixel (32 of its 64 channels: rows (r & 3) + 8 (r >> 2) + 4 h of column block ct) into the planar
// image; the zero padding around the 15x15 map is never touched
// (LOPLANE > 0: also the values' f16 remainders, LOPLANE bytes behind -- split mode)
// (SAT: the value is clamped to the f16 range first -- the keeping forms run their depthwise phase on f16 operands in
//  every mode, and a bf16 handle's residual stream may exceed 65504: saturate instead of inf -> NaN through the LayerNorm)
template <typename T, int LOPLANE = 0, bool SAT = false>
__device__ __forceinline__ void regs_to_planar(const f32x16 (&x)[CT], unsigned char* pl, int p, int h) {
  const int y = p / HW, xx = p - y * HW;
  unsigned char* dst = pl + (xx >> 2) * PL_XQ + (y + 3) * 8 + (xx & 3) * 2 + h * 4 * PL_CH;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const T hi = (T)(SAT ? __builtin_amdgcn_fmed3f(x[ct][r], -65504.0f, 65504.0f) : x[ct][r]);
      *reinterpret_cast<T*>(dst + (ct * 32 + 8 * (r >> 2) + (r & 3)) * PL_CH) = hi;
      if (LOPLANE > 0)
        *reinterpret_cast<T*>(dst + LOPLANE + (ct * 32 + 8 * (r >> 2) + (r & 3)) * PL_CH) = (T)(x[ct][r] - (float)hi);
    }
}

__device__ __forceinline__ void regs_to_tap(const f32x16 (&x)[CT], float* tap, int h) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
      *reinterpret_cast<float4*>(tap + ct * 32 + 8 * qd + 4 * h) =
          make_float4(x[ct][4 * qd], x[ct][4 * qd + 1], x[ct][4 * qd + 2], x[ct][4 * qd + 3]);
}

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Transposing sum over the 16 blocks (lane bits 2..5) of 64 values per lane (SQ: of their squares): level by
// level the lanes of a pair split the values between them, so 32 + 16 + 8 + 4 adds instead of 4 x 64.
// out[0..3] = the 16-lane totals of values 16 (lane >> 4) + 4 ((lane >> 2) & 3) + 0..3.
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

// X2 (BTSBOT_F16X2, T = f16): the LayerNorm outputs and the hidden activations go to the matrix pipe as f16 head +
// f16 remainder (two products per k-step against the f16 filters), the depthwise taps likewise (two 4x4x4 products per
// tap fragment against the f16 map), the downsample with both operands split (three products).  What stays plain f16
// -- the stem's operands, the map the depthwise phase reads, the pointwise filters -- is budgeted in DESIGN.md
// (tools/error_budget2.py): 1.4e-5 rms of the 1e-4 score tolerance for the whole network.
// WPS = waves per SIMD the register allocation leaves room for (2: two workgroups per CU, 256 registers; 1: one
// workgroup per CU with 512 registers -- the split mode's doubled fragments spill at 256)
// KEEP: the training forward (Stage0Args::keep_*): the same kernel plus the copies the backward reads
template <typename T, bool X2, int WPS = 2, bool KEEP = false>
__global__ __launch_bounds__(256, WPS) void stage0b_kernel(Stage0Args a) {
  using frag = typename SBM<T>::frag;
  // the depthwise phase's operand type: the training forward reads map and taps as f16 in the bf16 mode too -- the
  // map is the fp32 residual stream, and its rounding to bf16 in front of 49 products moved a 50-step bf16 training
  // run ten times further from the fp32 recipe than the per-op forward's fp32 convolution (test_16bit_training_follows_
  // the_fp32_recipe: worst loss difference 4.0e-2 against 4.0e-3); f16 keeps 11 bits of it
  using DT = typename std::conditional<KEEP, f16_t, T>::type;
  using frag4 = typename SBM<DT>::frag4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* map = smem;     // [pixel][channel] image (MLP operand, downsample input)
  unsigned char* pl = smem;      // planar image (depthwise operand): same bytes, never live together
  using L = S0L<X2>;
  constexpr int CHB = L::CHB;
  constexpr int PLO = X2 ? PLB : 0;                  // split mode: the planar image of the remainders
  unsigned char* ring = smem + L::OFF_RING;
  float* b1s = reinterpret_cast<float*>(smem + L::OFF_B1);
  float* b2s = b1s + HID;
  float* part = reinterpret_cast<float*>(ring + 2 * CHB);      // LayerNorm partial sums [2][4 waves][256 slots]:
                                                               // ring slot 2 is idle until the MLP's first chunk
  float* st = reinterpret_cast<float*>(smem + L::OFF_ST);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int alert = blockIdx.x;
  int pix[2];
  bool live[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    pix[t] = wave * 64 + t * 32 + lr;     // this lane's pixel slot of column block t
    live[t] = pix[t] < P;
  }
  // depthwise roles: lane = (block b = channel 16 wave + b, row offset j)
  const int dj = lane & 3, dch = wave * 16 + (lane >> 2);
  const int dyb = lane >> 4, dxb = (lane >> 2) & 3;   // where this lane's LayerNorm sums end up

  SB_STAMP(0);
  // a block's Toeplitz taps (16 channels x 4 rows x 21 fragments per wave, lane-linear 8-byte loads) and its
  // per-lane scalars: requested one phase ahead of the depthwise convolution that needs them
  frag4 tw[TW_R], twl[X2 ? TW_R : 1];
  float dwbias, lng, lnb2, b1v, b2v;
  auto load_block_params = [&](int j) {
    const Stage0Blk& bk = a.blk[j];
    const uint2* src = reinterpret_cast<const uint2*>(bk.par) + wave * 64 + lane;
#pragma unroll
    for (int r = 0; r < TW_R; ++r) tw[r] = __builtin_bit_cast(frag4, src[r * 256]);
    if (X2) {
#pragma unroll
      for (int r = 0; r < TW_R; ++r) twl[r] = __builtin_bit_cast(frag4, src[TW_BYTES / 8 + r * 256]);
    }
    const float* pf = reinterpret_cast<const float*>(bk.par + (X2 ? 2 : 1) * TW_BYTES);
    dwbias = pf[PF_DWB + dch];
    lng = pf[PF_LNW + dch];
    lnb2 = pf[PF_LNB + dch];
    b1v = pf[PF_B1 + tid];
    b2v = pf[PF_B2 + (tid & 63)];
  };
  load_block_params(0);   // lands under the stem
  auto zero_planar = [&]() {
    for (int i = tid; i < L::PLANES * PLB / 16; i += 256) reinterpret_cast<uint4*>(pl)[i] = make_uint4(0u, 0u, 0u, 0u);
  };
  zero_planar();
  __syncthreads();

  // ============================ stem: conv 4x4 s4 + LN =====================================
  f32x16 x[2][CT];
  {
    const float* src = a.img + (size_t)alert * 3 * 63 * 63;
    const T* sw = reinterpret_cast<const T*>(a.stem_w);
    frag af[3][CT], afl[X2 ? 3 : 1][CT];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        af[ci][ct] = *reinterpret_cast<const frag*>(sw + (ct * 32 + lr) * 48 + ci * 16 + h * 8);
        if (X2)
          afl[ci][ct] = *reinterpret_cast<const frag*>(reinterpret_cast<const T*>(a.stem_w_lo) + (ct * 32 + lr) * 48 + ci * 16 + h * 8);
      }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pc = live[t] ? pix[t] : 0;
      const int py = pc / HW, px = pc - py * HW;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 bv = *reinterpret_cast<const float4*>(a.stem_b + ct * 32 + 8 * qd + 4 * h);
          x[t][ct][4 * qd + 0] = bv.x;
          x[t][ct][4 * qd + 1] = bv.y;
          x[t][ct][4 * qd + 2] = bv.z;
          x[t][ct][4 * qd + 3] = bv.w;
        }
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {     // k-step = input channel: k = ci*16 + ky*4 + kx
        const float* r0 = src + (ci * 63 + 4 * py + 2 * h) * 63 + 4 * px;
        const f4u v0 = *reinterpret_cast<const f4u*>(r0);
        const f4u v1 = *reinterpret_cast<const f4u*>(r0 + 63);
        frag bf, bfl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bf[e] = (T)v0.v[e];
          bf[4 + e] = (T)v1.v[e];
          if (X2) {
            bfl[e] = (T)(v0.v[e] - (float)bf[e]);
            bfl[4 + e] = (T)(v1.v[e] - (float)bf[4 + e]);
          }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (X2) {
            x[t][ct] = SBM<T>::run(afl[ci][ct], bf, x[t][ct]);
            x[t][ct] = SBM<T>::run(af[ci][ct], bfl, x[t][ct]);
          }
          x[t][ct] = SBM<T>::run(af[ci][ct], bf, x[t][ct]);
        }
      }
      if (KEEP && live[t]) regs_to_tap(x[t], a.keep_stem_pre + ((size_t)alert * P + pix[t]) * C, h);
      ln_regs(x[t], a.stem_lnw, a.stem_lnb, h, x[t]);
      if (live[t]) regs_to_planar<DT, PLO, KEEP && !std::is_same<T, f16_t>::value>(x[t], pl, pix[t], h);
      if (a.tap_stem != nullptr && live[t])
        regs_to_tap(x[t], a.tap_stem + ((size_t)alert * P + pix[t]) * C, h);
    }
  }
  SB_STAMP(1);   // stem done

  // ============================ two ConvNeXt blocks ========================================
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    SB_STAMP(2 + 5 * j);
    __syncthreads();   // planar image complete (stem / previous MLP); ring free; b1s / st idle
    b1s[tid] = b1v;
    if (tid < C) b2s[tid] = b2v;

    // ---- pointwise filters: chunk = 32 hidden units = 8 pieces of 1 KiB, 2 per wave.
    //      pieces 0..3: W1 rows (LDS row m <- hidden unit 32*ch + swap23(m)), 128-byte rows,
    //                   16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
    //      pieces 4..7: gamma*W2 columns 32*ch .. +31 of the 64 channel rows, 64-byte rows,
    //                   chunk c of row r at position c ^ F[(r >> 2) & 3]
    const unsigned char* wsrc[2];
    const unsigned char* wsrcl[2];   // split mode: the same pieces of the remainder images, 8 KB further into the slot
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave * 2 + i;
      if (pc < 4) {
        const int m = pc * 8 + (lane >> 3);
        const int hid = (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);   // swap bits 2 and 3
        const size_t o = (size_t)hid * (C * 2) + (((lane & 7) ^ ((m >> 1) & 7)) << 4);
        wsrc[i] = bk.w1 + o;
        wsrcl[i] = X2 ? bk.w1_lo + o : nullptr;
      } else {
        const int r = (pc - 4) * 16 + (lane >> 2);
        const size_t o = (size_t)r * (HID * 2) + (((lane & 3) ^ swz4(r)) << 4);
        wsrc[i] = bk.w2g + o;
        wsrcl[i] = X2 ? bk.w2g_lo + o : nullptr;
      }
    }
    // chunk ch adds 32 W1 rows (4096 B) resp. 32 W2 columns (64 B)
    const int wstep0 = wave < 2 ? 32 * C * 2 : 64;
    auto issue = [&](int ch) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)ch * wstep0),
                                         (lptr_t)(ring + (ch % NSLOT) * CHB + (wave * 2 + i) * 1024),
                                         16, 0, 0);
      if (X2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrcl[i] + (size_t)ch * wstep0),
                                           (lptr_t)(ring + (ch % NSLOT) * CHB + CHUNKB + (wave * 2 + i) * 1024),
                                           16, 0, 0);
      }
    };
    SB_STAMP(3 + 5 * j);

    // ---- depthwise 7x7 on the matrix pipe: per channel (= MFMA block) and output tile (4 rows yb, 4 columns xb)
    //      D[i][j] = out[4 yb + j][4 xb + i] = sum over ky, rb, k of  W[ky][4 rb + k - i + 3] * in[4 yb + j + ky - 3][4 (xb + rb) + k]
    //      A = the Toeplitz taps (registers), B = 4 consecutive x of 4 consecutive rows (one ds_read_b64 per lane);
    //      a row step s = 4 yb + ky serves every (yb, ky) pair with that sum: 76 reads, 280 MFMAs per wave.
    float v[64];
    {
      f32x4 acc[4][4];
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb) acc[yb][xb] = f32x4{dwbias, dwbias, dwbias, dwbias};
      const unsigned char* lb = pl + dch * PL_CH + dj * 8;
      // (the row step's four reads are requested a whole step -- 10 to 48 products -- ahead of their use: read and used in
      //  the same step, every step began with an exposed LDS round trip, 19 per block, about as long as the products)
      frag4 bq[2][4], bql[X2 ? 2 : 1][X2 ? 4 : 1];
      auto read_step = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bq[buf][q] = __builtin_bit_cast(frag4, *reinterpret_cast<const uint2*>(lb + q * PL_XQ + s * 8));
          if (X2) bql[buf][q] = __builtin_bit_cast(frag4, *reinterpret_cast<const uint2*>(lb + PLB + q * PL_XQ + s * 8));
        }
      };
      read_step(0, 0);
#ifdef S0_EXP_NODW
#pragma unroll 1
      for (int s = 0; s < 1; ++s) {
#else
#pragma unroll
      for (int s = 0; s < 19; ++s) {
#endif
        // (touching this step's fragments makes hipcc wait for them HERE, while they are the only LDS reads in flight:
        //  placed by itself the wait lands behind the next step's reads -- lgkmcnt(0), their whole latency exposed)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          asm volatile("" : "+v"(bq[s & 1][q]));
          if (X2) asm volatile("" : "+v"(bql[X2 ? s & 1 : 0][X2 ? q : 0]));
        }
        if (s + 1 < 19) read_step(s + 1, (s + 1) & 1);
        if (s == 0) {
          // the taps are in registers by now: only now queue the filter chunks (a wait for an ordinary load
          // placed behind an LDS-DMA would wait for the DMA too)
          __builtin_amdgcn_sched_barrier(0);
          issue(0);
          issue(1);
        }
        __builtin_amdgcn_sched_barrier(0);
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
              if (X2) {   // remainders first (small terms into the accumulator before the large one)
                acc[yb][xb] = SBM<DT>::run4(twl[ky * 3 + rbi], bq[s & 1][q], acc[yb][xb]);
                acc[yb][xb] = SBM<DT>::run4(tw[ky * 3 + rbi], bql[X2 ? s & 1 : 0][X2 ? q : 0], acc[yb][xb]);
              }
              acc[yb][xb] = SBM<DT>::run4(tw[ky * 3 + rbi], bq[s & 1][q], acc[yb][xb]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[yb * 16 + xb * 4 + i] = acc[yb][xb][i];
    }
    if (KEEP && a.keep_d[j] != nullptr) {   // the depthwise output before the LayerNorm: lane = (channel, row j of the quad), 60 live
                                             // values (nullptr: the backward recomputes it, dwln_bwd.hip)
      float* dst = a.keep_d[j] + (size_t)alert * P * C + dch;
#pragma unroll
      for (int yb = 0; yb < 4; ++yb)
#pragma unroll
        for (int xb = 0; xb < 4; ++xb)
          if (yb < 3 || dj < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (xb == 3 && i == 3) continue;
              dst[((4 * yb + dj) * HW + 4 * xb + i) * C] = v[yb * 16 + xb * 4 + i];
            }
          }
    }
    SB_STAMP(4 + 5 * j);   // depthwise done
    // ---- LayerNorm over the 64 channels of a pixel: 16 blocks of this wave (transposing lane reduction), then
    //      the 4 waves through LDS; single-pass variance
    {
      float s1[4], s2[4];
#ifdef S0_EXP_NOLNRED
      for (int i = 0; i < 4; ++i) { s1[i] = v[i]; s2[i] = v[i + 4] * v[i + 4] + 1.f; }
#else
      block_reduce64<false>(v, lane, s1);
      block_reduce64<true>(v, lane, s2);
#endif
      const int slot = (4 * dyb + dj) * 16 + 4 * dxb;
      *reinterpret_cast<float4*>(part + wave * 256 + slot) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(part + 1024 + wave * 256 + slot) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    if (j == 0) SB_STAMP(14);
    __syncthreads();   // partial sums complete; nobody reads the planar image any more
    if (j == 0) SB_STAMP(15);
    {
      const float t1 = part[tid] + part[256 + tid] + part[512 + tid] + part[768 + tid];
      const float t2 = part[1024 + tid] + part[1280 + tid] + part[1536 + tid] + part[1792 + tid];
      const float mean = t1 * (1.0f / C);
      const float rstd = rsqrtf(fmaxf(t2 * (1.0f / C) - mean * mean, 0.0f) + LN_EPS);
      st[tid] = rstd;
      st[256 + tid] = -mean * rstd;
    }
    __syncthreads();
    // the LayerNorm output into the [pixel][channel] image: its f16 values, or (split mode, second pass through the same
    // bytes) their f16 remainders
    auto write_ln = [&](bool lo_pass) {
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
              const T yh = (T)y;
              mo[((4 * yb + dj) * HW + 4 * xb + i) * (PITCH / 2)] = lo_pass ? (T)(y - (float)yh) : yh;
            }
          }
        }
    };
    write_ln(false);
    SB_STAMP(5 + 5 * j);
    __syncthreads();   // LN image complete
    if (KEEP) {   // the LayerNorm output rows, 16-byte pieces (8 per pixel row), before the image is cleared in the MLP
      unsigned char* dst = reinterpret_cast<unsigned char*>(a.keep_xn[j]) + (size_t)alert * P * C * 2;
      for (int i = tid; i < P * 8; i += 256) {
        const int p = i >> 3, c16 = i & 7;
        *reinterpret_cast<uint4*>(dst + (size_t)p * C * 2 + 16 * c16) = *reinterpret_cast<const uint4*>(map + p * PITCH + 16 * c16);
      }
    }

    // ---- fc1 -> GELU -> fc2 over 8 chunks; fc2 accumulates into x (gamma is in the filter)
    {
      frag xf[2][4], xfl[X2 ? 2 : 1][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          xf[t][ks] = *reinterpret_cast<const frag*>(map + pix[t] * PITCH + ks * 32 + h * 16);
      if (X2) {   // the remainders through the same image bytes
        __syncthreads();
        write_ln(true);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            xfl[t][ks] = *reinterpret_cast<const frag*>(map + pix[t] * PITCH + ks * 32 + h * 16);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 bv = *reinterpret_cast<const float4*>(b2s + ct * 32 + 8 * qd + 4 * h);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            x[t][ct][4 * qd + 0] += bv.x;
            x[t][ct][4 * qd + 1] += bv.y;
            x[t][ct][4 * qd + 2] += bv.z;
            x[t][ct][4 * qd + 3] += bv.w;
          }
        }
#ifdef S0_EXP_NOLOOP
#pragma unroll 1
      for (int ch = 0; ch < 0; ++ch) {
#else
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
#endif
        // this wave's pieces of chunk ch have landed once only the younger chunk is outstanding
        if (ch + 1 < NCH) wait_vm<X2 ? 4 : 2>();
        else wait_vm<0>();
        // raw barrier: __syncthreads() would also wait vmcnt(0) while an LDS-DMA is in flight and so drain the two
        // chunks this ring keeps ahead (the chunk time then IS the DMA latency, ~2k cycles for ~1k of work)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // chunk ch has landed for everyone; chunk ch-1 is read out
        // the LN image is dead once every wave holds its xf: clear it for the next block's planar image
        // (whose padding must read as zero) while the matrix pipe works
        if (ch == 1 && j == 0) zero_planar();
        const unsigned char* w1s = ring + (ch % NSLOT) * CHB;
        const unsigned char* w2s = w1s + 4096;
        frag a1[4], a2[CT][2], a1l[X2 ? 4 : 1], a2l[X2 ? CT : 1][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          a1[ks] = *reinterpret_cast<const frag*>(w1s + lr * 128 + (((ks * 2 + h) ^ ((lr >> 1) & 7)) << 4));
          if (X2) a1l[ks] = *reinterpret_cast<const frag*>(w1s + CHUNKB + lr * 128 + (((ks * 2 + h) ^ ((lr >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int r = ct * 32 + lr;
            a2[ct][s2] = *reinterpret_cast<const frag*>(w2s + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
            if (X2) a2l[ct][s2] = *reinterpret_cast<const frag*>(w2s + CHUNKB + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
          }
        // (inline asm: behind a compiler-visible ds_read of anything but the ring hipcc waits vmcnt(0) -- it cannot
        //  tell that the bias words are not an LDS-DMA destination -- and the ring's two chunks in flight are gone)
        f32x4 bq[4];
        {
          const unsigned baddr = (unsigned)(size_t)(lptr_t)(b1s + ch * 32 + 8 * h);
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\t"
                       "ds_read_b128 %3, %4 offset:80\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]) : "v"(baddr) : "memory");
        }
        // every fragment read is queued before the first product (see stage1b.hip)
        __builtin_amdgcn_sched_barrier(0);
        // both column blocks' fc1 first: the second one's MFMAs run under the first one's GELU
        f32x16 hacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // accumulator row (r&3) + 8(r>>2) + 4h holds hidden unit 32ch + (r&3) + 4((r>>2)&1) + 8h + 16(r>>3)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
#pragma unroll
            for (int e = 0; e < 4; ++e) hacc[t][4 * qd + e] = bq[qd][e];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            if (X2) {
              hacc[t] = SBM<T>::run(a1l[ks], xf[t][ks], hacc[t]);
              hacc[t] = SBM<T>::run(a1[ks], xfl[t][ks], hacc[t]);
            }
            hacc[t] = SBM<T>::run(a1[ks], xf[t][ks], hacc[t]);
          }
        }
        // (issued here, not at the barrier: an LDS-DMA holds the issuing wave ~90 cycles per
        //  instruction, which now passes while the fc1 MFMAs drain)
        if (ch + 2 < NCH) issue(ch + 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // GELU in two halves, each followed by the fc2 MFMAs that consume it
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            frag hf, hfl;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#ifdef S0_EXP_NOGELU
              const float gv = hacc[t][8 * s2 + r];
#else
              const float gv = gelu_for<T>(hacc[t][8 * s2 + r]);
#endif
              hf[r] = (T)gv;
              if (X2) hfl[r] = (T)(gv - (float)hf[r]);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              if (X2) {
                x[t][ct] = SBM<T>::run(a2l[ct][s2], hf, x[t][ct]);
                x[t][ct] = SBM<T>::run(a2[ct][s2], hfl, x[t][ct]);
              }
              x[t][ct] = SBM<T>::run(a2[ct][s2], hf, x[t][ct]);
            }
          }
        }
      }
      // next block's depthwise operand (the region was cleared during this MLP)
      if (j == 0) {
        load_block_params(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (live[t]) regs_to_planar<DT, PLO, KEEP && !std::is_same<T, f16_t>::value>(x[t], pl, pix[t], h);
          if (KEEP && live[t]) regs_to_tap(x[t], a.keep_xin1 + ((size_t)alert * P + pix[t]) * C, h);
        }
      }
    }
    SB_STAMP(6 + 5 * j);   // MLP done
  }
  if (a.tap_stage != nullptr) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (live[t]) regs_to_tap(x[t], a.tap_stage + ((size_t)alert * P + pix[t]) * C, h);
  }

  // ============================ downsample: LN + conv 2x2 s2 (64 -> 128) ====================
  {
    // wave -> 32 output channels (cot = wave) x the 49 output pixels (2 column blocks); K = 4 x 64
    // (filter fragments requested first: their L2 latency passes under the LayerNorm below)
    const T* dw = reinterpret_cast<const T*>(a.ds_w) + (size_t)(wave * 32 + lr) * 256 + h * 8;
    frag af[16], afl[X2 ? 16 : 1];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const frag*>(dw + ks * 16);
    // (every wave loaded its xf from the last block's LN image before that MLP's second barrier)
    constexpr int MAPB = 256 * PITCH;   // split mode: the remainder image follows (into the filter ring's bytes)
    static_assert(2 * MAPB <= L::OFF_B1, "two [pixel][channel] images in front of the bias words");
    if (X2) __syncthreads();            // ... which every wave must have read out
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 xn[CT];
      ln_regs(x[t], a.ds_lnw, a.ds_lnb, h, xn);
      regs_to_map<T, X2 ? MAPB : 0>(xn, map, pix[t], h);
    }
    if (X2) {
      const T* dwl = reinterpret_cast<const T*>(a.ds_w_lo) + (size_t)(wave * 32 + lr) * 256 + h * 8;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) afl[ks] = *reinterpret_cast<const frag*>(dwl + ks * 16);
    }
    __syncthreads();
    if (KEEP) {   // the downsample's patch rows [output pixel][q = 2 ky + kx][64] = LayerNorm'd pixels (2 oy + ky, 2 ox + kx)
      unsigned char* dst = reinterpret_cast<unsigned char*>(a.keep_patches) + (size_t)alert * 49 * 4 * C * 2;
      for (int i = tid; i < 49 * 4 * 8; i += 256) {
        const int c16 = i & 7, q = (i >> 3) & 3, o = i >> 5;
        const int pin = (2 * (o / 7) + (q >> 1)) * HW + 2 * (o % 7) + (q & 1);
        *reinterpret_cast<uint4*>(dst + (size_t)(o * 4 + q) * C * 2 + 16 * c16) =
            *reinterpret_cast<const uint4*>(map + pin * PITCH + 16 * c16);
      }
    }
    SB_STAMP(12);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int o = t * 32 + lr;
      const bool olive = o < 49;
      const int oc = olive ? o : 0;
      const int oy = oc / 7, ox = oc - oy * 7;
      f32x16 acc;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 bv = *reinterpret_cast<const float4*>(a.ds_b + wave * 32 + 8 * qd + 4 * h);
        acc[4 * qd + 0] = bv.x;
        acc[4 * qd + 1] = bv.y;
        acc[4 * qd + 2] = bv.z;
        acc[4 * qd + 3] = bv.w;
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int q = ks >> 2;
        const int pin = (2 * oy + (q >> 1)) * HW + 2 * ox + (q & 1);
        const frag bf = *reinterpret_cast<const frag*>(map + pin * PITCH + (ks & 3) * 32 + h * 16);
        if (X2) {
          const frag bfl = *reinterpret_cast<const frag*>(map + MAPB + pin * PITCH + (ks & 3) * 32 + h * 16);
          acc = SBM<T>::run(afl[ks], bf, acc);
          acc = SBM<T>::run(af[ks], bfl, acc);
        }
        acc = SBM<T>::run(af[ks], bf, acc);
      }
      if (olive) {
        float* dst = a.out + ((size_t)alert * 49 + o) * 128 + wave * 32 + 4 * h;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(dst + 8 * qd) =
              make_float4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
      }
    }
    SB_STAMP(13);
  }
}

// one block's parameter image (layout at TW_R / PF_* above); taps tap-major [49][64] fp32
template <typename T, bool X2>
__global__ void pack_s0par_kernel(const float* __restrict__ taps, const float* __restrict__ dw_b,
                                  const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                  const float* __restrict__ b1, const float* __restrict__ b2,
                                  const float* __restrict__ gamma, unsigned char* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NTW = TW_R * C * 16;
  if (idx < NTW) {
    const int k = idx & 3, i = (idx >> 2) & 3, c = (idx >> 4) & (C - 1), r = idx >> 10;
    const int ky = r / 3, rb = r % 3 - 1, kx = 4 * rb + k - i + 3;
    const float w = (kx >= 0 && kx < 7) ? taps[(ky * 7 + kx) * C + c] : 0.f;
    const T wh = (T)w;
    reinterpret_cast<T*>(out)[idx] = wh;
    if (X2) reinterpret_cast<T*>(out)[NTW + idx] = (T)(w - (float)wh);
    return;
  }
  const int f = idx - NTW;
  if (f >= PF_FLOATS) return;
  float v;
  if (f < PF_LNW) v = dw_b[f - PF_DWB];
  else if (f < PF_LNB) v = ln_w[f - PF_LNW];
  else if (f < PF_B1) v = ln_b[f - PF_LNB];
  else if (f < PF_B2) v = b1[f - PF_B1];
  else v = gamma[f - PF_B2] * b2[f - PF_B2];
  reinterpret_cast<float*>(out + (X2 ? 2 : 1) * TW_BYTES)[f] = v;
}

template <typename T, bool X2 = false, int WPS = 2, bool KEEP = false> int launch_stage0b_t(const Stage0Args& a, hipStream_t st) {
  auto kern = stage0b_kernel<T, X2, WPS, KEEP>;
  static DevOnce attr_set;
  // (BTSBOT_AMD_S0_ONE_WG=1: developer probe -- the LDS request padded so that ONE workgroup fits a CU: how the kernel's
  //  time scales from one to two waves per SIMD says what two more would buy)
  static const int pad = [] {
    const char* e = getenv("BTSBOT_AMD_S0_ONE_WG");
    return e != nullptr && e[0] == '1' && !X2 ? 90 * 1024 - S0L<X2>::LDS_BYTES : 0;
  }();
  const int LDS_BYTES = S0L<X2>::LDS_BYTES + pad;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

#ifdef STAGE0B_X2_TU
// ---- this translation unit (stage0x.hip) holds only the split-operand instantiation: co-compiled instantiations of
//      one kernel template share the register allocator's context and move each other's spills
int launch_stage0b_x2(const Stage0Args& a, hipStream_t st) {
  // (one workgroup per CU -- the two planar images and the doubled filter ring take 155 KB of LDS -- with 512 registers)
  return launch_stage0b_t<f16_t, true, 1>(a, st);
}
int launch_pack_s0par_x2(const float* taps, const float* dw_b, const float* ln_w, const float* ln_b, const float* b1,
                         const float* b2, const float* gamma, void* out, hipStream_t st) {
  const dim3 grid((TW_R * C * 16 + PF_FLOATS + 255) / 256), blk(256);
  hipLaunchKernelGGL((pack_s0par_kernel<f16_t, true>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b, b1, b2, gamma,
                     reinterpret_cast<unsigned char*>(out));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
#else
int launch_stage0b_x2(const Stage0Args& a, hipStream_t st);
int launch_pack_s0par_x2(const float* taps, const float* dw_b, const float* ln_w, const float* ln_b, const float* b1,
                         const float* b2, const float* gamma, void* out, hipStream_t st);

size_t s0par_bytes() { return PARB_X2; }   // (the split mode's image is the larger one)

// taps: the block's depthwise filter tap-major [49][64] fp32; the others the master parameters
int launch_pack_s0par(int prec, const float* taps, const float* dw_b, const float* ln_w, const float* ln_b,
                      const float* b1, const float* b2, const float* gamma, void* out, hipStream_t st) {
  const dim3 grid((TW_R * C * 16 + PF_FLOATS + 255) / 256), blk(256);
  unsigned char* o = reinterpret_cast<unsigned char*>(out);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL((pack_s0par_kernel<bf16_t, false>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b, b1, b2, gamma, o);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL((pack_s0par_kernel<f16_t, false>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b, b1, b2, gamma, o);
  else if (prec == BTSBOT_F16X2)
    return launch_pack_s0par_x2(taps, dw_b, ln_w, ln_b, b1, b2, gamma, out, st);
  else {
    btsbot_set_error("pack_s0par: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

bool stage0_supported(int prec, int c0) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16 || prec == BTSBOT_F16X2) && c0 == 64;
}

// Needs Stage0Blk::par, ::w1 (plain [256][64]) and Stage0Blk::w2g (gamma-scaled [64][256]), 16-bit.
int launch_stage0b(int prec, const Stage0Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (a.keep_xn[0] != nullptr) {   // the training forward
    if (a.keep_stem_pre == nullptr || a.tap_stem == nullptr || a.keep_xin1 == nullptr || a.tap_stage == nullptr ||
        (a.keep_d[0] == nullptr) != (a.keep_d[1] == nullptr) || a.keep_xn[1] == nullptr || a.keep_patches == nullptr) {
      btsbot_set_error("stage0b: the training forward needs every kept buffer");
      return BTSBOT_ERR_INVALID_ARG;
    }
    if (prec == BTSBOT_BF16) return launch_stage0b_t<bf16_t, false, 2, true>(a, st);
    if (prec == BTSBOT_F16) return launch_stage0b_t<f16_t, false, 2, true>(a, st);
    btsbot_set_error("stage0b: the training forward runs in the bf16 / f16 modes, not %d", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16) return launch_stage0b_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage0b_t<f16_t>(a, st);
  if (prec == BTSBOT_F16X2) {
    if (a.ds_w_lo == nullptr || a.stem_w_lo == nullptr || a.blk[0].w1_lo == nullptr || a.blk[0].w2g_lo == nullptr ||
        a.blk[1].w1_lo == nullptr || a.blk[1].w2g_lo == nullptr) {
      btsbot_set_error("stage0b: the split mode needs the remainder images (ds_w_lo, stem_w_lo, w1_lo, w2g_lo)");
      return BTSBOT_ERR_INVALID_ARG;
    }
    return launch_stage0b_x2(a, st);
  }
  btsbot_set_error("stage0b: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
#endif
