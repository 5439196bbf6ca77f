// Stage-1 megakernel (gfx950): TWO alerts' 7x7x128 maps per 256-thread workgroup, two workgroups per CU:
//
//   2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]  ->  LN + conv 2x2 s2 (128 -> 256)
//
// (timm ConvNeXt stages[1].blocks / stages[2].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132).  HBM sees [49][128] f32 in and [9][256] f32
// out per alert.  At B = 1024 the 512 workgroups are all resident at once (2 per CU) and drift
// apart, so one's depthwise phase runs under the other's MLP (MFMA + GELU).
//   * residual stream fp32 in registers, 32x32 MFMA accumulator layout: wave = 32 pixel slots of
//     the 98, lane half h and 64 registers = the 128 channels;
//   * depthwise 7x7 on the matrix pipe, as in stage0b.hip: the 16-block 4x4x4 MFMA, one block per channel,
//     A = Toeplitz taps in registers (21 fragments per group of 16 channels), B = 4 consecutive x of 4
//     consecutive rows read with one ds_read_b64 from a planar image [alert][x quad][row][channel][4 x]: the
//     channel-minor order with a 136-entry row pitch makes the 8 blocks x 4 rows of a half wave hit 32 different
//     8-byte bank slots.  Rows outside the map are one shared zero row (per-lane row offsets), column 7 of
//     the second quad is kept zero.  Wave = 2 channel groups x both alerts: 224 MFMAs, 88 reads per block
//     (the VALU form this replaces: 25k cycles per block, 2 x 49 x 7 FMAs per lane plus conversions);
//   * LayerNorm: transposing lane reduction over the 16 blocks, the 4 waves meet through LDS (stage0b.hip);
//     its 16-bit output is the MLP's B operand image, in ring slots 0..1 until the MLP has loaded it;
//   * pointwise filters: chunks of 32 hidden units (8 KB of W1 rows + 8 KB of gamma*W2 columns) through two
//     3-slot LDS-DMA rings straight from the plain row-major filters (waves 0, 1 load W1, waves 2, 3 load W2); the
//     per-lane source address applies the bank swizzles and the bit-2/bit-3 row swap that makes the fc1 accumulator
//     the fc2 B operand in plain k order (stage0b.hip); fc2 accumulates into the residual registers.  The rings
//     live in the planar image's and the LN image's bytes (dead by then).  The chunk loop is software-pipelined
//     inside the wave: W1 fragments and the fc1 bias of chunk ch+1 are in registers before step ch starts, its fc1
//     MFMAs issue between the GELUs of chunk ch, chunk ch's fc2 MFMAs between the second half's GELUs.
#include "common.h"
#include "stage0.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct SCM;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <> struct SCM<bf16_t> {
  using frag = bf16x8;
  using frag4 = s16x4;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 run4(frag4 a, frag4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <> struct SCM<f16_t> {
  using frag = f16x8;
  using frag4 = f16x4;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 run4(frag4 a, frag4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 128, HW = 7, PA = 49, G = 2, NPX = G * PA, CT = 4, HID = 512, KS1 = 8;
constexpr int CN = 256, PO = 9;                   // downsample: output channels, pixels per alert
constexpr int PITCH = 2 * C + 16;                 // [pixel][channel] image: 272 bytes per pixel row
constexpr int MAPB = NPX * PITCH;                 // 26656: lives in ring slots 0..1
constexpr int CHUNKB = 16384, HALFB = CHUNKB / 2, NCH = HID / 32, NSLOT = 3;
// planar image of the depthwise phase: [alert][x quad 0..1][row 0..6, 7 = zeros][channel, pitch 136][4 x]
constexpr int PL_ROW = 136 * 8, PL_XQ = 8 * PL_ROW, PL_AL = 2 * PL_XQ, PLB = G * PL_AL;   // 1088, 8704, 17408, 34816
constexpr int OFF_PL = 2 * CHUNKB;                // ring slots 0, 1 | planar image = ring slot 2 + 18 KB
// Split mode (X2): a second planar image behind the first (the remainders of the map the depthwise phase reads), ring slots of
// twice the size (a filter piece's remainders 8 KB behind its heads): the W1 slots and W2 slot 0 (64 KB) lie in the two
// planar images, W2 slots 1, 2 in a region of their own -- 142,848 bytes, one workgroup per CU.
template <bool X2> struct S1L {
  static constexpr int PLANES = X2 ? 2 : 1;
  static constexpr int SLB = X2 ? 2 * HALFB : HALFB;            // bytes of a W1 / W2 ring slot
  static constexpr int OFF_W2X = OFF_PL + PLANES * PLB;         // X2 only: W2 slots 1, 2
  static constexpr int OFF_B1 = OFF_W2X + (X2 ? 2 * SLB : 0);   // 512 floats fc1 bias + 128 floats gamma*b2
  static constexpr int OFF_PART = OFF_B1 + (HID + C) * 4;       // LayerNorm partial sums [2][4 waves][128 slots]
  static constexpr int OFF_ST = OFF_PART + 2 * 4 * 128 * 4;     // (rstd, -mean * rstd) per padded pixel slot [2][128]
  static constexpr int LDS_BYTES = OFF_ST + 2 * 128 * 4;        // 75264: two workgroups per CU / 142848
  static_assert(4 * SLB <= PLANES * PLB, "W1 slots + W2 slot 0 inside the planar images");
};
static_assert(MAPB <= 2 * CHUNKB && S1L<false>::LDS_BYTES <= 80 * 1024 && S1L<true>::LDS_BYTES <= 160 * 1024, "LDS layout");
constexpr float LN_EPS = 1e-6f;
// per-block parameter image in HBM (launch_pack_s1par): Toeplitz taps in the operand type
//   [r = ky * 3 + (rb + 1)][channel][i][k] = W[channel][ky][4 rb + k - i + 3]   (0 outside the 7 taps)
// then fp32: dw bias [128] | LN weight [128] | LN bias [128]
constexpr int TW_R = 21, TW_BYTES = TW_R * C * 4 * 4 * 2;   // 86016
constexpr int PARB = TW_BYTES + 3 * C * 4;        // 87552
// split mode (BTSBOT_F16X2): the taps' f16 remainders follow their heads, the fp32 part comes last
constexpr int PARB_X2 = 2 * TW_BYTES + 3 * C * 4;

#define SC_STAMP(i)                                                                \
  do {                                                                             \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
    if (a.wgt != nullptr && threadIdx.x == 0 && ((i) == 0 || (i) == 13))           \
      a.wgt[2 * blockIdx.x + ((i) == 13)] = wall_clock64();                        \
  } while (0)

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
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Transposing sum over the 16 blocks (lane bits 2..5) of 32 values per lane: out[0..1] = the 16-lane totals of
// values 16 (lane >> 5) + 8 ((lane >> 4) & 1) + 4 ((lane >> 3) & 1) + 2 ((lane >> 2) & 1) + 0..1  (stage0b.hip)
__device__ __forceinline__ void block_reduce32(float (&w)[32], int lane, float (&out)[2]) {
#pragma unroll
  for (int n = 0; n < 16; ++n) w[n] = swap_add32(w[n], w[n + 16]);
#pragma unroll
  for (int n = 0; n < 8; ++n) w[n] = swap_add16(w[n], w[n + 8]);
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float own = b3 ? w[n + 4] : w[n], send = b3 ? w[n] : w[n + 4];
    w[n] = own + dpp_mov<0x128>(send);                       // row_ror:8 = lane ^ 8
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const float own = b2 ? w[n + 2] : w[n], send = b2 ? w[n] : w[n + 2];
    const float lo = dpp_mov<0x124>(send), hi = dpp_mov<0x12C>(send);   // row_ror:4 / :12 = lane - 4 / lane + 4
    out[n] = own + (b2 ? lo : hi);
  }
}

// LayerNorm over the 128 channels of this lane's pixel (x[4][16] here + the partner lane ^ 32)
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
  typedef T __attribute__((ext_vector_type(4))) T4;
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

// this lane's pixel (64 of its 128 channels: rows (r & 3) + 8 (r >> 2) + 4 h of column block ct) into the planar
// image; p = alert * 49 + y * 7 + x
// (LOPLANE > 0: also the values' f16 remainders, LOPLANE bytes behind -- split mode)
// (SAT: the value is clamped to the f16 range first -- the keeping forms run their depthwise phase on f16 operands in
//  every mode, and a bf16 handle's residual stream may exceed 65504: saturate instead of inf -> NaN through the LayerNorm)
template <typename T, int LOPLANE = 0, bool SAT = false>
__device__ __forceinline__ void regs_to_planar(const f32x16 (&x)[CT], unsigned char* pl, int p, int h) {
  const int al = p >= PA ? 1 : 0, pp = p - al * PA;
  const int y = pp / HW, xx = pp - y * HW;
  unsigned char* dst = pl + al * PL_AL + (xx >> 2) * PL_XQ + y * PL_ROW + (xx & 3) * 2 + h * 32;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const T hi = (T)(SAT ? __builtin_amdgcn_fmed3f(x[ct][r], -65504.0f, 65504.0f) : x[ct][r]);
      *reinterpret_cast<T*>(dst + (ct * 32 + 8 * (r >> 2) + (r & 3)) * 8) = hi;
      if (LOPLANE > 0) *reinterpret_cast<T*>(dst + LOPLANE + (ct * 32 + 8 * (r >> 2) + (r & 3)) * 8) = (T)(x[ct][r] - (float)hi);
    }
}

// X2 (BTSBOT_F16X2, T = f16): as in stage0b.hip -- LayerNorm outputs and hidden activations as f16 head + remainder
// (two products per k-step against the f16 filters), the depthwise taps likewise, the downsample with both operands split.
// WPS = waves per SIMD the register allocation leaves room for (2: two workgroups per CU, 256 registers; 1: one
// workgroup per CU with 512 registers -- the split mode's doubled fragments spill at 256)
// KEEP: the training forward (Stage1Args::keep_*): the same kernel plus the copies the backward reads
template <typename T, bool X2, int WPS = 2, bool KEEP = false>
__global__ __launch_bounds__(256, WPS) void stage1b_kernel(Stage1Args a) {
  using frag = typename SCM<T>::frag;
  // the depthwise phase's operand type: the training forward reads map and taps as f16 in the bf16 mode too -- the
  // map is the fp32 residual stream, and its rounding to bf16 in front of 49 products moved a 50-step bf16 training
  // run ten times further from the fp32 recipe than the per-op forward's fp32 convolution (test_16bit_training_follows_
  // the_fp32_recipe: worst loss difference 4.0e-2 against 4.0e-3); f16 keeps 11 bits of it
  using DT = typename std::conditional<KEEP, f16_t, T>::type;
  using frag4 = typename SCM<DT>::frag4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* pl = smem + OFF_PL;                 // planar image; its first 16 KB double as ring slot 2
  unsigned char* stg = smem;                         // LN image [98][PITCH] in ring slots 0..1
  using L = S1L<X2>;
  constexpr int SLB = L::SLB;
  constexpr int PLO = X2 ? PLB : 0;                  // split mode: the planar image of the remainders
  float* b1s = reinterpret_cast<float*>(smem + L::OFF_B1);
  float* b2s = b1s + HID;
  float* part = reinterpret_cast<float*>(smem + L::OFF_PART);
  float* st = reinterpret_cast<float*>(smem + L::OFF_ST);
  // filter rings, 3 slots of 8 KB each (split mode: 16 KB, the remainders behind the heads): the W1 slots and W2 slot 0
  // lie in the planar image(s) (dead once the depthwise phase is over, so the first chunk is requested under the
  // LayerNorm), W2 slots 1, 2 in the LN image's bytes (split mode: in a region of their own)
  auto w1slot = [&](int sl) { return pl + sl * SLB; };
  auto w2slot = [&](int sl) { return sl == 0 ? pl + 3 * SLB : (X2 ? smem + L::OFF_W2X : smem) + (sl - 1) * SLB; };

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const bool w1wave = wave < 2;
  const int a0 = blockIdx.x * G;
  const int nal = min(G, a.B - a0);
  const int p = wave * 32 + lr;                      // this lane's pixel slot (MFMA phases)
  const bool live = p < nal * PA;
  const bool inmap = p < NPX;
  const int pm = inmap ? p : 0;                      // row to read for slots beyond the image
  SC_STAMP(0);
  // what the live writes never touch and the depthwise products read: the zero rows and column 7
  auto zero_pads = [&]() {
#pragma unroll
    for (int pn = 0; pn < L::PLANES; ++pn) {
      unsigned char* pb = pl + pn * PLB;
      for (int i = tid; i < 4 * PL_ROW / 16; i += 256) {
        const int blk = i / (PL_ROW / 16), o = i - blk * (PL_ROW / 16);
        *reinterpret_cast<uint4*>(pb + (blk >> 1) * PL_AL + (blk & 1) * PL_XQ + 7 * PL_ROW + o * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
      for (int i = tid; i < G * HW * C; i += 256) {
        const int al = i / (HW * C), r = (i / C) % HW, c = i % C;
        *reinterpret_cast<unsigned short*>(pb + al * PL_AL + PL_XQ + r * PL_ROW + c * 8 + 6) = 0;
      }
    }
  };
  zero_pads();
  // ---- the residual stream does NOT stay in registers through the depthwise phase (64 registers next to that
  //      phase's 64 outputs and 42 tap registers spill): it is re-read at the start of each MLP, from the stage
  //      input for block 0 and from a scratch copy (written by the same lanes, L2-resident) for block 1
  const float* xsrc = a.x_in + ((size_t)a0 * PA + (live ? p : 0)) * C;
  float* xscr = a.scratch + ((size_t)a0 * PA + (live ? p : 0)) * C;
  auto load_x = [&](const float* src, f32x16 (&x)[CT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 v4 = *reinterpret_cast<const float4*>(src + ct * 32 + 8 * qd + 4 * h);
        x[ct][4 * qd + 0] = live ? v4.x : 0.f;
        x[ct][4 * qd + 1] = live ? v4.y : 0.f;
        x[ct][4 * qd + 2] = live ? v4.z : 0.f;
        x[ct][4 * qd + 3] = live ? v4.w : 0.f;
      }
  };
  {
    f32x16 x0[CT];
    load_x(xsrc, x0);
    if (inmap) regs_to_planar<DT, PLO, KEEP && !std::is_same<T, f16_t>::value>(x0, pl, p, h);
  }
  SC_STAMP(1);

  f32x16 x[CT];   // (written in full at the start of every MLP: dead through the depthwise phases)
  const int rot = (blockIdx.x * 5 + (blockIdx.x >> 4)) & (NCH - 1);   // chunk rotation, see issue()
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    // depthwise roles: lane = (block b, row offset j); wave = channel groups 2 wave, 2 wave + 1 of both alerts.
    // (The lane id is laundered per block: with the block loop unrolled hipcc otherwise computes every address of
    //  BOTH blocks' depthwise / LayerNorm phases once, keeps them across the first MLP and starves its filter
    //  fragments of registers -- fc1 then waits for an LDS read in front of every MFMA.)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int dj = ln & 3, db = ln >> 2;
    // ---- ordinary loads first (vmcnt retires in order: a load younger than a DMA would wait for it): fc1 bias,
    //      gamma*b2, the first channel group's Toeplitz taps and both groups' per-channel scalars
    const float b1v0 = bk.b1[tid], b1v1 = bk.b1[256 + tid];
    const float b2v = bk.gamma[tid & (C - 1)] * bk.b2[tid & (C - 1)];
    const uint2* twsrc = reinterpret_cast<const uint2*>(bk.par) + (2 * wave) * 64 + ln;
    frag4 tw[TW_R], twl[X2 ? TW_R : 1];
#pragma unroll
    for (int r = 0; r < TW_R; ++r) tw[r] = __builtin_bit_cast(frag4, twsrc[r * 512]);
    if (X2) {
#pragma unroll
      for (int r = 0; r < TW_R; ++r) twl[r] = __builtin_bit_cast(frag4, twsrc[TW_BYTES / 8 + r * 512]);
    }
    const float* pf = reinterpret_cast<const float*>(bk.par + (X2 ? 2 : 1) * TW_BYTES);
    float dwbias[2], lng[2], lnb2[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int c = 16 * (2 * wave + gi) + db;
      dwbias[gi] = pf[c];
      lng[gi] = pf[C + c];
      lnb2[gi] = pf[2 * C + c];
    }
    SC_STAMP(2 + 5 * j);
    __syncthreads();   // planar image complete (input / previous MLP + zero pads); ring slots 0, 1 free
    b1s[tid] = b1v0;
    b1s[256 + tid] = b1v1;
    if (tid < C) b2s[tid] = b2v;

    // ---- pointwise filters: chunk = 32 hidden units = 16 pieces of 1 KiB, 4 per wave.
    //      pieces 0..7 : W1 rows (LDS row m <- hidden unit 32*ch + swap23(m)), 256-byte rows,
    //                    16-byte chunk c of row m at position c ^ (m & 15)
    //      pieces 8..15: gamma*W2 columns 32*ch .. +31 of the 128 channel rows, 64-byte rows,
    //                    chunk c of row r at position c ^ F[(r >> 2) & 3]
    const unsigned char* wsrc[4];
    const unsigned char* wsrcl[4];   // split mode: the same pieces of the remainder images, 8 KB further into the slot
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave * 4 + i;
      if (pc < 8) {
        const int m = pc * 4 + (lane >> 4);
        const int hid = (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);   // swap bits 2 and 3
        const size_t o = (size_t)hid * (C * 2) + (((lane & 15) ^ (m & 15)) << 4);
        wsrc[i] = bk.w1 + o;
        wsrcl[i] = X2 ? bk.w1_lo + o : nullptr;
      } else {
        const int r = (pc - 8) * 16 + (lane >> 2);
        const size_t o = (size_t)r * (HID * 2) + (((lane & 3) ^ swz4(r)) << 4);
        wsrc[i] = bk.w2g + o;
        wsrcl[i] = X2 ? bk.w2g_lo + o : nullptr;
      }
    }
    // chunk k adds 32 W1 rows (8192 B) resp. 32 W2 columns (64 B)
    const int wstep0 = w1wave ? 32 * C * 2 : 64;
    // (every workgroup walks the 16 chunks in its own rotation -- fc2 sums over the hidden units,
    //  so the order is free -- which keeps the 512 workgroups off the same L2 lines)
    // waves 0, 1 load the W1 half of chunk k into W1 slot k % 3, waves 2, 3 the W2 half into W2 slot k % 3
    auto issue = [&](int k) {
      unsigned char* dst = (w1wave ? w1slot(k % NSLOT) : w2slot(k % NSLOT)) + (wave & 1) * 4096;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)((k + rot) & (NCH - 1)) * wstep0),
                                         (lptr_t)(dst + i * 1024), 16, 0, 0);
      if (X2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrcl[i] + (size_t)((k + rot) & (NCH - 1)) * wstep0),
                                           (lptr_t)(dst + HALFB + i * 1024), 16, 0, 0);
      }
    };
    // one of the four pieces (the chunk loop spreads them over its MFMAs: four LDS-DMAs back to back held the wave
    // for 230-420 cycles at the vector-memory port, a fifth of a step)
    auto issue_piece = [&](int k, int i) {   // (split mode: pieces 4..7 are the remainders)
      unsigned char* dst = (w1wave ? w1slot(k % NSLOT) : w2slot(k % NSLOT)) + (wave & 1) * 4096;
      if (X2 && i >= 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrcl[i & 3] + (size_t)((k + rot) & (NCH - 1)) * wstep0),
                                         (lptr_t)(dst + HALFB + (i & 3) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)((k + rot) & (NCH - 1)) * wstep0),
                                         (lptr_t)(dst + i * 1024), 16, 0, 0);
    };
    SC_STAMP(3 + 5 * j);

    // ---- depthwise 7x7 on the matrix pipe.  Per channel (= block) and output tile (rows 4 yb + j, columns 4 xb + i):
    //        D[i][j] = sum over ky, rb, k of  W[ky][4 rb + k - i + 3] * in[4 yb + j + ky - 3][4 (xb + rb) + k]
    //      a row step s = 4 yb + ky serves both yb with that sum: 22 reads, 56 MFMAs per (channel group, alert)
    // padded rows of the planar image: rofs[s] = byte offset of input row s + j - 3 (row 7 = zeros when outside)
    int rofs[11];
#pragma unroll
    for (int sx = 0; sx < 11; ++sx) {
      const int r = sx + dj - 3;
      rofs[sx] = ((unsigned)r < (unsigned)HW ? r : 7) * PL_ROW;
    }
    float v[2][2][16];   // [group][alert][yb * 8 + xb * 4 + i]
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      if (gi == 1) {   // the second group's taps into the same registers (their last use is behind us)
#pragma unroll
        for (int r = 0; r < TW_R; ++r) tw[r] = __builtin_bit_cast(frag4, twsrc[r * 512 + 64]);
        if (X2) {
#pragma unroll
          for (int r = 0; r < TW_R; ++r) twl[r] = __builtin_bit_cast(frag4, twsrc[TW_BYTES / 8 + r * 512 + 64]);
        }
      }
      // Both alerts of a row step together (8 to 16 products per step), the NEXT step's four reads requested in front of
      // them: read and used in the same step, each of a block's 44 steps began with an exposed LDS round trip.
      f32x4 acc[G][2][2];
#pragma unroll
      for (int al = 0; al < G; ++al)
#pragma unroll
        for (int yb = 0; yb < 2; ++yb)
#pragma unroll
          for (int xb = 0; xb < 2; ++xb) acc[al][yb][xb] = f32x4{dwbias[gi], dwbias[gi], dwbias[gi], dwbias[gi]};
      const unsigned char* lb = pl + (16 * (2 * wave + gi) + db) * 8;
      frag4 bq[2][G][2], bql[X2 ? 2 : 1][G][X2 ? 2 : 1];
      auto read_step = [&](int sx, int buf) {
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            bq[buf][al][q] = __builtin_bit_cast(frag4, *reinterpret_cast<const uint2*>(lb + al * PL_AL + q * PL_XQ + rofs[sx]));
            if (X2)
              bql[buf][al][q] =
                  __builtin_bit_cast(frag4, *reinterpret_cast<const uint2*>(lb + PLB + al * PL_AL + q * PL_XQ + rofs[sx]));
          }
      };
      read_step(0, 0);
#ifdef S1_EXP_NODW
#pragma unroll 1
      for (int sx = 0; sx < 0; ++sx) {
#else
#pragma unroll
      for (int sx = 0; sx < 11; ++sx) {
#endif
        // (touching this step's fragments makes hipcc wait for them here, while they are the only LDS reads in flight)
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            asm volatile("" : "+v"(bq[sx & 1][al][q]));
            if (X2) asm volatile("" : "+v"(bql[X2 ? sx & 1 : 0][al][X2 ? q : 0]));
          }
        if (sx + 1 < 11) read_step(sx + 1, (sx + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int yb = 0; yb < 2; ++yb) {
            const int ky = sx - 4 * yb;
            if (ky < 0 || ky > 6) continue;
#pragma unroll
            for (int rbi = 0; rbi < 3; ++rbi)
#pragma unroll
              for (int xb = 0; xb < 2; ++xb) {
                const int q = xb + rbi - 1;
                if (q < 0 || q > 1) continue;
                if (X2) {   // remainders first
                  acc[al][yb][xb] = SCM<DT>::run4(twl[ky * 3 + rbi], bq[sx & 1][al][q], acc[al][yb][xb]);
                  acc[al][yb][xb] = SCM<DT>::run4(tw[ky * 3 + rbi], bql[X2 ? sx & 1 : 0][al][X2 ? q : 0], acc[al][yb][xb]);
                }
                acc[al][yb][xb] = SCM<DT>::run4(tw[ky * 3 + rbi], bq[sx & 1][al][q], acc[al][yb][xb]);
              }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int al = 0; al < G; ++al)
#pragma unroll
        for (int yb = 0; yb < 2; ++yb)
#pragma unroll
          for (int xb = 0; xb < 2; ++xb)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[gi][al][yb * 8 + xb * 4 + i] = acc[al][yb][xb][i];
    }
    if (KEEP && a.keep_d[j] != nullptr) {   // the depthwise output before the LayerNorm: lane = (channel of the group, row j of
                                             // the quad) (nullptr: the backward recomputes it, dwln_bwd.hip)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int al = 0; al < G; ++al) {
          if (al >= nal) continue;
          float* dst = a.keep_d[j] + (size_t)(a0 + al) * PA * C + 16 * (2 * wave + gi) + db;
#pragma unroll
          for (int yb = 0; yb < 2; ++yb)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb)
              if (yb < 1 || dj < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  if (xb == 1 && i == 3) continue;
                  dst[((4 * yb + dj) * HW + 4 * xb + i) * C] = v[gi][al][yb * 8 + xb * 4 + i];
                }
              }
        }
    }
    SC_STAMP(4 + 5 * j);   // depthwise done
    // ---- LayerNorm over the 128 channels of a pixel: this wave's 2 groups in the lane, its 16 blocks by the
    //      transposing lane reduction, the 4 waves through LDS; single-pass variance
    {
      float o1[2], o2[2];
      {
        float s1[32];
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int n = 0; n < 16; ++n) s1[al * 16 + n] = v[0][al][n] + v[1][al][n];
        block_reduce32(s1, ln, o1);
      }
      {
        float s2[32];
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int n = 0; n < 16; ++n) s2[al * 16 + n] = fmaf(v[0][al][n], v[0][al][n], v[1][al][n] * v[1][al][n]);
        block_reduce32(s2, ln, o2);
      }
      // padded pixel slot [alert][row 0..7][column 0..7] of this lane's two totals
      const int slot = (ln >> 5) * 64 + (4 * ((ln >> 4) & 1) + dj) * 8 + 4 * ((ln >> 3) & 1) + 2 * ((ln >> 2) & 1);
      *reinterpret_cast<float2*>(part + wave * 128 + slot) = make_float2(o1[0], o1[1]);
      *reinterpret_cast<float2*>(part + 512 + wave * 128 + slot) = make_float2(o2[0], o2[1]);
    }
    __syncthreads();   // partial sums complete; nobody reads the planar image any more
    if (tid < 128) {
      const float t1 = part[tid] + part[128 + tid] + part[256 + tid] + part[384 + tid];
      const float t2 = part[512 + tid] + part[640 + tid] + part[768 + tid] + part[896 + tid];
      const float mean = t1 * (1.0f / C);
      const float rstd = rsqrtf(fmaxf(t2 * (1.0f / C) - mean * mean, 0.0f) + LN_EPS);
      st[tid] = rstd;
      st[128 + tid] = -mean * rstd;
    }
    __syncthreads();
    // the LayerNorm output into the [pixel][channel] image: its f16 values, or (split mode, second pass through the same
    // bytes) their f16 remainders
    auto write_ln = [&](bool lo_pass) {
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        T* mo = reinterpret_cast<T*>(stg) + 16 * (2 * wave + gi) + db;
#pragma unroll
        for (int al = 0; al < G; ++al)
#pragma unroll
          for (int yb = 0; yb < 2; ++yb)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
              const int slot = al * 64 + (4 * yb + dj) * 8 + 4 * xb;
              const float4 r4 = *reinterpret_cast<const float4*>(st + slot);
              const float4 m4 = *reinterpret_cast<const float4*>(st + 128 + slot);
              const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
              if (yb < 1 || dj < 3) {   // row 7 is padding
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  if (xb == 1 && i == 3) continue;   // column 7 is padding
                  const float y = fmaf(fmaf(v[gi][al][yb * 8 + xb * 4 + i], rr[i], mm[i]), lng[gi], lnb2[gi]);
                  const T yh = (T)y;
                  mo[(al * PA + (4 * yb + dj) * HW + 4 * xb + i) * (PITCH / 2)] = lo_pass ? (T)(y - (float)yh) : yh;
                }
              }
            }
      }
    };
    write_ln(false);
    __syncthreads();   // LN image complete
    if (KEEP) {   // the LayerNorm output rows, 16-byte pieces (16 per pixel row), before the ring takes the image's bytes
      unsigned char* dst = reinterpret_cast<unsigned char*>(a.keep_xn[j]) + (size_t)a0 * PA * C * 2;
      for (int i = tid; i < nal * PA * 16; i += 256) {
        const int pp = i >> 4, c16 = i & 15;
        *reinterpret_cast<uint4*>(dst + (size_t)pp * C * 2 + 16 * c16) = *reinterpret_cast<const uint4*>(stg + pp * PITCH + 16 * c16);
      }
    }
    issue(0);          // chunk 0 lives in slot 2 = the (dead) planar image's first 16 KB
    SC_STAMP(5 + 5 * j);

    // ---- fc1 -> GELU -> fc2 over 16 chunks, software-pipelined inside the wave: step ch issues chunk ch+1's fc1 MFMAs
    //      between the GELUs of chunk ch (one element per MFMA: ~7 VALU instructions pass while the matrix pipe
    //      works on a 32-cycle product), then chunk ch's fc2 MFMAs between the second half's GELUs.  Without this a
    //      wave runs LDS reads -> 8 dependent MFMAs -> 120 VALU -> 8 MFMAs strictly one after the other (2.9k cycles per
    //      chunk).  No pipe is saturated now (MFMA 20 % busy, LDS array 31 %, VALU active in 21 % of wave-cycles): a
    //      step is ~2k cycles of which ~0.5k are barrier, fragment-read issue and LDS-DMA issue.  fc2 accumulates into x
    //      (gamma is in the filter).
    load_x(j == 0 ? xsrc : xscr, x);
    {
      frag xf[KS1], xfl[X2 ? KS1 : 1];
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(stg + pm * PITCH + ks * 32 + h * 16);
      if (X2) {   // the remainders through the same image bytes
        __syncthreads();
        write_ln(true);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
          xfl[ks] = *reinterpret_cast<const frag*>(stg + pm * PITCH + ks * 32 + h * 16);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 bv = *reinterpret_cast<const float4*>(b2s + ct * 32 + 8 * qd + 4 * h);
          x[ct][4 * qd + 0] += bv.x;
          x[ct][4 * qd + 1] += bv.y;
          x[ct][4 * qd + 2] += bv.z;
          x[ct][4 * qd + 3] += bv.w;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // xf is in registers
      __syncthreads();   // ... everyone's: the LN image's bytes may take the W2 ring's slots 1, 2
      issue(1);
      if (w1wave) issue(2);
      // fc1 bias of chunk k into an accumulator: row (r&3) + 8(r>>2) + 4h holds hidden unit
      // 32k + (r&3) + 4((r>>2)&1) + 8h + 16(r>>3).  (Read as 16-bit vectors like the filter fragments: behind an LDS
      // read of float type hipcc waits vmcnt(0) while an LDS-DMA is in flight -- its type-based alias test takes the
      // DMA for a possible writer of those words -- and the chunks this ring keeps ahead are drained.)
      auto bias_acc = [&](int k, f32x16& acc) {
        const float* bp = b1s + ((k + rot) & (NCH - 1)) * 32 + 8 * h;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x4 bv = __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(bp + 4 * (qd & 1) + 16 * (qd >> 1)));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * qd + e] = bv[e];
        }
      };
      frag a1l[X2 ? KS1 : 1];   // split mode: the W1 fragments' remainders, read with their heads
      auto read_a1 = [&](int k, frag (&a1)[KS1]) {
        const unsigned char* w1s = w1slot(k % NSLOT);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          a1[ks] = *reinterpret_cast<const frag*>(w1s + lr * 256 + (((ks * 2 + h) ^ (lr & 15)) << 4));
          if (X2) a1l[ks] = *reinterpret_cast<const frag*>(w1s + HALFB + lr * 256 + (((ks * 2 + h) ^ (lr & 15)) << 4));
        }
      };
#ifdef S1_LOOPSTAMP
      unsigned long long lts[16];
#define LS(i) do { __builtin_amdgcn_sched_barrier(0); if (ch == 6 || ch == 7) lts[i + 8 * (ch - 6)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LS(i)
#endif
      // Register pipeline: entering step ch the wave holds hc = fc1 of chunk ch, hn = the fc1 bias of chunk ch + 1 and
      // a1 = the W1 fragments of chunk ch + 1 (both requested in the middle of step ch - 1, so their LDS latency --
      // 400+ cycles when eight waves read 20 KB each -- is not in anybody's way).
      f32x16 hacc[2];
      frag a1[KS1];
      {   // prologue: chunk 0's fc1, then the operands of step 0
        wait_vm<X2 ? 8 : 4>();   // W1(0) (and x) landed; at most this wave's youngest group is still out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bias_acc(0, hacc[0]);
        read_a1(0, a1);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          if (X2) {
            hacc[0] = SCM<T>::run(a1l[ks], xf[ks], hacc[0]);
            hacc[0] = SCM<T>::run(a1[ks], xfl[ks], hacc[0]);
          }
          hacc[0] = SCM<T>::run(a1[ks], xf[ks], hacc[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<0>();   // W1(1), W1(2) / W2(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // ... for everyone, and W1 slot 0 is read out
        bias_acc(1, hacc[1]);
        read_a1(1, a1);
        if (w1wave) issue(3);
      }
      // one step: hc = fc1 of chunk ch (complete), hn = bias of chunk ch + 1 <- fc1 of chunk ch + 1; hc <- bias of ch + 2
      auto step = [&](auto last_c, int ch, f32x16& hc, f32x16& hn) {
        constexpr bool LAST = decltype(last_c)::value;
        LS(0);
        // VM order of a wave: W1 waves W1(0..ch+3), W2 waves W2(0..ch+1); both need all but their youngest group
        wait_vm<X2 ? 8 : 4>();
        LS(1);
        // raw barrier: __syncthreads() would also wait vmcnt(0) while an LDS-DMA is in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of the last step is home
#ifndef S1_EXP_NOBAR
        __builtin_amdgcn_s_barrier();   // W1(ch+2), W2(ch) have landed for everyone; W1(ch+1), W2(ch-1) are read out
#endif
        LS(2);
        frag a2[CT][2], a2l[X2 ? CT : 1][2];
        const unsigned char* w2s = w2slot(ch % NSLOT);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int r = ct * 32 + lr;
            a2[ct][s2] = *reinterpret_cast<const frag*>(w2s + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
            if (X2) a2l[ct][s2] = *reinterpret_cast<const frag*>(w2s + HALFB + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
          }
        LS(3);
        const int kdma = ch + (w1wave ? 4 : 2);   // (indices past the last chunk wrap: harmless reloads into dead slots)
        __builtin_amdgcn_sched_barrier(0);
        LS(4);
        float g[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (!LAST) {
            if (X2) {
              hn = SCM<T>::run(a1l[r], xf[r], hn);
              hn = SCM<T>::run(a1[r], xfl[r], hn);
            }
            hn = SCM<T>::run(a1[r], xf[r], hn);
          }
#ifdef S1_EXP_NOGELU
          g[r] = hc[r];
#else
          g[r] = gelu_for<T>(hc[r]);
#endif
#ifndef S1_EXP_NODMA
          if (X2) issue_piece(kdma, r);
          else if (r & 1) issue_piece(kdma, r >> 1);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        LS(5);
        if (!LAST) read_a1(ch + 2, a1);
        frag hf, hfl;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          hf[r] = (T)g[r];
          if (X2) hfl[r] = (T)(g[r] - (float)hf[r]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (X2) {
            x[ct] = SCM<T>::run(a2l[ct][0], hf, x[ct]);
            x[ct] = SCM<T>::run(a2[ct][0], hfl, x[ct]);
          }
          x[ct] = SCM<T>::run(a2[ct][0], hf, x[ct]);
#ifdef S1_EXP_NOGELU
          g[8 + 2 * ct] = hc[8 + 2 * ct];
          g[9 + 2 * ct] = hc[9 + 2 * ct];
#else
          g[8 + 2 * ct] = gelu_for<T>(hc[8 + 2 * ct]);
          g[9 + 2 * ct] = gelu_for<T>(hc[9 + 2 * ct]);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!LAST) bias_acc(ch + 2, hc);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          hf[r] = (T)g[8 + r];
          if (X2) hfl[r] = (T)(g[8 + r] - (float)hf[r]);
        }
        LS(6);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (X2) {
            x[ct] = SCM<T>::run(a2l[ct][1], hf, x[ct]);
            x[ct] = SCM<T>::run(a2[ct][1], hfl, x[ct]);
          }
          x[ct] = SCM<T>::run(a2[ct][1], hf, x[ct]);
        }
        LS(7);
      };
#ifndef S1_EXP_NOLOOP
#pragma unroll 1
      for (int ch = 0; ch < NCH - 2; ch += 2) {
        step(std::false_type{}, ch, hacc[0], hacc[1]);
        step(std::false_type{}, ch + 1, hacc[1], hacc[0]);
      }
      step(std::false_type{}, NCH - 2, hacc[0], hacc[1]);
      step(std::true_type{}, NCH - 1, hacc[1], hacc[0]);
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last prefetches (unused) are not left in flight
#ifdef S1_LOOPSTAMP
      if (a.wgt != nullptr && blockIdx.x == 0 && tid == 0 && j == 0)
        for (int i = 0; i < 16; ++i) a.wgt[4096 + i] = lts[i];
#endif
      wait_vm<0>();   // the wrapped reloads: nothing may land in the ring once its bytes are reused
      if (j == 0) {      // next block's depthwise operand; chunk 15 (slot 2 = the same bytes) must be read out first
        __syncthreads();
        zero_pads();
        if (inmap) regs_to_planar<DT, PLO, KEEP && !std::is_same<T, f16_t>::value>(x, pl, p, h);
        if (live) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
              *reinterpret_cast<float4*>(xscr + ct * 32 + 8 * qd + 4 * h) =
                  make_float4(x[ct][4 * qd], x[ct][4 * qd + 1], x[ct][4 * qd + 2], x[ct][4 * qd + 3]);
        }
      }
    }
    SC_STAMP(6 + 5 * j);   // MLP done
  }
  if (a.tap_stage != nullptr && live) {
    float* tp = a.tap_stage + ((size_t)a0 * PA + p) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<float4*>(tp + ct * 32 + 8 * qd + 4 * h) =
            make_float4(x[ct][4 * qd], x[ct][4 * qd + 1], x[ct][4 * qd + 2], x[ct][4 * qd + 3]);
  }

  // ---- downsample: LN + conv 2x2 s2 (128 -> 256): 18 output pixels x 256 channels, K = 512
  {
    __syncthreads();   // every wave is out of the MLP: ring slots 0..1 take the LN image
    {
      f32x16 xn[CT];
      ln_regs(x, a.ds_lnw, a.ds_lnb, h, xn);
      if (inmap) regs_to_map<T, X2 ? MAPB : 0>(xn, stg, p, h);   // (split: the remainder image behind it, dead bytes too)
    }
    static_assert(2 * MAPB <= L::OFF_B1, "two [pixel][channel] images in front of the bias words");
    // this wave's output tiles wave, wave + 4 (32 channels each) x 32 k-steps: 64 filter fragments packed as
    // MFMA A operands (1 KiB each, launch_pack_frag32), a ring of 16 in flight
    constexpr int KSD = 4 * C / 16, RING = X2 ? 8 : 16, NSTEP = 2 * KSD;
    // (split: a packed fragment is 2 KiB, the heads' 1 KiB then the remainders')
    const frag* wsrc = reinterpret_cast<const frag*>(a.ds_w) + lane;
    auto fsrc = [&](int stp) {
      return wsrc + (size_t)((wave + 4 * (stp >> 5)) * KSD + (stp & (KSD - 1))) * (X2 ? 128 : 64);
    };
    frag wq[RING], wql[X2 ? RING : 1];
#pragma unroll
    for (int i = 0; i < RING; ++i) {
      wq[i] = *fsrc(i);
      if (X2) wql[i] = fsrc(i)[64];
    }
    __syncthreads();
    if (KEEP) {   // the downsample's patch rows [output pixel][q = 2 ky + kx][128] = LayerNorm'd pixels (2 oy + ky, 2 ox + kx)
      unsigned char* dst = reinterpret_cast<unsigned char*>(a.keep_patches) + (size_t)a0 * PO * 4 * C * 2;
      for (int i = tid; i < nal * PO * 4 * 16; i += 256) {
        const int c16 = i & 15, q = (i >> 4) & 3, o2 = i >> 6;
        const int g2 = o2 / PO, oo2 = o2 - g2 * PO;
        const int pin2 = g2 * PA + (2 * (oo2 / 3) + (q >> 1)) * HW + 2 * (oo2 % 3) + (q & 1);
        *reinterpret_cast<uint4*>(dst + (size_t)(o2 * 4 + q) * C * 2 + 16 * c16) =
            *reinterpret_cast<const uint4*>(stg + pin2 * PITCH + 16 * c16);
      }
    }
    SC_STAMP(12);
    const int o = lr;                                // output pixel slot: 18 of 32 used
    const bool olive = o < nal * PO;
    const int oc = o < G * PO ? o : 0;
    const int g = oc / PO, oo = oc - g * PO;
    const int oy = oo / 3, ox = oo - oy * 3;
    f32x16 acc;
    constexpr int RPT = KSD / RING;   // ring rounds per output tile
#ifdef S1_EXP_NODS
#pragma unroll 1
    for (int rd = 0; rd < 0; ++rd) {
#else
#pragma unroll 1
    for (int rd = 0; rd < NSTEP / RING; ++rd) {
#endif
      const int cot = wave + 4 * (rd / RPT);
      if (rd % RPT == 0) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 bv = *reinterpret_cast<const float4*>(a.ds_b + cot * 32 + 8 * qd + 4 * h);
          acc[4 * qd + 0] = bv.x;
          acc[4 * qd + 1] = bv.y;
          acc[4 * qd + 2] = bv.z;
          acc[4 * qd + 3] = bv.w;
        }
      }
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        const int stp = rd * RING + i, ks = stp & (KSD - 1);
        const int q = ks >> 3;                     // tap (ky*2 + kx): 8 k-steps of 16 channels each
        const int pin = g * PA + (2 * oy + (q >> 1)) * HW + 2 * ox + (q & 1);
        const frag bf = *reinterpret_cast<const frag*>(stg + pin * PITCH + (ks & 7) * 32 + h * 16);
        if (X2) {
          const frag bfl = *reinterpret_cast<const frag*>(stg + MAPB + pin * PITCH + (ks & 7) * 32 + h * 16);
          acc = SCM<T>::run(wql[i], bf, acc);
          acc = SCM<T>::run(wq[i], bfl, acc);
        }
        acc = SCM<T>::run(wq[i], bf, acc);
        if (stp + RING < NSTEP) {
          wq[i] = *fsrc(stp + RING);
          if (X2) wql[i] = fsrc(stp + RING)[64];
        }
      }
      if (rd % RPT == RPT - 1 && olive) {
        float* dst = a.out + ((size_t)a0 * PO + o) * CN + cot * 32 + 4 * h;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(dst + 8 * qd) =
              make_float4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
      }
    }
    SC_STAMP(13);
  }
}

// one block's parameter image (layout at TW_R above); taps: tap-major [49][128] fp32
template <typename T, bool X2>
__global__ void pack_s1par_kernel(const float* __restrict__ taps, const float* __restrict__ dw_b,
                                  const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                  unsigned char* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NTW = TW_R * C * 16;
  if (idx < NTW) {
    const int k = idx & 3, i = (idx >> 2) & 3, c = (idx >> 4) & (C - 1), r = idx / (16 * C);
    const int ky = r / 3, rb = r % 3 - 1, kx = 4 * rb + k - i + 3;
    const float w = (kx >= 0 && kx < 7) ? taps[(ky * 7 + kx) * C + c] : 0.f;
    const T wh = (T)w;
    reinterpret_cast<T*>(out)[idx] = wh;
    if (X2) reinterpret_cast<T*>(out)[NTW + idx] = (T)(w - (float)wh);
    return;
  }
  const int f = idx - NTW;
  if (f >= 3 * C) return;
  float v;
  if (f < C) v = dw_b[f];
  else if (f < 2 * C) v = ln_w[f - C];
  else v = ln_b[f - 2 * C];
  reinterpret_cast<float*>(out + (X2 ? 2 : 1) * TW_BYTES)[f] = v;
}

// fp32 downsample filter [Cout][Cin][2][2] -> 32x32x16 A fragments [row tile][k-step][lane][8], lane l holds row
// (l & 31), k = 16 s + 8 (l >> 5) + j of its tile, k = (2 ky + kx) * Cin + c
template <typename T, bool X2 = false>
__global__ void pack_frag32_kernel(const float* __restrict__ w, T* __restrict__ out, int rows, int cin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = 4 * cin;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = K / 16;
  const int sk = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 32 * tile + (l & 31), k = 16 * sk + 8 * (l >> 5) + j;
  const int q = k / cin, c = k - q * cin;
  const float v = w[((long)row * cin + c) * 4 + q];
  const T vh = (T)v;
  if (X2) {   // 2 KiB per fragment: [lane][8] heads, then [lane][8] remainders
    out[fs * 1024 + l * 8 + j] = vh;
    out[fs * 1024 + 512 + l * 8 + j] = (T)(v - (float)vh);
  } else {
    out[i] = vh;
  }
}

template <typename T, bool X2 = false, int WPS = 2, bool KEEP = false> int launch_stage1b_t(const Stage1Args& a, hipStream_t st) {
  auto kern = stage1b_kernel<T, X2, WPS, KEEP>;
  static DevOnce attr_set;
  static const int pad = [] {   // (BTSBOT_AMD_S1_ONE_WG=1: the same probe as stage0b.hip's)
    const char* e = getenv("BTSBOT_AMD_S1_ONE_WG");
    return e != nullptr && e[0] == '1' && !X2 ? 90 * 1024 - S1L<X2>::LDS_BYTES : 0;
  }();
  const int LDS_BYTES = S1L<X2>::LDS_BYTES + pad;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3((a.B + G - 1) / G), dim3(256), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

#ifdef STAGE1B_X2_TU
// ---- this translation unit (stage1x.hip) holds only the split-operand instantiations (see stage0b.hip)
int launch_stage1b_x2(const Stage1Args& a, hipStream_t st) {
  // (one workgroup per CU -- two planar images, doubled ring slots: 143 KB of LDS -- with 512 registers)
  if (a.blk[0].w1_lo == nullptr || a.blk[0].w2g_lo == nullptr || a.blk[1].w1_lo == nullptr || a.blk[1].w2g_lo == nullptr) {
    btsbot_set_error("stage1b: the split mode needs the filters' remainder images (w1_lo, w2g_lo)");
    return BTSBOT_ERR_INVALID_ARG;
  }
  return launch_stage1b_t<f16_t, true, 1>(a, st);
}
int launch_pack_s1par_x2(const float* taps, const float* dw_b, const float* ln_w, const float* ln_b, void* out,
                         hipStream_t st) {
  const dim3 grid((TW_R * C * 16 + 3 * C + 255) / 256), blk(256);
  hipLaunchKernelGGL((pack_s1par_kernel<f16_t, true>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b,
                     reinterpret_cast<unsigned char*>(out));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
int launch_pack_frag32_x2(const float* src, void* dst, int cout, int cin, hipStream_t st) {
  const long total = (long)cout * 4 * cin;
  hipLaunchKernelGGL((pack_frag32_kernel<f16_t, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src,
                     reinterpret_cast<f16_t*>(dst), cout, cin);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
#else
int launch_stage1b_x2(const Stage1Args& a, hipStream_t st);
int launch_pack_s1par_x2(const float* taps, const float* dw_b, const float* ln_w, const float* ln_b, void* out,
                         hipStream_t st);
int launch_pack_frag32_x2(const float* src, void* dst, int cout, int cin, hipStream_t st);

size_t s1par_bytes() { return PARB_X2; }   // (the split mode's image is the larger one)

int launch_pack_s1par(int prec, const float* taps, const float* dw_b, const float* ln_w,
                      const float* ln_b, void* out, hipStream_t st) {
  unsigned char* o = reinterpret_cast<unsigned char*>(out);
  const dim3 grid((TW_R * C * 16 + 3 * C + 255) / 256), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL((pack_s1par_kernel<bf16_t, false>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b, o);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL((pack_s1par_kernel<f16_t, false>), grid, blk, 0, st, taps, dw_b, ln_w, ln_b, o);
  else if (prec == BTSBOT_F16X2)
    return launch_pack_s1par_x2(taps, dw_b, ln_w, ln_b, out, st);
  else {
    btsbot_set_error("pack_s1par: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// downsample filter [Cout][Cin][2][2] fp32 -> 32x32x16 MFMA A fragments (stage1b.hip's / stage0b.hip's last phase)
int launch_pack_frag32(int prec, const float* src, void* dst, int cout, int cin, hipStream_t st) {
  const long total = (long)cout * 4 * cin;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_frag32_kernel<bf16_t>, grid, blk, 0, st, src, reinterpret_cast<bf16_t*>(dst), cout, cin);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_frag32_kernel<f16_t>, grid, blk, 0, st, src, reinterpret_cast<f16_t*>(dst), cout, cin);
  else if (prec == BTSBOT_F16X2)
    return launch_pack_frag32_x2(src, dst, cout, cin, st);
  else {
    btsbot_set_error("pack_frag32: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

bool stage1_supported(int prec, int c1, int c2) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16 || prec == BTSBOT_F16X2) && c1 == 128 && c2 == 256;
}

// Needs Stage0Blk::par (launch_pack_s1par), ::w1 (plain [512][128]) and Stage0Blk::w2g (gamma-scaled [128][512]), 16-bit,
// and Stage1Args::ds_w as MFMA fragments (launch_pack_frag32).
int launch_stage1b(int prec, const Stage1Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (a.keep_xn[0] != nullptr) {   // the training forward
    if ((a.keep_d[0] == nullptr) != (a.keep_d[1] == nullptr) || a.keep_xn[1] == nullptr || a.keep_patches == nullptr ||
        a.tap_stage == nullptr || a.scratch == nullptr) {
      btsbot_set_error("stage1b: the training forward needs every kept buffer");
      return BTSBOT_ERR_INVALID_ARG;
    }
    if (prec == BTSBOT_BF16) return launch_stage1b_t<bf16_t, false, 2, true>(a, st);
    if (prec == BTSBOT_F16) return launch_stage1b_t<f16_t, false, 2, true>(a, st);
    btsbot_set_error("stage1b: the training forward runs in the bf16 / f16 modes, not %d", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16) return launch_stage1b_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage1b_t<f16_t>(a, st);
  if (prec == BTSBOT_F16X2) return launch_stage1b_x2(a, st);
  btsbot_set_error("stage1b: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
#endif
