// Stage-3 pointwise kernels (gfx950).  The last ConvNeXt stage works on 1x1 maps, so a block
//
//     x += gamma * fc2( GELU( fc1( LN( dwconv7x7(x) ) ) ) )
//
// (timm ConvNeXtBlock, reached from /root/reference/btsbot/architectures.py:108,132) is two GEMMs over the batch:
// [B x C] x [C x 4C] and [B x 4C] x [4C x C], with the depthwise filter reduced to its centre tap.  Two launches per
// block, 256 workgroups of 8 waves each at B = 1024, every global operand a 1 KiB contiguous MFMA fragment:
//   s3_fc1  tile = 32 alerts x 256 hidden units, wave = 32 hidden units.  The 32 rows are requested first, then the
//           wave's filter fragments (straight into registers, all of them): centre tap + bias + LayerNorm of the rows
//           (4 per wave; redone by the 8 tiles that share them: 32 x C values, cheaper than a launch of its own) runs
//           while the filter streams in; 16-bit rows in LDS are the B operand.  Epilogue bias + GELU; the filter rows
//           are packed with bits 2/3 of the row index swapped, so a lane's accumulator holds 2 x 8 consecutive hidden
//           units of its alert = exactly fc2's B fragment: h leaves as [alert block][k-step][lane][8], 1 KiB per store.
//   s3_fc2  tile = 64 alerts x 32 channels, the 8 waves split K = 4C; A (gamma * W2) and B (h) fragments are all
//           requested up front (48 per wave); the partial tiles meet in LDS, + gamma * b2 + x, in place.
// Bound: what one CU can pull through its TA port, ~64 B/clk: 320 KB per fc1 tile, 384 KB per fc2 tile = 5k / 6k
// cycles, behind one MALL round trip (the L2s start every kernel empty); algorithmic FLOPs 2 x 2.1 GFLOP per block.
// (One cooperative launch for the whole stage was tried: a grid-wide barrier costs 8-20 us on this part -- 256
//  same-address device-scope atomics, or cooperative_groups' grid.sync() -- against ~3 us for a kernel boundary.)
#include <stdlib.h>

#include "common.h"
#include "stage3.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// operand traits: fragment type (8 elements per lane), element size, MFMA, and the conversions from fp32
template <typename T> struct S3M;
// Every trait also says how a fragment lies in memory: ldg / stg = fragment number f of a packed fragment array in HBM
// (64 lanes x 8 values), lds_ld8 / lds_st2 = values c .. c + 7 / c, c + 1 of an operand row in LDS (`plane` = the
// row's width in values; only the split mode, which keeps a row's remainders `plane` values behind its heads, uses it).
template <typename T> struct S3M16 {
  static constexpr int ESZ = 2;
  static constexpr int KSTEP = 16;        // k per MFMA; a lane holds KSTEP / 2 consecutive values of its row / column
  static constexpr bool MX = false;
  static constexpr int MAXRING = 32;      // fragments a wave keeps in flight per operand stream
  typedef T pair_t __attribute__((ext_vector_type(2)));
  typedef T frag __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ frag ldg(const void* base, size_t f, int lane) {
    return reinterpret_cast<const frag*>(base)[f * 64 + lane];
  }
  static __device__ __forceinline__ void stg(void* base, size_t f, int lane, frag v) {
    reinterpret_cast<frag*>(base)[f * 64 + lane] = v;
  }
  static __device__ __forceinline__ frag lds_ld8(const unsigned char* row, int c, int plane) {
    return *reinterpret_cast<const frag*>(row + c * 2);
  }
  static __device__ __forceinline__ void lds_st2(unsigned char* row, int c, int plane, float a, float b) {
    *reinterpret_cast<pair_t*>(row + c * 2) = pack2(a, b);
  }
  static __device__ __forceinline__ pair_t pack2(float a, float b) {
    pair_t o;
    o[0] = (T)a;
    o[1] = (T)b;
    return o;
  }
  static __device__ __forceinline__ frag pack8(const float (&v)[8]) {
    frag o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
    return o;
  }
};
template <> struct S3M<bf16_t> : S3M16<bf16_t> {
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct S3M<f16_t> : S3M16<f16_t> {
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
// fp8 (OCP e4m3) on the block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4: 64 k per instruction, twice the bf16 rate
// per clock (stage2p.hip has the 16x16x128 form and the note on the scales, both the constant 2^0 here too).  A lane
// holds 32 consecutive k of its row / column: lane l = row l & 31, k = 32 (l >> 5) + j.  Activations are clamped to the
// format's range before the conversion (its largest finite value is 448; what lies beyond would become NaN).
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
template <> struct S3M<fp8_t> {
  static constexpr int ESZ = 1;
  static constexpr int KSTEP = 64;
  static constexpr bool MX = true;
  static constexpr int MAXRING = 16;      // 8 registers per fragment
  typedef unsigned short pair_t;
  typedef v8i32 frag;
  static __device__ __forceinline__ float clamp8(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }
  static __device__ __forceinline__ pair_t pack2(float a, float b) {
    return (unsigned short)__builtin_amdgcn_cvt_pk_fp8_f32(clamp8(a), clamp8(b), 0, false);
  }
  // a packed fragment in HBM is 2 KiB: [lane][k 0..15 of its 32], then [lane][k 16..31]
  static __device__ __forceinline__ frag ldg(const void* base, size_t f, int lane) {
    const v4i32* p = reinterpret_cast<const v4i32*>(base) + f * 128 + lane;
    const v4i32 lo = p[0], hi = p[64];
    return frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  // 16 values (registers 0..15 of a 32x32 accumulator) as half `half` of fragment f: one coalesced 1 KiB store per wave
  static __device__ __forceinline__ void stg_half(void* base, size_t f, int half, int lane, const float (&v)[16]) {
    v4i32 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(v[4 * q]), clamp8(v[4 * q + 1]), 0, false);
      o[q] = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(v[4 * q + 2]), clamp8(v[4 * q + 3]), w, true);
    }
    reinterpret_cast<v4i32*>(base)[f * 128 + half * 64 + lane] = o;
  }
  static __device__ __forceinline__ frag lds_ld8(const unsigned char* row, int c, int plane) {   // 32 values from c on
    const v4i32 lo = *reinterpret_cast<const v4i32*>(row + c), hi = *reinterpret_cast<const v4i32*>(row + c + 16);
    return frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  static __device__ __forceinline__ void lds_st2(unsigned char* row, int c, int plane, float a, float b) {
    *reinterpret_cast<pair_t*>(row + c) = pack2(a, b);
  }
  static __device__ __forceinline__ f32x16 run(const frag& a, const frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  }
};
// split operands (BTSBOT_F16X2): value = f16 head + f16 remainder; a product is hi*hi + hi*lo + lo*hi on the f16 MFMA
// (the lo*lo term is below 2^-22 of the product).  A packed fragment is 2 KiB: the heads' 1 KiB, then the remainders'.
template <> struct S3M<f16x2_t> {
  static constexpr int ESZ = 4;
  static constexpr int KSTEP = 16;
  static constexpr bool MX = false;
  static constexpr int MAXRING = 16;      // 8 registers per fragment
  typedef h2x8 frag;
  static __device__ __forceinline__ frag pack8(const float (&v)[8]) { return split8(v); }
  static __device__ __forceinline__ frag ldg(const void* base, size_t f, int lane) {
    const f16x8* p = reinterpret_cast<const f16x8*>(base) + f * 128 + lane;
    frag o;
    o.hi = p[0];
    o.lo = p[64];
    return o;
  }
  static __device__ __forceinline__ void stg(void* base, size_t f, int lane, frag v) {
    f16x8* p = reinterpret_cast<f16x8*>(base) + f * 128 + lane;
    p[0] = v.hi;
    p[64] = v.lo;
  }
  static __device__ __forceinline__ frag lds_ld8(const unsigned char* row, int c, int plane) {
    frag o;
    o.hi = *reinterpret_cast<const f16x8*>(row + c * 2);
    o.lo = *reinterpret_cast<const f16x8*>(row + (plane + c) * 2);
    return o;
  }
  static __device__ __forceinline__ void lds_st2(unsigned char* row, int c, int plane, float a, float b) {
    f16x2v hi, lo;
    _Float16 h0, l0, h1, l1;
    split_f16(a, h0, l0);
    split_f16(b, h1, l1);
    hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
    *reinterpret_cast<f16x2v*>(row + c * 2) = hi;
    *reinterpret_cast<f16x2v*>(row + (plane + c) * 2) = lo;
  }
  static __device__ __forceinline__ f32x16 run(const frag& a, const frag& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
  }
};
// GELU of the operand mode: fp8 rides on the bf16 schedule (degree-3 polynomial form)
template <typename T> struct GeluOf { using type = T; };
template <> struct GeluOf<fp8_t> { using type = bf16_t; };

#define S3_STAMP(i)                                                                       \
  do {                                                                                    \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)

constexpr int NT = 512;           // threads per workgroup (8 waves)
constexpr int M1 = 32, N1 = 256;  // fc1 tile: alerts x hidden units
constexpr int M2 = 64, N2 = 32;   // fc2 tile: alerts x channels
constexpr float LN_EPS = 1e-6f;
constexpr int RED_BYTES = 8 * 2 * 4 * 64 * 16;   // fc2: 8 K slices x 2 alert blocks x 4 quads x 64 lanes x float4

// MB = 32-alert row blocks per workgroup (1; 2 at C = 640, where 32 x 10 tiles would be 320 workgroups = two rounds of the
// chip's 256 CUs and 16 x 10 = 160 with twice the rows per filter fragment is one)
template <typename T, int C, int MB = 1>
__device__ __forceinline__ void fc1_tile(const Stage3Args& a, const Stage3Blk& bk, int ab, int nt, unsigned char* smem) {
  using frag = typename S3M<T>::frag;
  constexpr int KSTEP = S3M<T>::KSTEP, VPL = KSTEP / 2;
  constexpr int HID = 4 * C, KS = C / KSTEP, NV = C / 128;
  constexpr int RING0 = KS > 32 ? KS / 2 : KS, RING1 = RING0 < S3M<T>::MAXRING ? RING0 : S3M<T>::MAXRING;
  constexpr int RING = RING1 / 4 * 4;   // (four quarters, one requested in front of each row's LayerNorm)
  static_assert(RING >= 4, "ring quarters");
  constexpr int ESZ = S3M<T>::ESZ;
  constexpr int PITCH = C * ESZ + 16;   // bytes per LDS row: C operand values + 16 (8 rows cover the 32 banks)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int ht = nt * (N1 / 32) + wave;                 // this wave's tile of 32 hidden units
  S3_STAMP(0);
  // ---- the 32 rows first (4 per wave; lane = channels 128 i + 2 lane + {0, 1}), then the per-channel constants
  constexpr int RPW = 4 * MB;   // rows per wave
  float2 v[RPW][NV];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int al = ab * M1 * MB + RPW * wave + u;
    const float* src = a.x + (size_t)(al < a.B ? al : a.B - 1) * C + 2 * lane;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[u][i] = *reinterpret_cast<const float2*>(src + 128 * i);
  }
  float2 wc[NV], bc[NV], lw[NV], lb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 128 * i + 2 * lane;
    wc[i] = *reinterpret_cast<const float2*>(bk.dw_c + c);
    bc[i] = *reinterpret_cast<const float2*>(bk.dw_b + c);
    lw[i] = *reinterpret_cast<const float2*>(bk.ln_w + c);
    lb[i] = *reinterpret_cast<const float2*>(bk.ln_b + c);
  }
  // ---- the wave's filter fragments, a quarter in front of each row's LayerNorm: the requests queue at the TA port
  //      (a wave cannot run ahead of its own unissued loads) while the VALU works on the row before
  const size_t wf0 = (size_t)ht * KS;      // this wave's first filter fragment
  frag wq[RING];
  S3_STAMP(9);
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    if (u < 4) {
#pragma unroll
      for (int i = u * (RING / 4); i < (u + 1) * (RING / 4); ++i) wq[i] = S3M<T>::ldg(bk.w1p, wf0 + i, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[u][i].x = fmaf(v[u][i].x, wc[i].x, bc[i].x);
      v[u][i].y = fmaf(v[u][i].y, wc[i].y, bc[i].y);
      s += v[u][i].x + v[u][i].y;
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[u][i].x -= mean;
      v[u][i].y -= mean;
      q = fmaf(v[u][i].x, v[u][i].x, fmaf(v[u][i].y, v[u][i].y, q));
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + LN_EPS);
    unsigned char* row = smem + (RPW * wave + u) * PITCH;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      S3M<T>::lds_st2(row, 128 * i + 2 * lane, C, fmaf(v[u][i].x * rstd, lw[i].x, lb[i].x),
                      fmaf(v[u][i].y * rstd, lw[i].y, lb[i].y));
    __builtin_amdgcn_sched_barrier(0);
  }
  S3_STAMP(10);
  // ---- fc1 bias into the accumulator: register r of a lane = hidden unit 32 ht + (r & 7) + 8 h + 16 (r >> 3).
  //      (fp8: the filter is packed times a power of two S1 -- bk.scales = {S1, 1/S1, S2, 1/S2} -- so the bias goes in
  //      times S1 and the sum comes out times 1/S1; the 16-bit modes have no scales)
  constexpr bool F8 = std::is_same<T, fp8_t>::value;
  const float s1 = F8 ? bk.scales[0] : 1.0f, is1 = F8 ? bk.scales[1] : 1.0f;
  f32x16 acc[MB];
  {
    const float* bp = bk.b1 + 32 * ht + 8 * h;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * (qd & 1) + 16 * (qd >> 1));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][4 * qd + 0] = bv.x * s1;
        acc[mb][4 * qd + 1] = bv.y * s1;
        acc[mb][4 * qd + 2] = bv.z * s1;
        acc[mb][4 * qd + 3] = bv.w * s1;
      }
    }
  }
  __syncthreads();
  S3_STAMP(11);
  const unsigned char* bp = smem + lr * PITCH;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const frag bf = S3M<T>::lds_ld8(bp + mb * 32 * PITCH, ks * KSTEP + h * VPL, C);
      acc[mb] = S3M<T>::run(wq[ks % RING], bf, acc[mb]);
    }
    if (ks + RING < KS) wq[ks % RING] = S3M<T>::ldg(bk.w1p, wf0 + ks + RING, lane);
  }
  S3_STAMP(12);
  // ---- GELU, out as fc2's B fragments: lane (alert lr, half h) holds k = 16 ks2 + 8 h + 0..7 of k-step ks2 = 2 ht + hh
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const size_t ab32 = (size_t)ab * MB + mb;   // the 32-alert block these rows are
    if constexpr (S3M<T>::MX) {
      // fc2 sums over all hidden units, so their order inside its k-steps is free: this lane's 16 values are bytes
      // 16 (ht & 1) .. + 15 of its slot in fragment ht >> 1 -- position (S, half h, j) of fc2's k holds hidden unit
      // 64 S + 32 (j >> 4) + (j & 7) + 8 h + 16 ((j >> 3) & 1), which is how launch_pack_s3 orders gamma * W2 (kperm)
      float g[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) g[j] = gelu_for<typename GeluOf<T>::type>(acc[mb][j] * is1);
      S3M<T>::stg_half(a.hfrag, ab32 * (HID / 64) + (ht >> 1), ht & 1, lane, g);
    } else {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = gelu_for<typename GeluOf<T>::type>(acc[mb][8 * hh + j] * is1);
        S3M<T>::stg(a.hfrag, ab32 * (HID / 16) + 2 * ht + hh, lane, S3M<T>::pack8(g));
      }
    }
  }
  __syncthreads();   // the rows in LDS are read out: the next tile may overwrite them
}

// NC = 32-channel tiles per workgroup (1; 2 at C = 640: 16 x 10 = 160 workgroups instead of 320, see fc1_tile)
template <typename T, int C, int NC = 1>
__device__ __forceinline__ void fc2_tile(const Stage3Args& a, const Stage3Blk& bk, int mt, int ct, unsigned char* smem) {
  using frag = typename S3M<T>::frag;
  constexpr int HID = 4 * C, KSA = HID / S3M<T>::KSTEP, KSW = KSA / 8;
  // two channel tiles: four fragment streams -- at most 10 fragments each (160 registers), a divisor of the wave's k-steps
  constexpr int RING0 = NC == 1 ? (KSW <= 16 ? KSW : KSW / 2) : KSW % 10 == 0 ? 10 : KSW % 8 == 0 ? 8 : KSW % 4 == 0 ? 4 : KSW;
  // (split / fp8: a fragment is 8 registers, 3 streams x 8 fragments at most; two channel tiles: 4 x 4)
  constexpr int RING = S3M<T>::MAXRING >= 32 ? RING0 : NC == 2 && KSW % 4 == 0 ? 4 : KSW % 8 == 0 ? 8 : KSW % 4 == 0 ? 4 : KSW;
  static_assert(KSW % RING == 0, "k-steps per wave");
  float4* red = reinterpret_cast<float4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = mt * M2;
  const size_t wf0 = (size_t)(ct * NC) * KSA + wave * KSW, hf0 = (size_t)(m0 / 32) * KSA + wave * KSW, hf1 = hf0 + KSA;
  frag wa[NC][RING], ha[RING], hb[RING];
  S3_STAMP(1);
#pragma unroll
  for (int i = 0; i < RING; ++i) {
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) wa[nc][i] = S3M<T>::ldg(bk.w2p, wf0 + (size_t)nc * KSA + i, lane);
    ha[i] = S3M<T>::ldg(a.hfrag, hf0 + i, lane);
    hb[i] = S3M<T>::ldg(a.hfrag, hf1 + i, lane);
  }
  // the epilogue's operands (this thread's four residual values, layer scale, bias) are requested behind the fragments:
  // asked for only after the K slices had met in LDS they were one more exposed memory round trip per launch (~2.5k cycles
  // of the workgroup's 14.7k).  (One channel tile per workgroup only: with two the fragments leave no registers for them.)
  constexpr bool EPI_PRE = NC == 1;
  float4 xv_pre = make_float4(0.f, 0.f, 0.f, 0.f), g_pre = xv_pre, b_pre = xv_pre;
  if constexpr (EPI_PRE) {
    const int q = tid, t = q >> 8, qd = (q >> 6) & 3, ln = q & 63;
    const int al = m0 + 32 * t + (ln & 31), c = 32 * ct + 8 * qd + 4 * (ln >> 5);
    xv_pre = *reinterpret_cast<const float4*>(a.x + (size_t)(al < a.B ? al : a.B - 1) * C + c);
    g_pre = *reinterpret_cast<const float4*>(bk.gamma + c);
    b_pre = *reinterpret_cast<const float4*>(bk.b2 + c);
  }
  __builtin_amdgcn_sched_barrier(0);   // (left alone, hipcc sinks the loads to their MFMAs: ~10 in flight instead of 48)
  f32x16 acc[NC][2];
#pragma unroll
  for (int nc = 0; nc < NC; ++nc)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nc][0][r] = acc[nc][1][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) {
      acc[nc][0] = S3M<T>::run(wa[nc][ks % RING], ha[ks % RING], acc[nc][0]);
      acc[nc][1] = S3M<T>::run(wa[nc][ks % RING], hb[ks % RING], acc[nc][1]);
    }
    if (ks + RING < KSW) {
#pragma unroll
      for (int nc = 0; nc < NC; ++nc) wa[nc][ks % RING] = S3M<T>::ldg(bk.w2p, wf0 + (size_t)nc * KSA + ks + RING, lane);
      ha[ks % RING] = S3M<T>::ldg(a.hfrag, hf0 + ks + RING, lane);
      hb[ks % RING] = S3M<T>::ldg(a.hfrag, hf1 + ks + RING, lane);
    }
  }
  S3_STAMP(13);
  // ---- the eight K slices meet in LDS; register r = channel 32 ct + (r & 3) + 8 (r >> 2) + 4 h of alert (lane & 31)
#pragma unroll
  for (int nc = 0; nc < NC; ++nc)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        red[(((wave * NC + nc) * 2 + t) * 4 + qd) * 64 + lane] =
            make_float4(acc[nc][t][4 * qd], acc[nc][t][4 * qd + 1], acc[nc][t][4 * qd + 2], acc[nc][t][4 * qd + 3]);
  __syncthreads();
#pragma unroll
  for (int nc = 0; nc < NC; ++nc) {
    const int q = tid, t = q >> 8, qd = (q >> 6) & 3, ln = q & 63;
    const int al = m0 + 32 * t + (ln & 31), c = 32 * (ct * NC + nc) + 8 * qd + 4 * (ln >> 5);
    float4 s = red[nc * 512 + q];
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float4 p = red[(w * NC + nc) * 512 + q];
      s.x += p.x;
      s.y += p.y;
      s.z += p.z;
      s.w += p.w;
    }
    if (std::is_same<T, fp8_t>::value) {   // fp8: gamma * W2 was packed times S2
      const float is2 = bk.scales[3];
      s.x *= is2;
      s.y *= is2;
      s.z *= is2;
      s.w *= is2;
    }
    if (al < a.B) {
      float* xp = a.x + (size_t)al * C + c;
      const float4 xv = EPI_PRE ? xv_pre : *reinterpret_cast<const float4*>(xp);
      const float4 g = EPI_PRE ? g_pre : *reinterpret_cast<const float4*>(bk.gamma + c);
      const float4 b = EPI_PRE ? b_pre : *reinterpret_cast<const float4*>(bk.b2 + c);
      *reinterpret_cast<float4*>(xp) = make_float4(xv.x + fmaf(g.x, b.x, s.x), xv.y + fmaf(g.y, b.y, s.y),
                                                   xv.z + fmaf(g.z, b.z, s.z), xv.w + fmaf(g.w, b.w, s.w));
    }
  }
  S3_STAMP(14);
  __syncthreads();   // the partial sums are read out
}


template <typename T, int C, int MB = 1>
__global__ __launch_bounds__(NT) void s3_fc1_kernel(Stage3Args a, int j) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTL = 4 * C / N1;
  // (blockIdx % 8 = the XCD: consecutive workgroups take different hidden tiles of one alert block, so an XCD's L2
  //  holds 1/8 of the filter)
  fc1_tile<T, C, MB>(a, a.blk[j], blockIdx.x / NTL, blockIdx.x % NTL, smem);
}

template <typename T, int C, int NC = 1>
__global__ __launch_bounds__(NT) void s3_fc2_kernel(Stage3Args a, int j) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTL = C / (N2 * NC);
  fc2_tile<T, C, NC>(a, a.blk[j], blockIdx.x / NTL, blockIdx.x % NTL, smem);
}

// fp32 [rows][K] (x rowscale[row]) -> 32x32x16 A fragments [row tile][k-step][lane][8]: lane l holds tile row (l & 31),
// k = 16 s + 8 (l >> 5) + j; with swap23 tile row r carries source row (r with bits 2 and 3 exchanged)
template <typename T>
__global__ void pack_s3_kernel(const float* __restrict__ w, const float* __restrict__ rowscale, T* __restrict__ out,
                               int rows, int K, int swap23) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = K / 16;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  int r = l & 31;
  if (swap23) r = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
  const int row = 32 * tile + r, k = 16 * s + 8 * (l >> 5) + j;
  float v = w[(long)row * K + k];
  if (rowscale != nullptr) v *= rowscale[row];
  out[i] = (T)v;
}

// split mode: the same fragments, 2 KiB each: [lane][8] heads, then [lane][8] remainders
__global__ void pack_s3_split_kernel(const float* __restrict__ w, const float* __restrict__ rowscale,
                                     _Float16* __restrict__ out, int rows, int K, int swap23) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = K / 16;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  int r = l & 31;
  if (swap23) r = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
  const int row = 32 * tile + r, k = 16 * s + 8 * (l >> 5) + j;
  float v = w[(long)row * K + k];
  if (rowscale != nullptr) v *= rowscale[row];
  _Float16 hi, lo;
  split_f16(v, hi, lo);
  out[fs * 1024 + l * 8 + j] = hi;
  out[fs * 1024 + 512 + l * 8 + j] = lo;
}

// ---- fp8: max |w| -> power-of-two scale -> packed bytes
__global__ void absmax_kernel(const float* __restrict__ w, const float* __restrict__ rowscale, unsigned* __restrict__ bits,
                              long n, int K) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i] * (rowscale != nullptr ? rowscale[i / K] : 1.0f)));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(bits, __float_as_uint(m));   // (non-negative floats order like their bits)
}
__global__ void scale_from_max_kernel(float* scale) {
  const float m = __uint_as_float(*reinterpret_cast<const unsigned*>(scale));
  const float s = m > 0.f ? exp2f(floorf(log2f(240.0f / m))) : 1.0f;
  scale[0] = s;
  scale[1] = 1.0f / s;
}
__global__ void pack_s3_fp8_kernel(const float* __restrict__ w, const float* __restrict__ rowscale,
                                   const float* __restrict__ scale, unsigned char* __restrict__ out, int rows, int K,
                                   int swap23, int kperm) {
  // 32x32x64 A fragments, 2 KiB each: lane l holds tile row (l & 31), 32 values of k-step s; in memory [lane][j 0..15]
  // then [lane][j 16..31] (S3M<fp8_t>::ldg).  Plain order: k = 64 s + 32 (l >> 5) + j.  kperm (fc2, whose k = the
  // hidden unit): position (s, h = l >> 5, j) holds k = 64 s + 32 (j >> 4) + (j & 7) + 8 h + 16 ((j >> 3) & 1), the
  // order in which s3_fc1's accumulators leave as fragments
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 31), l = (int)((i >> 5) & 63);
  const long fs = i >> 11;
  const int ksteps = K / 64;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  int r = l & 31;
  if (swap23) r = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
  const int h = l >> 5;
  const int row = 32 * tile + r;
  const int k = kperm ? 64 * s + 32 * (j >> 4) + (j & 7) + 8 * h + 16 * ((j >> 3) & 1) : 64 * s + 32 * h + j;
  float v = w[(long)row * K + k] * scale[0];
  if (rowscale != nullptr) v *= rowscale[row];
  out[fs * 2048 + (j >> 4) * 1024 + l * 16 + (j & 15)] =
      (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}

template <typename T, int C, int MB> int launch_tm(const Stage3Args& a, int j, int phase, hipStream_t st) {
  constexpr int NC = MB;
  constexpr int LDS1 = M1 * MB * (C * S3M<T>::ESZ + 16);
  if (phase == 0) {
    auto kern1 = s3_fc1_kernel<T, C, MB>;
    if (LDS1 > 65536) {
      static DevOnce attr1_set;
      if (attr1_set.need()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
        attr1_set.done();
      }
    }
    hipLaunchKernelGGL(kern1, dim3(((a.B + M1 * MB - 1) / (M1 * MB)) * (4 * C / N1)), dim3(NT), LDS1, st, a, j);
  } else {
    auto kern = s3_fc2_kernel<T, C, NC>;
    static DevOnce attr_set;
    if (attr_set.need()) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RED_BYTES * NC));
      attr_set.done();
    }
    hipLaunchKernelGGL(kern, dim3(((a.B + M2 - 1) / M2) * (C / (N2 * NC))), dim3(NT), RED_BYTES * NC, st, a, j);
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// Tile shapes: 32-alert fc1 tiles / 32-channel fc2 tiles; at 640 channels (16-bit modes) 64 alerts / 64 channels -- 160
// workgroups instead of 320, one round of the chip's 256 CUs with twice the rows per filter fragment: 19 -> 16.7 / 14.7 us per
// launch at 1024 alerts.  (Measured and NOT taken at 512 channels for the 2048-alert chunks of a large call, 512 -> 256
// workgroups: 2.255 against 2.265 ms per 8192 alerts in bf16, 2.09 against 2.07 in fp8.  The split mode's fragments are 8
// registers each: narrow tiles only.)
template <typename T, int C> int launch_t(const Stage3Args& a, int j, int phase, hipStream_t st) {
  if constexpr (S3M<T>::ESZ <= 2) {
    static const int forced = [] {   // BTSBOT_AMD_S3_TILES=1 / 2: always narrow / always wide (A/B timing, parity)
      const char* e = getenv("BTSBOT_AMD_S3_TILES");
      return e != nullptr ? atoi(e) : 0;
    }();
    const bool wide = forced == 2 || (forced != 1 && C == 640 && S3M<T>::ESZ == 2);
    if (wide) return launch_tm<T, C, 2>(a, j, phase, st);
  }
  return launch_tm<T, C, 1>(a, j, phase, st);
}

}  // namespace

bool stage3_supported(int prec, int c3, int depth) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16 || prec == BTSBOT_FP8 || prec == BTSBOT_F16X2) && (c3 == 512 || c3 == 640) &&
         depth >= 1 && depth <= S3_MAX_DEPTH;
}

size_t stage3_hfrag_bytes(int prec, int c3, int B) {
  return (size_t)((B + M2 - 1) / M2) * M2 * 4 * c3 * (prec == BTSBOT_F32 || prec == BTSBOT_F16X2 ? 4 : 2);
}

int launch_stage3(int prec, int c3, const Stage3Args& a, int block, int phase, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16 && c3 == 512) return launch_t<bf16_t, 512>(a, block, phase, st);
  if (prec == BTSBOT_F16 && c3 == 512) return launch_t<f16_t, 512>(a, block, phase, st);
  if (prec == BTSBOT_BF16 && c3 == 640) return launch_t<bf16_t, 640>(a, block, phase, st);
  if (prec == BTSBOT_F16 && c3 == 640) return launch_t<f16_t, 640>(a, block, phase, st);
  if (prec == BTSBOT_FP8 && c3 == 512) return launch_t<fp8_t, 512>(a, block, phase, st);
  if (prec == BTSBOT_FP8 && c3 == 640) return launch_t<fp8_t, 640>(a, block, phase, st);
  if (prec == BTSBOT_F16X2 && c3 == 512) return launch_t<f16x2_t, 512>(a, block, phase, st);
  if (prec == BTSBOT_F16X2 && c3 == 640) return launch_t<f16x2_t, 640>(a, block, phase, st);
  btsbot_set_error("stage3: precision %d / width %d not supported", prec, c3);
  return BTSBOT_ERR_INVALID_ARG;
}

int launch_fp8_scale(const float* src, const float* rowscale, long n, int K, float* scale, hipStream_t st) {
  HIP_TRY(hipMemsetAsync(scale, 0, 8, st));
  hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, src, rowscale, reinterpret_cast<unsigned*>(scale), n, K);
  hipLaunchKernelGGL(scale_from_max_kernel, dim3(1), dim3(1), 0, st, scale);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_pack_s3(int prec, const float* src, const float* rowscale, void* dst, int rows, int K, int swap23,
                   float* scale, hipStream_t st) {
  const long total = (long)rows * K;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_s3_kernel<bf16_t>, grid, blk, 0, st, src, rowscale, reinterpret_cast<bf16_t*>(dst), rows, K,
                       swap23);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_s3_kernel<f16_t>, grid, blk, 0, st, src, rowscale, reinterpret_cast<f16_t*>(dst), rows, K,
                       swap23);
  else if (prec == BTSBOT_F16X2)
    hipLaunchKernelGGL(pack_s3_split_kernel, grid, blk, 0, st, src, rowscale, reinterpret_cast<_Float16*>(dst), rows, K,
                       swap23);
  else if (prec == BTSBOT_FP8) {
    if (scale == nullptr) {
      btsbot_set_error("pack_s3: the fp8 mode needs a scale slot");
      return BTSBOT_ERR_INVALID_ARG;
    }
    const int rc = launch_fp8_scale(src, rowscale, total, K, scale, st);
    if (rc != BTSBOT_OK) return rc;
    // (fc1's filter: rows in the bit-2/3-swapped order, k as it is; fc2's: rows as they are, k in the fragment order)
    hipLaunchKernelGGL(pack_s3_fp8_kernel, grid, blk, 0, st, src, rowscale, scale, reinterpret_cast<unsigned char*>(dst),
                       rows, K, swap23, swap23 ? 0 : 1);
  } else {
    btsbot_set_error("pack_s3: precision %d is not a packed-fragment mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
