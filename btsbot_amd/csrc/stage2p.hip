// Stage 2 (3x3 maps, C = 256; convnext_nano: C = 320, template parameter CW below) as ONE persistent launch (gfx950, 16-bit modes):
//
//   depth x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]  ->  LN + conv 2x2 s2 (256 -> 512)
//
// (timm ConvNeXt stages[2].blocks / stages[3].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132).  HBM sees [9][256] f32 in and [512] f32 out per alert;
// the residual stream, the LayerNorm outputs and the 1024-wide hidden activations never leave the CU.
//
// Why this shape.  At the benchmark's 1024 alerts stage 2 has 9216 pixel rows against 1 MB of filters per
// block: too few rows to amortise the filters inside one workgroup, too many launches (12 + 2) to keep them
// short.  What this chip does well is let EVERY CU stream the SAME bytes out of L2: 125 GB/s per CU, 32 TB/s
// chip-wide, straight into registers (tools/unit/l2stream.hip).  So: one 512-thread workgroup per CU keeps
// G = 4 alerts (36 pixel rows, padded to 48 = three 16-column MFMA blocks) resident for the whole stage and
// streams all 6 x 1 MB of filters past them, 8.4 us per block at the measured rate, which is also about
// what the 48-column products cost on the matrix pipe (6 us): the two overlap.
//
//   * filters are packed per MFMA fragment (launch_pack_s2p): a wave's A operand for one k-step is ONE
//     contiguous 1 KiB global_load_dwordx4, no LDS staging, no barrier between load and use.  fc1 chunk of
//     128 hidden units: wave w takes hidden tile 8 ch + w (8 k-steps); fc2: wave w owns output channels
//     32 w .. 32 w + 31 (2 tiles x 4 k-steps of the chunk).  Two register sets: a chunk's fragments are requested
//     one whole chunk ahead.
//   * the residual stream IS the fc2 accumulator: 32 channels x 48 pixels per wave in the 16x16 C/D layout
//     (24 registers), layer scale folded into the packed fc2 filter, gamma * b2 added once per block;
//   * fc1's B operand is the LayerNorm output ([pixel][channel] in LDS, re-read per chunk: the registers it would
//     take are the second fragment set); its result goes through GELU into a double-buffered [pixel][hidden]
//     LDS image that is fc2's B operand: one barrier per chunk;
//   * depthwise + LayerNorm: the map goes to LDS in fp32, thread = (channel, alert pair) applies the central
//     5 x 5 taps (all a 3 x 3 map can touch) in place, wave = pixel normalises (two wave reductions);
//   * the downsample's 4 MB-per-launch GEMM (M = alerts) runs here as 16-column products with 4 live columns:
//     wasteful on the matrix pipe, free in time -- it is the 1 MB filter stream that bounds it.
#include <string.h>
#include <type_traits>

#include "common.h"
#include "stage2p.h"
#include "stage3.h"

namespace {

// operand traits: fragment (8 elements per lane), element size, MFMA, conversions from fp32
template <typename T> struct MP;
// Besides the arithmetic a trait says where a fragment lies: gld = fragment number f of a packed array in HBM (64 lanes
// x 8 values); ld8 / st8 / st4 = 8 / 8 / 4 consecutive values of an operand image in LDS at byte address p of its
// first (for the split mode: only) plane.  PLANE = byte distance of the split mode's remainder plane.
template <typename T> struct MP16 {
  static constexpr int ESZ = 2;
  static constexpr int KSTEP = 32;        // k per MFMA; a lane holds KSTEP / 4 consecutive values of its row / column
  static constexpr bool SPLIT = false;
  typedef T frag __attribute__((ext_vector_type(8)));
  typedef T quad __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ frag gld(const void* base, size_t f, int lane) {
    return reinterpret_cast<const frag*>(base)[f * 64 + lane];
  }
  template <int PLANE> static __device__ __forceinline__ frag ld8(const unsigned char* p) {
    return *reinterpret_cast<const frag*>(p);
  }
  template <int PLANE> static __device__ __forceinline__ void st8(unsigned char* p, const float (&v)[8]) {
    *reinterpret_cast<frag*>(p) = pack8(v);
  }
  template <int PLANE> static __device__ __forceinline__ void st4(unsigned char* p, const float (&v)[4]) {
    *reinterpret_cast<quad*>(p) = pack4(v);
  }
  static __device__ __forceinline__ float roundtrip(float v) { return (float)(T)v; }
  static __device__ __forceinline__ frag pack8(const float (&v)[8]) {
    frag o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
    return o;
  }
  static __device__ __forceinline__ quad pack4(const float (&v)[4]) {
    quad o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
    return o;
  }
};
template <> struct MP<bf16_t> : MP16<bf16_t> {
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MP<f16_t> : MP16<f16_t> {
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
// split operands (BTSBOT_F16X2): value = f16 head + f16 remainder, product = lo*hi + hi*lo + hi*hi on the f16 MFMA.
// LDS images keep the 16-bit geometry for the heads and a second plane of the same geometry for the remainders (so
// the bank mapping of every read is the 16-bit one); a packed fragment in HBM is 2 KiB: heads, then remainders.
template <> struct MP<f16x2_t> {
  static constexpr int ESZ = 2;
  static constexpr int KSTEP = 32;
  static constexpr bool SPLIT = true;
  typedef h2x8 frag;
  typedef h2x4 quad;
  static __device__ __forceinline__ frag gld(const void* base, size_t f, int lane) {
    const f16x8* p = reinterpret_cast<const f16x8*>(base) + f * 128 + lane;
    frag o;
    o.hi = p[0];
    o.lo = p[64];
    return o;
  }
  template <int PLANE> static __device__ __forceinline__ frag ld8(const unsigned char* p) {
    frag o;
    o.hi = *reinterpret_cast<const f16x8*>(p);
    o.lo = *reinterpret_cast<const f16x8*>(p + PLANE);
    return o;
  }
  template <int PLANE> static __device__ __forceinline__ void st8(unsigned char* p, const float (&v)[8]) {
    const frag o = split8(v);
    *reinterpret_cast<f16x8*>(p) = o.hi;
    *reinterpret_cast<f16x8*>(p + PLANE) = o.lo;
  }
  template <int PLANE> static __device__ __forceinline__ void st4(unsigned char* p, const float (&v)[4]) {
    const quad o = split4(v);
    *reinterpret_cast<f16x4v*>(p) = o.hi;
    *reinterpret_cast<f16x4v*>(p + PLANE) = o.lo;
  }
  static __device__ __forceinline__ float roundtrip(float v) { return v; }   // (no training form)
  static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, b.hi, c, 0, 0, 0);
  }
};
// fp8 (OCP e4m3) on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4: 128 k per instruction at twice the bf16
// rate per clock (the non-scaled fp8 MFMA of round 2 runs at the bf16 rate).  A lane holds 32 consecutive k of its row /
// column (lane l: row l & 15, k = 32 (l >> 4) + j; checked with exact integer data, tools/unit/mx_probe.hip).  Both block
// scales are the constant 2^0 (E8M0 127): the mode keeps ONE power-of-two scale per filter, applied in fp32 around the
// products (stage3.hip), and unscaled activations clamped to +-448 (beyond its largest finite value the format has only NaN).
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
template <> struct MP<fp8_t> {
  static constexpr int ESZ = 1;
  static constexpr int KSTEP = 128;
  static constexpr bool SPLIT = false;
  typedef v8i32 frag;
  typedef unsigned quad;
  // a packed fragment in HBM is 2 KiB: [lane][k 0..15 of its 32], then [lane][k 16..31]: two coalesced 1 KiB pieces
  static __device__ __forceinline__ frag gld(const void* base, size_t f, int lane) {
    const v4i32* p = reinterpret_cast<const v4i32*>(base) + f * 128 + lane;
    const v4i32 lo = p[0], hi = p[64];
    return frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  template <int PLANE> static __device__ __forceinline__ frag ld8(const unsigned char* p) {
    const v4i32 lo = *reinterpret_cast<const v4i32*>(p), hi = *reinterpret_cast<const v4i32*>(p + 16);
    return frag{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  static __device__ __forceinline__ float c8(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }
  static __device__ __forceinline__ quad pack4(const float (&v)[4]) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c8(v[0]), c8(v[1]), 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c8(v[2]), c8(v[3]), w, true);
  }
  template <int PLANE> static __device__ __forceinline__ void st8(unsigned char* p, const float (&v)[8]) {
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<uint2*>(p) = make_uint2(pack4(a), pack4(b));
  }
  template <int PLANE> static __device__ __forceinline__ void st4(unsigned char* p, const float (&v)[4]) {
    *reinterpret_cast<quad*>(p) = pack4(v);
  }
  static __device__ __forceinline__ float roundtrip(float v) { return v; }   // (no training form)
  static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  }
};
template <typename T> struct GeluOf2 { using type = T; };
template <> struct GeluOf2<fp8_t> { using type = bf16_t; };   // fp8 rides on the bf16 schedule's GELU

// The stage's width is a template parameter CW: 256 (pico, every mode) or 320 (convnext_nano, 16-bit modes, inference).
// 320 channels are 20 MFMA row tiles for 8 waves: every wave owns two (as at 256) and the four left over are shared by
// wave pairs (w, w + 4), each of which runs half of a chunk's fc2 k-steps on its tile -- the residual of those tiles is the
// SUM of the two waves' accumulators (they meet in the LDS map at every block start).
// G alerts are resident per workgroup: NPX = 9 G pixel rows in NCOL = 16 NB MFMA columns.  G = 4 (36 of 48 columns,
// the form for one batch of 1024: one workgroup per CU); G = 7 (63 of 64 columns) for batches large enough that it takes
// fewer rounds of 256 workgroups: the 1 MB filter stream per block per workgroup, which bounds the kernel, is then
// shared by 7 alerts instead of 4
template <int G> struct Geo {
  static constexpr int NPX = 9 * G, NCOL = (NPX + 15) / 16 * 16, NB = NCOL / 16;
};
constexpr int NT = 512, NW = NT / 64;                 // 8 waves: 2 per SIMD
constexpr int CHUNK = 128;                            // hidden units per fc1 / fc2 step
// (k-steps of fc1 / of fc2 per chunk: C / KSTEP and CHUNK / KSTEP of the operand mode: 8 and 4, fp8: 2 and 1)
template <int CW> struct Shp {
  static constexpr int C = CW, HID = 4 * CW, NCHUNK = HID / CHUNK;
  static constexpr int CO = 2 * CW, KD = 4 * CW, KSD = KD / 32;   // downsample: 512 outputs, K = 1024 (640, 1280)
  static constexpr int XLP = CW;                                   // fp32 map: floats per pixel row
  static constexpr int NTILE = CW / 16, MF = NTILE / NW, NX = NTILE - MF * NW;   // row tiles; per wave; shared by wave pairs
  static_assert(NX == 0 || 2 * NX == NW, "the tiles left over are shared by wave pairs");
  // 16-bit LN image: bytes per pixel row, 32 (mod 256) -- see below (544; 800)
  static constexpr int XNP2 = CW == 256 ? 544 : 800;
  static_assert(XNP2 % 256 == 32 && XNP2 >= 2 * CW + 32, "LN image pitch");
};
// Operand images [pixel][k]: a ds_read_b128 is served in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,
// 28-31} and the same + 32 -- each of which must cover the 16 slots of a 256-byte bank row.  A group holds every
// pixel column once, at two neighbouring k-groups; with rows 32 bytes (mod 256) apart the slot is 2 col + kg: a
// bijection.  (Rows 16 bytes apart, the classic padding, made every group 2-way on one slot: half of the kernel's LDS
// cycles were conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50.)  fp8 reads 8 bytes per lane in two groups
// of 32 lanes: 16 bytes apart is the bijection there.
constexpr int HP2 = CHUNK * 2 + 32;                   // hidden image: bytes per pixel row (288)
constexpr int OFF_XL = 0;                             // [NCOL][256] f32 (rows >= NPX stay zero)
template <typename T, int G, int CW = 256> struct Lds {
  static constexpr int NCOL = Geo<G>::NCOL;
  static constexpr int XLP = Shp<CW>::XLP, XNP2 = Shp<CW>::XNP2, HID = Shp<CW>::HID;
  static constexpr int OFF_XN = OFF_XL + NCOL * XLP * 4;      // G = 4: 49152
  static constexpr int XN_PLANE = NCOL * XNP2, H_PLANE = NCOL * HP2;   // 26112, 13824 (the split mode has two planes of each)
  static constexpr int NPL = MP<T>::SPLIT ? 2 : 1;
  static constexpr int H_IMG = NPL * H_PLANE;                 // one hidden image (two of them)
  static constexpr int OFF_H = OFF_XN + NPL * XN_PLANE;
  static constexpr int OFF_B1 = OFF_H + 2 * H_IMG;            // fc1 bias [1024] f32
  static constexpr int BYTES = OFF_B1 + HID * 4;              // G = 4: 104704 (split: 160768); G = 7: 141312
  // training forward: two more hidden-sized images (the rounded fc1 pre-activation of a chunk, kept for the backward)
  static constexpr int OFF_A = BYTES;
  static constexpr int BYTES_TRAIN = BYTES + 2 * H_IMG;
};
static_assert(Lds<f16x2_t, 4>::BYTES <= 160 * 1024 && Lds<bf16_t, 7>::BYTES <= 160 * 1024, "the images fit one CU");
constexpr float LN_EPS = 1e-6f;
#ifndef XRES_TRAIN
#define XRES_TRAIN 1
#endif
#define S2P_STAMP(i)                                                                      \
  do {                                                                                    \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)

// Kept rows of the training forward leave through buffer stores: the base is a scalar descriptor, the lane's part a
// 32-bit offset, the chunk's part a scalar offset -- no per-lane 64-bit pointers (six of them, strength-reduced over the
// chunk loop, were 24 registers the loop does not have)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t keep_rsrc(void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
}
// cache policy of the kept rows' stores (the aux immediate of buffer_store: 1 sc0, 2 nt, 16 sc1).  Measured at 1024
// alerts: 227 us default, 227 nt, 220 sc0 | nt | sc1, 222 sc0 | sc1 -- no policy matters; the default stays
#ifndef KEEP_AUX
#define KEEP_AUX 0
#endif


template <typename Q> __device__ __forceinline__ void keep_st8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, Q q) {
  static_assert(sizeof(Q) == 8, "a quad of 16-bit values");
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, q), r, (int)voff, (int)soff, KEEP_AUX);
}
template <typename Q> __device__ __forceinline__ void keep_st16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, Q q) {
  static_assert(sizeof(Q) == 16, "four fp32 values / a 16-byte piece");
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, q), r, (int)voff, (int)soff, KEEP_AUX);
}

// sum over the 32 lanes of a half wave (lanes 0-31 / 32-63 separately); every lane ends with its half's total
__device__ __forceinline__ float half_sum(float v) {
  v = group16_sum(v);
  float w = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return v + w;
}

// TRAIN: 0 inference; 1 the training forward's keeping form: per block the input map (fp32), the LayerNorm output, the
// ROUNDED fc1 pre-activation and its GELU (operand type) -- what s2mlp_bwd / dw3ln_bwd / the filter-gradient GEMMs read;
// the depthwise output is not kept (dw3ln_bwd_kernel recomputes it from the input) -- plus the stage output and the
// downsample's patch rows.  Every kept row leaves by an UNCONDITIONAL store: rows of alerts beyond the batch (a ragged
// last workgroup) and the pad columns 9 G .. 16 NB - 1 go to padding the caller provides behind the live rows (buffers
// hold (B + 9) alerts' rows; pad columns land on the rows of alerts B + 7, B + 8) -- with conditional stores or copy loops of
// data-dependent length in the chunk loop hipcc's counted vmcnt waits for the filter fragments degrade to vmcnt(0):
// rounds 4-5 measured 252-259 us for a keeping form written that way (and + 57 us for one that kept the block inputs
// only) against 100 us for the inference form.  Stores are buffer stores (scalar descriptor + 32-bit lane offset + scalar
// chunk offset): per-lane 64-bit pointers, strength-reduced over the chunk loop, cost the loop 24 registers it does not
// have, and a spilled value's reload is again a vmcnt(0) in front of the fragments.
// ROWS > 0: the row-tile form -- a workgroup owns ROWS consecutive rows of a [rows][CW] fp32 stream instead of G alerts' 3x3
// maps, every block is  x += W2 gelu(W1 LN(x) + b1) + b2  (no depthwise phase, no downsample): the MLP half of a MaxViT
// partition-attention layer at 256 channels (timm PartitionAttentionCl.mlp behind norm2, reached from
// /root/reference/btsbot/architectures.py:51,97), updated in place (x_in == tap_stage); a.B counts rows.
template <typename T, int G, int TRAIN = 0, int CW = 256, int ROWS = 0>
__global__ __launch_bounds__(NT, 2) void stage2p_kernel(Stage2pArgs a) {
  using SH = Shp<CW>;
  constexpr int C = SH::C, HID = SH::HID, NCHUNK = SH::NCHUNK, CO = SH::CO, KSD = SH::KSD, XLP = SH::XLP;
  constexpr int MF = SH::MF, NX = SH::NX;
  static_assert(NX == 0 || (TRAIN == 0 && !MP<T>::SPLIT && MP<T>::ESZ == 2), "320 channels: 16-bit inference only");
  using LD = Lds<T, G, CW>;
  constexpr bool RM = ROWS > 0;
  static_assert(!RM || (TRAIN == 0 && ROWS == Geo<G>::NCOL && CW == 256), "row-tile form: inference, whole column blocks");
  constexpr int NPX = RM ? ROWS : Geo<G>::NPX, NCOL = Geo<G>::NCOL, NB = Geo<G>::NB;
  constexpr int KSTEP = MP<T>::KSTEP, VPL = KSTEP / 4, KS1 = C / KSTEP, KS2 = CHUNK / KSTEP, KH = HID / KSTEP;
  constexpr int OFF_XN = LD::OFF_XN, XN_PLANE = LD::XN_PLANE, H_PLANE = LD::H_PLANE;
  using frag = typename MP<T>::frag;
  constexpr int ESZ = MP<T>::ESZ;
  constexpr bool F8 = std::is_same<T, fp8_t>::value;
  // operand images: bytes per pixel row (the regions keep their 16-bit sizes)
  constexpr int XNP = SH::XNP2 * ESZ / 2, HP = CHUNK * ESZ + 16 * ESZ;
  // fc1's B operand (the block's LN image, the same for all 8 chunks): k-steps kept in registers for the whole block;
  // every wave re-reading it from LDS per chunk was 2/3 of the kernel's LDS traffic (196 of 288 KB per chunk)
  // (split: a fragment is 8 registers and the filter streams take 128 of them: nothing stays resident)
  // (16-bit, 3 column blocks: 5 or 6 spill in the block prologue and lose more than they save; 4 column blocks: 1 -- 2 spill)
  // (training forward: ONE k-step -- the copy-out of the kept rows needs the registers: with two the chunk loop spills 5-8
  //  of them, with four 33, and a spilled value's reload puts a vmcnt(0) in front of the filter fragments' counted waits)
  constexpr int XRES = MP<T>::SPLIT ? 0 : NB > 3 ? 1 : NX > 0 ? 0 : F8 ? KS1 : TRAIN ? XRES_TRAIN : 4;
  constexpr int H_IMG = LD::H_IMG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xl = reinterpret_cast<float*>(smem + OFF_XL);
  unsigned char* xn = smem + OFF_XN;
  unsigned char* hb = smem + LD::OFF_H;   // two hidden images, H_IMG bytes apart
  float* b1s = reinterpret_cast<float*>(smem + LD::OFF_B1);
  unsigned char* ab = smem + LD::OFF_A;   // (training forward) two pre-activation images, H_IMG bytes apart

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int alert0 = blockIdx.x * G;
  const size_t row0 = RM ? (size_t)blockIdx.x * ROWS : (size_t)alert0 * 9;   // first row of this workgroup in x_in / tap_stage
  const int nlive = RM ? (int)min((long)ROWS, (long)a.B - (long)row0) : min(G, a.B - alert0) * 9;   // its live rows
  // 320 channels: the tile this wave shares with wave (wave ^ 4), its first channel for this lane, the half of a chunk's
  // fc2 k-steps it runs on it; xlead = the wave of the pair that carries the tile's input, bias and layer-scale terms
  constexpr int KX = KS2 / 2;
  const int xtile = MF * NW + (NX > 0 ? wave & (NX - 1) : 0), cx0 = 16 * xtile + 4 * kg, xh = NX > 0 ? wave / NX : 0;
  const bool xlead = xh == 0;
  S2P_STAMP(0);
  // pad rows of the operand images: zero once (they are never written again)
  for (int i = tid; i < (NCOL - NPX) * XNP / 4; i += NT) {
    reinterpret_cast<unsigned*>(xn + NPX * XNP)[i] = 0u;
    if (MP<T>::SPLIT) reinterpret_cast<unsigned*>(xn + XN_PLANE + NPX * XNP)[i] = 0u;
  }
  for (int i = tid; i < NCOL * XLP; i += NT) xl[i] = 0.f;

  // ---- residual stream: this wave's 32 channels x 48 pixels, acc[m][n][r] = x[16 n + col][32 wave + 16 m + 4 kg + r]
  f32x4 acc[MF][NB], accx[NB];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int p = 16 * n + col;
      acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p < nlive)
        acc[m][n] = *reinterpret_cast<const f32x4*>(a.x_in + (row0 + p) * C + 16 * (MF * wave + m) + 4 * kg);
    }
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int p = 16 * n + col;
    accx[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (NX > 0 && xlead && p < nlive) accx[n] = *reinterpret_cast<const f32x4*>(a.x_in + (row0 + p) * C + cx0);
  }
  __syncthreads();   // zero fill done before the first map is written

  // this wave's fragment streams: fc1 tile (8 ch + wave) of a chunk: 8 KiB contiguous; fc2 tiles 2 wave, 2 wave + 1:
  // 2 x 4 KiB.  ONE register set, refilled in place: the slot a product has just read is requested again at once with
  // the same k-step of the next chunk, which arrives a whole chunk (~3k cycles) before it is needed (the loads of all
  // 256 CUs hit the same L2 lines at about the same time: ~1.1k cycles of transfer per 64 KB burst per CU on top of
  // the L2 latency).  The registers a second set would take hold the fc1 B operand instead (below).
  frag a1[KS1], a2[MF][KS2], a2x[KX > 0 ? KX : 1];
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // (split: the two streams are 128 registers; they are NOT carried through the depthwise / LayerNorm phases -- every
  //  block requests its chunk 0 at the start of its chunk loop instead, one exposed L2 round trip per block)
  constexpr bool CARRY = !MP<T>::SPLIT;
  if (a.depth > 0 && CARRY) {   // chunk 0 of both filters (fc2 runs one step behind fc1: its chunk 0 is first used in step 1)
#pragma unroll
    for (int s = 0; s < KS1; ++s) a1[s] = MP<T>::gld(a.blk[0].w1p, (size_t)wave * KS1 + s, lane);
#pragma unroll
    for (int m = 0; m < MF; ++m) {
#pragma unroll
      for (int s = 0; s < KS2; ++s) a2[m][s] = MP<T>::gld(a.blk[0].w2p, (size_t)(MF * wave + m) * KH + s, lane);
    }
    if (NX > 0) {
#pragma unroll
      for (int s = 0; s < KX; ++s) a2x[s] = MP<T>::gld(a.blk[0].w2p, (size_t)xtile * KH + xh * KX + s, lane);
    }
  }

#pragma unroll 1
  for (int j = 0; j < a.depth; ++j) {
    const Stage2pBlk& bk = a.blk[j];
    S2P_STAMP(1 + 8 * j);
    // ---- every small parameter of the block is requested first: the L2 latency passes under the phases below
    // (320 channels: the layer-scale / bias / LayerNorm terms wait until the depthwise phase is over -- 44 registers that
    //  otherwise pushed filter fragments still in flight out to scratch, whose stores then waited for them: 5k cycles per block)
    f32x4 g4[MF], b4[MF], gx4 = f32x4{0.f, 0.f, 0.f, 0.f}, bx4 = gx4;
    constexpr int CXL = C > 256 ? 256 : 0;   // (320 channels: a lane's 9th and 10th channel are 256 + 2 (lane & 31) + {0, 1})
    float2 lwx = make_float2(0.f, 0.f), lbx = lwx;
    // (the training forward too: its kept rows' stores take the registers at this point, and a spilled value's reload is a
    //  vmcnt(0) in front of the filter fragments still in flight)
    constexpr bool LATE = NX > 0 || TRAIN != 0;
    if constexpr (!LATE) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        g4[m] = *reinterpret_cast<const f32x4*>(bk.gamma + 16 * (MF * wave + m) + 4 * kg);
        b4[m] = *reinterpret_cast<const f32x4*>(bk.b2 + 16 * (MF * wave + m) + 4 * kg);
      }
    }
    constexpr int NB1 = (HID + NT - 1) / NT;
    float b1r[NB1];
#pragma unroll
    for (int i = 0; i < NB1; ++i) b1r[i] = bk.b1[tid + i * NT < HID ? tid + i * NT : 0];
    // depthwise role: 256 channels: (channel, alert pair); 320: thread = channel, every alert of the workgroup
    const int dc = C == 256 ? tid & 255 : tid < C ? tid : 0, dhalf = C == 256 ? tid >> 8 : 0;
    float w[25];
    float dbias = 0.f;
    if constexpr (!RM) {
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) w[ky * 5 + kx] = bk.dw_w[((ky + 1) * 7 + kx + 1) * C + dc];
      dbias = bk.dw_b[dc];
    }
    f32x4 lw, lw2, lb, lb2;
    if constexpr (!LATE) {
      lw = *reinterpret_cast<const f32x4*>(bk.ln_w + 8 * (lane & 31));
      lw2 = *reinterpret_cast<const f32x4*>(bk.ln_w + 8 * (lane & 31) + 4);
      lb = *reinterpret_cast<const f32x4*>(bk.ln_b + 8 * (lane & 31));
      lb2 = *reinterpret_cast<const f32x4*>(bk.ln_b + 8 * (lane & 31) + 4);
    }
    // ---- the map to LDS in fp32
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int c0 = 16 * (MF * wave + m) + 4 * kg;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int p = 16 * n + col;
        if (p < NPX) *reinterpret_cast<f32x4*>(xl + p * XLP + c0) = acc[m][n];
        // (training forward: the block's input rows, straight from the accumulators; pad columns: the rows of alerts B + 7,
        //  B + 8 -- the caller's padding.  A coalesced copy out of the LDS map behind the barrier measured no faster and was
        //  not reproducible from pass to pass -- 3 of 24 identical passes differed in one block's gradients; not understood)
        if constexpr (TRAIN != 0)
          keep_st16(keep_rsrc(a.keep[j].xin), (unsigned)(((p < NPX ? alert0 * 9 + p : (a.B + 7) * 9 + (p - NPX)) * C + c0) * 4), 0u, acc[m][n]);
      }
    }
    if (NX > 0) {   // the shared tiles: the pair's two partial residuals, one after the other
      if (xlead) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
          if (16 * n + col < NPX) *reinterpret_cast<f32x4*>(xl + (16 * n + col) * XLP + cx0) = accx[n];
      }
      __syncthreads();
      if (!xlead) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
          if (16 * n + col < NPX) *reinterpret_cast<f32x4*>(xl + (16 * n + col) * XLP + cx0) += accx[n];
      }
    }
    __syncthreads();
    S2P_STAMP(2 + 8 * j);
    // ---- depthwise 7x7 on the 3x3 maps, in place: thread = (channel, alert pair)
    if constexpr (!RM) {
      constexpr int AH = C == 256 ? (G + 1) / 2 : G;   // alerts per half of the workgroup (odd G: the second half has one fewer)
#pragma unroll
      for (int g = 0; g < AH; ++g) {
        if (dhalf * AH + g >= G || (C != 256 && tid >= C)) continue;
        float* px = xl + (size_t)((dhalf * AH + g) * 9) * XLP + dc;
        float in[9], o[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) in[i] = px[i * XLP];
#pragma unroll
        for (int oy = 0; oy < 3; ++oy)
#pragma unroll
          for (int ox = 0; ox < 3; ++ox) {
            float sum = dbias;
#pragma unroll
            for (int iy = 0; iy < 3; ++iy)
#pragma unroll
              for (int ix = 0; ix < 3; ++ix) sum = fmaf(w[(iy - oy + 2) * 5 + (ix - ox + 2)], in[iy * 3 + ix], sum);
            o[oy * 3 + ox] = sum;
          }
#pragma unroll
        for (int i = 0; i < 9; ++i) px[i * XLP] = o[i];
      }
    }
    if constexpr (LATE) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        g4[m] = *reinterpret_cast<const f32x4*>(bk.gamma + 16 * (MF * wave + m) + 4 * kg);
        b4[m] = *reinterpret_cast<const f32x4*>(bk.b2 + 16 * (MF * wave + m) + 4 * kg);
      }
      lw = *reinterpret_cast<const f32x4*>(bk.ln_w + 8 * (lane & 31));
      lw2 = *reinterpret_cast<const f32x4*>(bk.ln_w + 8 * (lane & 31) + 4);
      lb = *reinterpret_cast<const f32x4*>(bk.ln_b + 8 * (lane & 31));
      lb2 = *reinterpret_cast<const f32x4*>(bk.ln_b + 8 * (lane & 31) + 4);
      if constexpr (NX > 0) {
        gx4 = *reinterpret_cast<const f32x4*>(bk.gamma + cx0);
        bx4 = *reinterpret_cast<const f32x4*>(bk.b2 + cx0);
        lwx = *reinterpret_cast<const float2*>(bk.ln_w + CXL + 2 * (lane & 31));
        lbx = *reinterpret_cast<const float2*>(bk.ln_b + CXL + 2 * (lane & 31));
      }
    }
#pragma unroll
    for (int i = 0; i < NB1; ++i)   // (last read by the previous block's last fc1, two barriers ago)
      if (tid + i * NT < HID) b1s[tid + i * NT] = b1r[i];
    __syncthreads();
    S2P_STAMP(3 + 8 * j);
    // ---- LayerNorm over the 256 channels of a pixel: half wave = pixel, lane = 8 channels
    for (int p = 2 * wave + (lane >> 5); p < NPX; p += 2 * NW) {
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(xl + p * XLP + 8 * (lane & 31));
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(xl + p * XLP + 8 * (lane & 31) + 4);
      float2 dx = make_float2(0.f, 0.f);
      if (C > 256) dx = *reinterpret_cast<const float2*>(xl + p * XLP + CXL + 2 * (lane & 31));
      const float mean = half_sum(d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3] + dx.x + dx.y) * (1.0f / C);
      const f32x4 e0 = d0 - mean, e1 = d1 - mean;
      const float ex0 = C > 256 ? dx.x - mean : 0.f, ex1 = C > 256 ? dx.y - mean : 0.f;
      const float var = half_sum(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2] + e0[3] * e0[3] + e1[0] * e1[0] +
                                 e1[1] * e1[1] + e1[2] * e1[2] + e1[3] * e1[3] + ex0 * ex0 + ex1 * ex1) * (1.0f / C);
      const float rstd = rsqrtf(var + LN_EPS);
      float y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] = e0[i] * rstd * lw[i] + lb[i];
        y[4 + i] = e1[i] * rstd * lw2[i] + lb2[i];
      }
      MP<T>::template st8<XN_PLANE>(xn + p * XNP + 8 * ESZ * (lane & 31), y);
      if constexpr (C > 256) {
        typedef T pair_t __attribute__((ext_vector_type(2)));
        pair_t yx;
        yx[0] = (T)(ex0 * rstd * lwx.x + lbx.x);
        yx[1] = (T)(ex1 * rstd * lwx.y + lbx.y);
        *reinterpret_cast<pair_t*>(xn + p * XNP + (CXL + 2 * (lane & 31)) * ESZ) = yx;
      }
    }
    __syncthreads();
    S2P_STAMP(4 + 8 * j);
    if (TRAIN) {   // the LayerNorm output rows, 16-byte pieces (C * ESZ / 16 per row): a fixed number of trips per thread,
                   // the last pieces written twice rather than under a condition (rows of absent alerts: the padding)
      const __amdgpu_buffer_rsrc_t rx = keep_rsrc(a.keep[j].xn);
      constexpr int PPR = C * ESZ / 16, NPC = NPX * PPR;
#pragma unroll
      for (int it = 0; it < (NPC + NT - 1) / NT; ++it) {
        const int i = min(tid + it * NT, NPC - 1);
        const int p = i / PPR, c = i - p * PPR;
        keep_st16(rx, (unsigned)(p * C * ESZ + 16 * c), (unsigned)(alert0 * 9 * C * ESZ), *reinterpret_cast<const uint4*>(xn + p * XNP + 16 * c));
      }
    }
    frag xr[XRES > 0 ? XRES : 1][NB];
#pragma unroll
    for (int s = 0; s < XRES; ++s)
#pragma unroll
      for (int n = 0; n < NB; ++n)
        xr[s][n] = MP<T>::template ld8<XN_PLANE>(xn + (16 * n + col) * XNP + (KSTEP * s + VPL * kg) * ESZ);
    // residual + gamma * b2 (the bias of the folded fc2)
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] += g4[m] * b4[m];
    if (NX > 0 && xlead) {
#pragma unroll
      for (int n = 0; n < NB; ++n) accx[n] += gx4 * bx4;
    }
    // fp8: gamma * W2 was packed times the power of two S2, so the residual rides through the chunk loop times S2
    // (exact) and comes back times 1/S2 behind the block's last fc2
    const float s1 = F8 ? bk.scales[0] : 1.0f, is1 = F8 ? bk.scales[1] : 1.0f;
    if (F8) {
      const float s2 = bk.scales[2];
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] *= s2;
    }

    // One step = fc1 of chunk ch, then GELU of chunk ch BETWEEN the fc2 products of chunk ch - 1 (software pipeline
    // inside the wave: the two waves of a SIMD meet at the same barrier every chunk, so they are in the same phase
    // all the time and nothing overlaps unless a wave overlaps its own VALU and MFMA work; with fc1 -> GELU -> barrier
    // -> fc2 in line a chunk took 3.35k cycles against 1.5k of MFMA work and a 2.3k filter-streaming floor).
    // P = which of the two hidden images this chunk writes (the other one holds chunk ch - 1's).
    // The next chunk's 16 fragment loads are spread over the k-steps: issued in one burst they hold every wave at the
    // CU's one vector-memory port for ~2k cycles before its first product.
    const __amdgpu_buffer_rsrc_t ra = keep_rsrc(TRAIN ? a.keep[j].a : nullptr), rh = keep_rsrc(TRAIN ? a.keep[j].hh : nullptr);
    // training forward: chunk kc's two images (rounded pre-activation, GELU) -> columns kc of the kept [row][1024] arrays,
    // 16-byte pieces, 16 per row: a wave instruction writes four whole 256-byte row pieces.  (Quads straight from the
    // registers -- lane = (row, 4 hidden units), 64 separate 8-byte writes per instruction -- took the kernel from 100 to
    // 279 us.)  A fixed number of trips per thread, the last pieces written twice; rows of absent alerts: the padding
    auto keep_chunk = [&](int kc, int img) {
      constexpr int PPR = CHUNK * ESZ / 16, NPC = NPX * PPR, TRIPS = (NPC + NT - 1) / NT;
      const unsigned so = (unsigned)((alert0 * 9 * HID + kc * CHUNK) * ESZ);
#pragma unroll
      for (int it = 0; it < TRIPS; ++it) {   // (one trip's two pieces at a time: eight registers, not sixteen)
        const int i = min(tid + it * NT, NPC - 1);
        const int pr = i / PPR, c = i - pr * PPR;
        const uint4 va = *reinterpret_cast<const uint4*>(ab + img * H_IMG + pr * HP + 16 * c);
        const uint4 vh = *reinterpret_cast<const uint4*>(hb + img * H_IMG + pr * HP + 16 * c);
        const unsigned vo = (unsigned)(pr * HID * ESZ + 16 * c);
        keep_st16(ra, vo, so, va);
        keep_st16(rh, vo, so, vh);
      }
    };
    auto step = [&](auto P, auto FIRST, int ch) {
      constexpr int p = decltype(P)::value;
      constexpr bool first = decltype(FIRST)::value;

      // (after the stage's last chunk the fc1 loads below re-read this block's first chunk: an unconditional load
      //  keeps the k-step loops free of branches -- hipcc waits vmcnt(0) behind every conditional load)
      const Stage2pBlk& nb = ch + 1 < NCHUNK ? bk : a.blk[j + 1 < a.depth ? j + 1 : j];
      const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;
      const void* src1 = nb.w1p;
      const size_t f1 = (size_t)(nch * NW + wave) * KS1;
      // fc2 runs one step behind: its slots are refilled with THIS chunk's fragments (used in the next step)
      const void* src2 = bk.w2p;
      const size_t f2 = (size_t)(MF * wave) * KH + ch * KS2;
      const size_t fx = (size_t)xtile * KH + ch * KS2 + xh * KX;   // (320 channels) this wave's k-steps of the shared tile
      // fc1: hidden tile (8 ch + wave) x 48 pixels, bias in the accumulator; B = [k = channel][n = pixel]: k-steps
      // 0 .. XRES-1 from the registers filled after the LayerNorm, the rest from LDS
      f32x4 hacc[NB];
      {
        f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + ch * CHUNK + 16 * wave + 4 * kg);
        if (F8) bv *= s1;   // (the filter was packed times S1: the sum leaves times 1/S1 in front of the GELU)
#pragma unroll
        for (int n = 0; n < NB; ++n) hacc[n] = bv;
      }
      // (split: B fragments are 8 registers each and the filter streams hold 128: ONE buffer, read at the start of its
      //  k-step -- nine products per k-step and the SIMD's other wave cover the LDS latency)
      constexpr int DB = MP<T>::SPLIT ? 1 : 2;
      frag xb[DB][NB];
      if (XRES < KS1 && DB == 2) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
          xb[XRES & 1][n] = MP<T>::template ld8<XN_PLANE>(xn + (16 * n + col) * XNP + (KSTEP * XRES + VPL * kg) * ESZ);
      }
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        if (DB == 2 && s >= XRES && s + 1 < KS1) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            xb[(s + 1) & 1][n] = MP<T>::template ld8<XN_PLANE>(xn + (16 * n + col) * XNP + (KSTEP * (s + 1) + VPL * kg) * ESZ);
        }
        if (DB == 1 && s >= XRES) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            xb[0][n] = MP<T>::template ld8<XN_PLANE>(xn + (16 * n + col) * XNP + (KSTEP * s + VPL * kg) * ESZ);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) hacc[n] = MP<T>::run(a1[s], s < XRES ? xr[s < XRES ? s : 0][n] : xb[s & (DB - 1)][n], hacc[n]);
        a1[s] = MP<T>::gld(src1, f1 + s, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (training forward) chunk ch - 1's images are complete and live through this step: their rows leave behind the
      // fc1 products' issue, under their execution
      if constexpr (TRAIN == 1 && !first) keep_chunk(ch - 1, 1 - p);
      // fc2 of the previous chunk: out channels 32 wave .. + 31, K = its 128 hidden units (image hb[1 - p]), into the
      // residual; between its k-steps GELU of this chunk -> image hb[p] [pixel][hidden]; rows 4 kg .. + 3 of tile `wave`
      unsigned char* hcur = hb + p * H_IMG;
      const unsigned char* hprev = hb + (1 - p) * H_IMG;
      frag hbf[DB][NB];
      if (!first && DB == 2) {
#pragma unroll
        for (int n = 0; n < NB; ++n) hbf[0][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (VPL * kg) * ESZ);
      }
      // one GELU column block per fc2 k-step; a mode with fewer k-steps than column blocks (fp8: one k-step of 128)
      // runs the remaining blocks behind the last product
      constexpr int NS2 = KS2 > NB ? KS2 : NB;
#pragma unroll
      for (int s = 0; s < NS2; ++s) {
        if (s >= KS2) {
        } else if (!first) {
          if (DB == 2 && s + 1 < KS2) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
              hbf[(s + 1) & 1][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * (s + 1) + VPL * kg) * ESZ);
          }
          if (DB == 1) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
              hbf[0][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * s + VPL * kg) * ESZ);
          }
          frag hbx[NB];
          if (NX > 0 && s < KX) {   // the shared tile: this wave's k-step xh KX + s of the chunk, in program step s
#pragma unroll
            for (int n = 0; n < NB; ++n)
              hbx[n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * (xh * KX + s) + VPL * kg) * ESZ);
          }
#pragma unroll
          for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = MP<T>::run(a2[m][s], hbf[s & (DB - 1)][n], acc[m][n]);
#pragma unroll
          for (int m = 0; m < MF; ++m) a2[m][s] = MP<T>::gld(src2, f2 + m * KH + s, lane);
          if (NX > 0 && s < KX) {
#pragma unroll
            for (int n = 0; n < NB; ++n) accx[n] = MP<T>::run(a2x[s], hbx[n], accx[n]);
            a2x[s] = MP<T>::gld(src2, fx + s, lane);
          }
        } else if (!CARRY) {   // (first step of a block: this chunk's fc2 fragments, used in the next step)
#pragma unroll
          for (int m = 0; m < MF; ++m) a2[m][s] = MP<T>::gld(src2, f2 + m * KH + s, lane);
        }
        if (s < NB) {
          float hv[4];
          if constexpr (TRAIN == 1) {
            // the backward differentiates GELU at the ROUNDED pre-activation (as gemm2.hip's GELU_SAVE epilogue does)
            float av[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              av[r] = MP<T>::roundtrip(hacc[s][r]);
              hv[r] = gelu_for<typename GeluOf2<T>::type>(av[r]);
            }
            // (the rounded pre-activation into its own image; both images leave as whole 256-byte row pieces one step later)
            MP<T>::template st4<H_PLANE>(ab + p * H_IMG + (16 * s + col) * HP + (16 * wave + 4 * kg) * ESZ, av);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = gelu_for<typename GeluOf2<T>::type>(F8 ? hacc[s][r] * is1 : hacc[s][r]);
          }
          MP<T>::template st4<H_PLANE>(hcur + (16 * s + col) * HP + (16 * wave + 4 * kg) * ESZ, hv);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();   // image hb[p] complete; hb[1 - p] is read out
    };
    // the fc2 half of a step of its own: the block's last chunk (the residual must be complete before the next
    // block's depthwise phase); it refills the a2 slots it empties with the next block's chunk 0
    auto fc2_tail = [&]() {
      if constexpr (TRAIN == 1) keep_chunk(NCHUNK - 1, 1);
      const Stage2pBlk& nb = a.blk[j + 1 < a.depth ? j + 1 : j];
      const void* src2 = nb.w2p;
      const size_t f2 = (size_t)(MF * wave) * KH;
      const size_t fx = (size_t)xtile * KH + xh * KX;
      const unsigned char* hprev = hb + 1 * H_IMG;   // chunk NCHUNK - 1 is odd: image 1
      constexpr int DB = MP<T>::SPLIT ? 1 : 2;
      frag hbf[DB][NB];
      if (DB == 2) {
#pragma unroll
        for (int n = 0; n < NB; ++n) hbf[0][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (VPL * kg) * ESZ);
      }
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        if (DB == 2 && s + 1 < KS2) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            hbf[(s + 1) & 1][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * (s + 1) + VPL * kg) * ESZ);
        }
        if (DB == 1) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            hbf[0][n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * s + VPL * kg) * ESZ);
        }
        frag hbx[NB];
        if (NX > 0 && s < KX) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            hbx[n] = MP<T>::template ld8<H_PLANE>(hprev + (16 * n + col) * HP + (KSTEP * (xh * KX + s) + VPL * kg) * ESZ);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = MP<T>::run(a2[m][s], hbf[s & (DB - 1)][n], acc[m][n]);
        if (CARRY) {
#pragma unroll
          for (int m = 0; m < MF; ++m) a2[m][s] = MP<T>::gld(src2, f2 + m * KH + s, lane);
        }
        if (NX > 0 && s < KX) {
#pragma unroll
          for (int n = 0; n < NB; ++n) accx[n] = MP<T>::run(a2x[s], hbx[n], accx[n]);
          a2x[s] = MP<T>::gld(src2, fx + s, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(NCHUNK % 2 == 0, "the last chunk writes hidden image 1");
    if (!CARRY) {
#pragma unroll
      for (int s = 0; s < KS1; ++s) a1[s] = MP<T>::gld(bk.w1p, (size_t)wave * KS1 + s, lane);
    }
    step(P0{}, std::true_type{}, 0);
    step(P1{}, std::false_type{}, 1);
    S2P_STAMP(5 + 8 * j);
    // (no stamp inside this loop: its store sits under a branch on the thread id, and behind divergent control flow
    //  hipcc turned the loop's counted vmcnt waits into vmcnt(0))
#pragma unroll 1
    for (int ch = 2; ch < NCHUNK; ch += 2) {
      step(P0{}, std::false_type{}, ch);
      step(P1{}, std::false_type{}, ch + 1);
    }
    fc2_tail();
    if (F8) {
      const float is2 = bk.scales[3];
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] *= is2;
    }
    S2P_STAMP(7 + 8 * j);
  }

  S2P_STAMP(56);
  // ---- stage output (validation copy), then the downsample: LN per pixel + conv 2x2 s2 on pixels 0, 1, 3, 4
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int c0 = 16 * (MF * wave + m) + 4 * kg;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int p = 16 * n + col;
      if (p < NPX) *reinterpret_cast<f32x4*>(xl + p * XLP + c0) = acc[m][n];
      if (a.tap_stage != nullptr && p < nlive)
        *reinterpret_cast<f32x4*>(a.tap_stage + (row0 + p) * C + c0) = acc[m][n];
    }
  }
  if constexpr (RM) return;   // (the row-tile form ends with its rows written back)
  if (NX > 0) {   // the shared tiles: sum of the pair's partial residuals (the validation copy leaves from the map)
    if (xlead) {
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (16 * n + col < NPX) *reinterpret_cast<f32x4*>(xl + (16 * n + col) * XLP + cx0) = accx[n];
    }
    __syncthreads();
    if (!xlead) {
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int p = 16 * n + col;
        if (p < NPX) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xl + p * XLP + cx0) + accx[n];
          *reinterpret_cast<f32x4*>(xl + p * XLP + cx0) = v;
          if (a.tap_stage != nullptr && p < nlive) *reinterpret_cast<f32x4*>(a.tap_stage + (row0 + p) * C + cx0) = v;
        }
      }
    }
  }
  __syncthreads();
  // (the downsample keeps 16-bit operands in the fp8 mode too: it is 5 % of the stage's FLOPs and its LN'd input in
  //  fp8 tripled the mode's score error with trained-like layer scales)
  using TD = typename std::conditional<F8, bf16_t, T>::type;
  using fragd = typename MP<TD>::frag;
  constexpr int XND = SH::XNP2 * MP<TD>::ESZ / 2;
  {
    // (320 channels: a lane's fifth channel is 256 + lane)
    constexpr int CXD = C > 256 ? 256 : 0;
    const f32x4 lw = *reinterpret_cast<const f32x4*>(a.ds_lnw + 4 * lane);
    const f32x4 lb = *reinterpret_cast<const f32x4*>(a.ds_lnb + 4 * lane);
    const float lwx = a.ds_lnw[CXD + lane], lbx = a.ds_lnb[CXD + lane];
    for (int p = wave; p < NPX; p += NW) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(xl + p * XLP + 4 * lane);
      const float dx = C > 256 ? xl[p * XLP + CXD + lane] : 0.f;
      const float mean = wave_sum(d[0] + d[1] + d[2] + d[3] + dx) * (1.0f / C);
      const f32x4 e = d - mean;
      const float ex = C > 256 ? dx - mean : 0.f;
      const float var = wave_sum(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3] + ex * ex) * (1.0f / C);
      const float rstd = rsqrtf(var + LN_EPS);
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = e[i] * rstd * lw[i] + lb[i];
      MP<TD>::template st4<XN_PLANE>(xn + p * XND + 4 * MP<TD>::ESZ * lane, y);
      if constexpr (C > 256) *reinterpret_cast<TD*>(xn + p * XND + (CXD + lane) * MP<TD>::ESZ) = (TD)(ex * rstd * lwx + lbx);
    }
  }
  __syncthreads();
  if (TRAIN) {   // the downsample's patch rows [alert][q = 2 ky + kx][256] = LayerNorm'd pixels 3 ky + kx
    constexpr int PPR = C * MP<TD>::ESZ / 16;
    unsigned char* dst = reinterpret_cast<unsigned char*>(a.ds_patches) + (size_t)alert0 * 4 * C * MP<TD>::ESZ;
    const int nal = nlive / 9;
    for (int i = tid; i < nal * 4 * PPR; i += NT) {
      const int r = i / PPR, c = i - r * PPR, al2 = r >> 2, q = r & 3;
      *reinterpret_cast<uint4*>(dst + (size_t)r * C * MP<TD>::ESZ + 16 * c) =
          *reinterpret_cast<const uint4*>(xn + (9 * al2 + 3 * (q >> 1) + (q & 1)) * XND + 16 * c);
    }
  }
  {
    S2P_STAMP(57);
    // out[alert][co] = b[co] + sum_k Wd[co][k] patch[alert][k],  k = (2 ky + kx) * 256 + c  ->  pixel 3 ky + kx.
    // Column = alert (4 live of 16); wave w: output tiles 4 w .. 4 w + 3 (16 channels each), 32 k-steps.
    const int al = col < G ? col : 0;
    // this wave's 4 (5) tiles x 32 (40) k-steps = 128 (200) fragments, contiguous in memory (tile-major): a ring of 16 (20) in flight
    constexpr int TPW = CO / 16 / NW, KSC = C / 32;
    const size_t fd0 = (size_t)(TPW * wave) * KSD;
    constexpr int RING = MP<TD>::SPLIT ? 8 : C == 256 ? 16 : 20, NSTEP = TPW * KSD;
    static_assert(KSD % RING == 0, "a tile's k-steps are whole ring rounds");
    constexpr int RPT = KSD / RING;   // ring rounds per tile
    fragd wq[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) wq[i] = MP<TD>::gld(a.ds_wp, fd0 + i, lane);
    f32x4 o = *reinterpret_cast<const f32x4*>(a.ds_b + 16 * (TPW * wave) + 4 * kg);
#pragma unroll 1
    for (int g = 0; g < NSTEP / RING; ++g) {
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        // (s from the round's position in its tile: `s == KSD - 1` below is then false at compile time for every i but the last)
        const int st = g * RING + i, tq = g / RPT, s = (g - tq * RPT) * RING + i, tile = TPW * wave + tq;
        const int q = s / KSC, pq = 3 * (q >> 1) + (q & 1);
        const fragd bf = MP<TD>::template ld8<XN_PLANE>(xn + (9 * al + pq) * XND + (32 * (s - q * KSC) + 8 * kg) * MP<TD>::ESZ);
        o = MP<TD>::run(wq[i], bf, o);
        if (st + RING < NSTEP) wq[i] = MP<TD>::gld(a.ds_wp, fd0 + st + RING, lane);
        if (i == RING - 1 && s == KSD - 1) {   // tile finished (every KSD steps = RPT ring rounds)
          if (col < G && alert0 + col < a.B)
            *reinterpret_cast<f32x4*>(a.out + (size_t)(alert0 + col) * CO + 16 * tile + 4 * kg) = o;
          if (st + 1 < NSTEP) o = *reinterpret_cast<const f32x4*>(a.ds_b + 16 * (tile + 1) + 4 * kg);
        }
      }
    }
  }
  S2P_STAMP(58);
}

// ---- operand packing: one 16x16x32 A fragment = 64 lanes x 8 elements = 1 KiB, lane l holds row (l & 15),
//      k = 32 s + 8 (l >> 4) + j of its tile
template <typename T>
__global__ void pack_frag_kernel(const float* __restrict__ w, const float* __restrict__ rowscale, T* __restrict__ out,
                                 int rows, int K, int reorder_down, int cin) {
  // element index -> (tile, k-step, lane, j); out is tile-major, k-step next: a tile's k-steps are contiguous
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)rows * K;
  if (i >= total) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;                 // fragment index = tile * (K / 32) + s
  const int ksteps = K / 32;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 16 * tile + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
  float v;
  if (reorder_down) {
    // downsample filter [Cout][Cin][2][2] -> k = (2 ky + kx) * Cin + c
    const int q = k / cin, c = k - q * cin;
    v = w[((long)row * cin + c) * 4 + q];
  } else {
    v = w[(long)row * K + k];
  }
  if (rowscale != nullptr) v *= rowscale[row];
  out[i] = (T)v;
}

// split mode: 2 KiB per fragment, [lane][8] heads then [lane][8] remainders
__global__ void pack_frag_split_kernel(const float* __restrict__ w, const float* __restrict__ rowscale,
                                       _Float16* __restrict__ out, int rows, int K, int reorder_down, int cin) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = K / 32;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 16 * tile + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
  float v;
  if (reorder_down) {
    const int q = k / cin, c = k - q * cin;
    v = w[((long)row * cin + c) * 4 + q];
  } else {
    v = w[(long)row * K + k];
  }
  if (rowscale != nullptr) v *= rowscale[row];
  _Float16 hi, lo;
  split_f16(v, hi, lo);
  out[fs * 1024 + l * 8 + j] = hi;
  out[fs * 1024 + 512 + l * 8 + j] = lo;
}

__global__ void pack_frag_fp8_kernel(const float* __restrict__ w, const float* __restrict__ rowscale,
                                     const float* __restrict__ scale, unsigned char* __restrict__ out, int rows, int K,
                                     int reorder_down, int cin) {
  // 16x16x128 A fragments, 2 KiB each: lane l holds row (l & 15), k = 128 s + 32 (l >> 4) + j (j < 32); in memory
  // [lane][j 0..15] then [lane][j 16..31] (MP<fp8_t>::gld)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 31), l = (int)((i >> 5) & 63);
  const long fs = i >> 11;
  const int ksteps = K / 128;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 16 * tile + (l & 15), k = 128 * s + 32 * (l >> 4) + j;
  float v;
  if (reorder_down) {
    const int q = k / cin, c = k - q * cin;
    v = w[((long)row * cin + c) * 4 + q];
  } else {
    v = w[(long)row * K + k];
  }
  if (rowscale != nullptr) v *= rowscale[row];
  out[fs * 2048 + (j >> 4) * 1024 + l * 16 + (j & 15)] =
      (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v * scale[0], 0.f, 0, false) & 0xff);
}

template <typename T, int G = S2P_ALERTS, int TRAIN = 0, int CW = 256, int ROWS = 0> int launch_stage2p_t(const Stage2pArgs& a, hipStream_t st) {
  auto kern = stage2p_kernel<T, G, TRAIN, CW, ROWS>;
  constexpr int lds_bytes = TRAIN == 1 ? Lds<T, G, CW>::BYTES_TRAIN : Lds<T, G, CW>::BYTES;
  static_assert(lds_bytes <= 160 * 1024, "the images fit one CU");
  static DevOnce attr_set;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set.done();
  }
  hipLaunchKernelGGL(kern, dim3(ROWS > 0 ? (a.B + ROWS - 1) / ROWS : (a.B + G - 1) / G), dim3(NT), lds_bytes, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool stage2p_supported(int prec, int c2, int c3, int depth) {
  if (depth < 1 || depth > S2P_MAX_DEPTH || c3 != 2 * c2) return false;
  if (c2 == 256) return prec == BTSBOT_BF16 || prec == BTSBOT_F16 || prec == BTSBOT_FP8 || prec == BTSBOT_F16X2;
  return c2 == 320 && (prec == BTSBOT_BF16 || prec == BTSBOT_F16);   // convnext_nano: the 16-bit modes, inference
}

// src [rows][K] fp32 (row-major; reorder_down: a [Cout][Cin][2][2] downsample filter) -> MFMA A fragments
int launch_pack_s2p(int prec, const float* src, const float* rowscale, void* dst, int rows, int K, int reorder_down,
                    int cin, float* scale, hipStream_t st) {
  const long total = (long)rows * K;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_frag_kernel<bf16_t>, grid, blk, 0, st, src, rowscale, reinterpret_cast<bf16_t*>(dst), rows, K,
                       reorder_down, cin);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_frag_kernel<f16_t>, grid, blk, 0, st, src, rowscale, reinterpret_cast<f16_t*>(dst), rows, K,
                       reorder_down, cin);
  else if (prec == BTSBOT_F16X2)
    hipLaunchKernelGGL(pack_frag_split_kernel, grid, blk, 0, st, src, rowscale, reinterpret_cast<_Float16*>(dst), rows, K,
                       reorder_down, cin);
  else if (prec == BTSBOT_FP8) {
    if (scale == nullptr) {
      btsbot_set_error("pack_s2p: the fp8 mode needs a scale slot");
      return BTSBOT_ERR_INVALID_ARG;
    }
    // (the downsample filter [Cout][Cin][2][2] is scanned in its own order: the maximum does not care)
    const int rc = launch_fp8_scale(src, rowscale, total, K, scale, st);
    if (rc != BTSBOT_OK) return rc;
    hipLaunchKernelGGL(pack_frag_fp8_kernel, grid, blk, 0, st, src, rowscale, scale, reinterpret_cast<unsigned char*>(dst),
                       rows, K, reorder_down, cin);
  } else {
    btsbot_set_error("pack_s2p: precision %d is not a packed-fragment mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// 7 alerts per workgroup when that takes fewer rounds of one workgroup per CU than 4 (256 CUs): the kernel's time per
// round does not depend on the alerts resident -- the filter stream bounds it
int stage2p_alerts_per_workgroup(int B, int hint) {
  static const int forced = [] {
    const char* e = getenv("BTSBOT_AMD_S2P_G");
    return e != nullptr ? atoi(e) : 0;
  }();
  if (forced == 4 || forced == 5 || forced == 7) return forced;
  // the caller's hint (btsbot_set_option): a scoring loop that keeps several forwards in flight on different streams
  // asks for 7 at every batch size -- at 1024 alerts the kernel then occupies 147 CUs instead of 256 for 116 instead of
  // 105 us, and the other stream's kernels take the rest (measured: +6 % through ScoreStream, -4 % for serial calls)
  if (hint == 4 || hint == 5 || hint == 7) return hint;
  // 5 alerts (45 of the same 48 columns as 4: no more matrix work per workgroup, a fifth fewer workgroups streaming the
  // filters -- 100.7 against 105.7 us at 1024 alerts, where 205 workgroups leave 51 CUs and a fifth of the L2 traffic
  // unused) unless 7 take fewer rounds of one workgroup per CU
  const int r5 = ((B + 4) / 5 + 255) / 256, r7 = ((B + 6) / 7 + 255) / 256;
  return r7 < r5 ? 7 : 5;
}

// The row-tile form: x [rows][256] fp32 += W2 gelu(W1 LN(x) + b1) + b2 in place, 64 rows per workgroup; blk carries the
// LayerNorm, the two biases, gamma (ones where the layer has no layer scale) and both filters as launch_pack_s2p fragments
bool stage2p_rows_supported(int prec, int c) { return c == 256 && (prec == BTSBOT_BF16 || prec == BTSBOT_F16); }
int launch_stage2p_rows(int prec, float* x, long rows, const Stage2pBlk& blk, hipStream_t st) {
  if (rows <= 0) return BTSBOT_OK;
  if (!stage2p_rows_supported(prec, 256) || rows > 0x7fffffffL) {
    btsbot_set_error("stage2p_rows: precision %d / %ld rows not supported", prec, rows);
    return BTSBOT_ERR_INVALID_ARG;
  }
  Stage2pArgs a;
  memset(&a, 0, sizeof(a));
  a.x_in = x;
  a.tap_stage = x;
  a.blk[0] = blk;
  a.depth = 1;
  a.B = (int)rows;
  a.cw = 256;
  return prec == BTSBOT_BF16 ? launch_stage2p_t<bf16_t, 7, 0, 256, 64>(a, st) : launch_stage2p_t<f16_t, 7, 0, 256, 64>(a, st);
}

int launch_stage2p(int prec, const Stage2pArgs& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (a.cw == 320) {   // convnext_nano: 5 alerts per workgroup (7 would not fit the LDS next to the 320-wide map)
    if (a.train != 0 || (prec != BTSBOT_BF16 && prec != BTSBOT_F16)) {
      btsbot_set_error("stage2p: 320 channels run the bf16 / f16 inference form only (precision %d, train %d)", prec, a.train);
      return BTSBOT_ERR_INVALID_ARG;
    }
    return prec == BTSBOT_BF16 ? launch_stage2p_t<bf16_t, 5, 0, 320>(a, st) : launch_stage2p_t<f16_t, 5, 0, 320>(a, st);
  }
  if (a.cw != 256) {
    btsbot_set_error("stage2p: width %d not supported", a.cw);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (a.train) {   // the training forward: 16-bit modes, alerts per workgroup as in inference (4 / 5 / 7)
    for (int j = 0; j < a.depth; ++j)
      if (a.keep[j].xin == nullptr || a.keep[j].xn == nullptr || a.keep[j].a == nullptr || a.keep[j].hh == nullptr) {
        btsbot_set_error("stage2p: the training forward needs every kept buffer of block %d (each with room for %d + 9 alerts)",
                         j, a.B);
        return BTSBOT_ERR_INVALID_ARG;
      }
    if (a.ds_patches == nullptr || a.tap_stage == nullptr) {
      btsbot_set_error("stage2p: the training forward needs ds_patches and the stage output (tap_stage)");
      return BTSBOT_ERR_INVALID_ARG;
    }
    // (4 or 5 alerts per workgroup: with 7 -- four column blocks -- the form spills 22-25 registers in its chunk loop)
    const bool g4 = stage2p_alerts_per_workgroup(a.B, a.alerts_hint) == 4;
    if (prec == BTSBOT_BF16) return g4 ? launch_stage2p_t<bf16_t, 4, 1>(a, st) : launch_stage2p_t<bf16_t, 5, 1>(a, st);
    if (prec == BTSBOT_F16) return g4 ? launch_stage2p_t<f16_t, 4, 1>(a, st) : launch_stage2p_t<f16_t, 5, 1>(a, st);
    btsbot_set_error("stage2p: the training forward runs in the bf16 / f16 modes, not %d", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int gsel = stage2p_alerts_per_workgroup(a.B, a.alerts_hint);
  const bool g7 = gsel == 7, g5 = gsel == 5;
  if (prec == BTSBOT_BF16)
    return g7 ? launch_stage2p_t<bf16_t, 7>(a, st) : g5 ? launch_stage2p_t<bf16_t, 5>(a, st) : launch_stage2p_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16)
    return g7 ? launch_stage2p_t<f16_t, 7>(a, st) : g5 ? launch_stage2p_t<f16_t, 5>(a, st) : launch_stage2p_t<f16_t>(a, st);
  if (prec == BTSBOT_FP8)
    return g7 ? launch_stage2p_t<fp8_t, 7>(a, st) : g5 ? launch_stage2p_t<fp8_t, 5>(a, st) : launch_stage2p_t<fp8_t>(a, st);
  // (split mode: the doubled planes of 7 alerts do not fit the LDS: 5 wherever more than 4 were chosen)
  if (prec == BTSBOT_F16X2) return gsel == 4 ? launch_stage2p_t<f16x2_t>(a, st) : launch_stage2p_t<f16x2_t, 5>(a, st);
  btsbot_set_error("stage2p: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
