// Backward of a ConvNeXt block's MLP as ONE kernel (gfx950, 16-bit operand modes, C in {64, 128}: the wide-map
// stages, where every 4C-wide tensor of the unfused backward -- fc1 pre-activation, GELU output, da -- is 118 MB per
// 1024 alerts and each is written once and read twice).
//
//     y = x + gamma * (W2 g + b2),   g = gelu(a),   a = W1 xn + b1
//
//   given   xn [R][C] (kept by the forward), dy [R][C] = d(loss)/dy, both in the operand type
//   writes  dxn [R][C] fp32            = da W1,              da = (dy (gamma W2)) * gelu'(a)
//           G   [C][4C]   partial tiles = dy^T g             (fc2 / layer-scale gradients follow from it: fc2_grads_kernel)
//           dW1 [4C][C]   partial tiles = da^T xn
//           db1 [4C]     += colsum(da)
// a is RECOMPUTED from xn (one more K = C GEMM, fp32 accumulators: the unfused path differentiated GELU at the
// 16-bit rounding of a); g and da never leave the CU.  Replaces, for nn.Linear x 2 + GELU of timm's ConvNeXtBlock
// (reached at /root/reference/btsbot/architectures.py:108,132), what autograd does at /root/reference/btsbot/train.py:526.
//
// Mapping.  A wave owns 32 hidden units for the whole kernel; a workgroup of NW waves owns HS = 32 NW of them and walks
// row tiles of 64 pixels (grid.y = 4C / HS hidden slices, grid.x workgroups stride over the row tiles).  Per row tile:
//   (1) a  [64 x 32] = xn-tile . W1_w^T      (3) t [64 x 32] = dy-tile . (gamma W2)_w      v_mfma_f32_32x32x16, K = C
//   (2) g = gelu(a + b1), g' likewise (one v_exp for both)      (4) da = t g'
//   (5) G_w   [C x 32] += dy-tile^T . g      (6) dW1_w [32 x C] += da^T . xn-tile          K = 64 pixels
//       The accumulator tiles of (1)/(3) ARE the B / A operands of (5)/(6): the sum runs over their row index, so they
//       are only converted to 16 bit.  Position 8h + e of k-step s then holds pixel 16 s + 8 (e >> 2) + 4 h + (e & 3);
//       the other operand -- the transposed row tile -- is read with ds_read_b64_tr_b16 in the same order.
//   (7) dxn [64 x C] = da [64 x HS] . W1 [HS x C]: da goes through LDS ([hidden][pixel], 8-byte stores) because here the
//       sum runs over the hidden units of ALL waves; v_mfma_f32_16x16x32, both operands by transposing reads (da, and the
//       same LDS image of W1 that feeds (1)).  With more than one hidden slice (C = 128: four) every slice stores
//       its own addend plane of dxn and the reader -- dwln_bwd_kernel -- adds them (fp32 atomics on a cleared buffer
//       were 47 of the kernel's 107 us at 1024 alerts, and not reproducible).
// The filter-gradient accumulators (2 x C x HS floats per workgroup, half the register budget) leave once, as dense
// partial tiles in wgrad.hip's layout; wgrad_reduce_kernel adds the workgroups in a fixed order.  colsum(dy) (the
// layer-scale / fc2-bias gradients need it) and colsum(da) = db1 are summed on the way.
// ONE barrier per row tile: the row tiles and the da image are double-buffered, dxn of tile t - 1 is computed in
// iteration t, the next tile's rows are requested at the top of an iteration and stored behind its last read.  Per wave
// the products of one 32-pixel unit are issued between the GELU pieces of the other (A(0) | A(1) + GELU(0) | C(0) +
// GELU(1) | C(1)).  Measured and what bounds it: DESIGN.md section 5b, profiles/r03_mlp_bwd_kernel.txt.
#include <stdlib.h>

#include "common.h"

#ifndef MLP_BWD_STAMP
#define MLP_BWD_STAMP 0   // development only: per-phase s_memtime stamps of the 4th tile of workgroup 0 (into `stamps`)
#endif
#if MLP_BWD_STAMP
__device__ unsigned long long mlp_bwd_stamps[8 * 16];
#define MB_STAMP(i)                                                                                 \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (blockIdx.x == 0 && blockIdx.y == 0 && it == 3 && lane == 0) mlp_bwd_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#else
#define MB_STAMP(i)
#endif
#ifndef MLP_BWD_ABL
#define MLP_BWD_ABL 0   // development only (tools/unit/mlp_bwd_time.hip): bit mask of phases left out
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <typename T> struct MB;
template <> struct MB<bf16_t> {
  static constexpr int DEG = GeluDeg<bf16_t>::value;
  static __device__ __forceinline__ f32x16 m32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 m16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float v) { return __builtin_bit_cast(short, (bf16_t)v); }
  static __device__ __forceinline__ float tofloat(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
};
template <> struct MB<f16_t> {
  static constexpr int DEG = GeluDeg<f16_t>::value;
  static __device__ __forceinline__ f32x16 m32(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 m16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short cvt(float v) { return __builtin_bit_cast(short, (f16_t)v); }
  static __device__ __forceinline__ float tofloat(unsigned short v) { return (float)__builtin_bit_cast(f16_t, v); }
};

// gelu_poly and its derivative from one exponential (common.h: gelu_poly / gelu_poly_grad)
template <int DEG> __device__ __forceinline__ void gelu_both(float x, float& g, float& gp) {
  const float a = __builtin_fabsf(x);
  const float e = __builtin_amdgcn_exp2f(gelu_q<DEG>(a));
  g = fmaf(-a, e, relu_f(x));
  const float d = e * fmaf(a * 0.6931471805599453f, gelu_dq<DEG>(a), 1.0f);
  gp = x > 0.0f ? 1.0f - d : d;
}

// ds_read_b64_tr_b16: the 16 lanes of a group address a 4-row x 16-column block (lane = (row q, 4-column piece p)) and
// lane i of the group receives column i's four row values
__device__ __forceinline__ s16x4 tr4(const unsigned char* tile, int pitchb, int row0, int col0, int lane) {
  const int q = (lane >> 2) & 3, p = lane & 3;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + (row0 + q) * pitchb + (col0 + 4 * p) * 2));
}
// 16x16x32 operand: column col0 + (lane & 15), reduction rows row0 + 8 (lane >> 4) .. +7
__device__ __forceinline__ s16x8 tr16(const unsigned char* tile, int pitchb, int row0, int col0, int lane) {
  const int g = lane >> 4;
  const s16x4 lo = tr4(tile, pitchb, row0 + 8 * g, col0, lane);
  const s16x4 hi = tr4(tile, pitchb, row0 + 8 * g + 4, col0, lane);
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// 32x32x16 operand in the accumulator's row order: column col0 + (lane & 31), reduction rows
// row0 + 4 h + {0..3} and row0 + 8 + 4 h + {0..3}  (h = lane >> 5)
__device__ __forceinline__ s16x8 trp(const unsigned char* tile, int pitchb, int row0, int col0, int lane) {
  const int h = lane >> 5, g1 = (lane >> 4) & 1;
  const s16x4 lo = tr4(tile, pitchb, row0 + 4 * h, col0 + 16 * g1, lane);
  const s16x4 hi = tr4(tile, pitchb, row0 + 8 + 4 * h, col0 + 16 * g1, lane);
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int C, int NW> struct BwdGeo {
  static constexpr int H = 4 * C, HS = 32 * NW, NH = H / HS, TR = 64, MT = TR / 32;
  static constexpr int NT = 64 * NW;
  static constexpr int PX = 2 * C + 16;         // bytes per staged row of xn / dy / W1 (C values + 16: b128 row reads of 16 rows
                                                // start on 16 different 4-bank groups)
  static constexpr int PD = 2 * TR + 16;        // bytes per da row [hidden][pixel]
  static constexpr int XB = TR * PX;            // one row tile of one tensor
  static constexpr int XY = 2 * XB;             // xn tile + dy tile; two of those (the next tile lands while this one is read)
  static constexpr int DAB = HS * PD;           // one da image; two of those (tile t's is read while tile t+1's is written)
  static constexpr int OFF_W1 = 2 * XY, OFF_DA = OFF_W1 + HS * PX;
  static constexpr int BYTES = OFF_DA + 2 * DAB;
  static constexpr int LCH = TR * C / 8 / NT;   // 16-byte pieces per thread, tile and tensor
  static constexpr int CT = C / 32, KC = C / 16;
  static constexpr int TPW = (TR / 16) * (C / 16) / NW;   // 16x16 tiles of dxn per wave (one row of tiles)
  static_assert(H % HS == 0 && (TR * C / 8) % NT == 0 && (C / 16) % TPW == 0 && NT % (C / NH) == 0 && TR % (NT / (C / NH)) == 0, "geometry");
};

template <typename T, int C, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_bwd_kernel(const T* __restrict__ xn, const T* __restrict__ dy,
                                                          const T* __restrict__ w1, const T* __restrict__ w2g,
                                                          const float* __restrict__ b1, float* __restrict__ dxn,
                                                          float* __restrict__ partG, float* __restrict__ partW,
                                                          float* __restrict__ db1, float* __restrict__ Ssum, int R,
                                                          float* detb, float* dets) {
  using G = BwdGeo<C, NW>;
  using M = MB<T>;
  constexpr int PX = G::PX, PD = G::PD;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int hs0 = (int)blockIdx.y * G::HS;
  const int hid = hs0 + 32 * wave + lr;   // the lane's hidden unit in steps (1)-(6)

  for (int q = tid; q < G::HS * C / 8; q += G::NT) {
    const int r = q / (C / 8), cc = q % (C / 8);
    *reinterpret_cast<u32x4*>(sm + G::OFF_W1 + r * PX + cc * 16) =
        *reinterpret_cast<const u32x4*>(w1 + (size_t)(hs0 + r) * C + 8 * cc);
  }
  s16x8 w2f[G::KC];
#pragma unroll
  for (int kc = 0; kc < G::KC; ++kc)
    w2f[kc] = *reinterpret_cast<const s16x8*>(w2g + (size_t)hid * C + 16 * kc + 8 * lh);
  float b1v = b1[hid];
  // Re-define the loop-invariant operands behind their loads: hipcc's wait-count pass otherwise keeps them "possibly in
  // flight" around the loop's back edge and puts s_waitcnt vmcnt(0) in front of their first use in EVERY iteration --
  // directly behind the row prefetch, i.e. a whole HBM round trip per tile.
  asm volatile("" : "+v"(b1v));
#pragma unroll
  for (int kc = 0; kc < G::KC; ++kc) {
    u32x4 v = __builtin_bit_cast(u32x4, w2f[kc]);
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    w2f[kc] = __builtin_bit_cast(s16x8, v);
  }

  f32x16 gG[G::CT], gW[G::CT];
#pragma unroll
  for (int i = 0; i < G::CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gG[i][r] = 0.f;
      gW[i][r] = 0.f;
    }
  float sb1 = 0.f, ssum = 0.f;
  // colsum(dy): every hidden slice sums C / NH of the channels (one slice doing all of them was 6 % behind the others
  // per tile); thread = (channel of the slice's share, row group)
  constexpr int SC = C / G::NH, SGRP = G::NT / SC;
  const int scol = (int)blockIdx.y * SC + tid % SC, srg = tid / SC;

  const int ntiles = (R + G::TR - 1) / G::TR;
  u32x4 rx[G::LCH], ry[G::LCH];
  // rows beyond R: the load repeats row 0 and the dy piece is cleared on its way into LDS (dy = 0 makes the row's
  // t, da and its share of G, dW1, db1 and colsum(dy) vanish whatever xn holds).  The select sits in stash(), behind
  // the tile's work: in fetch() it put an s_waitcnt vmcnt(0) -- a whole HBM round trip -- at the top of every tile
  auto fetch = [&](int t) {
#pragma unroll
    for (int s = 0; s < G::LCH; ++s) {
      const int q = tid + G::NT * s, r = q / (C / 8), cc = q % (C / 8);
      const int m = t * G::TR + r;
      const size_t off = (size_t)(m < R ? m : 0) * C + 8 * cc;
      rx[s] = *reinterpret_cast<const u32x4*>(xn + off);
      ry[s] = *reinterpret_cast<const u32x4*>(dy + off);
    }
  };
  auto stash = [&](int buf, int t) {
#pragma unroll
    for (int s = 0; s < G::LCH; ++s) {
      const int q = tid + G::NT * s, r = q / (C / 8), cc = q % (C / 8);
      const bool ok = t * G::TR + r < R;
      *reinterpret_cast<u32x4*>(sm + buf * G::XY + r * PX + cc * 16) = rx[s];
      *reinterpret_cast<u32x4*>(sm + buf * G::XY + G::XB + r * PX + cc * 16) = ok ? ry[s] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // ---- (7) dxn = da W1 for the tile whose da image is complete: the wave's TPW tiles of 16 pixels x 16 channels
  constexpr int NTC = C / 16;
  const int mi = wave * G::TPW / NTC, ni0 = wave * G::TPW % NTC;
  auto dxn_tile = [&](int tile, const unsigned char* DA) {
    f32x4 dx[G::TPW];
#pragma unroll
    for (int i = 0; i < G::TPW; ++i) dx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(MLP_BWD_ABL & 4)) {
      // fragments of k-step ks + 1 are requested before the products of k-step ks (left to itself the compiler waits
      // for each k-step's six transposing reads right in front of its products: 8 x ~250 cycles of LDS latency per
      // tile, a quarter of the tile's time in the in-kernel stamps)
      constexpr int NKS = G::HS / 32;
      s16x8 af[2], bf[2][G::TPW];
      af[0] = tr16(DA, PD, 0, 16 * mi, lane);
#pragma unroll
      for (int i = 0; i < G::TPW; ++i) bf[0][i] = tr16(sm + G::OFF_W1, PX, 0, 16 * (ni0 + i), lane);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < NKS) {
          af[nxt] = tr16(DA, PD, 32 * (ks + 1), 16 * mi, lane);
#pragma unroll
          for (int i = 0; i < G::TPW; ++i) bf[nxt][i] = tr16(sm + G::OFF_W1, PX, 32 * (ks + 1), 16 * (ni0 + i), lane);
        }
#pragma unroll
        for (int i = 0; i < G::TPW; ++i) dx[i] = M::m16(af[cur], bf[cur][i], dx[i]);
      }
    }
    // (every slice stores its own addend plane, R * C floats apart; one plane when the workgroup owns all hidden units)
    const size_t oi = (size_t)blockIdx.y * R * C + (size_t)(tile * G::TR + 16 * mi + 4 * (lane >> 4)) * C + 16 * ni0 + (lane & 15);
    float* o = dxn + oi;
    if (MLP_BWD_ABL & 8) return;
    if (tile * G::TR + G::TR <= R) {   // whole tile inside the map (workgroup-uniform): no per-row masks
#pragma unroll
      for (int i = 0; i < G::TPW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * C + 16 * i] = dx[i][r];
    } else {
#pragma unroll
      for (int i = 0; i < G::TPW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (tile * G::TR + 16 * mi + 4 * (lane >> 4) + r < R) o[r * C + 16 * i] = dx[i][r];
    }
  };

  // One barrier per row tile.  Iteration t: the rows of tile t are in row buffer p, the da image of tile t - 1 in da
  // buffer p ^ 1.  Program order per wave (MT = 2 units of 32 pixels):
  //     dxn(t - 1), A(0) | A(1) with GELU(0) | C(0) with GELU(1) | C(1)        A = steps (1)(3), C = steps (5)(6)
  // so that the matrix work of one unit is issued between the VALU work of the other.
  int t = blockIdx.x, p = 0, tprev = -1;
  if (t < ntiles) {
    fetch(t);
    stash(0, t);
  }
  __syncthreads();
  int it = 0;
  for (; t < ntiles; t += gridDim.x, p ^= 1, ++it) {
    MB_STAMP(0);
    const int tn = t + (int)gridDim.x;
    const bool more = tn < ntiles;   // workgroup-uniform
    if (more) fetch(tn);
    const unsigned char* X = sm + p * G::XY;
    const unsigned char* Y = X + G::XB;
    unsigned char* DA = sm + G::OFF_DA + p * G::DAB;
    // Two waves share a SIMD (NW = 8) and the tile is VALU-issue bound on GELU + GELU': the older wave (0-3) wins the
    // arbitration, so in lockstep the younger one's GELU waits for it and the older one then idles at the barrier
    // (stamps: 1650 vs 3300 cycles for the same work).  The older half therefore runs the LDS / matrix-pipe bound
    // dxn(t - 1) BEHIND its GELU work and the younger half in front: the older half's VALU phase faces the younger
    // half's dxn phase and the other way round.
    const bool dxn_late = NW == 8 && wave < NW / 2;   // wave-uniform
    if (!dxn_late && tprev >= 0) dxn_tile(tprev, sm + G::OFF_DA + (p ^ 1) * G::DAB);
    MB_STAMP(1);

    f32x16 a[G::MT], tt[G::MT];
    s16x8 gq[G::MT][2], dq[G::MT][2];
    // the k-th of unit mt's 2 KC products of steps (1)(3)
    auto stepA = [&](int mt, int k) {
      if (MLP_BWD_ABL & 16) return;
      const int kc = k >> 1, o = (32 * mt + lr) * PX + (16 * kc + 8 * lh) * 2;
      if (!(k & 1)) {
        const s16x8 bw = *reinterpret_cast<const s16x8*>(sm + G::OFF_W1 + (32 * wave + lr) * PX + (16 * kc + 8 * lh) * 2);
        a[mt] = M::m32(*reinterpret_cast<const s16x8*>(X + o), bw, a[mt]);
      } else {
        tt[mt] = M::m32(*reinterpret_cast<const s16x8*>(Y + o), w2f[kc], tt[mt]);
      }
    };
    // the k-th of unit mt's 4 CT products of steps (5)(6): k-step ks = 16 pixels = 8 accumulator registers
    auto stepC = [&](int mt, int k) {
      if (MLP_BWD_ABL & 2) return;
      const int ks = k / (2 * G::CT), ct = (k / 2) % G::CT, r0 = 32 * mt + 16 * ks;
      if (!(k & 1))
        gG[ct] = M::m32(trp(Y, PX, r0, 32 * ct, lane), gq[mt][ks], gG[ct]);
      else
        gW[ct] = M::m32(dq[mt][ks], trp(X, PX, r0, 32 * ct, lane), gW[ct]);
    };
    // (2)(4) for accumulator registers r0 .. r0 + n - 1 of unit mt
    auto gelu = [&](int mt, int r0, int n) {
#pragma unroll
      for (int r = r0; r < r0 + n; ++r) {
        float g, gp;
        if (MLP_BWD_ABL & 1) {
          g = a[mt][r] + b1v;
          gp = 1.f;
        } else
          gelu_both<M::DEG>(a[mt][r] + b1v, g, gp);
        const float da = tt[mt][r] * gp;
        sb1 += da;
        gq[mt][r >> 3][r & 7] = M::cvt(g);
        dq[mt][r >> 3][r & 7] = M::cvt(da);
      }
    };
    // da -> LDS [hidden][pixel]: accumulator registers 4 rq .. 4 rq + 3 are pixels 32 mt + 8 rq + 4 h + {0..3}
    auto da_out = [&](int mt) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const s16x8 v = dq[mt][rq >> 1];
        const int e = (rq & 1) * 4;
        *reinterpret_cast<s16x4*>(DA + (32 * wave + lr) * PD + (32 * mt + 8 * rq + 4 * lh) * 2) =
            s16x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
      }
    };
    constexpr int NA = 2 * G::KC, NC = 4 * G::CT;   // products per unit
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        a[mt][r] = 0.f;
        tt[mt][r] = 0.f;
      }
#pragma unroll
    for (int k = 0; k < NA; ++k) stepA(0, k);
    MB_STAMP(2);
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt) {
      // GELU of unit mt in 8 pieces of 2 values, the matrix products of the neighbouring units between them
      constexpr int NP = 8;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        if (mt + 1 < G::MT) {
#pragma unroll
          for (int k = pc * NA / NP; k < (pc + 1) * NA / NP; ++k) stepA(mt + 1, k);
        }
        if (mt > 0) {
#pragma unroll
          for (int k = pc * NC / NP; k < (pc + 1) * NC / NP; ++k) stepC(mt - 1, k);
        }
        gelu(mt, 2 * pc, 2);
      }
      da_out(mt);
      MB_STAMP(3 + mt);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) stepC(G::MT - 1, k);
    MB_STAMP(5);
#pragma unroll
    for (int r = 0; r < G::TR / SGRP; ++r)
      ssum += M::tofloat(*reinterpret_cast<const unsigned short*>(Y + (srg + SGRP * r) * PX + scol * 2));
    __builtin_amdgcn_sched_barrier(0);   // (keeps the wait for the fetched rows down here)
    if (dxn_late && tprev >= 0) dxn_tile(tprev, sm + G::OFF_DA + (p ^ 1) * G::DAB);
    MB_STAMP(6);
    if (more) stash(p ^ 1, tn);
    MB_STAMP(7);
    __syncthreads();   // da image p and row buffer p ^ 1 are complete; everybody is done with row buffer p and da image p ^ 1
    MB_STAMP(8);
    tprev = t;
  }
  if (tprev >= 0) dxn_tile(tprev, sm + G::OFF_DA + (p ^ 1) * G::DAB);
  __syncthreads();   // (the colsum below reuses the row buffers)

  // ---- the workgroup's filter-gradient tiles (wgrad.hip's partial layout: [slice][hidden slice][rows][cols])
  const size_t pslot = (size_t)blockIdx.x * gridDim.y + blockIdx.y;
  float* pg = partG + pslot * (C * G::HS);
  float* pw = partW + pslot * (G::HS * C);
#pragma unroll
  for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
      pg[(32 * ct + rr) * G::HS + 32 * wave + lr] = gG[ct][r];   // G  [channel][hidden]
      pw[(32 * wave + rr) * C + 32 * ct + lr] = gW[ct][r];       // dW1 [hidden][channel]
    }
  sb1 += __shfl_xor(sb1, 32);
  // (deterministic mode: one partial row per row slice for db1 [4 C] and one for colsum(dy) [C], launch_det_reduce)
  if (lh == 0) det_add(db1 + hid, sb1, detb, (size_t)blockIdx.x * (4 * C) + hid);
  {   // (every LDS read of the loop is behind its last barrier)
    float* red = reinterpret_cast<float*>(sm);
    red[tid] = ssum;
    __syncthreads();
    if (tid < SC) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < SGRP; ++g) v += red[tid + SC * g];
      det_add(Ssum + (int)blockIdx.y * SC + tid, v, dets, (size_t)blockIdx.x * C + (int)blockIdx.y * SC + tid);
    }
  }
}

template <typename T, int C, int NW>
int mlp_bwd_launch(const void* xn, const void* dy, const void* w1, const void* w2g, const float* b1, float* dxn,
                   float* part, float* Gacc, float* Ssum, float* dW1, float* db1, int R, hipStream_t st, WgradReduceJob* jobs) {
  using G = BwdGeo<C, NW>;
  auto kern = mlp_bwd_kernel<T, C, NW>;
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::BYTES));
    attr.done();
  }
  const int gx = mlp_bwd_slices(C, R);
  float* partG = part;
  float* partW = part + (size_t)gx * G::NH * C * G::HS;
  float* detb = det_alloc((size_t)gx * 4 * C);
  float* dets = detb != nullptr ? det_alloc((size_t)gx * C) : nullptr;
  if (dets == nullptr) detb = nullptr;
  hipLaunchKernelGGL(kern, dim3(gx, G::NH), dim3(G::NT), G::BYTES, st, reinterpret_cast<const T*>(xn),
                     reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(w1), reinterpret_cast<const T*>(w2g), b1,
                     dxn, partG, partW, db1, Ssum, R, detb, dets);
  LAUNCH_CHECK();
  if (detb != nullptr) {
    const DetOut ob{db1, 1}, os{Ssum, 1};
    if (launch_det_reduce(detb, gx, 4 * C, 1, &ob, st) != BTSBOT_OK || launch_det_reduce(dets, gx, C, 1, &os, st) != BTSBOT_OK)
      return BTSBOT_ERR_HIP;
  }
  jobs[0] = WgradReduceJob{partG, Gacc, C, 4 * C, 4 * C, 1, G::NH, gx, C, G::HS};
  jobs[1] = WgradReduceJob{partW, dW1, 4 * C, C, C, G::NH, 1, gx, G::HS, C};
  return BTSBOT_OK;
}

template <typename T>
int mlp_bwd_t(int C, const void* xn, const void* dy, const void* w1, const void* w2g, const float* b1, float* dxn,
              float* part, float* Gacc, float* Ssum, float* dW1, float* db1, int R, hipStream_t st, WgradReduceJob* jobs) {
  if (C == 64) return mlp_bwd_launch<T, 64, 8>(xn, dy, w1, w2g, b1, dxn, part, Gacc, Ssum, dW1, db1, R, st, jobs);
  return mlp_bwd_launch<T, 128, 4>(xn, dy, w1, w2g, b1, dxn, part, Gacc, Ssum, dW1, db1, R, st, jobs);
}

}  // namespace

bool mlp_bwd_supported(int prec, int C) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (C == 64 || C == 128);
}

int mlp_bwd_planes(int C) { return C == 128 ? 4 : 1; }   // = BwdGeo<128, 4>::NH

// workgroups along the rows (= slices of the partial tiles)
int mlp_bwd_slices(int C, int R) {
  const int ntiles = (R + 63) / 64;
  const int want = C == 64 ? 256 : 64;   // x the hidden slices (1 / 4): one workgroup per CU
  return ntiles < want ? (ntiles > 0 ? ntiles : 1) : want;
}

size_t mlp_bwd_part_floats(int C, int R) { return (size_t)mlp_bwd_slices(C, R) * 2 * C * 4 * C; }

// jobs[0] (G += ...) and jobs[1] (dW1 += ...) describe the slice reductions for launch_wgrad_reduce(); db1 and
// Ssum [C] (+= colsum(dy)) are accumulated here (atomics).  dxn is overwritten: mlp_bwd_planes(C) addend planes of
// R * C floats (C = 128: its four hidden slices; the reader adds them, launch_dwln_bwd's nplanes).
int launch_mlp_bwd(int prec, int C, const void* xn, const void* dy, const void* w1, const void* w2g, const float* b1,
                   float* dxn, float* part, float* Gacc, float* Ssum, float* dW1, float* db1, int R, hipStream_t st,
                   WgradReduceJob* jobs) {
  jobs[0].nsl = jobs[1].nsl = 0;
  if (R <= 0) return BTSBOT_OK;
  if (!mlp_bwd_supported(prec, C)) {
    btsbot_set_error("mlp_bwd: unsupported (prec %d, C %d)", prec, C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16) return mlp_bwd_t<bf16_t>(C, xn, dy, w1, w2g, b1, dxn, part, Gacc, Ssum, dW1, db1, R, st, jobs);
  return mlp_bwd_t<f16_t>(C, xn, dy, w1, w2g, b1, dxn, part, Gacc, Ssum, dW1, db1, R, st, jobs);
}
