// MaxViT MBConv front half in one kernel (gfx950, 16-bit modes, wide stages):
//
//   m2 = silu(BN2(dw3x3_s(silu(BN1(conv1_1x1(xn))))))        + per-tile partial sums of the SE pool
//
// timm MbConvBlock.conv1_1x1 / norm1 / conv2_kxk / norm2 (reached from
// /root/reference/btsbot/architectures.py:51,97).  Unfused, the expanded map m1 (4 x C_in channels at the
// INPUT resolution: 6.4 MB per alert in the first block) is written by the 1x1 GEMM and read back by the
// depthwise kernel -- 40 % of the whole network's HBM traffic.  Here a workgroup owns one output tile of one
// alert: it stages the tile's input halo (xn, C_in channels) in LDS once, then for every chunk of 64
// expanded channels
//   phase 1  m1 chunk [halo pixels][64] = silu(W1' . xn + b1')   v_mfma_f32_16x16x32, filter rows = A operand
//            from L2, pixel rows = B operand from the LDS halo; pixels outside the image are written as 0
//            (the depthwise conv pads m1, not xn); the chunk lands in a second LDS image
//   phase 2  depthwise 3x3 (stride 1 / 2) + folded BN2 + SiLU from that image, 8 channels per thread with the
//            9 x 8 taps in registers; 16-byte stores of m2; the thread's outputs are summed for the
//            squeeze-excite pool and reduced per workgroup -> part [B][tiles][MID]
// so m1 never leaves the CU.  Tiles: stride 2 -> 7x7 outputs (15x15 halo = 15 MFMA pixel tiles), stride 1 ->
// 14x14 outputs (16x16 halo = 16 pixel tiles exactly); halo recompute costs 15 % / 30 % more conv1 work.
#include "maxvit.h"

namespace {

template <typename T> struct FM;
template <> struct FM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct FM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <int STRIDE> struct Geo {
  static constexpr int TO = STRIDE == 2 ? 7 : 14;          // output tile side
  static constexpr int HALO = STRIDE == 2 ? 15 : 16;       // input tile side
  static constexpr int NPX = HALO * HALO;                   // 225 / 256
  static constexpr int PT = (NPX + 15) / 16;                // MFMA pixel tiles: 15 / 16
  static constexpr int NPXP = PT * 16;                      // padded pixel count
  static constexpr int NOUT = TO * TO;                      // 49 / 196
};

constexpr int CH = 64;                    // expanded channels per chunk
constexpr int M1PITCH = CH * 2 + 16;      // bytes per pixel row of the m1 chunk image

template <typename T, int STRIDE, int CIN>
__global__ __launch_bounds__(256) void mv_mbconv_front_kernel(const T* __restrict__ xn,
                                                              const T* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ w9,
                                                              const float* __restrict__ b2,
                                                              T* __restrict__ m2,
                                                              float* __restrict__ part, int H, int MID) {
  using G = Geo<STRIDE>;
  using frag = typename FM<T>::frag;
  constexpr int KS = CIN / 32;
  constexpr int XPITCH = CIN * 2 + 16;    // bytes per pixel row of the input halo image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                   // [NPXP][XPITCH]
  unsigned char* m1s = smem + G::NPXP * XPITCH;               // [NPXP][M1PITCH]
  float* red = reinterpret_cast<float*>(m1s + G::NPXP * M1PITCH);   // [32][CH] pool partials

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int Ho = H / STRIDE, TPR = Ho / G::TO;                // tiles per row
  const int tile = blockIdx.x, ty = tile / TPR, tx = tile - ty * TPR;
  const long b = blockIdx.y;
  const int iy0 = ty * G::TO * STRIDE - 1, ix0 = tx * G::TO * STRIDE - 1;   // halo origin in the input map

  // ---- stage the input halo (zeros outside the image and in the padding rows)
  constexpr int CPR = CIN / 8;                                // 16-byte chunks per pixel
  {
    // all of a thread's pieces are requested before the first is stored (a load under a branch, or one per trip of a
    // rolled loop, is waited for on the spot: eight HBM round trips in a row instead of one): the address is clamped
    // into the image, the value zeroed by a select
    constexpr int TRIPS = (G::NPXP * CPR + 255) / 256;
    uint4 hv[TRIPS];
    bool ok[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int i = tid + 256 * t;
      const int px = i / CPR, ck = i - px * CPR;
      const int hy = px / G::HALO, hx = px - hy * G::HALO;
      const int iy = iy0 + hy, ix = ix0 + hx;
      ok[t] = px < G::NPX && iy >= 0 && iy < H && ix >= 0 && ix < H;
      const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), H - 1);
      hv[t] = *reinterpret_cast<const uint4*>(xn + ((b * H + cy) * H + cx) * CIN + ck * 8);
    }
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      asm volatile("" : "+v"(hv[t].x), "+v"(hv[t].y), "+v"(hv[t].z), "+v"(hv[t].w));   // (keeps the load unconditional)
      if (!ok[t]) hv[t] = make_uint4(0, 0, 0, 0);
      const int i = tid + 256 * t;
      const int px = i / CPR, ck = i - px * CPR;
      if (i < G::NPXP * CPR) *reinterpret_cast<uint4*>(xs + px * XPITCH + ck * 16) = hv[t];
    }
  }
  __syncthreads();

  // which halo pixels of this lane's column (pixel l15 of each pixel tile) lie inside the image
  // (bit pt of `inside`): the depthwise conv zero-pads m1
  unsigned inside = 0;
#pragma unroll
  for (int pt = 0; pt < G::PT; ++pt) {
    const int px = pt * 16 + l15;
    const int hy = px / G::HALO, hx = px - hy * G::HALO;
    const int iy = iy0 + hy, ix = ix0 + hx;
    if (px < G::NPX && iy >= 0 && iy < H && ix >= 0 && ix < H) inside |= 1u << pt;
  }

  const int cg = tid & 7;                                     // phase 2: this thread's 8-channel group
  typedef T __attribute__((ext_vector_type(8))) T8;
  // filter fragments / biases of phase 1 for the current chunk (the next chunk's are requested before
  // phase 2, whose VALU work hides the L2 latency; phase 2's taps are requested before phase 1)
  frag wf[4][KS];
  float4 bv[4];
  auto load_p1 = [&](int ch0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wf[mt][ks] = *reinterpret_cast<const frag*>(w1 + (size_t)(ch0 + mt * 16 + l15) * CIN + ks * 32 +
                                                    g * 8);
      bv[mt] = *reinterpret_cast<const float4*>(b1 + ch0 + mt * 16 + 4 * g);
    }
  };
  load_p1(0);
  const int nchunk = MID / CH;
  for (int cc = 0; cc < nchunk; ++cc) {
    const int ch0 = cc * CH;
    const int c = ch0 + cg * 8;
    // ---- phase 2 operands first (taps + bias of this thread's 8 channels)
    float4 wq[9][2], bq[2];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      wq[t][0] = *reinterpret_cast<const float4*>(w9 + (size_t)t * MID + c);
      wq[t][1] = *reinterpret_cast<const float4*>(w9 + (size_t)t * MID + c + 4);
    }
    bq[0] = *reinterpret_cast<const float4*>(b2 + c);
    bq[1] = *reinterpret_cast<const float4*>(b2 + c + 4);
    // ---- phase 1: m1 chunk = silu(W1' xn + b1'), wave w takes pixel tiles w, w+4, ...
    for (int pt = wave; pt < G::PT; pt += 4) {
      frag xf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(xs + (pt * 16 + l15) * XPITCH + ks * 64 + g * 16);
      const float inf = ((inside >> pt) & 1u) ? 1.0f : 0.0f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = FM<T>::run(wf[mt][ks], xf[ks], acc);
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 v;
        // (a factor, not a branch: straight-line code lets the next tile's MFMAs issue under this one's SiLU; the
        //  adds and multiplies of silu_fast as packed fp32 pairs -- the same operations in the same order)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 inf2 = f2{inf, inf};
        auto silu2 = [&](f2 x) {
          const f2 t = x * f2{-1.4426950408889634f, -1.4426950408889634f};
          const f2 d = f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f2{1.0f, 1.0f};
          return x * f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)} * inf2;
        };
        const f2 s01 = silu2(f2{acc[0], acc[1]} + f2{bv[mt].x, bv[mt].y});
        const f2 s23 = silu2(f2{acc[2], acc[3]} + f2{bv[mt].z, bv[mt].w});
        v[0] = (T)s01.x;
        v[1] = (T)s01.y;
        v[2] = (T)s23.x;
        v[3] = (T)s23.y;
        *reinterpret_cast<T4*>(m1s + (pt * 16 + l15) * M1PITCH + (mt * 16 + 4 * g) * 2) = v;
      }
    }
    // phase 2's taps arrived long ago; naming them here puts their wait BEFORE the next chunk's requests -- behind them
    // it would be a wait for everything (loads and this kernel's stores share one counter, which then cannot count)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      asm volatile("" ::"v"(wq[t][0].x), "v"(wq[t][0].y), "v"(wq[t][0].z), "v"(wq[t][0].w), "v"(wq[t][1].x), "v"(wq[t][1].y),
                   "v"(wq[t][1].z), "v"(wq[t][1].w));
    asm volatile("" ::"v"(bq[0].x), "v"(bq[0].y), "v"(bq[0].z), "v"(bq[0].w), "v"(bq[1].x), "v"(bq[1].y), "v"(bq[1].z),
                 "v"(bq[1].w));
    if (cc + 1 < nchunk) load_p1(ch0 + CH);      // in flight during phase 2
    __syncthreads();
    // ---- phase 2: depthwise 3x3 + BN2 + SiLU on the chunk; items = (output pixel, 8-channel group)
    {
      float psum[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) psum[e] = 0.f;
      // (fixed trips, the last one ragged and predicated: a rolled loop's header waits for every load in flight -- the
      //  next chunk's filter fragments, requested just above to arrive DURING this phase)
      constexpr int P2T = (G::NOUT + 31) / 32;
#pragma unroll
      for (int t2 = 0; t2 < P2T; ++t2) {
        const int op = (tid >> 3) + 32 * t2;
        if (op >= G::NOUT) break;
        const int oy = op / G::TO, ox = op - oy * G::TO;
        // (the 9 x 8 products as packed fp32 FMAs, two channels per instruction)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 acc[4] = {f2{bq[0].x, bq[0].y}, f2{bq[0].z, bq[0].w}, f2{bq[1].x, bq[1].y}, f2{bq[1].z, bq[1].w}};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int px = (oy * STRIDE + ky) * G::HALO + ox * STRIDE + kx;
            const T8 v = *reinterpret_cast<const T8*>(m1s + px * M1PITCH + cg * 16);
            const float4 wa = wq[ky * 3 + kx][0], wb = wq[ky * 3 + kx][1];
            acc[0] = f2{(float)v[0], (float)v[1]} * f2{wa.x, wa.y} + acc[0];
            acc[1] = f2{(float)v[2], (float)v[3]} * f2{wa.z, wa.w} + acc[1];
            acc[2] = f2{(float)v[4], (float)v[5]} * f2{wb.x, wb.y} + acc[2];
            acc[3] = f2{(float)v[6], (float)v[7]} * f2{wb.z, wb.w} + acc[3];
          }
        T8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (T)silu_fast(acc[e / 2][e & 1]);
          psum[e] += (float)o[e];
        }
        const long orow = (b * Ho + ty * G::TO + oy) * Ho + tx * G::TO + ox;
        *reinterpret_cast<T8*>(m2 + orow * MID + c) = o;
      }
      // pool partials: 32 threads share a channel group
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(tid >> 3) * CH + cg * 8 + e] = psum[e];
    }
    __syncthreads();
    if (tid < CH) {
      float a = 0.f;
      for (int q = 0; q < 32; ++q) a += red[q * CH + tid];
      part[(b * gridDim.x + tile) * MID + ch0 + tid] = a;
    }
    // (the next chunk's phase 1 rewrites m1s: phase 2 above finished reading it at the barrier; `red` is
    //  rewritten only after the next chunk's phase-1 barrier)
  }
}

template <typename T, int STRIDE, int CIN>
int launch_front(const void* xn, const void* w1, const float* b1, const float* w9, const float* b2,
                 void* m2, float* part, int B, int H, int MID, hipStream_t st) {
  using G = Geo<STRIDE>;
  constexpr size_t lds = (size_t)G::NPXP * (CIN * 2 + 16) + (size_t)G::NPXP * M1PITCH + 32 * CH * 4;
  auto kern = mv_mbconv_front_kernel<T, STRIDE, CIN>;
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr.done();
  }
  const int Ho = H / STRIDE, tiles = (Ho / G::TO) * (Ho / G::TO);
  hipLaunchKernelGGL(kern, dim3(tiles, B), dim3(256), lds, st, reinterpret_cast<const T*>(xn),
                     reinterpret_cast<const T*>(w1), b1, w9, b2, reinterpret_cast<T*>(m2), part, H, MID);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T>
int launch_front_t(const void* xn, const void* w1, const float* b1, const float* w9, const float* b2,
                   void* m2, float* part, int B, int H, int CIN, int MID, int stride, hipStream_t st) {
  if (stride == 2 && CIN == 64) return launch_front<T, 2, 64>(xn, w1, b1, w9, b2, m2, part, B, H, MID, st);
  if (stride == 1 && CIN == 64) return launch_front<T, 1, 64>(xn, w1, b1, w9, b2, m2, part, B, H, MID, st);
  if (stride == 2 && CIN == 128) return launch_front<T, 2, 128>(xn, w1, b1, w9, b2, m2, part, B, H, MID, st);
  return launch_front<T, 1, 128>(xn, w1, b1, w9, b2, m2, part, B, H, MID, st);
}

}  // namespace

bool mv_mbconv_front_supported(int prec, int H, int CIN, int MID, int stride) {
  if (prec != BTSBOT_BF16 && prec != BTSBOT_F16) return false;
  if (CIN != 64) return false;          // C_in = 128 (stage 1, second block) measured slower than the unfused pair
  if (MID % 64 != 0 || (stride != 1 && stride != 2) || H % stride != 0) return false;
  const int Ho = H / stride, to = stride == 2 ? 7 : 14;
  return Ho % to == 0 && Ho >= 28;      // the wide maps only: narrow stages move little m1 traffic
}

int mv_mbconv_front_tiles(int H, int stride) {
  const int Ho = H / stride, to = stride == 2 ? 7 : 14;
  return (Ho / to) * (Ho / to);
}

int launch_mv_mbconv_front(int prec, const void* xn, const void* w1, const float* b1, const float* w9,
                           const float* b2, void* m2, float* part, int B, int H, int CIN, int MID,
                           int stride, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (!mv_mbconv_front_supported(prec, H, CIN, MID, stride)) {
    btsbot_set_error("mv_mbconv_front: unsupported (prec %d, H %d, CIN %d, MID %d, stride %d)", prec, H,
                     CIN, MID, stride);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16)
    return launch_front_t<bf16_t>(xn, w1, b1, w9, b2, m2, part, B, H, CIN, MID, stride, st);
  return launch_front_t<f16_t>(xn, w1, b1, w9, b2, m2, part, B, H, CIN, MID, stride, st);
}
