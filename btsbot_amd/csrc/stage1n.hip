// convnext_nano's stage 1 (7x7 maps, C = 160, two blocks) and the downsample in front of stage 2 as ONE launch
// (gfx950, bf16 / f16, inference):
//
//   2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]  ->  LN + conv 2x2 s2 (160 -> 320)
//
// (timm ConvNeXt stages[1].blocks / stages[2].downsample of `convnext_nano`, the reference classes' default model_kind:
// /root/reference/btsbot/architectures.py:107-108,128,132).  Replaces 2 x (dwconv_ln + fused_mlp) + ln_patch + GEMM of the
// per-op schedule.  The design is stage2p.hip's: one 512-thread workgroup keeps TWO alerts' 98 pixel rows (seven 16-column
// MFMA blocks) resident for the whole stage (one alert per workgroup was the first cut: 149 us per 1024 alerts against 178 for
// the launches it replaces -- every workgroup streams the stage's 1.2 MB of filters, and the per-workgroup phases around the
// chunk loop cost as much as the loop), the residual stream is the fc2 accumulator, and the filters stream past
// as packed MFMA A fragments (launch_pack_s2p: 1 KiB contiguous per wave instruction, straight into registers).
// 160 channels are ten 16-row tiles for eight waves: wave w owns tile w; tiles 8 and 9 are shared by four waves each
// (wave & 1 picks the tile, wave >> 1 the one of a chunk's four fc2 k-steps the wave runs on it), so a shared tile's
// residual is the sum of four accumulators, which meet in the LDS map at every block start.
// HBM sees [49][160] f32 in and [9][320] f32 out per alert.
// OPT-IN (BTSBOT_AMD_STAGE1N=1; held to the oracle by tests/test_gpu_parity.py::test_nano_ragged_batches_match_oracle[*-fused_stage1]):
// 150 us per 1024 alerts against 178 for the launches it replaces and no measurable change of the forward.  In-kernel stamps
// (tools/stamps_nano.py) per workgroup of two alerts: chunk loop 2 x 30 k cycles (6 k per step: every wave reads all seven
// 1 KiB B fragments of a k-step from LDS for ONE 16-row tile -- 504 KB of LDS reads per step on a CU: the design is LDS-bound
// where a wave owns a single row tile; at 256 / 320 channels it owns two / 2.5), downsample 31 k, depthwise 2 x 17 k (60 k
// with the taps read from L2 inside the loop), residual -> map 2 x 12 k, LayerNorm 2 x 8 k, prologue 13 k.
#include <stdlib.h>

#include "common.h"
#include "stage2p.h"

namespace {

template <typename T> struct MQ;
template <typename T> struct MQ16 {
  typedef T frag __attribute__((ext_vector_type(8)));
  typedef T quad __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ frag gld(const void* base, size_t f, int lane) {
    return reinterpret_cast<const frag*>(base)[f * 64 + lane];
  }
  static __device__ __forceinline__ quad pack4(const float (&v)[4]) {
    quad o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
    return o;
  }
};
template <> struct MQ<bf16_t> : MQ16<bf16_t> {
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MQ<f16_t> : MQ16<f16_t> {
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 160, HID = 640, HW = 7, G = 2, PA = HW * HW, NPX = G * PA, NCOL = 112, NB = 7;
constexpr int NT = 512, NW = 8;
constexpr int CHUNK = 128, NCHUNK = HID / CHUNK;            // 5 chunks of 128 hidden units
constexpr int KS1 = C / 32, KS2 = CHUNK / 32, KH = HID / 32; // 5, 4, 20 k-steps
constexpr int CO = 320, KSD = 4 * C / 32;                    // downsample: 320 outputs, K = 640 (20 k-steps)
constexpr int XLP = C;                                       // fp32 map: floats per pixel row
constexpr int XNP = 544, HP = CHUNK * 2 + 32;                // LN image / hidden image: bytes per pixel row (32 mod 256)
// the fp32 map (depthwise / LayerNorm phases) and the two hidden images (chunk loop) are never live together: one region
constexpr int OFF_XN = 0, OFF_XL = OFF_XN + NCOL * XNP, OFF_H = OFF_XL, H_IMG = NCOL * HP;
constexpr int U_BYTES = NCOL * XLP * 4 > 2 * H_IMG ? NCOL * XLP * 4 : 2 * H_IMG;
constexpr int OFF_B1 = OFF_XL + U_BYTES, LDS_BYTES = OFF_B1 + HID * 4;   // 60928 + 71680 + 2560 = 135168
constexpr float LN_EPS = 1e-6f;
#define S1N_STAMP(i)                                                                      \
  do {                                                                                    \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)

__device__ __forceinline__ float half_sum1(float v) {   // sum over the 32 lanes of a half wave
  v = group16_sum(v);
  float w = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return v + w;
}

// depthwise 7x7 on a 7x7 map, output rows R0 .. R1 - 1 of channel c, into registers (the map is updated in place behind
// a barrier of the whole workgroup, by the caller)
template <int R0, int R1>
__device__ __forceinline__ void dw_rows(const float* __restrict__ xl, const float* __restrict__ w, float bias, int c,
                                        float (&o)[3][HW]) {
  constexpr int I0 = R0 - 3 < 0 ? 0 : R0 - 3, I1 = R1 + 3 > HW ? HW : R1 + 3;   // input rows touched
  float in[I1 - I0][HW];
#pragma unroll
  for (int y = I0; y < I1; ++y)
#pragma unroll
    for (int x = 0; x < HW; ++x) in[y - I0][x] = xl[(y * HW + x) * XLP + c];
#pragma unroll
  for (int y = R0; y < R1; ++y)
#pragma unroll
    for (int x = 0; x < HW; ++x) o[y - R0][x] = bias;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const float t = w[(ky * 7 + kx) * C + c];
#pragma unroll
      for (int y = R0; y < R1; ++y) {
        const int iy = y + ky - 3;
        if (iy < 0 || iy >= HW) continue;
#pragma unroll
        for (int x = 0; x < HW; ++x) {
          const int ix = x + kx - 3;
          if (ix >= 0 && ix < HW) o[y - R0][x] = fmaf(t, in[iy - I0][ix], o[y - R0][x]);
        }
      }
    }
}

template <typename T>
__global__ __launch_bounds__(NT, 2) void stage1n_kernel(Stage2pArgs a) {
  using frag = typename MQ<T>::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xl = reinterpret_cast<float*>(smem + OFF_XL);
  unsigned char* xn = smem + OFF_XN;
  unsigned char* hb = smem + OFF_H;
  float* b1s = reinterpret_cast<float*>(smem + OFF_B1);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int alert0 = blockIdx.x * G;                  // two alerts per workgroup
  const int nal = min(G, a.B - alert0), nlive = nal * PA;
  const float* xin = a.x_in + (size_t)alert0 * PA * C;
  // this wave's own tile and the shared one: tile, first channel for this lane, fc2 k-step of a chunk it runs there
  const int c0 = 16 * wave + 4 * kg;
  const int xtile = NW + (wave & 1), cx0 = 16 * xtile + 4 * kg, xq = wave >> 1;

  S1N_STAMP(0);
  for (int i = tid; i < (NCOL - NPX) * XNP / 4; i += NT) reinterpret_cast<unsigned*>(xn + NPX * XNP)[i] = 0u;
  for (int i = tid; i < NCOL * XLP; i += NT) xl[i] = 0.f;
  f32x4 acc[NB], accx[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int p = 16 * n + col;
    acc[n] = accx[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p < nlive) {
      acc[n] = *reinterpret_cast<const f32x4*>(xin + p * C + c0);
      if (xq == 0) accx[n] = *reinterpret_cast<const f32x4*>(xin + p * C + cx0);
    }
  }
  __syncthreads();   // zero fill done before the first map is written

  // the four partial residuals of the shared tiles (or one complete map), one after the other, into xl
  auto residual_to_map = [&]() {
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (16 * n + col < NPX) *reinterpret_cast<f32x4*>(xl + (16 * n + col) * XLP + c0) = acc[n];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (xq == q) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
          if (16 * n + col < NPX) {
            f32x4* d = reinterpret_cast<f32x4*>(xl + (16 * n + col) * XLP + cx0);
            *d = q == 0 ? accx[n] : *d + accx[n];
          }
      }
      __syncthreads();
    }
  };
  // LayerNorm of the map's live pixel rows into the 16-bit image: half wave = pixel, lane = channels 4 l .. + 3 and 128 + l
  auto layernorm = [&](const float* lnw, const float* lnb) {
    const int l = lane & 31;
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(lnw + 4 * l), b4 = *reinterpret_cast<const f32x4*>(lnb + 4 * l);
    const float wx = lnw[128 + l], bx = lnb[128 + l];
    for (int p = 2 * wave + (lane >> 5); p < NPX; p += 2 * NW) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(xl + p * XLP + 4 * l);
      const float dx = xl[p * XLP + 128 + l];
      const float mean = half_sum1(d[0] + d[1] + d[2] + d[3] + dx) * (1.0f / C);
      const f32x4 e = d - mean;
      const float ex = dx - mean;
      const float var = half_sum1(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3] + ex * ex) * (1.0f / C);
      const float rstd = rsqrtf(var + LN_EPS);
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = e[i] * rstd * w4[i] + b4[i];
      *reinterpret_cast<typename MQ<T>::quad*>(xn + p * XNP + 8 * l) = MQ<T>::pack4(y);
      *reinterpret_cast<T*>(xn + p * XNP + (128 + l) * 2) = (T)(ex * rstd * wx + bx);
    }
  };

#pragma unroll 1
  for (int j = 0; j < a.depth; ++j) {
    const Stage2pBlk& bk = a.blk[j];
    S1N_STAMP(1 + 5 * j);
    // the block's 49 x 160 taps into the LN image's bytes (dead until this block's LayerNorm): read from HBM / L2 inside
    // the depthwise loop every tap was an exposed memory round trip -- 60 k cycles per block for 1.1 k FMAs per thread
    float* taps = reinterpret_cast<float*>(xn);
    static_assert(49 * C * 4 <= NPX * XNP, "the taps fit the live rows of the LN image");
    for (int i = tid; i < 49 * C / 4; i += NT)
      reinterpret_cast<f32x4*>(taps)[i] = reinterpret_cast<const f32x4*>(bk.dw_w)[i];
    residual_to_map();   // (its barriers also publish the taps)
    S1N_STAMP(2 + 5 * j);
    // ---- depthwise 7x7 in place: thread = (channel, row group: rows 0-2 / 3-4 / 5-6); 480 of the 512 threads
    {
      const int dc = tid % C, rg = tid / C;
      const float dbias = bk.dw_b[dc];
#pragma unroll 1
      for (int al = 0; al < G; ++al) {
        float* xa = xl + al * PA * XLP;
        float o[3][HW];
        int r0 = 0, nr = 0;
        if (rg == 0) {
          dw_rows<0, 3>(xa, taps, dbias, dc, o);
          nr = 3;
        } else if (rg == 1) {
          dw_rows<3, 5>(xa, taps, dbias, dc, o);
          r0 = 3, nr = 2;
        } else if (rg == 2) {
          dw_rows<5, 7>(xa, taps, dbias, dc, o);
          r0 = 5, nr = 2;
        }
        __syncthreads();   // every thread holds its outputs: the map may change
#pragma unroll
        for (int y = 0; y < 3; ++y)
          if (y < nr) {
#pragma unroll
            for (int x = 0; x < HW; ++x) xa[((r0 + y) * HW + x) * XLP + dc] = o[y][x];
          }
      }
    }
    for (int i = tid; i < HID; i += NT) b1s[i] = bk.b1[i];
    __syncthreads();
    S1N_STAMP(3 + 5 * j);
    layernorm(bk.ln_w, bk.ln_b);
    // residual + gamma * b2 (the bias of the folded fc2); the shared tiles' lead waves carry theirs
    {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(bk.gamma + c0), b4 = *reinterpret_cast<const f32x4*>(bk.b2 + c0);
      const f32x4 gx4 = *reinterpret_cast<const f32x4*>(bk.gamma + cx0), bx4 = *reinterpret_cast<const f32x4*>(bk.b2 + cx0);
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        acc[n] += g4 * b4;
        if (xq == 0) accx[n] += gx4 * bx4;
      }
    }
    // this block's chunk 0 of both filter streams (not carried through the phases above: two blocks only)
    frag a1[KS1], a2[KS2], a2x;
#pragma unroll
    for (int s = 0; s < KS1; ++s) a1[s] = MQ<T>::gld(bk.w1p, (size_t)wave * KS1 + s, lane);
    // (fc2 k-steps in the order xq, xq + 1, ... of a chunk: the shared tile's one k-step is then this wave's program step 0)
#pragma unroll
    for (int s = 0; s < KS2; ++s) a2[s] = MQ<T>::gld(bk.w2p, (size_t)wave * KH + ((xq + s) & 3), lane);
    a2x = MQ<T>::gld(bk.w2p, (size_t)xtile * KH + xq, lane);
    __syncthreads();   // LN image complete (and the map is read out: the hidden images take its place)
    S1N_STAMP(4 + 5 * j);

    // One step = fc1 of chunk ch, then GELU of chunk ch between the fc2 products of chunk ch - 1 (stage2p.hip's pipeline);
    // P = the hidden image this chunk writes.  Fragment slots are refilled in place with the next chunk's.
    auto step = [&](auto P, auto FIRST, int ch) {
      constexpr int p = decltype(P)::value;
      constexpr bool first = decltype(FIRST)::value;
      const int nch = ch + 1 < NCHUNK ? ch + 1 : ch;   // (the last step re-reads its own fragments: no branch in the loops)
      const size_t f1 = (size_t)(nch * NW + wave) * KS1;
      const size_t f2 = (size_t)wave * KH + ch * KS2, fx = (size_t)xtile * KH + ch * KS2 + xq;
      f32x4 hacc[NB];
      {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + ch * CHUNK + 16 * wave + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) hacc[n] = bv;
      }
      frag xb[2][NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) xb[0][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (8 * kg) * 2);
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        if (s + 1 < KS1) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            xb[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (32 * (s + 1) + 8 * kg) * 2);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) hacc[n] = MQ<T>::run(a1[s], xb[s & 1][n], hacc[n]);
        a1[s] = MQ<T>::gld(bk.w1p, f1 + s, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
      unsigned char* hcur = hb + p * H_IMG;
      const unsigned char* hprev = hb + (1 - p) * H_IMG;
      frag hbf[2][NB];
      if (!first) {
#pragma unroll
        for (int n = 0; n < NB; ++n) hbf[0][n] = *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (32 * xq + 8 * kg) * 2);
      }
#pragma unroll
      for (int s = 0; s < NB; ++s) {   // one GELU column block per iteration, an fc2 k-step in the first four
        if (!first && s < KS2) {
          if (s + 1 < KS2) {
#pragma unroll
            for (int n = 0; n < NB; ++n)
              hbf[(s + 1) & 1][n] =
                  *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (32 * ((xq + s + 1) & 3) + 8 * kg) * 2);
          }
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[n] = MQ<T>::run(a2[s], hbf[s & 1][n], acc[n]);
          a2[s] = MQ<T>::gld(bk.w2p, f2 + ((xq + s) & 3), lane);
          if (s == 0) {   // the shared tile: k-step xq of the previous chunk
#pragma unroll
            for (int n = 0; n < NB; ++n) accx[n] = MQ<T>::run(a2x, hbf[0][n], accx[n]);
            a2x = MQ<T>::gld(bk.w2p, fx, lane);
          }
        }
        {   // GELU of column block s -> image hb[p] [pixel][hidden]: rows 4 kg .. + 3 of hidden tile `wave`
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = gelu_for<T>(hacc[s][r]);
          *reinterpret_cast<typename MQ<T>::quad*>(hcur + (16 * s + col) * HP + (16 * wave + 4 * kg) * 2) = MQ<T>::pack4(hv);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();   // image hb[p] complete; hb[1 - p] is read out
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    static_assert(NCHUNK == 5 && KS2 <= NB, "five steps; the fc2 k-steps fit the GELU column blocks' iterations");
    step(P0{}, std::true_type{}, 0);
    step(P1{}, std::false_type{}, 1);
    S1N_STAMP(5 + 5 * j);
    step(P0{}, std::false_type{}, 2);
    step(P1{}, std::false_type{}, 3);
    step(P0{}, std::false_type{}, 4);
    {   // fc2 of the last chunk (image 0)
      const unsigned char* hprev = hb;
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        frag hbf[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n)
          hbf[n] = *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (32 * ((xq + s) & 3) + 8 * kg) * 2);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = MQ<T>::run(a2[s], hbf[n], acc[n]);
        if (s == 0) {
#pragma unroll
          for (int n = 0; n < NB; ++n) accx[n] = MQ<T>::run(a2x, hbf[n], accx[n]);
        }
      }
    }
    __syncthreads();   // the images are read out (the map takes their place again)
  }

  // ---- stage output, then the downsample: LN per pixel + conv 2x2 s2 -> 3x3 pixels of 320 channels
  S1N_STAMP(11);
  residual_to_map();
  if (a.tap_stage != nullptr) {
    for (int i = tid; i < nlive * (C / 4); i += NT) {
      const int p = i / (C / 4), c4 = i - p * (C / 4);
      *reinterpret_cast<f32x4*>(a.tap_stage + ((size_t)alert0 * PA + p) * C + 4 * c4) = *reinterpret_cast<const f32x4*>(xl + p * XLP + 4 * c4);
    }
  }
  layernorm(a.ds_lnw, a.ds_lnb);
  __syncthreads();
  S1N_STAMP(12);
  {
    // out[opix][co] = b[co] + sum_k Wd[co][k] patch[opix][k],  k = (2 ky + kx) * 160 + c  ->  pixel (2 oy + ky, 2 ox + kx).
    // Columns = the two alerts' 9 output pixels each (18 live of 32); wave w: output tiles w, w + 8, w + 16 (20 tiles of
    // 16 channels), 20 k-steps each
    int pbase[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int o18 = 16 * nb + col, ov = o18 < 9 * G ? o18 : 0, al = ov / 9, op = ov - 9 * al, oy = op / 3, ox = op - 3 * oy;
      pbase[nb] = al * PA + 2 * oy * HW + 2 * ox;
    }
#pragma unroll 1
    for (int t = wave; t < CO / 16; t += NW) {
      frag wq[KSD];
#pragma unroll
      for (int s = 0; s < KSD; ++s) wq[s] = MQ<T>::gld(a.ds_wp, (size_t)t * KSD + s, lane);
      f32x4 o[2];
      o[0] = o[1] = *reinterpret_cast<const f32x4*>(a.ds_b + 16 * t + 4 * kg);
#pragma unroll
      for (int s = 0; s < KSD; ++s) {
        const int q = s / KS1, dp = (q >> 1) * HW + (q & 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const frag bf = *reinterpret_cast<const frag*>(xn + (pbase[nb] + dp) * XNP + (32 * (s - q * KS1) + 8 * kg) * 2);
          o[nb] = MQ<T>::run(wq[s], bf, o[nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int o18 = 16 * nb + col;
        if (o18 < 9 * nal) *reinterpret_cast<f32x4*>(a.out + ((size_t)alert0 * 9 + o18) * CO + 16 * t + 4 * kg) = o[nb];
      }
    }
  }
  S1N_STAMP(13);
}

}  // namespace

bool stage1n_supported(int prec, int c1, int c2, int depth) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && c1 == C && c2 == CO && depth == 2;
}

// x_in [B][49][160] f32 -> out [B][9][320] f32; blk[0..1] / ds_* as for launch_stage2p (fragments from launch_pack_s2p)
int launch_stage1n(int prec, const Stage2pArgs& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (!stage1n_supported(prec, C, CO, a.depth)) {
    btsbot_set_error("stage1n: precision %d / depth %d not supported", prec, a.depth);
    return BTSBOT_ERR_INVALID_ARG;
  }
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(stage1n_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(stage1n_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr.done();
  }
  const dim3 grid((a.B + G - 1) / G);
  if (prec == BTSBOT_BF16) hipLaunchKernelGGL(stage1n_kernel<bf16_t>, grid, dim3(NT), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(stage1n_kernel<f16_t>, grid, dim3(NT), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
