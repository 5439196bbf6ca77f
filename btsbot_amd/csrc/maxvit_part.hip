// A MaxViT partition block as ONE kernel (gfx950, 16-bit modes; C = 128 / 256 here, C = 64 in the second half of the file):
//
//   x += proj( MHSA_7x7( qkv( LN1(x) ) ) )            the attention half
//   x += fc2( gelu( fc1( LN2(x) ) ) )                  the MLP half (MLP = true)
//
// timm PartitionAttentionCl (norm1, attn.qkv, rel-pos attention over a 7x7 window / the 7x7 dilated grid, attn.proj,
// norm2, mlp.fc1, GELU, mlp.fc2 and both residual adds), reached from /root/reference/btsbot/architectures.py:51,97.
// Unfused the attention half alone is four launches (LayerNorm, qkv GEMM, attention, proj GEMM) that move 7.5 KB per
// 256-channel row through HBM -- the qkv rows written and read back are 3 KB of them; fused, the partition's fp32 rows are
// read once and written once.
//
// The design is stage2p.hip's: the rows stay on the CU, the filters stream past them from L2 as packed MFMA A fragments
// (1 KiB contiguous per wave instruction, launch_pack_s2p), the residual stream IS the accumulator of proj and fc2.
// A workgroup of HEADS = C / 32 waves owns one partition (49 tokens, padded to 64 = four 16-column MFMA blocks); wave wo
// owns its residual rows' channels 32 wo .. 32 wo + 31 and its head wo.  C = 256: 8 waves, 155 KB of LDS, one workgroup
// per CU; C = 128: 4 waves, 80 KB, two per CU, which run into each other's barriers and load latencies.  (PARTS > 1 puts
// PARTS neighbouring partitions of one alert into a workgroup: wave (wt, wo) = (wave / HEADS, wave % HEADS).)
//   phase 0  the rows (an index map on the row address: window or grid) -> residual registers
//            acc[m][n][r] = x[partition wt, token 16 n + col][channel 32 wo + 16 m + 4 kg + r]
//   phase 1  LayerNorm in registers: a wave reduces its 32 channels of a token to (mean, centred squares) -- over the lane's
//            8 channels, then over kg by permlane swaps --, the HEADS pairs meet in a small LDS table and combine exactly
//            (one exchange, one barrier) -> 16-bit image xn
//   phase 2  qkv^T = Wqkv . xn^T + b in three chunks of QT = 2 output tiles per wave, the next chunk's fragments
//            requested as the current ones are used -> 16-bit image IMG [token][3C]
//   phase 3  attention of head wo: S^T = K Q^T (16 MFMAs), softmax over the keys in registers (bias and key mask in lane
//            order), O^T = V^T P^T (16 MFMAs, V^T by ds_read_b64_tr_b16); O overwrites the head's own Q columns -- two
//            passes of two query tiles
//   phase 4  residual += Wproj . O + b: k-step = head (its 32 O columns are contiguous in IMG)
//   phase 5  (MLP) LayerNorm (norm2) as phase 1 -> xn; 128 hidden units per step: fc1 (HT tiles per wave) + GELU -> one
//            of two hidden images (they overlay IMG), one barrier, fc2 into the residual
//   phase 6  rows back to x (tokens 49..63 of the padded tile repeat token 48, are masked as keys and never stored); behind
//            a block's grid half also the next block's pre-norm copy (T)(x * s + t) (post_out)
#include "maxvit.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <typename T> struct PM;
template <> struct PM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct PM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int NB = 4;
constexpr float LN_EPS = 1e-6f;

// sum over the wave's four 16-lane rows (the lanes that share lane & 15); every lane ends with the total.  The rows meet
// through v_permlane16_swap / v_permlane32_swap as in common.h's wave_sum (inline asm for the reason given there).
__device__ __forceinline__ float rows_sum(float v) {
  float w = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  v += w;
  w = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return v + w;
}
__device__ __forceinline__ float rows_max(float v) {
  float w = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  v = fmaxf(v, w);
  w = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return fmaxf(v, w);
}

template <int C_, int PARTS_ = 1> struct PG {
  static constexpr int C = C_, HEADS = C / 32, PARTS = PARTS_, NW = HEADS * PARTS, NT = 64 * NW, NTOK = 64 * PARTS;
  static constexpr int C3 = 3 * C, KS = C / 32;
  static constexpr int QT = 2;                               // qkv output tiles per wave per chunk: every wave reads the whole xn
                                                             // image per chunk, so two tiles halve the LDS traffic of one
  static constexpr int NCHQ = C3 / 16 / (HEADS * QT);        // qkv chunks: 3
  static constexpr int HT = 8 / HEADS;                       // fc1 hidden tiles per wave per step (8 fragments; two tiles at C = 256
                                                             // halve fc1's LDS reads but spill: 623 -> 735 us per launch)
  static constexpr int HCH = HT * HEADS * 16;                // hidden units per MLP step: 128
  static constexpr int KF = HCH / 32;                        // fc2 k-steps per MLP step
  static constexpr int HP = HCH * 2 + 32;                    // bytes per token row of a hidden image (288)
  static constexpr int NCHM = 4 * C / HCH;                   // MLP steps
  static constexpr int IP = C3 * 2 + 32;                     // bytes per token row of the qkv image (32 mod 256: conflict-free)
  static constexpr int XNP = C * 2 + 32;                     // ... of the LayerNorm image
  static constexpr int OFF_IMG = 0, IMG_BYTES = NTOK * IP;
  static constexpr int OFF_XN = OFF_IMG + IMG_BYTES, XN_BYTES = NTOK * XNP;
  static constexpr int OFF_RED = OFF_XN + XN_BYTES, RED_BYTES = 2 * HEADS * NTOK * 8;   // [2 (LN1 | LN2)][HEADS][NTOK] (mean, M2)
  // the per-channel constants, staged once per workgroup (a load from L2 at the point of use costs its whole latency):
  // norm1 w, b | qkv bias [3C] | proj bias | norm2 w, b | fc1 bias [4C] | fc2 bias
  static constexpr int K_LN1W = 0, K_LN1B = C, K_BQKV = 2 * C, K_BPROJ = 5 * C, K_LN2W = 6 * C, K_LN2B = 7 * C, K_B1 = 8 * C,
                       K_B2 = 12 * C, K_PS = 13 * C, K_PB = 14 * C, K_N = 15 * C;
  static constexpr int OFF_K = OFF_RED + RED_BYTES;
  static constexpr int LDS_BYTES = OFF_K + K_N * 4;
  static constexpr int WGS = NW <= 4 ? 2 : 1;                // workgroups per CU (8 waves, 256 registers each, either way)
  static_assert(NW <= 8 && NCHM * HCH == 4 * C && NCHQ * HEADS * QT * 16 == C3 && KF % 2 == 0, "whole chunks");
  static_assert(IP % 256 == 32 && XNP % 256 == 32 && HP % 256 == 32, "image pitches");
  static_assert(2 * NTOK * HP <= IMG_BYTES, "the hidden images fit under the qkv image");
  static_assert(WGS * LDS_BYTES <= 160 * 1024, "one CU");
};

struct PartArgs {
  float* x;
  const float *ln1w, *ln1b, *bqkv, *bproj, *biasl;
  const void *wqkvp, *wprojp;
  const float *ln2w, *ln2b, *b1, *b2;
  const void *w1p, *w2p;
  const float *post_s, *post_b;   // with post_out: the rows are also written as (T)(x * post_s[c] + post_b[c]) -- the next
  void* post_out;                 // block's pre-norm BatchNorm copy, from the registers that hold the finished rows
  int H, grid_mode, units;
  unsigned long long* stamps;   // developer diagnostic (tools/stamps_maxvit.py): phase clocks of workgroup 0's first unit
};

#define PT_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (a.stamps != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && u == blockIdx.x) a.stamps[i] = clock64();    \
  } while (0)

template <typename T, int C_, bool MLP>
__global__ __launch_bounds__(PG<C_>::NT, 2) void mv_part_kernel(PartArgs a) {
  using P = PG<C_>;
  constexpr int C = C_, HEADS = P::HEADS, PARTS = P::PARTS, KS = P::KS, QT = P::QT, NCHQ = P::NCHQ, HT = P::HT;
  constexpr int NCHM = P::NCHM, IP = P::IP, XNP = P::XNP, NTOK = P::NTOK, HCH = P::HCH, HP = P::HP, KF = P::KF;
  using frag = typename PM<T>::frag;
  typedef T __attribute__((ext_vector_type(4))) T4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kg = lane >> 4;
  const int wo = wave % HEADS, wt = wave / HEADS;
  unsigned char* img = smem + P::OFF_IMG + wt * 64 * IP;     // this wave's partition: its rows of the qkv image,
  unsigned char* xn = smem + P::OFF_XN + wt * 64 * XNP;      // of the LayerNorm image,
  unsigned char* hb = smem + P::OFF_IMG + wt * 64 * HP;      // of hidden image 0 (image 1: + NTOK * HP)
  float* red = reinterpret_cast<float*>(smem + P::OFF_RED) + wt * 128;  // LayerNorm exchange tables [2][HEADS][NTOK] float2
  const int H = a.H, G = H / 7, nW = G * G;
  const frag* wq0 = reinterpret_cast<const frag*>(a.wqkvp) + lane;   // fragment f at [f * 64]
  const frag* wp0 = reinterpret_cast<const frag*>(a.wprojp) + lane;
  const frag* w10 = reinterpret_cast<const frag*>(a.w1p) + lane;
  const frag* w20 = reinterpret_cast<const frag*>(a.w2p) + lane;
  const f32x4* bp0 = reinterpret_cast<const f32x4*>(a.biasl) + (size_t)wo * 16 * 64 + lane;

  const float* kc = reinterpret_cast<const float*>(smem + P::OFF_K);
  {
    float* kw = reinterpret_cast<float*>(smem + P::OFF_K);
    for (int i = tid; i < C; i += P::NT) {
      kw[P::K_LN1W + i] = a.ln1w[i];
      kw[P::K_LN1B + i] = a.ln1b[i];
      kw[P::K_BPROJ + i] = a.bproj[i];
      if (a.post_out != nullptr) {
        kw[P::K_PS + i] = a.post_s[i];
        kw[P::K_PB + i] = a.post_b[i];
      }
      if constexpr (MLP) {
        kw[P::K_LN2W + i] = a.ln2w[i];
        kw[P::K_LN2B + i] = a.ln2b[i];
        kw[P::K_B2 + i] = a.b2[i];
      }
    }
    for (int i = tid; i < 3 * C; i += P::NT) kw[P::K_BQKV + i] = a.bqkv[i];
    if constexpr (MLP)
      for (int i = tid; i < 4 * C; i += P::NT) kw[P::K_B1 + i] = a.b1[i];
    __syncthreads();
  }
  for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
    // the filters and the bias do not depend on the partition: an opaque zero keeps their loads where they are written
    // (hoisted out of this loop they would live in scratch)
    int zo = 0;
    asm volatile("" : "+s"(zo));
    const frag* wq = wq0 + zo;
    const frag* wp = wp0 + zo;
    const f32x4* bp = bp0 + zo;
    PT_STAMP(0);
    // ---- phase 0: the partition's rows
    const int pi = u * PARTS + wt;            // (the PARTS partitions of a unit belong to one alert: nW % PARTS == 0)
    const int w = pi % nW;
    const long b = pi / nW;
    const int wy = w / G, wx = w - wy * G;
    float* xa = a.x + b * H * H * C + 32 * wo + 4 * kg;   // the alert's map (uniform per wave) + this lane's channel quad
    int rowt[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int t = min(16 * n + col, 48);
      const int ty = t / 7, tx = t - ty * 7;
      const int py = a.grid_mode ? ty * G + wy : wy * 7 + ty;
      const int px = a.grid_mode ? tx * G + wx : wx * 7 + tx;
      rowt[n] = (py * H + px) * C;
    }
    f32x4 acc[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = *reinterpret_cast<const f32x4*>(xa + rowt[n] + 16 * m);
    // chunk 0 of the qkv filter: requested before the LayerNorm, used behind it
    frag a1[QT][KS];
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int s = 0; s < KS; ++s) a1[q][s] = wq[(size_t)((wo * QT + q) * KS + s) * 64];

    // LayerNorm of the residual registers -> xn (phases 1 and 5).  One exchange: a wave reduces its 32 channels of a token
    // to (mean, sum of centred squares), the HEADS pairs meet in LDS and combine exactly (equal counts):
    //   mean = avg mean_w,   M2 = sum M2_w + 32 sum (mean_w - mean)^2
    // -- the conditioning of the two-pass form with one barrier less.  `which` picks the table (norm1 | norm2), so that
    // the second LayerNorm of a unit does not wait for the readers of the first.
    auto layer_norm = [&](const float* lnw, const float* lnb, int which) {
      float2* tab = reinterpret_cast<float2*>(red) + which * (HEADS * NTOK);
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) s += (acc[m][n][0] + acc[m][n][1]) + (acc[m][n][2] + acc[m][n][3]);
        const float mw = rows_sum(s) * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[m][n][r] - mw;
            q = fmaf(d, d, q);
          }
        q = rows_sum(q);
        if (kg == 0) tab[wo * NTOK + 16 * n + col] = make_float2(mw, q);
      }
      __syncthreads();
      f32x4 lw[2], lb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        lw[m] = *reinterpret_cast<const f32x4*>(lnw + 32 * wo + 16 * m + 4 * kg);
        lb[m] = *reinterpret_cast<const f32x4*>(lnb + 32 * wo + 16 * m + 4 * kg);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float2 pr[HEADS];
        float mean = 0.f;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
          pr[h] = tab[h * NTOK + 16 * n + col];
          mean += pr[h].x;
        }
        mean *= 1.0f / HEADS;
        float m2 = 0.f;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
          const float d = pr[h].x - mean;
          m2 += fmaf(32.0f * d, d, pr[h].y);
        }
        const float rstd = rsqrtf(m2 * (1.0f / C) + LN_EPS);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          T4 y;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = (T)((acc[m][n][r] - mean) * rstd * lw[m][r] + lb[m][r]);
          *reinterpret_cast<T4*>(xn + (16 * n + col) * XNP + (32 * wo + 16 * m + 4 * kg) * 2) = y;
        }
      }
      __syncthreads();   // xn complete
    };
    PT_STAMP(1);
    // ---- phase 1
    layer_norm(kc + P::K_LN1W, kc + P::K_LN1B, 0);
    PT_STAMP(2);
    // ---- phase 2: qkv^T tile by tile -> IMG
#pragma unroll 1
    for (int ch = 0; ch < NCHQ; ++ch) {
      const int nch = ch + 1 < NCHQ ? ch + 1 : 0;   // (behind the last chunk: an unconditional reload of chunk 0)
      f32x4 bv[QT];
#pragma unroll
      for (int q = 0; q < QT; ++q)
        bv[q] = *reinterpret_cast<const f32x4*>(kc + P::K_BQKV + ((ch * HEADS + wo) * QT + q) * 16 + 4 * kg);
      f32x4 hacc[QT][NB];
#pragma unroll
      for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int n = 0; n < NB; ++n) hacc[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      frag xb[2][NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) xb[0][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (8 * kg) * 2);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            xb[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (32 * (s + 1) + 8 * kg) * 2);
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) {
#pragma unroll
          for (int n = 0; n < NB; ++n) hacc[q][n] = PM<T>::run(a1[q][s], xb[s & 1][n], hacc[q][n]);
          a1[q][s] = wq[(size_t)(((nch * HEADS + wo) * QT + q) * KS + s) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          T4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (T)(hacc[q][n][r] + bv[q][r]);
          *reinterpret_cast<T4*>(img + (16 * n + col) * IP + (((ch * HEADS + wo) * QT + q) * 16 + 4 * kg) * 2) = v;
        }
    }
    PT_STAMP(3);
    // the proj filter of this wave's first channel tile: requested here, used behind the attention phase
    frag a2[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) a2[0][s] = wp[(size_t)((2 * wo) * KS + s) * 64];
    // relative-position bias (+ key mask) of this head's first two query tiles, in lane order: [head][it][jt][lane][4]
    f32x4 bl[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) bl[q][jt] = bp[(q * 4 + jt) * 64];
    __syncthreads();   // IMG complete
    PT_STAMP(4);
    // ---- phase 3: attention of head wo, two query tiles per pass
    {
      const int head = wo;
      frag kf[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
        kf[jt] = *reinterpret_cast<const frag*>(img + (jt * 16 + col) * IP + (head * 96 + 32 + kg * 8) * 2);
      frag vf[2][2];   // [d tile][k step]: V^T by transposing reads
      {
        const int qq = col >> 2, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* ap = img + (32 * ks + 4 * kg + qq) * IP + (head * 96 + 64 + dt * 16 + 4 * p) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 16 * IP));
            union { short h[8]; frag f; } cv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cv.h[e] = lo[e];
              cv.h[4 + e] = hi[e];
            }
            vf[dt][ks] = cv.f;
          }
      }
#pragma unroll 1
      for (int it0 = 0; it0 < 4; it0 += 2) {
        frag qf[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          qf[q] = *reinterpret_cast<const frag*>(img + ((it0 + q) * 16 + col) * IP + (head * 96 + kg * 8) * 2);
        f32x4 s[4][2];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int q = 0; q < 2; ++q) s[jt][q] = PM<T>::run(kf[jt], qf[q], f32x4{0.f, 0.f, 0.f, 0.f});
        float inv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float mx = -3.0e38f;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s[jt][q][r] = fmaf(s[jt][q][r], 0.17677669529663687f, bl[q][jt][r]);
              mx = fmaxf(mx, s[jt][q][r]);
            }
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          float sum = 0.f;
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s[jt][q][r] = __expf(s[jt][q][r] - mx);
              sum += s[jt][q][r];
            }
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          inv[q] = 1.0f / sum;
        }
        // the other pass's bias, requested behind the last use of this one's (pass 1 re-reads pass 0's: unconditional)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) bl[q][jt] = bp[((((it0 + 2) & 2) + q) * 4 + jt) * 64];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          frag pf[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              pf[ks][e] = (T)s[2 * ks][q][e];
              pf[ks][4 + e] = (T)s[2 * ks + 1][q][e];
            }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            f32x4 o = PM<T>::run(vf[dt][0], pf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            o = PM<T>::run(vf[dt][1], pf[1], o);
            // O[token (it0 + q) * 16 + col][head * 32 + dt * 16 + 4 kg ..] -> this head's (dead) Q columns of that token
            T4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (T)(o[r] * inv[q]);
            *reinterpret_cast<T4*>(img + ((it0 + q) * 16 + col) * IP + (head * 96 + dt * 16 + 4 * kg) * 2) = v;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) a2[1][s] = wp[(size_t)((2 * wo + 1) * KS + s) * 64];
    PT_STAMP(5);
    __syncthreads();   // every head's O is in place
    PT_STAMP(6);
    // ---- phase 4: residual += Wproj . O + b  (k-step = head: its 32 O values of a token are contiguous)
    {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + P::K_BPROJ + 32 * wo + 16 * m + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] += bv;
      }
      frag of[2][NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) of[0][n] = *reinterpret_cast<const frag*>(img + (16 * n + col) * IP + (8 * kg) * 2);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            of[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(img + (16 * n + col) * IP + ((s + 1) * 96 + 8 * kg) * 2);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = PM<T>::run(a2[m][s], of[s & 1][n], acc[m][n]);
      }
    }
    PT_STAMP(7);
    if constexpr (MLP) {
      // ---- phase 5: x += fc2(gelu(fc1(LN2(x))))
      const frag* w1 = w10 + zo;
      const frag* w2 = w20 + zo;
      frag f1[HT][KS], f2[2][KF];
#pragma unroll
      for (int q = 0; q < HT; ++q)
#pragma unroll
        for (int s = 0; s < KS; ++s) f1[q][s] = w1[(size_t)((wo * HT + q) * KS + s) * 64];
      // (the first barrier of the LayerNorm also closes phase 4's reads of IMG, which the hidden images overlay)
      layer_norm(kc + P::K_LN2W, kc + P::K_LN2B, 1);
      PT_STAMP(8);
#pragma unroll 1
      for (int ch = 0; ch < NCHM; ++ch) {
        const int nch = ch + 1 < NCHM ? ch + 1 : 0;
        unsigned char* hw = hb + (ch & 1) * (NTOK * HP);
        // fc2's fragments of this step: requested ahead of fc1, used behind the barrier
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int s = 0; s < KF; ++s) f2[m][s] = w2[(size_t)((2 * wo + m) * (4 * C / 32) + ch * KF + s) * 64];
        f32x4 bv[HT];
#pragma unroll
        for (int q = 0; q < HT; ++q)
          bv[q] = *reinterpret_cast<const f32x4*>(kc + P::K_B1 + ch * HCH + (wo * HT + q) * 16 + 4 * kg);
        f32x4 hacc[HT][NB];
#pragma unroll
        for (int q = 0; q < HT; ++q)
#pragma unroll
          for (int n = 0; n < NB; ++n) hacc[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
          frag xb[2][NB];
#pragma unroll
          for (int n = 0; n < NB; ++n) xb[0][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (8 * kg) * 2);
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
              for (int n = 0; n < NB; ++n)
                xb[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (32 * (s + 1) + 8 * kg) * 2);
            }
#pragma unroll
            for (int q = 0; q < HT; ++q) {
#pragma unroll
              for (int n = 0; n < NB; ++n) hacc[q][n] = PM<T>::run(f1[q][s], xb[s & 1][n], hacc[q][n]);
              f1[q][s] = w1[(size_t)(((nch * HEADS + wo) * HT + q) * KS + s) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int q = 0; q < HT; ++q)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            T4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (T)gelu_for<T>(hacc[q][n][r] + bv[q][r]);
            *reinterpret_cast<T4*>(hw + (16 * n + col) * HP + ((wo * HT + q) * 16 + 4 * kg) * 2) = v;
          }
        if (ch < 2) PT_STAMP(9 + 3 * ch);
        __syncthreads();   // hidden image ch & 1 complete (the other one is free: its readers passed this barrier)
        if (ch < 2) PT_STAMP(10 + 3 * ch);
        {
          frag hf[2][NB];
#pragma unroll
          for (int n = 0; n < NB; ++n) hf[0][n] = *reinterpret_cast<const frag*>(hw + (16 * n + col) * HP + (8 * kg) * 2);
#pragma unroll
          for (int s = 0; s < KF; ++s) {
            if (s + 1 < KF) {
#pragma unroll
              for (int n = 0; n < NB; ++n)
                hf[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(hw + (16 * n + col) * HP + (32 * (s + 1) + 8 * kg) * 2);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int n = 0; n < NB; ++n) acc[m][n] = PM<T>::run(f2[m][s], hf[s & 1][n], acc[m][n]);
          }
        }
        if (ch < 2) PT_STAMP(11 + 3 * ch);
      }
      PT_STAMP(15);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + P::K_B2 + 32 * wo + 16 * m + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] += bv;
      }
    }
    // ---- phase 6: the partition's rows back
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (16 * n + col < 49) *reinterpret_cast<f32x4*>(xa + rowt[n] + 16 * m) = acc[m][n];
    if (a.post_out != nullptr) {
      T* pa = reinterpret_cast<T*>(a.post_out) + (xa - a.x);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 ps = *reinterpret_cast<const f32x4*>(kc + P::K_PS + 32 * wo + 16 * m + 4 * kg);
        const f32x4 pb = *reinterpret_cast<const f32x4*>(kc + P::K_PB + 32 * wo + 16 * m + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          T4 y;
#pragma unroll
          for (int r = 0; r < 4; ++r) y[r] = (T)(acc[m][n][r] * ps[r] + pb[r]);
          if (16 * n + col < 49) *reinterpret_cast<T4*>(pa + rowt[n] + 16 * m) = y;
        }
      }
    }
    PT_STAMP(16);
    // (no barrier here: the next unit's LayerNorm has three before anything this unit still reads is rewritten)
  }
}

// ------------------------------------------------------------------------------------------------------------------
// C = 64 (stage 0: 64 partitions per alert, 65536 per 1024-alert forward): the same block with the filters RESIDENT.  All
// four of them are 96 KB -- 24 fragments per wave of a 4-wave workgroup -- so a persistent workgroup loads them once and
// nothing streams: a unit's critical path has no L2 round trip left but its rows, and those are requested one unit ahead.
// Wave w = (head h = w >> 1, half j = w & 1) owns the residual channels 16 w .. 16 w + 15 (one output tile) of all four
// token tiles, the qkv output tiles 6 h + 3 j + {0, 1, 2} (its head's q, k, v split between the head's two waves), the
// attention of query tiles 2 j, 2 j + 1 of head h, and the fc1 hidden tiles 4 w .. 4 w + 3.  Two workgroups per CU.
namespace p64 {
constexpr int C = 64, NT = 256, KS = 2;
constexpr int XNP = C * 2 + 16;            // 144: bytes per token row of the LayerNorm image (36 banks per row: conflict-free)
constexpr int IP = 3 * C * 2 + 16;         // 400: ... of the qkv image
constexpr int HPP = 4 * C * 2 + 16;        // 528: ... of the hidden image (it overlays the qkv image)
constexpr int OFF_A = 0, A_BYTES = 64 * HPP;                        // 33792 (the qkv image needs 25600)
constexpr int OFF_XN = OFF_A + A_BYTES, XN_BYTES = 64 * XNP;        // 9216
constexpr int OFF_TAB = OFF_XN + XN_BYTES, TAB_BYTES = 4 * 64 * 8;  // LayerNorm exchange [4 waves][64 tokens] (mean, M2)
constexpr int K_LN1W = 0, K_LN1B = C, K_BQKV = 2 * C, K_BPROJ = 5 * C, K_LN2W = 6 * C, K_LN2B = 7 * C, K_B1 = 8 * C,
              K_B2 = 12 * C, K_PS = 13 * C, K_PB = 14 * C, K_N = 15 * C;
constexpr int OFF_K = OFF_TAB + TAB_BYTES;
constexpr int OFF_BIAS = OFF_K + K_N * 4, BIAS_BYTES = 2 * 16 * 64 * 16;   // the lane-ordered rel-pos bias of both heads
constexpr int LDS_BYTES = OFF_BIAS + BIAS_BYTES;                           // 81664
static_assert(2 * LDS_BYTES <= 160 * 1024 && 64 * IP <= A_BYTES, "two workgroups per CU");
}  // namespace p64

template <typename T, bool MLP>
__global__ __launch_bounds__(p64::NT, 2) void mv_part64_kernel(PartArgs a) {
  using namespace p64;
  using frag = typename PM<T>::frag;
  typedef T __attribute__((ext_vector_type(4))) T4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem + OFF_A;
  unsigned char* hid = smem + OFF_A;
  unsigned char* xn = smem + OFF_XN;
  float2* tab = reinterpret_cast<float2*>(smem + OFF_TAB);
  const float* kc = reinterpret_cast<const float*>(smem + OFF_K);
  const f32x4* bl = reinterpret_cast<const f32x4*>(smem + OFF_BIAS);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 15, kg = lane >> 4;
  const int h = w >> 1, j = w & 1;
  const int H = a.H, G = H / 7, nW = G * G;

  // ---- once per workgroup: the filters into registers, the constants and the bias into LDS
  frag fq[3][KS], fp[KS], f1[4][KS], f2[8];
  {
    const frag* wq = reinterpret_cast<const frag*>(a.wqkvp) + lane;
    const frag* wp = reinterpret_cast<const frag*>(a.wprojp) + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int s = 0; s < KS; ++s) fq[q][s] = wq[(size_t)((6 * h + 3 * j + q) * KS + s) * 64];
#pragma unroll
    for (int s = 0; s < KS; ++s) fp[s] = wp[(size_t)(w * KS + s) * 64];
    if constexpr (MLP) {
      const frag* w1 = reinterpret_cast<const frag*>(a.w1p) + lane;
      const frag* w2 = reinterpret_cast<const frag*>(a.w2p) + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < KS; ++s) f1[q][s] = w1[(size_t)((4 * w + q) * KS + s) * 64];
#pragma unroll
      for (int s = 0; s < 8; ++s) f2[s] = w2[(size_t)(w * 8 + s) * 64];
    }
    float* kw = reinterpret_cast<float*>(smem + OFF_K);
    if (tid < C) {
      kw[K_LN1W + tid] = a.ln1w[tid];
      kw[K_LN1B + tid] = a.ln1b[tid];
      kw[K_BPROJ + tid] = a.bproj[tid];
      if (a.post_out != nullptr) {
        kw[K_PS + tid] = a.post_s[tid];
        kw[K_PB + tid] = a.post_b[tid];
      }
      if constexpr (MLP) {
        kw[K_LN2W + tid] = a.ln2w[tid];
        kw[K_LN2B + tid] = a.ln2b[tid];
        kw[K_B2 + tid] = a.b2[tid];
      }
    }
    if (tid < 3 * C) kw[K_BQKV + tid] = a.bqkv[tid];
    if constexpr (MLP) kw[K_B1 + tid] = a.b1[tid];
    f32x4* bw = reinterpret_cast<f32x4*>(smem + OFF_BIAS);
    for (int i = tid; i < BIAS_BYTES / 16; i += NT) bw[i] = reinterpret_cast<const f32x4*>(a.biasl)[i];
    __syncthreads();
  }

  // the rows of unit u: the alert's map + this lane's channel quad, and the four token rows' offsets
  auto rows_of = [&](int u, float*& xa, int (&rowt)[NB]) {
    const int pw = u % nW;
    const long b = u / nW;
    const int wy = pw / G, wx = pw - wy * G;
    xa = a.x + b * H * H * C + 16 * w + 4 * kg;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int t = min(16 * n + col, 48);
      const int ty = t / 7, tx = t - ty * 7;
      const int py = a.grid_mode ? ty * G + wy : wy * 7 + ty;
      const int px = a.grid_mode ? tx * G + wx : wx * 7 + tx;
      rowt[n] = (py * H + px) * C;
    }
  };
  // LayerNorm of the residual registers -> xn: one exchange of (mean, centred squares) per wave, combined exactly
  auto layer_norm = [&](const f32x4 (&acc)[NB], const float* lnw, const float* lnb) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float mw = rows_sum((acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3])) * (1.0f / 16.0f);
      float q = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[n][r] - mw;
        q = fmaf(d, d, q);
      }
      q = rows_sum(q);
      if (kg == 0) tab[w * 64 + 16 * n + col] = make_float2(mw, q);
    }
    __syncthreads();
    const f32x4 lw = *reinterpret_cast<const f32x4*>(lnw + 16 * w + 4 * kg);
    const f32x4 lb = *reinterpret_cast<const f32x4*>(lnb + 16 * w + 4 * kg);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float2 pr[4];
      float mean = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        pr[v] = tab[v * 64 + 16 * n + col];
        mean += pr[v].x;
      }
      mean *= 0.25f;
      float m2 = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d = pr[v].x - mean;
        m2 += fmaf(16.0f * d, d, pr[v].y);
      }
      const float rstd = rsqrtf(m2 * (1.0f / C) + LN_EPS);
      T4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = (T)((acc[n][r] - mean) * rstd * lw[r] + lb[r]);
      *reinterpret_cast<T4*>(xn + (16 * n + col) * XNP + (16 * w + 4 * kg) * 2) = y;
    }
    __syncthreads();   // xn complete
  };

  float* xa_n;
  int rowt_n[NB];
  f32x4 pre[NB];
  if ((int)blockIdx.x < a.units) {
    rows_of(blockIdx.x, xa_n, rowt_n);
#pragma unroll
    for (int n = 0; n < NB; ++n) pre[n] = *reinterpret_cast<const f32x4*>(xa_n + rowt_n[n]);
  }
  for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
    // ---- phase 0: this unit's rows (requested one unit ago); the next unit's are requested now
    f32x4 acc[NB];
    float* xa = xa_n;
    int rowt[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      acc[n] = pre[n];
      rowt[n] = rowt_n[n];
    }
    {
      const int un = u + (int)gridDim.x < a.units ? u + (int)gridDim.x : u;   // (behind the last unit: its own rows again)
      rows_of(un, xa_n, rowt_n);
#pragma unroll
      for (int n = 0; n < NB; ++n) pre[n] = *reinterpret_cast<const f32x4*>(xa_n + rowt_n[n]);
    }
    // ---- phase 1
    layer_norm(acc, kc + K_LN1W, kc + K_LN1B);
    // ---- phase 2: this wave's three qkv tiles -> IMG
    {
      frag xb[KS][NB];
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int n = 0; n < NB; ++n) xb[s][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (32 * s + 8 * kg) * 2);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int tile = 6 * h + 3 * j + q;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + K_BQKV + tile * 16 + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          f32x4 hacc = bv;
#pragma unroll
          for (int s = 0; s < KS; ++s) hacc = PM<T>::run(fq[q][s], xb[s][n], hacc);
          T4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (T)hacc[r];
          *reinterpret_cast<T4*>(img + (16 * n + col) * IP + (tile * 16 + 4 * kg) * 2) = v;
        }
      }
    }
    __syncthreads();   // IMG complete
    // ---- phase 3: attention of head h, query tiles 2 j and 2 j + 1
    {
      frag kf[4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
        kf[jt] = *reinterpret_cast<const frag*>(img + (jt * 16 + col) * IP + (h * 96 + 32 + kg * 8) * 2);
      frag vf[2][2];   // [d tile][k step]: V^T by transposing reads
      {
        const int qq = col >> 2, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* ap = img + (32 * ks + 4 * kg + qq) * IP + (h * 96 + 64 + dt * 16 + 4 * p) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap + 16 * IP));
            union { short hh[8]; frag f; } cv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cv.hh[e] = lo[e];
              cv.hh[4 + e] = hi[e];
            }
            vf[dt][ks] = cv.f;
          }
      }
      frag qf[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
        qf[q] = *reinterpret_cast<const frag*>(img + ((2 * j + q) * 16 + col) * IP + (h * 96 + kg * 8) * 2);
      f32x4 sc[4][2];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 2; ++q) sc[jt][q] = PM<T>::run(kf[jt], qf[q], f32x4{0.f, 0.f, 0.f, 0.f});
      float inv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const f32x4 bb = bl[((h * 4 + 2 * j + q) * 4 + jt) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[jt][q][r] = fmaf(sc[jt][q][r], 0.17677669529663687f, bb[r]);
            mx = fmaxf(mx, sc[jt][q][r]);
          }
        }
        mx = rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[jt][q][r] = __expf(sc[jt][q][r] - mx);
            sum += sc[jt][q][r];
          }
        inv[q] = 1.0f / rows_sum(sum);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        frag pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pf[ks][e] = (T)sc[2 * ks][q][e];
            pf[ks][4 + e] = (T)sc[2 * ks + 1][q][e];
          }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          f32x4 o = PM<T>::run(vf[dt][0], pf[0], f32x4{0.f, 0.f, 0.f, 0.f});
          o = PM<T>::run(vf[dt][1], pf[1], o);
          T4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (T)(o[r] * inv[q]);
          // O -> the head's (dead) Q columns of this wave's own query rows
          *reinterpret_cast<T4*>(img + ((2 * j + q) * 16 + col) * IP + (h * 96 + dt * 16 + 4 * kg) * 2) = v;
        }
      }
    }
    __syncthreads();   // both heads' O in place
    // ---- phase 4: residual += Wproj . O + b  (k-step = head)
    {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + K_BPROJ + 16 * w + 4 * kg);
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        acc[n] += bv;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const frag of = *reinterpret_cast<const frag*>(img + (16 * n + col) * IP + (s * 96 + 8 * kg) * 2);
          acc[n] = PM<T>::run(fp[s], of, acc[n]);
        }
      }
    }
    if constexpr (MLP) {
      // ---- phase 5: x += fc2(gelu(fc1(LN2(x))))  (the LayerNorm's barriers close phase 4's reads of IMG, which the hidden
      // image overlays)
      layer_norm(acc, kc + K_LN2W, kc + K_LN2B);
      {
        frag xb[KS][NB];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int n = 0; n < NB; ++n) xb[s][n] = *reinterpret_cast<const frag*>(xn + (16 * n + col) * XNP + (32 * s + 8 * kg) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + K_B1 + (4 * w + q) * 16 + 4 * kg);
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            f32x4 hacc = bv;
#pragma unroll
            for (int s = 0; s < KS; ++s) hacc = PM<T>::run(f1[q][s], xb[s][n], hacc);
            T4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (T)gelu_for<T>(hacc[r]);
            *reinterpret_cast<T4*>(hid + (16 * n + col) * HPP + ((4 * w + q) * 16 + 4 * kg) * 2) = v;
          }
        }
      }
      __syncthreads();   // the hidden image complete
      {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(kc + K_B2 + 16 * w + 4 * kg);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] += bv;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const frag hf = *reinterpret_cast<const frag*>(hid + (16 * n + col) * HPP + (32 * s + 8 * kg) * 2);
            acc[n] = PM<T>::run(f2[s], hf, acc[n]);
          }
      }
    }
    // ---- phase 6: the partition's rows back
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (16 * n + col < 49) *reinterpret_cast<f32x4*>(xa + rowt[n]) = acc[n];
    if (a.post_out != nullptr) {
      T* pa = reinterpret_cast<T*>(a.post_out) + (xa - a.x);
      const f32x4 ps = *reinterpret_cast<const f32x4*>(kc + K_PS + 16 * w + 4 * kg);
      const f32x4 pb = *reinterpret_cast<const f32x4*>(kc + K_PB + 16 * w + 4 * kg);
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        T4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (T)(acc[n][r] * ps[r] + pb[r]);
        if (16 * n + col < 49) *reinterpret_cast<T4*>(pa + rowt[n]) = y;
      }
    }
    // (no barrier: the next unit's LayerNorm has two before anything this unit still reads is rewritten -- its exchange
    //  table was last read before this unit's last LayerNorm barrier)
  }
}

// out [heads][4 query tiles][4 key tiles][64 lanes][4]: the values lane (kg, col) of the head's wave adds to its S^T
// accumulator (key jt*16 + 4 kg + r, query it*16 + col), key padding mask folded in -- one 16-byte load per tile pair.
__global__ void mv_pack_relbias_lanes_kernel(const float* table, float* out, int heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= heads * 4096) return;
  const int r = i & 3, ln = (i >> 2) & 63, jt = (i >> 8) & 3, it = (i >> 10) & 3, hd = i >> 12;
  const int kj = jt * 16 + 4 * (ln >> 4) + r, qi = it * 16 + (ln & 15);
  float v = 0.f;
  if (kj >= 49) {
    v = -1.0e30f;
  } else if (qi < 49) {
    const int dy = qi / 7 - kj / 7, dx = qi % 7 - kj % 7;
    v = table[((dy + 6) * 13 + dx + 6) * heads + hd];
  }
  out[i] = v;
}

template <typename T, int C_, bool MLP> int launch_part_t(const PartArgs& a, int units, hipStream_t st) {
  using P = PG<C_>;
  auto kern = mv_part_kernel<T, C_, MLP>;
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P::LDS_BYTES));
    attr.done();
  }
  // C = 256 (one workgroup per CU): persistent, 256 workgroups walk 16 units each at 1024 alerts -- the constants are
  // staged and the workgroup launched once (622 -> 578 us); C = 128 (two per CU) measured better with a unit or two per
  // workgroup (770 against 785 us)
  const int cap = P::WGS == 1 ? 256 : 4096 * P::WGS;
  const int grid = units < cap ? units : cap;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(P::NT), P::LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T, bool MLP> int launch_part64_t(const PartArgs& a, hipStream_t st) {
  auto kern = mv_part64_kernel<T, MLP>;
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, p64::LDS_BYTES));
    attr.done();
  }
  // persistent: two workgroups per CU load the filters once and walk the units
  const int grid = a.units < 512 ? a.units : 512;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(p64::NT), p64::LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

int launch_mv_pack_relbias_lanes(const float* table, float* out, int heads, hipStream_t st) {
  hipLaunchKernelGGL(mv_pack_relbias_lanes_kernel, dim3((heads * 4096 + 255) / 256), dim3(256), 0, st, table, out, heads);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

bool mv_part_supported(int prec, int C) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (C == 256 || C == 128 || C == 64);
}

// x [B*H*H, C] f32 updated in place, per 7x7 window (grid_mode 0) / dilated grid (1):
//   x += proj(attn(qkv(LN1(x))));  with p.w1p != nullptr also  x += fc2(gelu(fc1(LN2(x)))).
// The filters are launch_pack_s2p fragments (attn.qkv.weight [3C][C], attn.proj.weight [C][C], mlp.fc1.weight [4C][C],
// mlp.fc2.weight [C][4C]); biasl the lane-ordered image of launch_mv_pack_relbias_lanes.
int launch_mv_part(int prec, float* x, const MvPartW& p, int B, int H, int C, int grid_mode, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  const int G = H / 7, nW = G * G, parts = 1;
  if (!mv_part_supported(prec, C) || H % 7 != 0 || nW % parts != 0) {
    btsbot_set_error("mv_part: unsupported (prec %d, C %d, H %d)", prec, C, H);
    return BTSBOT_ERR_INVALID_ARG;
  }
  PartArgs a;
  a.x = x;
  a.ln1w = p.ln1w; a.ln1b = p.ln1b; a.bqkv = p.bqkv; a.bproj = p.bproj; a.biasl = p.biasl;
  a.wqkvp = p.wqkvp; a.wprojp = p.wprojp;
  a.ln2w = p.ln2w; a.ln2b = p.ln2b; a.b1 = p.b1; a.b2 = p.b2;
  a.w1p = p.w1p; a.w2p = p.w2p;
  a.H = H; a.grid_mode = grid_mode;
  a.stamps = p.stamps;
  a.post_s = p.post_s; a.post_b = p.post_b; a.post_out = p.post_out;
  const long units = (long)B * nW / parts;
  if (units > 0x7fffffffL) {
    btsbot_set_error("mv_part: %ld units", units);
    return BTSBOT_ERR_INVALID_ARG;
  }
  a.units = (int)units;
  const bool mlp = p.w1p != nullptr;
  if (C == 64) {
    if (prec == BTSBOT_BF16) return mlp ? launch_part64_t<bf16_t, true>(a, st) : launch_part64_t<bf16_t, false>(a, st);
    return mlp ? launch_part64_t<f16_t, true>(a, st) : launch_part64_t<f16_t, false>(a, st);
  }
#define PART(TT)                                                                                              \
  (C == 256 ? (mlp ? launch_part_t<TT, 256, true>(a, a.units, st) : launch_part_t<TT, 256, false>(a, a.units, st)) \
            : (mlp ? launch_part_t<TT, 128, true>(a, a.units, st) : launch_part_t<TT, 128, false>(a, a.units, st)))
  return prec == BTSBOT_BF16 ? PART(bf16_t) : PART(f16_t);
#undef PART
}
