// Stage-0 megakernel (gfx950): one workgroup carries one alert from the raw 63x63x3 triplet to the
// input of stage 1 without touching HBM in between:
//
//   stem (conv 4x4 s4 + LN)  ->  2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]
//                            ->  downsample (LN + conv 2x2 s2)  ->  [49][128] f32
//
// (timm ConvNeXt stem / stages[0] / stages[1].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132.)  Unfused, these eight launches move ~0.9 MB per
// alert through HBM (fp32 residual stream, 16-bit LN outputs, patch matrices); here the alert's
// 15x15x64 map lives on the CU: HBM sees 47.6 KB in and 25 KB out.
//
// Data placement (512 threads = 8 waves, wave w owns pixels 32w .. 32w+31 of the 225):
//   * residual stream x: fp32 in REGISTERS, in the 32x32 MFMA accumulator layout (lane = pixel,
//     register = channel (r&3) + 8(r>>2) + 4(lane>>5) of a 32-channel tile).  The stem, both MLPs
//     and the downsample produce / consume it in that layout, so the residual add, the stem LN and
//     the downsample LN are register-only (a pixel's 64 channels sit in 32 registers of 2 lanes).
//   * two 16-bit map images in LDS ([256 px][64 ch], 144-byte rows): the depthwise conv reads one
//     (lane = channel, sliding window along x in registers) and writes LN(conv) to the other, which
//     the MLP then reads as its MFMA B operand; the MLP epilogue writes the new x back to the first.
//   * the block's pointwise filters (64 KB + padding) arrive by LDS-DMA while the depthwise phase
//     runs; same packed image and register-chained fc1 -> GELU -> fc2 as fused_mlp.hip.
#include "common.h"
#include "stage0.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct S0M;
template <> struct S0M<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct S0M<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 64, HW = 15, P = 225, CT = 2;
constexpr int PITCH = 2 * C + 16;                 // bytes per map row in LDS
constexpr int MAPB = 256 * PITCH;                 // one map image
constexpr int W1ROW = 2 * C + 16, W2ROW = 80;     // fused_mlp.hip FusedGeom<64>
constexpr int SUBBYTES = 32 * W1ROW + C * W2ROW;  // 9728
constexpr int NSUB = 8;
constexpr int WBYTES = NSUB * SUBBYTES;           // 77824: the whole block's pointwise filters
constexpr float LN_EPS = 1e-6f;

// diagnostic builds of the timeline: workgroup 0, thread 0 stores the shader clock
#define STAMP(i)                                                                  \
  do {                                                                            \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
    if (a.wgt != nullptr && threadIdx.x == 0 && ((i) == 0 || (i) == 13))           \
      a.wgt[2 * blockIdx.x + ((i) == 13)] = wall_clock64();                        \
  } while (0)

struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// 16 values per lane -> v[0..3] = 64-lane totals of values (lane>>4)*4 + j  (see convnext.hip)
__device__ __forceinline__ void treduce16(float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = swap_add32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = swap_add16(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = group16_sum(v[i]);
}

// channel owned by accumulator register r of 32-channel tile ct in lane half h
__device__ __forceinline__ int chan(int ct, int r, int h) { return ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; }

// LayerNorm over the 64 channels of this lane's pixel, held as x[2][16] in this lane and its
// partner lane^32; affine with w/b; result in place.
__device__ __forceinline__ void ln_regs(f32x16 (&x)[CT], const float* __restrict__ w,
                                        const float* __restrict__ b, int h, f32x16 (&y)[CT],
                                        bool noload = false) {
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
      const float4 wv = noload ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(w + c);
      const float4 bv = noload ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(b + c);
      y[ct][4 * qd + 0] = (x[ct][4 * qd + 0] - mean) * rstd * wv.x + bv.x;
      y[ct][4 * qd + 1] = (x[ct][4 * qd + 1] - mean) * rstd * wv.y + bv.y;
      y[ct][4 * qd + 2] = (x[ct][4 * qd + 2] - mean) * rstd * wv.z + bv.z;
      y[ct][4 * qd + 3] = (x[ct][4 * qd + 3] - mean) * rstd * wv.w + bv.w;
    }
}

// registers (accumulator layout) -> 16-bit map image row `p`
template <typename T>
__device__ __forceinline__ void regs_to_map(const f32x16 (&x)[CT], unsigned char* map, int p, int h) {
  typedef T __attribute__((ext_vector_type(4))) T4;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      T4 v;
      v[0] = (T)x[ct][4 * qd + 0];
      v[1] = (T)x[ct][4 * qd + 1];
      v[2] = (T)x[ct][4 * qd + 2];
      v[3] = (T)x[ct][4 * qd + 3];
      *reinterpret_cast<T4*>(map + p * PITCH + (ct * 32 + 8 * qd + 4 * h) * 2) = v;
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

}  // namespace

namespace {

template <typename T>
__global__ __launch_bounds__(512, 2) void stage0_kernel(Stage0Args a) {
  using frag = typename S0M<T>::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* map0 = smem;                     // x (16-bit) for the depthwise conv
  unsigned char* map1 = smem + MAPB;              // LN outputs for the MFMA B operand
  unsigned char* wring = smem + 2 * MAPB;         // pointwise filters of the current block
  float* b1s = reinterpret_cast<float*>(wring + WBYTES);   // [256]
  float* red = b1s + 256;                                    // [2][8][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int alert = blockIdx.x;
  const int p = wave * 32 + lr;                   // this lane's pixel (MFMA phases)
  const bool live = p < P;
  const int pc = live ? p : 0;
  const int py = pc / HW, px = pc - py * HW;

  STAMP(0);
  // rows 225..255 of both images are padding: keep them finite
  for (int i = tid; i < (256 - P) * PITCH / 4; i += 512) {
    reinterpret_cast<unsigned*>(map0 + P * PITCH)[i] = 0u;
    reinterpret_cast<unsigned*>(map1 + P * PITCH)[i] = 0u;
  }

  // ============================ stem: conv 4x4 s4 + LN =====================================
  f32x16 x[CT];
  {
    const float* src = a.img + (size_t)alert * 3 * 63 * 63;
    const T* sw = reinterpret_cast<const T*>(a.stem_w);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 bv = *reinterpret_cast<const float4*>(a.stem_b + ct * 32 + 8 * qd + 4 * h);
        x[ct][4 * qd + 0] = bv.x;
        x[ct][4 * qd + 1] = bv.y;
        x[ct][4 * qd + 2] = bv.z;
        x[ct][4 * qd + 3] = bv.w;
      }
#pragma unroll
    for (int ci = 0; ci < ((a.diag & 16) ? 0 : 3); ++ci) {     // k-step = input channel: k = ci*16 + ky*4 + kx
      const float* r0 = src + (ci * 63 + 4 * py + 2 * h) * 63 + 4 * px;
      const f4u v0 = *reinterpret_cast<const f4u*>(r0);
      const f4u v1 = *reinterpret_cast<const f4u*>(r0 + 63);
      frag bf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf[e] = (T)v0.v[e];
        bf[4 + e] = (T)v1.v[e];
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const frag af = *reinterpret_cast<const frag*>(sw + (ct * 32 + lr) * 48 + ci * 16 + h * 8);
        x[ct] = S0M<T>::run(af, bf, x[ct]);
      }
    }
    ln_regs(x, a.stem_lnw, a.stem_lnb, h, x, a.diag & 64);
    regs_to_map<T>(x, map0, p, h);
    if (a.tap_stem != nullptr && live)
      regs_to_tap(x, a.tap_stem + ((size_t)alert * P + p) * C, h);
  }

  STAMP(1);   // stem done
  // ============================ two ConvNeXt blocks ========================================
  // Depthwise filters (lane = channel: 49 taps + bias + LN affine) are fetched one block AHEAD with
  // ordinary loads, so that no ordinary load is outstanding while the pointwise filters' LDS-DMA
  // is in flight (hipcc waits vmcnt(0) for an ordinary load's first use, which would drain the DMA).
  float w[49], dwbias, lng, lnb2, b1v;
  {
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = (a.diag & 64) ? 0.01f : a.blk[0].dw_w[t * C + lane];
    dwbias = a.blk[0].dw_b[lane];
    lng = a.blk[0].ln_w[lane];
    lnb2 = a.blk[0].ln_b[lane];
    b1v = a.blk[0].b1[tid & 255];
  }
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    // first use of the prefetched depthwise filters: hipcc's wait for them lands HERE
#pragma unroll
    for (int t = 0; t < 49; ++t) asm volatile("" ::"v"(w[t]));
    asm volatile("" ::"v"(dwbias), "v"(lng), "v"(lnb2), "v"(b1v));
    STAMP(2 + 5 * j);   // filters touched
    __syncthreads();   // map0 complete; previous block's filter / fc1-bias reads finished
    if (tid < 256) b1s[tid] = b1v;
    // ---- the block's pointwise filters: 76 x 1 KiB LDS-DMA pieces, wave w takes w, w+8, ...
    for (int pc2 = wave; pc2 < ((a.diag & 4) ? 0 : WBYTES / 1024); pc2 += 8)
      __builtin_amdgcn_global_load_lds((gptr_t)(bk.wpk + (size_t)pc2 * 1024 + lane * 16),
                                       (lptr_t)(wring + pc2 * 1024), 16, 0, 0);

    STAMP(3 + 5 * j);   // DMA issued
    // ---- depthwise 7x7 + bias + LN: lane = channel, wave = map row, 2 rounds of 8 rows
    {
      const int c = lane;
      const float bias = dwbias, g = lng, bb = lnb2;
      const T* mi = reinterpret_cast<const T*>(map0);
#pragma unroll 1
      for (int rd = 0; rd < 2; ++rd) {
        const int y = rd * 8 + wave;
        const bool valid = y < HW;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (i < HW) ? bias : 0.f;
        if (valid && !(a.diag & 1)) {
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= HW) continue;
            const T* row = mi + (iy * HW) * (PITCH / 2) + c;
            float in[HW];
#pragma unroll
            for (int xx = 0; xx < HW; ++xx) in[xx] = (float)row[xx * (PITCH / 2)];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int xx = 0; xx < HW; ++xx) {
                const int ix = xx + kx - 3;
                if (ix >= 0 && ix < HW) acc[xx] = fmaf(in[ix], w[ky * 7 + kx], acc[xx]);
              }
          }
        }
        // LN over the 64 channels (= the 64 lanes) of each of the row's 15 pixels; two-pass
        // variance; acc is centred in place to keep the live register set small
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = valid ? acc[i] : 0.f;
        treduce16(s);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) red[wave * 16 + (lane >> 4) * 4 + jj] = s[jj];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] -= red[wave * 16 + i] * (1.0f / C);
          s[i] = valid ? acc[i] * acc[i] : 0.f;
        }
        treduce16(s);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) red[128 + wave * 16 + (lane >> 4) * 4 + jj] = s[jj];
        }
        __syncthreads();
        if (valid) {
          T* dst = reinterpret_cast<T*>(map1) + (y * HW) * (PITCH / 2) + c;
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const float rstd = rsqrtf(red[128 + wave * 16 + xx] * (1.0f / C) + LN_EPS);
            dst[xx * (PITCH / 2)] = (T)(acc[xx] * rstd * g + bb);
          }
        }
      }
    }
    STAMP(4 + 5 * j);   // depthwise done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's filter pieces have landed
    __syncthreads();                                    // map1 + filters complete for everyone
    if (j == 0) {   // next block's depthwise filters: in flight under the MLP (no DMA pending now)
#pragma unroll
      for (int t = 0; t < 49; ++t) w[t] = (a.diag & 64) ? 0.01f : a.blk[1].dw_w[t * C + lane];
      dwbias = a.blk[1].dw_b[lane];
      lng = a.blk[1].ln_w[lane];
      lnb2 = a.blk[1].ln_b[lane];
      b1v = a.blk[1].b1[tid & 255];
    }

    STAMP(5 + 5 * j);   // DMA landed + barrier
    // ---- fc1 -> GELU -> fc2 (register-chained, see fused_mlp.hip), then x += gamma*(y + b2)
    {
      frag xf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(map1 + p * PITCH + ks * 32 + h * 16);
      f32x16 yacc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;
#pragma unroll 1
      for (int sub = 0; sub < ((a.diag & 2) ? 0 : NSUB); ++sub) {
        const unsigned char* w1s = wring + sub * SUBBYTES;
        const unsigned char* w2s = w1s + 32 * W1ROW;
        // all 8 filter fragments of this sub-chunk are requested up front (explicit arrays: hipcc
        // otherwise reuses one register quad and serialises ds_read -> wait -> MFMA per fragment)
        frag a1[4], a2[CT][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          a1[ks] = *reinterpret_cast<const frag*>(w1s + lr * W1ROW + ks * 32 + h * 16);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            a2[ct][s2] = *reinterpret_cast<const frag*>(w2s + (ct * 32 + lr) * W2ROW + s2 * 32 + h * 16);
        f32x16 hacc;
        const float* bp = b1s + sub * 32 + 4 * h;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {          // bias rides in as the initial accumulator
          const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * qd);
          hacc[4 * qd + 0] = bv.x;
          hacc[4 * qd + 1] = bv.y;
          hacc[4 * qd + 2] = bv.z;
          hacc[4 * qd + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) hacc = S0M<T>::run(a1[ks], xf[ks], hacc);
        frag hf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) hf[r >> 3][r & 7] = (T)((a.diag & 8) ? hacc[r] : gelu_fast(hacc[r]));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) yacc[ct] = S0M<T>::run(a2[ct][s2], hf[s2], yacc[ct]);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = ct * 32 + 8 * qd + 4 * h;
          const float4 bv = (a.diag & 64) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(bk.b2 + c);
          const float4 gv = (a.diag & 64) ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(bk.gamma + c);
          x[ct][4 * qd + 0] += gv.x * (yacc[ct][4 * qd + 0] + bv.x);
          x[ct][4 * qd + 1] += gv.y * (yacc[ct][4 * qd + 1] + bv.y);
          x[ct][4 * qd + 2] += gv.z * (yacc[ct][4 * qd + 2] + bv.z);
          x[ct][4 * qd + 3] += gv.w * (yacc[ct][4 * qd + 3] + bv.w);
        }
      if (j == 0) regs_to_map<T>(x, map0, p, h);   // next block's depthwise input
    }
    STAMP(6 + 5 * j);   // MLP done
  }
  if (a.tap_stage != nullptr && live)
    regs_to_tap(x, a.tap_stage + ((size_t)alert * P + p) * C, h);

  // ============================ downsample: LN + conv 2x2 s2 (64 -> 128) ====================
  {
    f32x16 xn[CT];
    ln_regs(x, a.ds_lnw, a.ds_lnb, h, xn, a.diag & 64);
    regs_to_map<T>(xn, map0, p, h);   // map0: last read by block 1's depthwise phase
    __syncthreads();
    STAMP(12);   // downsample LN done
    // wave -> (pixel tile pt of the 49 outputs, 32-channel output tile cot); K = 4 taps x 64 ch
    const int pt = wave & 1, cot = wave >> 1;
    const int o = pt * 32 + lr;
    const bool olive = o < 49;
    const int oc = olive ? o : 0;
    const int oy = oc / 7, ox = oc - oy * 7;
    f32x16 acc;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 bv = *reinterpret_cast<const float4*>(a.ds_b + cot * 32 + 8 * qd + 4 * h);
      acc[4 * qd + 0] = bv.x;
      acc[4 * qd + 1] = bv.y;
      acc[4 * qd + 2] = bv.z;
      acc[4 * qd + 3] = bv.w;
    }
    const T* dw = reinterpret_cast<const T*>(a.ds_w) + (size_t)(cot * 32 + lr) * 256 + h * 8;
#pragma unroll
    for (int ks = 0; ks < ((a.diag & 32) ? 0 : 16); ++ks) {
      const int q = ks >> 2;
      const int pin = (2 * oy + (q >> 1)) * HW + 2 * ox + (q & 1);
      const frag bf = *reinterpret_cast<const frag*>(map0 + pin * PITCH + (ks & 3) * 32 + h * 16);
      const frag af = *reinterpret_cast<const frag*>(dw + ks * 16);
      acc = S0M<T>::run(af, bf, acc);
    }
    if (olive) {
      float* dst = a.out + ((size_t)alert * 49 + o) * 128 + cot * 32 + 4 * h;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<float4*>(dst + 8 * qd) =
            make_float4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
    }
    STAMP(13);   // end
  }
}

template <typename T> int launch_stage0_t(const Stage0Args& a, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)MAPB + WBYTES + 256 * 4 + 2 * 8 * 16 * 4;
  auto kern = stage0_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(512), lds, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool stage0_supported(int prec, int c0) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && c0 == 64;
}

int launch_stage0(int prec, const Stage0Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16) return launch_stage0_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage0_t<f16_t>(a, st);
  btsbot_set_error("stage0: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
