// Stage-1 megakernel (gfx950): one workgroup carries FOUR alerts' 7x7x128 maps through
//
//   2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]  ->  LN + conv 2x2 s2 (128 -> 256)
//
// (timm ConvNeXt stages[1].blocks / stages[2].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132) with the maps on-chip: HBM sees the stage input
// [49][128] f32 and the stage-2 input [9][256] f32 per alert.  Same placement as stage0.hip:
//   * residual stream fp32 in registers, 32x32 MFMA accumulator layout (lane = pixel of the 196,
//     64 registers = the 128 channels of 2 lane halves), so residual add and the downsample LN are
//     register-only;
//   * ONE 16-bit map image in LDS ([256 px][128 ch], 272-byte rows): the depthwise phase reads it
//     (lane = channel, 2 waves per map row), keeps its LN outputs in registers until every wave has
//     finished reading, then overwrites the image, which the MLP reads as its MFMA B operand;
//   * the pointwise filters (2 x 256 KB per block) stream L2 -> LDS by LDS-DMA through a 2-slot
//     ring, one chunk (64 hidden units) in flight under the MFMAs of the previous one, one barrier
//     per chunk; fc1 accumulators -> GELU -> fc2 operand stay in registers (fused_mlp.hip).
#include "common.h"
#include "stage0.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct S1M;
template <> struct S1M<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct S1M<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 128, HW = 7, P = 49, G = 4, CT = 4, KS1 = 8;
constexpr int CN = 256;                            // channels after the downsample
constexpr int PO = 9;                              // output pixels per alert
constexpr int PITCH = 2 * C + 16;                  // 272
constexpr int MAPB = 256 * PITCH;                  // 69632
constexpr int W1ROW = 2 * C + 16, W2ROW = 80;      // FusedGeom<128>
constexpr int SUBBYTES = 32 * W1ROW + C * W2ROW;   // 18944
constexpr int SUBS = 2, NCHUNK = 8;
constexpr int CHUNKB = SUBS * SUBBYTES;            // 37888 = 37 x 1 KiB
constexpr int PIECES = CHUNKB / 1024;
constexpr float LN_EPS = 1e-6f;

// diagnostic builds of the timeline: workgroup 0, thread 0 stores the shader clock
#define STAMP(i)                                                                  \
  do {                                                                            \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
    if (a.wgt != nullptr && threadIdx.x == 0 && ((i) == 0 || (i) == 13))           \
      a.wgt[2 * blockIdx.x + ((i) == 13)] = wall_clock64();                        \
  } while (0)
static_assert(CHUNKB % 1024 == 0, "chunk must be whole LDS-DMA pieces");

__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// 8 values per lane -> v[0..1] = 64-lane totals of values (lane>>4)*2 + j
__device__ __forceinline__ void treduce8(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = swap_add32(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = swap_add16(v[i], v[i + 2]);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = group16_sum(v[i]);
}

// LayerNorm over the 128 channels of this lane's pixel (x[4][16] here + partner lane^32)
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

template <typename T>
__global__ __launch_bounds__(512, 2) void stage1_kernel(Stage1Args a) {
  using frag = typename S1M<T>::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* map = smem;
  unsigned char* ring = smem + MAPB;                              // 2 x CHUNKB
  float* b1s = reinterpret_cast<float*>(ring + 2 * CHUNKB);       // [512]
  float* red = b1s + 4 * C;                                       // [2][8][8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int a0 = blockIdx.x * G;
  const int nal = min(G, a.B - a0);
  const int p = wave * 32 + lr;                      // this lane's pixel slot (MFMA phases)
  const bool live = p < nal * P;
  const int pc = live ? p : 0;

  STAMP(0);
  // ---- stage input -> registers (accumulator layout) and the 16-bit map image
  f32x16 x[CT];
  {
    const float* src = a.x_in + ((size_t)a0 * P + pc) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 v = *reinterpret_cast<const float4*>(src + ct * 32 + 8 * qd + 4 * h);
        x[ct][4 * qd + 0] = live ? v.x : 0.f;
        x[ct][4 * qd + 1] = live ? v.y : 0.f;
        x[ct][4 * qd + 2] = live ? v.z : 0.f;
        x[ct][4 * qd + 3] = live ? v.w : 0.f;
      }
    regs_to_map<T>(x, map, p, h);
  }

  STAMP(1);   // input loaded
  const int chunk = wave & 1;                        // channel half owned in the depthwise phase
  const int cdw = chunk * 64 + lane;
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    // depthwise filters of this block (ordinary loads, completed BEFORE any LDS-DMA is issued)
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = bk.dw_w[t * C + cdw];
    const float dwbias = bk.dw_b[cdw], lng = bk.ln_w[cdw], lnb2 = bk.ln_b[cdw];
    const float b1v = bk.b1[tid];
#pragma unroll
    for (int t = 0; t < 49; ++t) asm volatile("" ::"v"(w[t]));
    asm volatile("" ::"v"(dwbias), "v"(lng), "v"(lnb2), "v"(b1v));
    STAMP(2 + 5 * j);   // filters touched
    __syncthreads();            // map complete; previous block's ring / b1s reads finished
    b1s[tid] = b1v;
    // chunk 0 of the pointwise filters -> ring slot 0, in flight under the depthwise phase
    const int npieces = (a.diag & 4) ? 0 : PIECES;
    for (int pc2 = wave; pc2 < npieces; pc2 += 8)
      __builtin_amdgcn_global_load_lds((gptr_t)(bk.wpk + (size_t)pc2 * 1024 + lane * 16),
                                       (lptr_t)(ring + pc2 * 1024), 16, 0, 0);

    STAMP(3 + 5 * j);   // DMA issued
    // ---- depthwise 7x7 + bias + LN: 56 (alert, row, channel-half) items over 8 waves = 7 rounds;
    //      LN outputs wait in registers (xnv) until every wave is done reading the image
    typedef T T8 __attribute__((ext_vector_type(8)));
    T8 xnv[7];                                       // packed 16-bit: 4 registers per round
    {
      const T* mi = reinterpret_cast<const T*>(map);
#pragma unroll
      for (int rd = 0; rd < 7; ++rd) {
        const int slot = rd * 4 + (wave >> 1);       // (alert, row)
        const int g = slot / HW, y = slot - g * HW;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (i < HW) ? dwbias : 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
          const int iy = y + ky - 3;
          if (iy < 0 || iy >= HW || (a.diag & 1)) continue;
          const T* row = mi + ((g * HW + iy) * HW) * (PITCH / 2) + cdw;
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
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = acc[i];
        treduce8(s);
        if ((lane & 15) == 0) {
          red[wave * 8 + (lane >> 4) * 2 + 0] = s[0];
          red[wave * 8 + (lane >> 4) * 2 + 1] = s[1];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] -= (red[wave * 8 + i] + red[(wave ^ 1) * 8 + i]) * (1.0f / C);
          s[i] = i < HW ? acc[i] * acc[i] : 0.f;
        }
        treduce8(s);
        if ((lane & 15) == 0) {
          red[64 + wave * 8 + (lane >> 4) * 2 + 0] = s[0];
          red[64 + wave * 8 + (lane >> 4) * 2 + 1] = s[1];
        }
        __syncthreads();
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) {
          const float var = (red[64 + wave * 8 + xx] + red[64 + (wave ^ 1) * 8 + xx]) * (1.0f / C);
          xnv[rd][xx] = (T)(acc[xx] * rsqrtf(var + LN_EPS) * lng + lnb2);
        }
      }
      // every wave has passed the last barrier above => nobody reads the image any more
      T* mo = reinterpret_cast<T*>(map);
#pragma unroll
      for (int rd = 0; rd < 7; ++rd) {
        const int slot = rd * 4 + (wave >> 1);
        T* dst = mo + (slot * HW) * (PITCH / 2) + cdw;
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) dst[xx * (PITCH / 2)] = xnv[rd][xx];
      }
    }
    STAMP(4 + 5 * j);   // depthwise done
    __syncthreads();            // LN image complete
    STAMP(5 + 5 * j);

    // ---- fc1 -> GELU -> fc2 over 8 chunks of 64 hidden units (ring), then x += gamma*(y + b2)
    {
      frag xf[KS1];
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(map + p * PITCH + ks * 32 + h * 16);
      f32x16 yacc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;
      // Ring: chunk ch+1 travels global -> registers (ordinary, non-blocking loads issued before the
      // MFMAs of chunk ch) -> LDS slot (ch+1)&1 after them; ONE barrier per chunk.  (LDS-DMA here
      // stalled the issuing wave for the whole transfer: ~4.4k cycles per 37 KB chunk.)
      constexpr int SPT = (CHUNKB / 16 + 511) / 512;     // 16-byte pieces per thread per chunk
      uint4 stg[SPT];
#pragma unroll 1
      for (int ch = 0; ch < NCHUNK; ++ch) {
        if (ch == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk 0 came by LDS-DMA
        __syncthreads();        // chunk ch is in its slot for everyone; slot (ch+1)&1 is free
        {   // (the fetch after the last chunk wraps to chunk 0: redundant, harmless, branch-free)
          const uint4* srcp =
              reinterpret_cast<const uint4*>(bk.wpk + (size_t)((ch + 1) & (NCHUNK - 1)) * CHUNKB);
#pragma unroll
          for (int i = 0; i < SPT; ++i) {
            const int q = tid + i * 512;
            stg[i] = srcp[q < CHUNKB / 16 ? q : CHUNKB / 16 - 1];
          }
        }
        const unsigned char* cb = ring + (ch & 1) * CHUNKB;
#pragma unroll 1
        for (int sub = 0; sub < ((a.diag & 2) ? 0 : SUBS); ++sub) {
          const unsigned char* w1s = cb + sub * SUBBYTES;
          const unsigned char* w2s = w1s + 32 * W1ROW;
          f32x16 hacc;
          const float* bp = b1s + (ch * SUBS + sub) * 32 + 4 * h;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * qd);
            hacc[4 * qd + 0] = bv.x;
            hacc[4 * qd + 1] = bv.y;
            hacc[4 * qd + 2] = bv.z;
            hacc[4 * qd + 3] = bv.w;
          }
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks) {
            const frag af = *reinterpret_cast<const frag*>(w1s + lr * W1ROW + ks * 32 + h * 16);
            hacc = S1M<T>::run(af, xf[ks], hacc);
          }
          frag hf[2];
#pragma unroll
          for (int r = 0; r < 16; ++r) hf[r >> 3][r & 7] = (T)((a.diag & 8) ? hacc[r] : gelu_fast(hacc[r]));
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const frag af = *reinterpret_cast<const frag*>(w2s + (ct * 32 + lr) * W2ROW + s2 * 32 + h * 16);
              yacc[ct] = S1M<T>::run(af, hf[s2], yacc[ct]);
            }
        }
        {
          uint4* dstp = reinterpret_cast<uint4*>(ring + ((ch + 1) & 1) * CHUNKB);
#pragma unroll
          for (int i = 0; i < SPT; ++i) {
            const int q = tid + i * 512;
            if (i < SPT - 1 || q < CHUNKB / 16) dstp[q] = stg[i];
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = ct * 32 + 8 * qd + 4 * h;
          const float4 bv = *reinterpret_cast<const float4*>(bk.b2 + c);
          const float4 gv = *reinterpret_cast<const float4*>(bk.gamma + c);
          x[ct][4 * qd + 0] += gv.x * (yacc[ct][4 * qd + 0] + bv.x);
          x[ct][4 * qd + 1] += gv.y * (yacc[ct][4 * qd + 1] + bv.y);
          x[ct][4 * qd + 2] += gv.z * (yacc[ct][4 * qd + 2] + bv.z);
          x[ct][4 * qd + 3] += gv.w * (yacc[ct][4 * qd + 3] + bv.w);
        }
      // the LN image was last read (xf) before the first chunk barrier: free to overwrite
      if (j == 0) regs_to_map<T>(x, map, p, h);
    }
    STAMP(6 + 5 * j);   // MLP done
  }
  if (a.tap_stage != nullptr && live) {
    float* tp = a.tap_stage + ((size_t)a0 * P + p) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<float4*>(tp + ct * 32 + 8 * qd + 4 * h) =
            make_float4(x[ct][4 * qd], x[ct][4 * qd + 1], x[ct][4 * qd + 2], x[ct][4 * qd + 3]);
  }

  // ---- downsample: LN + conv 2x2 s2 (128 -> 256): 36 output pixels x 256 channels, K = 512
  {
    f32x16 xn[CT];
    ln_regs(x, a.ds_lnw, a.ds_lnb, h, xn);
    regs_to_map<T>(xn, map, p, h);
    __syncthreads();
    STAMP(12);
    const int pt = wave & 1;                         // output-pixel tile (36 -> 2 tiles of 32)
    const int o = pt * 32 + lr;
    const bool olive = o < nal * PO;
    const int oc = olive ? o : 0;
    const int g = oc / PO, oo = oc - g * PO;
    const int oy = oo / 3, ox = oo - oy * 3;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int cot = (wave >> 1) + 4 * half;        // 8 output-channel tiles of 32 over 4 wave pairs
      f32x16 acc;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 bv = *reinterpret_cast<const float4*>(a.ds_b + cot * 32 + 8 * qd + 4 * h);
        acc[4 * qd + 0] = bv.x;
        acc[4 * qd + 1] = bv.y;
        acc[4 * qd + 2] = bv.z;
        acc[4 * qd + 3] = bv.w;
      }
      const T* dw = reinterpret_cast<const T*>(a.ds_w) + (size_t)(cot * 32 + lr) * (4 * C) + h * 8;
#pragma unroll 8
      for (int ks = 0; ks < ((a.diag & 32) ? 0 : 32); ++ks) {
        const int q = ks >> 3;                       // tap (ky*2 + kx): 8 k-steps of 16 channels each
        const int pin = g * P + (2 * oy + (q >> 1)) * HW + 2 * ox + (q & 1);
        const frag bf = *reinterpret_cast<const frag*>(map + pin * PITCH + (ks & 7) * 32 + h * 16);
        const frag af = *reinterpret_cast<const frag*>(dw + ks * 16);
        acc = S1M<T>::run(af, bf, acc);
      }
      if (olive) {
        float* dst = a.out + ((size_t)a0 * PO + o) * CN + cot * 32 + 4 * h;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<float4*>(dst + 8 * qd) =
              make_float4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
      }
    }
    STAMP(13);
  }
}

template <typename T> int launch_stage1_t(const Stage1Args& a, hipStream_t st) {
  constexpr size_t lds = (size_t)MAPB + 2 * CHUNKB + 4 * C * 4 + 2 * 8 * 8 * 4;
  auto kern = stage1_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.B + G - 1) / G), dim3(512), lds, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool stage1_supported(int prec, int c1, int c2) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && c1 == 128 && c2 == 256;
}

int launch_stage1(int prec, const Stage1Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16) return launch_stage1_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage1_t<f16_t>(a, st);
  btsbot_set_error("stage1: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
