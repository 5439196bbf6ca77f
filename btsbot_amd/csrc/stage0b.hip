// Stage-0 megakernel, second layout (gfx950): the same computation as stage0.hip --
//
//   stem (conv 4x4 s4 + LN)  ->  2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]
//                            ->  downsample (LN + conv 2x2 s2)  ->  [49][128] f32
//
// (timm ConvNeXt stem / stages[0] / stages[1].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132) -- re-cut so that TWO workgroups share a CU:
// stage0.hip's single 512-thread workgroup per CU runs its phases in lockstep (depthwise = VALU +
// LDS, MLP = MFMA + VALU, parameter fetches = latency) and nothing overlaps; two independent
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
#include "common.h"
#include "stage0.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct SBM;
template <> struct SBM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct SBM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 64, HW = 15, P = 225, CT = 2, HID = 256;
constexpr int PITCH = 2 * C + 16;                 // 144 bytes per map row
constexpr int MAPB = 256 * PITCH;                 // 36864
constexpr int CHUNKB = 8192, NCH = HID / 32, NSLOT = 3;
constexpr int RINGB = NSLOT * CHUNKB;             // 24576
// per-block fp32 parameter image (packed once by launch_pack_s0par, fetched by LDS-DMA):
// [49][64] depthwise taps | dw bias | LN weight | LN bias | fc1 bias [256] | gamma*b2 [64] | pad
constexpr int PAR_DWB = 49 * C, PAR_LNW = PAR_DWB + C, PAR_LNB = PAR_LNW + C, PAR_B1 = PAR_LNB + C,
              PAR_B2 = PAR_B1 + HID, PAR_FLOATS = 4096;
static_assert(PAR_B2 + C <= PAR_FLOATS, "parameter image layout");
constexpr int PARB = PAR_FLOATS * 4;              // 16384 = 16 LDS-DMA pieces, 4 per wave
constexpr int OFF_RING = MAPB;
constexpr int OFF_PAR = OFF_RING + RINGB;
constexpr int OFF_B1 = OFF_PAR + PARB;            // 256 + 64 floats: this block's fc1 bias, gamma*b2
constexpr int OFF_RED = OFF_B1 + (HID + C) * 4;   // 4 waves x 32 floats
constexpr int LDS_BYTES = OFF_RED + 4 * 32 * 4;   // 79616: two workgroups per CU
constexpr float LN_EPS = 1e-6f;

#define SB_STAMP(i)                                                                \
  do {                                                                             \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
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
// 16 values per lane -> v[0..3] = 64-lane totals of values (lane>>4)*4 + j
__device__ __forceinline__ void treduce16(float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = swap_add32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = swap_add16(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = group16_sum(v[i]);
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

template <typename T>
__global__ __launch_bounds__(256, 2) void stage0b_kernel(Stage0Args a) {
  using frag = typename SBM<T>::frag;
  typedef T T8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* map = smem;
  unsigned char* ring = smem + OFF_RING;
  const float* par = reinterpret_cast<const float*>(smem + OFF_PAR);
  float* b1s = reinterpret_cast<float*>(smem + OFF_B1);
  float* b2s = b1s + HID;
  float* red = reinterpret_cast<float*>(smem + OFF_RED);

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

  SB_STAMP(0);
  // per-block parameter image: 16 pieces, wave w takes w, w+4, w+8, w+12
  auto issue_params = [&](int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.blk[j].par + (wave + 4 * i) * 1024 + lane * 16),
                                       (lptr_t)(smem + OFF_PAR + (wave + 4 * i) * 1024), 16, 0, 0);
  };
  issue_params(0);   // lands under the stem
  // rows 225..255 of the image are padding: keep them finite
  for (int i = tid; i < (256 - P) * PITCH / 4; i += 256)
    reinterpret_cast<unsigned*>(map + P * PITCH)[i] = 0u;

  // ============================ stem: conv 4x4 s4 + LN =====================================
  f32x16 x[2][CT];
  {
    const float* src = a.img + (size_t)alert * 3 * 63 * 63;
    const T* sw = reinterpret_cast<const T*>(a.stem_w);
    frag af[3][CT];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        af[ci][ct] = *reinterpret_cast<const frag*>(sw + (ct * 32 + lr) * 48 + ci * 16 + h * 8);
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
        frag bf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bf[e] = (T)v0.v[e];
          bf[4 + e] = (T)v1.v[e];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) x[t][ct] = SBM<T>::run(af[ci][ct], bf, x[t][ct]);
      }
      ln_regs(x[t], a.stem_lnw, a.stem_lnb, h, x[t]);
      regs_to_map<T>(x[t], map, pix[t], h);
      if (a.tap_stem != nullptr && live[t])
        regs_to_tap(x[t], a.tap_stem + ((size_t)alert * P + pix[t]) * C, h);
    }
  }
  SB_STAMP(1);   // stem done

  // ============================ two ConvNeXt blocks ========================================
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    SB_STAMP(2 + 5 * j);
    wait_vm<0>();      // this wave's quarter of the block's parameter image has landed
    __syncthreads();   // ... everyone's; map complete (stem / previous MLP); ring free
    b1s[tid] = par[PAR_B1 + tid];                  // the MLP reads these while the NEXT block's
    if (tid < C) b2s[tid] = par[PAR_B2 + tid];     // image is already arriving

    // ---- pointwise filters: chunk = 32 hidden units = 8 pieces of 1 KiB, 2 per wave.
    //      pieces 0..3: W1 rows (LDS row m <- hidden unit 32*ch + swap23(m)), 128-byte rows,
    //                   16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
    //      pieces 4..7: gamma*W2 columns 32*ch .. +31 of the 64 channel rows, 64-byte rows,
    //                   chunk c of row r at position c ^ F[(r >> 2) & 3]
    const unsigned char* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave * 2 + i;
      if (pc < 4) {
        const int m = pc * 8 + (lane >> 3);
        const int hid = (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);   // swap bits 2 and 3
        wsrc[i] = bk.w1 + (size_t)hid * (C * 2) + (((lane & 7) ^ ((m >> 1) & 7)) << 4);
      } else {
        const int r = (pc - 4) * 16 + (lane >> 2);
        wsrc[i] = bk.w2g + (size_t)r * (HID * 2) + (((lane & 3) ^ swz4(r)) << 4);
      }
    }
    // chunk ch adds 32 W1 rows (4096 B) resp. 32 W2 columns (64 B)
    const int wstep0 = wave < 2 ? 32 * C * 2 : 64;
    auto issue = [&](int ch) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)ch * wstep0),
                                         (lptr_t)(ring + (ch % NSLOT) * CHUNKB + (wave * 2 + i) * 1024),
                                         16, 0, 0);
    };
    issue(0);
    issue(1);
    SB_STAMP(3 + 5 * j);
    const float dwbias = par[PAR_DWB + lane], lng = par[PAR_LNW + lane], lnb2 = par[PAR_LNB + lane];

    // ---- depthwise 7x7 + bias + LN: lane = channel, wave = map rows wave, wave+4, ...; the LN
    //      outputs wait in registers (xnv) until every wave is done reading the image
    T8 xnv[4][2];
    {
      const T* mi = reinterpret_cast<const T*>(map);
      float* myred = red + wave * 32;
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {
        const int y = rd * 4 + wave;
        const bool valid = y < HW;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (i < HW) ? dwbias : 0.f;
        if (valid) {
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) {
            const int iy = y + ky - 3;
            if (iy < 0 || iy >= HW) continue;
            const T* row = mi + (iy * HW) * (PITCH / 2) + lane;
            float in[HW], w[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) w[kx] = par[(ky * 7 + kx) * C + lane];
#pragma unroll
            for (int xx = 0; xx < HW; ++xx) in[xx] = (float)row[xx * (PITCH / 2)];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
              for (int xx = 0; xx < HW; ++xx) {
                const int ix = xx + kx - 3;
                if (ix >= 0 && ix < HW) acc[xx] = fmaf(in[ix], w[kx], acc[xx]);
              }
          }
        }
        // LN over the 64 channels (= lanes) of each of the row's 15 pixels: transposing reduction,
        // totals broadcast through this wave's own LDS words (no workgroup barrier)
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = acc[i];
        treduce16(s);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) myred[(lane >> 4) * 4 + jj] = s[jj];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local hand-off through LDS
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] -= myred[i] * (1.0f / C);
          s[i] = acc[i] * acc[i];
        }
        treduce16(s);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) myred[16 + (lane >> 4) * 4 + jj] = s[jj];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int xx = 0; xx < 16; ++xx) {
          const float rstd = rsqrtf(myred[16 + xx] * (1.0f / C) + LN_EPS);
          xnv[rd][xx >> 3][xx & 7] = (T)(acc[xx] * rstd * lng + lnb2);
        }
        __builtin_amdgcn_sched_barrier(0);   // one round at a time: registers (see stage1b.hip)
      }
    }
    __syncthreads();   // nobody reads the image (or the taps) any more
    if (j == 0) issue_params(1);   // lands under this block's MLP
    {
      T* mo = reinterpret_cast<T*>(map);
#pragma unroll
      for (int rd = 0; rd < 4; ++rd) {
        const int y = rd * 4 + wave;
        if (y < HW) {
          T* dst = mo + (y * HW) * (PITCH / 2) + lane;
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) dst[xx * (PITCH / 2)] = xnv[rd][xx >> 3][xx & 7];
        }
      }
    }
    SB_STAMP(4 + 5 * j);   // depthwise done
    __syncthreads();   // LN image complete
    SB_STAMP(5 + 5 * j);

    // ---- fc1 -> GELU -> fc2 over 8 chunks; fc2 accumulates into x (gamma is in the filter)
    {
      frag xf[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          xf[t][ks] = *reinterpret_cast<const frag*>(map + pix[t] * PITCH + ks * 32 + h * 16);
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
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        // this wave's pieces of chunk ch have landed once only the younger chunks are outstanding
        // (VM order of a wave: chunk 0, chunk 1, [block 0: next parameter image, 4], chunk 2, ...)
        if (ch < 2 && j == 0) wait_vm<6>();
        else if (ch + 1 < NCH) wait_vm<2>();
        else wait_vm<0>();
        __syncthreads();   // ... everyone's; chunk ch-1 is read out (and xf is loaded, ch == 0)
        const unsigned char* w1s = ring + (ch % NSLOT) * CHUNKB;
        const unsigned char* w2s = w1s + 4096;
        frag a1[4], a2[CT][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          a1[ks] = *reinterpret_cast<const frag*>(w1s + lr * 128 + (((ks * 2 + h) ^ ((lr >> 1) & 7)) << 4));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int r = ct * 32 + lr;
            a2[ct][s2] = *reinterpret_cast<const frag*>(w2s + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
          }
        // both column blocks' fc1 first: the second one's MFMAs run under the first one's GELU
        f32x16 hacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // accumulator row (r&3) + 8(r>>2) + 4h holds hidden unit 32ch + (r&3) + 4((r>>2)&1) + 8h + 16(r>>3)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const float4 bv = *reinterpret_cast<const float4*>(b1s + ch * 32 + 4 * (qd & 1) + 8 * h +
                                                               16 * (qd >> 1));
            hacc[t][4 * qd + 0] = bv.x;
            hacc[t][4 * qd + 1] = bv.y;
            hacc[t][4 * qd + 2] = bv.z;
            hacc[t][4 * qd + 3] = bv.w;
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) hacc[t] = SBM<T>::run(a1[ks], xf[t][ks], hacc[t]);
        }
        // (issued here, not at the barrier: an LDS-DMA holds the issuing wave ~90 cycles per
        //  instruction, which now passes while the fc1 MFMAs drain)
        if (ch + 2 < NCH) issue(ch + 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // GELU in two halves, each followed by the fc2 MFMAs that consume it
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            frag hf;
#pragma unroll
            for (int r = 0; r < 8; ++r) hf[r] = (T)gelu_for<T>(hacc[t][8 * s2 + r]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) x[t][ct] = SBM<T>::run(a2[ct][s2], hf, x[t][ct]);
          }
        }
      }
      // the LN image was last read (xf) before the first chunk barrier: free to overwrite
      if (j == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) regs_to_map<T>(x[t], map, pix[t], h);
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
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 xn[CT];
      ln_regs(x[t], a.ds_lnw, a.ds_lnb, h, xn);
      regs_to_map<T>(xn, map, pix[t], h);
    }
    // wave -> 32 output channels (cot = wave) x the 49 output pixels (2 column blocks); K = 4 x 64
    const T* dw = reinterpret_cast<const T*>(a.ds_w) + (size_t)(wave * 32 + lr) * 256 + h * 8;
    frag af[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) af[ks] = *reinterpret_cast<const frag*>(dw + ks * 16);
    __syncthreads();
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

// one block's fp32 parameter image (see PAR_* above)
__global__ void pack_s0par_kernel(const float* __restrict__ taps, const float* __restrict__ dw_b,
                                  const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                  const float* __restrict__ b1, const float* __restrict__ b2,
                                  const float* __restrict__ gamma, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PAR_FLOATS) return;
  float v = 0.f;
  if (i < PAR_DWB) v = taps[i];
  else if (i < PAR_LNW) v = dw_b[i - PAR_DWB];
  else if (i < PAR_LNB) v = ln_w[i - PAR_LNW];
  else if (i < PAR_B1) v = ln_b[i - PAR_LNB];
  else if (i < PAR_B2) v = b1[i - PAR_B1];
  else if (i < PAR_B2 + C) v = gamma[i - PAR_B2] * b2[i - PAR_B2];
  out[i] = v;
}

template <typename T> int launch_stage0b_t(const Stage0Args& a, hipStream_t st) {
  auto kern = stage0b_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

size_t s0par_bytes() { return PARB; }

// taps: the block's depthwise filter tap-major [49][64] fp32; the others the master parameters
int launch_pack_s0par(const float* taps, const float* dw_b, const float* ln_w, const float* ln_b,
                      const float* b1, const float* b2, const float* gamma, void* out,
                      hipStream_t st) {
  hipLaunchKernelGGL(pack_s0par_kernel, dim3(PAR_FLOATS / 256), dim3(256), 0, st, taps, dw_b, ln_w,
                     ln_b, b1, b2, gamma, reinterpret_cast<float*>(out));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

bool stage0_supported(int prec, int c0) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && c0 == 64;
}

// Needs Stage0Blk::par, ::w1 (plain [256][64]) and Stage0Blk::w2g (gamma-scaled [64][256]), 16-bit.
int launch_stage0b(int prec, const Stage0Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16) return launch_stage0b_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage0b_t<f16_t>(a, st);
  btsbot_set_error("stage0b: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
