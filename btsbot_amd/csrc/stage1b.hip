// Stage-1 megakernel, second layout (gfx950): TWO alerts' 7x7x128 maps per 256-thread workgroup,
// two workgroups per CU (same scheme as stage0b.hip):
//
//   2 x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]  ->  LN + conv 2x2 s2 (128 -> 256)
//
// (timm ConvNeXt stages[1].blocks / stages[2].downsample, reached from
// /root/reference/btsbot/architectures.py:108,132).  HBM sees [49][128] f32 in and [9][256] f32
// out per alert.  At B = 1024 the 512 workgroups are all resident at once (2 per CU) and drift
// apart, so one's depthwise phase (VALU + LDS) runs under the other's MLP (MFMA + GELU).
//   * residual stream fp32 in registers, 32x32 MFMA accumulator layout: wave = 32 pixel slots of
//     the 98, lane half h and 64 registers = the 128 channels;
//   * a 16-bit map image in LDS ([98 px][128 ch], 272-byte rows) holds x for the depthwise phase
//     (lane = channel, wave = (alert, channel half), 7 rounds = rows); its LN outputs go to a
//     second image that borrows ring slots 0..1 until the MLP has loaded its B operand from it;
//     LayerNorm sums meet the other channel half's through LDS, one barrier per round
//     (single-pass variance);
//   * the depthwise taps (in the operand type) arrive by LDS-DMA in ring slot 2, which the filter
//     ring only needs from the MLP's third chunk on: 7 tap registers instead of 49, which is what
//     keeps the residual tile out of scratch;
//   * pointwise filters: 16 KB chunks of 32 hidden units (W1 rows + gamma*W2 columns) through a
//     3-slot LDS-DMA ring straight from the plain row-major filters; the per-lane source address
//     applies the bank swizzles and the bit-2/bit-3 row swap that makes the fc1 accumulator the
//     fc2 B operand in plain k order (stage0b.hip); fc2 accumulates into the residual registers.
#include "common.h"
#include "stage0.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct SCM;
template <> struct SCM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct SCM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 128, HW = 7, PA = 49, G = 2, NPX = G * PA, CT = 4, HID = 512, KS1 = 8;
constexpr int CN = 256, PO = 9;                   // downsample: output channels, pixels per alert
constexpr int PITCH = 2 * C + 16;                 // 272 bytes per map row
constexpr int MAPB = NPX * PITCH;                 // 26656
constexpr int CHUNKB = 16384, NCH = HID / 32, NSLOT = 3;
constexpr int OFF_RING = MAPB;
constexpr int OFF_B1 = OFF_RING + NSLOT * CHUNKB; // 512 floats fc1 bias + 128 floats gamma*b2
constexpr int OFF_RED = OFF_B1 + (HID + C) * 4;   // [2 parities][4 waves][16]
constexpr int LDS_BYTES = OFF_RED + 2 * 4 * 16 * 4;   // 78880: two workgroups per CU
static_assert(MAPB % 16 == 0 && LDS_BYTES <= 80 * 1024, "LDS layout");
constexpr float LN_EPS = 1e-6f;
// per-block depthwise parameter image (launch_pack_s1par), fetched by LDS-DMA into ring slot 2,
// which the filter ring does not need before the MLP's first chunk is consumed:
// [49][128] taps in the operand type | dw bias | LN weight | LN bias (fp32) | zero pad to 16 KiB
constexpr int TAPB = 49 * C * 2;                  // 12544
constexpr int PARB = CHUNKB;                      // 16384 = 16 pieces, 4 per wave
static_assert(TAPB + 3 * C * 4 <= PARB, "parameter image layout");

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
// 16 values per lane -> v[0..3] = 64-lane totals of values (lane>>4)*4 + j
__device__ __forceinline__ void treduce16(float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = swap_add32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = swap_add16(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = group16_sum(v[i]);
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
__global__ __launch_bounds__(256, 2) void stage1b_kernel(Stage1Args a) {
  using frag = typename SCM<T>::frag;
  typedef T T8 __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* map = smem;
  unsigned char* ring = smem + OFF_RING;
  float* b1s = reinterpret_cast<float*>(smem + OFF_B1);
  float* b2s = b1s + HID;
  float* red = reinterpret_cast<float*>(smem + OFF_RED);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int a0 = blockIdx.x * G;
  const int nal = min(G, a.B - a0);
  const int p = wave * 32 + lr;                      // this lane's pixel slot (MFMA phases)
  const bool live = p < nal * PA;
  const bool inmap = p < NPX;
  const int pm = inmap ? p : 0;                      // row to read for slots beyond the image

  SC_STAMP(0);
  unsigned char* pimg = ring + 2 * CHUNKB;          // parameter image = ring slot 2 (see PARB)
  auto issue_params = [&](int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.blk[j].par + (wave * 4 + i) * 1024 + lane * 16),
                                       (lptr_t)(pimg + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  issue_params(0);   // lands under the input load
  // ---- stage input -> registers (accumulator layout) and the 16-bit map image
  f32x16 x[CT];
  {
    const float* src = a.x_in + ((size_t)a0 * PA + (live ? p : 0)) * C;
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
    if (inmap) regs_to_map<T>(x, map, p, h);
  }
  SC_STAMP(1);

  const int rot = (blockIdx.x * 5 + (blockIdx.x >> 4)) & (NCH - 1);   // chunk rotation, see issue()
  const int half = wave & 1;                         // channel half owned in the depthwise phase
  const int cdw = half * 64 + lane;
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const Stage0Blk& bk = a.blk[j];
    // ---- fc1 bias and gamma*b2: ordinary loads, issued BEFORE the block's first filter DMA
    //      (vmcnt retires in order: a load younger than a DMA would have to wait for it)
    const float b1v0 = bk.b1[tid], b1v1 = bk.b1[256 + tid];
    const float b2v = bk.gamma[tid & (C - 1)] * bk.b2[tid & (C - 1)];
    SC_STAMP(2 + 5 * j);
    wait_vm<0>();      // this wave's quarter of the parameter image (and the loads above) landed
    __syncthreads();   // ... everyone's; map complete (input / previous MLP); ring slots 0, 1 free
    b1s[tid] = b1v0;
    b1s[256 + tid] = b1v1;
    if (tid < C) b2s[tid] = b2v;
    const T* taps = reinterpret_cast<const T*>(pimg) + cdw;
    const float* pf = reinterpret_cast<const float*>(pimg + TAPB);
    const float dwbias = pf[cdw], lng = pf[C + cdw], lnb2 = pf[2 * C + cdw];

    // ---- pointwise filters: chunk = 32 hidden units = 16 pieces of 1 KiB, 4 per wave.
    //      pieces 0..7 : W1 rows (LDS row m <- hidden unit 32*ch + swap23(m)), 256-byte rows,
    //                    16-byte chunk c of row m at position c ^ (m & 15)
    //      pieces 8..15: gamma*W2 columns 32*ch .. +31 of the 128 channel rows, 64-byte rows,
    //                    chunk c of row r at position c ^ F[(r >> 2) & 3]
    const unsigned char* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave * 4 + i;
      if (pc < 8) {
        const int m = pc * 4 + (lane >> 4);
        const int hid = (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);   // swap bits 2 and 3
        wsrc[i] = bk.w1 + (size_t)hid * (C * 2) + (((lane & 15) ^ (m & 15)) << 4);
      } else {
        const int r = (pc - 8) * 16 + (lane >> 2);
        wsrc[i] = bk.w2g + (size_t)r * (HID * 2) + (((lane & 3) ^ swz4(r)) << 4);
      }
    }
    // chunk ch adds 32 W1 rows (8192 B) resp. 32 W2 columns (64 B)
    const int wstep0 = wave < 2 ? 32 * C * 2 : 64;
    // (every workgroup walks the 16 chunks in its own rotation -- fc2 sums over the hidden units,
    //  so the order is free -- which keeps the 512 workgroups off the same L2 lines)
    auto issue = [&](int ch) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (size_t)((ch + rot) & (NCH - 1)) * wstep0),
                                         (lptr_t)(ring + ((ch + 2) % NSLOT) * CHUNKB + (wave * 4 + i) * 1024),
                                         16, 0, 0);
    };
    SC_STAMP(3 + 5 * j);

    // ---- depthwise 7x7 + bias + LN: wave = (alert, channel half), round = map row (a real loop:
    //      unrolled, the seven rounds cost ~36 live registers each and the residual tile went to
    //      scratch).  The LN outputs go straight to a second image in ring slots 0..1, which the
    //      filter ring only claims after the MLP has taken its B operand from it.
    unsigned char* stg = ring;                       // [98][PITCH] = 26656 B <= 2 slots
    {
      const T* mi = reinterpret_cast<const T*>(map);
      const int g = wave >> 1;                       // alert
#pragma unroll 1
      for (int y = 0; y < HW; ++y) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (i < HW) ? dwbias : 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
          const int iy = y + ky - 3;
          if (iy < 0 || iy >= HW) continue;          // wave-uniform
          const T* row = mi + ((g * HW + iy) * HW) * (PITCH / 2) + cdw;
          float in[HW], w[7];
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) w[kx] = (float)taps[(ky * 7 + kx) * C];
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
        // LN over 128 channels = this wave's 64 lanes + the partner wave's: sums and sums of
        // squares of the 7 pixels in one transposing reduction, exchanged through LDS
        float s[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s[i] = acc[i];
          s[8 + i] = acc[i] * acc[i];
        }
        treduce16(s);
        float* myred = red + ((y & 1) * 4 + wave) * 16;
        if ((lane & 15) == 0) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) myred[(lane >> 4) * 4 + jj] = s[jj];
        }
        __syncthreads();
        const float* pred = red + ((y & 1) * 4 + (wave ^ 1)) * 16;
        T* dst = reinterpret_cast<T*>(stg) + ((g * HW + y) * HW) * (PITCH / 2) + cdw;
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) {
          const float mean = (myred[xx] + pred[xx]) * (1.0f / C);
          const float var = (myred[8 + xx] + pred[8 + xx]) * (1.0f / C) - mean * mean;
          dst[xx * (PITCH / 2)] = (T)((acc[xx] - mean) * rsqrtf(var + LN_EPS) * lng + lnb2);
        }
      }
    }
    SC_STAMP(4 + 5 * j);   // depthwise done
    __syncthreads();   // LN image complete; the taps (ring slot 2) are dead
    issue(0);          // chunk ch lives in slot (ch + 2) % 3: chunk 0 can start right away
    SC_STAMP(5 + 5 * j);

    // ---- fc1 -> GELU -> fc2 over 16 chunks; fc2 accumulates into x (gamma is in the filter)
    {
      frag xf[KS1];
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(stg + pm * PITCH + ks * 32 + h * 16);
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
      __syncthreads();   // ... everyone's: slots 0..1 may be overwritten
      issue(1);
      issue(2);
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        // VM order of a wave: chunk 0, chunk 1, chunk 2, then chunk ch+2 at iteration ch >= 1
        if (ch == 0) wait_vm<8>();
        else if (ch + 1 < NCH) wait_vm<4>();
        else wait_vm<0>();
        __syncthreads();   // chunk ch has landed for everyone; chunk ch-1 is read out
        const unsigned char* w1s = ring + ((ch + 2) % NSLOT) * CHUNKB;
        const unsigned char* w2s = w1s + 8192;
        frag a1[KS1], a2[CT][2];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
          a1[ks] = *reinterpret_cast<const frag*>(w1s + lr * 256 + (((ks * 2 + h) ^ (lr & 15)) << 4));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int r = ct * 32 + lr;
            a2[ct][s2] = *reinterpret_cast<const frag*>(w2s + r * 64 + (((s2 * 2 + h) ^ swz4(r)) << 4));
          }
        f32x16 hacc;
        // accumulator row (r&3) + 8(r>>2) + 4h holds hidden unit 32ch + (r&3) + 4((r>>2)&1) + 8h + 16(r>>3)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 bv = *reinterpret_cast<const float4*>(b1s + ((ch + rot) & (NCH - 1)) * 32 +
                                                             4 * (qd & 1) + 8 * h + 16 * (qd >> 1));
          hacc[4 * qd + 0] = bv.x;
          hacc[4 * qd + 1] = bv.y;
          hacc[4 * qd + 2] = bv.z;
          hacc[4 * qd + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) hacc = SCM<T>::run(a1[ks], xf[ks], hacc);
        // the next chunk's LDS-DMA is issued HERE: an LDS-DMA holds the issuing wave for ~90 cycles
        // per instruction, which now passes while the fc1 MFMA chain drains
        if (ch >= 1 && ch + 2 < NCH) issue(ch + 2);
        // GELU in two halves, each followed by the fc2 MFMAs that consume it: the first half's
        // MFMAs run while the second half's GELU issues
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          frag hf;
#pragma unroll
          for (int r = 0; r < 8; ++r) hf[r] = (T)gelu_for<T>(hacc[8 * s2 + r]);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) x[ct] = SCM<T>::run(a2[ct][s2], hf, x[ct]);
        }
      }
      if (j == 0) {      // next block's parameter image -> ring slot 2 (chunk 15 must be read out)
        __syncthreads();
        issue_params(1);
      }
      // the map still holds the block's INPUT; the next block's depthwise phase wants the new x
      if (j == 0 && inmap) regs_to_map<T>(x, map, p, h);
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
    {
      f32x16 xn[CT];
      ln_regs(x, a.ds_lnw, a.ds_lnb, h, xn);
      if (inmap) regs_to_map<T>(xn, map, p, h);   // (the image was last read before chunk 0's barrier)
    }
    __syncthreads();
    SC_STAMP(12);
    const int o = lr;                                // output pixel slot: 18 of 32 used
    const bool olive = o < nal * PO;
    const int oc = o < G * PO ? o : 0;
    const int g = oc / PO, oo = oc - g * PO;
    const int oy = oo / 3, ox = oo - oy * 3;
#pragma unroll 1
    for (int tl = 0; tl < 2; ++tl) {
      const int cot = wave + 4 * tl;                 // 8 output-channel tiles of 32 over 4 waves
      const T* dw = reinterpret_cast<const T*>(a.ds_w) + (size_t)(cot * 32 + lr) * (4 * C) + h * 8;
      f32x16 acc;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 bv = *reinterpret_cast<const float4*>(a.ds_b + cot * 32 + 8 * qd + 4 * h);
        acc[4 * qd + 0] = bv.x;
        acc[4 * qd + 1] = bv.y;
        acc[4 * qd + 2] = bv.z;
        acc[4 * qd + 3] = bv.w;
      }
#pragma unroll 1
      for (int kh = 0; kh < 2; ++kh) {               // two halves of K: 16 filter fragments in flight
        frag af[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) af[k2] = *reinterpret_cast<const frag*>(dw + (kh * 16 + k2) * 16);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const int ks = kh * 16 + k2;
          const int q = ks >> 3;                     // tap (ky*2 + kx): 8 k-steps of 16 channels each
          const int pin = g * PA + (2 * oy + (q >> 1)) * HW + 2 * ox + (q & 1);
          const frag bf = *reinterpret_cast<const frag*>(map + pin * PITCH + (ks & 7) * 32 + h * 16);
          acc = SCM<T>::run(af[k2], bf, acc);
        }
      }
      if (olive) {
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

// one block's depthwise parameter image (see TAPB / PARB above); taps: tap-major [49][128] fp32
template <typename T>
__global__ void pack_s1par_kernel(const float* __restrict__ taps, const float* __restrict__ dw_b,
                                  const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                  unsigned char* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 49 * C) reinterpret_cast<T*>(out)[i] = (T)taps[i];
  if (i < (PARB - TAPB) / 4) {
    float v = 0.f;
    if (i < C) v = dw_b[i];
    else if (i < 2 * C) v = ln_w[i - C];
    else if (i < 3 * C) v = ln_b[i - 2 * C];
    reinterpret_cast<float*>(out + TAPB)[i] = v;
  }
}

template <typename T> int launch_stage1b_t(const Stage1Args& a, hipStream_t st) {
  auto kern = stage1b_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.B + G - 1) / G), dim3(256), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

size_t s1par_bytes() { return PARB; }

int launch_pack_s1par(int prec, const float* taps, const float* dw_b, const float* ln_w,
                      const float* ln_b, void* out, hipStream_t st) {
  unsigned char* o = reinterpret_cast<unsigned char*>(out);
  const dim3 grid((49 * C + 255) / 256), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_s1par_kernel<bf16_t>, grid, blk, 0, st, taps, dw_b, ln_w, ln_b, o);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_s1par_kernel<f16_t>, grid, blk, 0, st, taps, dw_b, ln_w, ln_b, o);
  else {
    btsbot_set_error("pack_s1par: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

bool stage1_supported(int prec, int c1, int c2) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && c1 == 128 && c2 == 256;
}

// Needs Stage0Blk::par (launch_pack_s1par), ::w1 (plain [512][128]) and Stage0Blk::w2g (gamma-scaled [128][512]), 16-bit.
int launch_stage1b(int prec, const Stage1Args& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16) return launch_stage1b_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage1b_t<f16_t>(a, st);
  btsbot_set_error("stage1b: unsupported precision %d", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
