// Stage-2 megakernel (gfx950, 16-bit modes, C = 256): ALL blocks of the 3x3 stage in one launch,
//
//   depth x [ dwconv 7x7 + LN -> fc1 -> GELU -> fc2 -> layer-scale -> +x ]       x: [B][9][256] f32
//
// (timm ConvNeXt stages[2].blocks, reached from /root/reference/btsbot/architectures.py:108,132)
// with FOUR alerts (36 pixels) resident per workgroup: the residual stream stays in registers and
// the 18.9 MB hidden tensor of every block never exists.  What moves instead is the filters: each
// workgroup streams W1 and gamma*W2 of every block (1 MB per block) L2 -> LDS through a 4-slot
// LDS-DMA ring of 32 KB chunks (32 hidden units: 32 W1 rows + 32 W2 columns), three chunks in
// flight -- the launch is bound by that stream (~66 GB/s per CU when every CU reads the same
// L2-resident filters), ~1100 cycles per chunk, against the two launches per block it replaces
// whose hidden tensor round trip and ramps cost about twice that.
//
// 512 threads: waves 4..7 only issue the LDS-DMA (a deep DMA queue stalls the issuing wave, so the
// waves that compute must not issue); waves 0..2 own 16 pixel slots each (48 >= 36): x in the 16x16 MFMA accumulator
// layout (lane = pixel, 4 lane groups x 4 registers x 16 tiles = 256 channels), v_mfma 16x16x32
// with the filters as the A operand; the W1 rows of a chunk are fetched in an order that makes a
// lane's two fc1 accumulator tiles exactly its fc2 B operand (hidden 8q .. 8q+7), so fc1 -> GELU
// -> fc2 never leaves registers, and fc2 accumulates straight into x (layer scale is folded into
// the filter).  Depthwise + LayerNorm: wave 0..3 = alert, lane = 4
// channels, the 3x3 map through a 16-bit LDS image that the LN output overwrites in place.
#include "common.h"

#include "stage2m.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct MM;
template <> struct MM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 256, HID = 1024, GA = 4, NPX = GA * 9, NT16 = C / 16;   // 36 pixels, 16 channel tiles
constexpr int IROW = C * 2;                        // 512-byte image rows
constexpr int IMGB = 48 * IROW;                    // 24576
constexpr int W1CB = 32 * IROW;                    // 16384: 32 hidden rows x 256 channels
constexpr int W2CB = C * 64;                       // 16384: 256 channel rows x 32 hidden
constexpr int CHUNKB = W1CB + W2CB, NCH = HID / 32, NSLOT = 4;
constexpr int OFF_RING = IMGB;
constexpr int OFF_B1 = OFF_RING + NSLOT * CHUNKB;  // 1024 floats
constexpr int LDS_BYTES = OFF_B1 + HID * 4;        // 159744
static_assert(LDS_BYTES <= 160 * 1024, "LDS layout");
constexpr float LN_EPS = 1e-6f;

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int swz4(int row) {   // F[(row >> 2) & 3], F = {0,3,2,1}
  return (4 - ((row >> 2) & 3)) & 3;
}
// byte offset of channel c (multiple of 4) of pixel row r in the image: 16-byte chunk c/8 of row r at
// position (c/8) ^ (r & 15)
__device__ __forceinline__ int img_off(int r, int c) {
  return r * IROW + ((((c >> 3) ^ (r & 15))) << 4) + (c & 7) * 2;
}

template <typename T>
__global__ __launch_bounds__(512) void stage2m_kernel(Stage2Args a) {
  using frag = typename MM<T>::frag;
  typedef T T4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem;
  unsigned char* ring = smem + OFF_RING;
  float* b1s = reinterpret_cast<float*>(smem + OFF_B1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;                     // waves 4..7 only feed the ring
  const int lw = wave & 3;                           // loader index / alert of the depthwise phase
  const int lp = lane & 15, q = lane >> 4;
  const int a0 = blockIdx.x * GA;
  const int nal = min(GA, a.B - a0);
  const int p = wave * 16 + lp;                      // pixel slot (waves 0..2)
  const bool owner = wave < 3;                       // waves 0..2 hold the residual tile
  const bool live = owner && p < nal * 9;
  const int NG = a.depth * NCH;                      // chunks of the whole launch

  // ---- filter ring: chunk g = (block g / 32, hidden units 32 (g % 32) .. +31) = 32 pieces of
  //      1 KiB, 8 per wave.  Pieces 0..15: W1 rows, two 512-byte rows per piece, LDS row m holds
  //      hidden unit perm(m) (below), 16-byte chunk c of row m at position c ^ (m & 15).
  //      Pieces 16..31: gamma*W2, sixteen 64-byte rows (channels) per piece, chunk c of row r at
  //      position c ^ F[(r >> 2) & 3].
  //      perm: rows 0..15 = fc1 tile 0, rows 16..31 = tile 1; row 16 t + 4 qq + r <- hidden
  //      8 qq + 4 t + r, so that lane group qq's accumulators (tile 0, tile 1) = hidden 8qq .. 8qq+7.
  int srcoff[8];     // byte offset inside (w1 | w2g) of this lane's 16 bytes, chunk 0
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pc = lw * 8 + i;
    if (pc < 16) {
      const int m = pc * 2 + (lane >> 5);
      const int t = m >> 4, qq = (m >> 2) & 3, r = m & 3;
      const int hid = 8 * qq + 4 * t + r;
      srcoff[i] = hid * IROW + (((lane & 31) ^ (m & 15)) << 4);
    } else {   // chunk-major image [32 chunks][256 rows][32 hidden]: a piece is 1 KiB contiguous
      const int r = (pc - 16) * 16 + (lane >> 2);
      srcoff[i] = r * 64 + (((lane & 3) ^ swz4(r)) << 4);
    }
  }
  const bool w1wave = lw < 2;                        // loaders 0,1 fetch W1 pieces, loaders 2,3 W2 pieces
  // Every workgroup walks the 32 chunks of a block in its own rotation (fc2 sums over the hidden
  // units, so the order is free): at any moment the 256 CUs pull different lines out of L2
  // instead of all hammering the same 32 KB.
  const int rot = (blockIdx.x * 5 + (blockIdx.x >> 3)) & 31;
  auto issue = [&](int g) {
    const Stage2Blk& bk = a.blk[g >> 5];
    const int ch = (g + rot) & 31;
    const unsigned char* base = w1wave ? bk.w1 + (size_t)ch * W1CB : bk.w2g + (size_t)ch * W2CB;
    unsigned char* slot = ring + (g % NSLOT) * CHUNKB;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + srcoff[i]),
                                       (lptr_t)(slot + (lw * 8 + i) * 1024), 16, 0, 0);
  };
  if (loader && !(a.diag & 2)) {
    issue(0);
    if (1 < NG) issue(1);
    if (2 < NG) issue(2);
  }

  // ---- stage input -> registers: x[t][r] = channel 16 t + 4 q + r of this lane's pixel
  f32x4 x[NT16];
#pragma unroll
  for (int t = 0; t < NT16; ++t) x[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float* src = a.x_in + ((size_t)a0 * 9 + p) * C + 4 * q;
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(src + 16 * t);
      x[t] = f32x4{v.x, v.y, v.z, v.w};
    }
  }

  int g = 0;                                         // next chunk to consume
#pragma unroll 1
  for (int j = 0; j < a.depth; ++j) {
    const Stage2Blk& bk = a.blk[j];
    // ---- x (16-bit) -> image, fc1 bias -> LDS, x += gamma * b2
    if (owner) {
#pragma unroll
      for (int t = 0; t < NT16; ++t) {
        T4 v;
        v[0] = (T)x[t][0];
        v[1] = (T)x[t][1];
        v[2] = (T)x[t][2];
        v[3] = (T)x[t][3];
        *reinterpret_cast<T4*>(img + img_off(p, 16 * t + 4 * q)) = v;
      }
#pragma unroll
      for (int t = 0; t < NT16; ++t) {
        const float4 gv = *reinterpret_cast<const float4*>(bk.gamma + 16 * t + 4 * q);
        const float4 bv = *reinterpret_cast<const float4*>(bk.b2 + 16 * t + 4 * q);
        x[t][0] += gv.x * bv.x;
        x[t][1] += gv.y * bv.y;
        x[t][2] += gv.z * bv.z;
        x[t][3] += gv.w * bv.w;
      }
    }
    if (!loader)
      *reinterpret_cast<float4*>(b1s + 4 * tid) = *reinterpret_cast<const float4*>(bk.b1 + 4 * tid);
    __syncthreads();   // image rows of every pixel tile are written

    // ---- depthwise 7x7 (central 5x5 taps) + LN: wave = alert, lane = channels 4 lane .. +3;
    //      the LN output overwrites the image rows of this alert in place
    if (!loader && wave < nal) {
      const int c4 = 4 * lane;
      f32x2 in[9][2];
#pragma unroll
      for (int pp = 0; pp < 9; ++pp) {
        const T4 v = *reinterpret_cast<const T4*>(img + img_off(wave * 9 + pp, c4));
        in[pp][0] = f32x2{(float)v[0], (float)v[1]};
        in[pp][1] = f32x2{(float)v[2], (float)v[3]};
      }
      const float4 b4 = *reinterpret_cast<const float4*>(bk.dw_b + c4);
      f32x2 acc[9][2];
#pragma unroll
      for (int pp = 0; pp < 9; ++pp) {
        acc[pp][0] = f32x2{b4.x, b4.y};
        acc[pp][1] = f32x2{b4.z, b4.w};
      }
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const float4 w4 = *reinterpret_cast<const float4*>(bk.dw_w + ((dy + 3) * 7 + dx + 3) * C + c4);
          const f32x2 w0 = {w4.x, w4.y}, w1 = {w4.z, w4.w};
#pragma unroll
          for (int oy = 0; oy < 3; ++oy)
#pragma unroll
            for (int ox = 0; ox < 3; ++ox) {
              const int iy = oy + dy, ix = ox + dx;
              if (iy >= 0 && iy < 3 && ix >= 0 && ix < 3) {
                acc[oy * 3 + ox][0] = __builtin_elementwise_fma(in[iy * 3 + ix][0], w0, acc[oy * 3 + ox][0]);
                acc[oy * 3 + ox][1] = __builtin_elementwise_fma(in[iy * 3 + ix][1], w1, acc[oy * 3 + ox][1]);
              }
            }
        }
      float mean[9], var[9];
#pragma unroll
      for (int pp = 0; pp < 9; ++pp) {
        const f32x2 t2 = acc[pp][0] + acc[pp][1];
        mean[pp] = wave_sum(t2[0] + t2[1]) * (1.0f / C);
      }
#pragma unroll
      for (int pp = 0; pp < 9; ++pp) {
        acc[pp][0] -= (f32x2)(mean[pp]);
        acc[pp][1] -= (f32x2)(mean[pp]);
        const f32x2 qq = __builtin_elementwise_fma(acc[pp][0], acc[pp][0], acc[pp][1] * acc[pp][1]);
        var[pp] = wave_sum(qq[0] + qq[1]) * (1.0f / C);
      }
      const float4 g4 = *reinterpret_cast<const float4*>(bk.ln_w + c4);
      const float4 bb4 = *reinterpret_cast<const float4*>(bk.ln_b + c4);
      const f32x2 g0 = {g4.x, g4.y}, g1 = {g4.z, g4.w}, bb0 = {bb4.x, bb4.y}, bb1 = {bb4.z, bb4.w};
#pragma unroll
      for (int pp = 0; pp < 9; ++pp) {
        const float rstd = rsqrtf(var[pp] + LN_EPS);
        const f32x2 o0 = __builtin_elementwise_fma(acc[pp][0] * (f32x2)(rstd), g0, bb0);
        const f32x2 o1 = __builtin_elementwise_fma(acc[pp][1] * (f32x2)(rstd), g1, bb1);
        T4 o;
        o[0] = (T)o0[0];
        o[1] = (T)o0[1];
        o[2] = (T)o1[0];
        o[3] = (T)o1[1];
        *reinterpret_cast<T4*>(img + img_off(wave * 9 + pp, c4)) = o;
      }
    }
    __syncthreads();   // LN image complete

    // ---- fc1 B operand of this lane's pixel: 8 k-steps of 32 channels
    frag xf[8];
    if (owner) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        xf[ks] = *reinterpret_cast<const frag*>(img + p * IROW + (((ks * 4 + q) ^ lp) << 4));
    }

    // ---- 32 chunks: fc1 (2 tiles of 16 hidden) -> GELU -> fc2 into x
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch, ++g) {
      // a loader's 8 pieces of chunk g have landed once only the two younger chunks' are outstanding
      // (the loaders issue nothing else, so their vmcnt counts LDS-DMA only; a deep DMA queue
      //  stalls the ISSUING wave, which is why the waves that compute do not issue)
      if (loader) {
        if (g + 2 < NG) wait_vm<16>();
        else if (g + 1 < NG) wait_vm<8>();
        else wait_vm<0>();
      }
      __syncthreads();   // chunk g has landed for everyone; chunk g-1 is read out by everyone
      if (loader && g + 3 < NG && !(a.diag & 2)) issue(g + 3);
      if (owner && !(a.diag & 1)) {
        const unsigned char* w1s = ring + (g % NSLOT) * CHUNKB;
        const unsigned char* w2s = w1s + W1CB;
        f32x4 hacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // accumulator row 4q + r of tile t = LDS row 16t + 4q + r = hidden 32 ch + 8q + 4t + r
          const float4 bv = *reinterpret_cast<const float4*>(b1s + ((ch + rot) & 31) * 32 + 8 * q + 4 * t);
          hacc[t] = f32x4{bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const int m = 16 * t + lp;
            const frag af = *reinterpret_cast<const frag*>(w1s + m * IROW + (((ks * 4 + q) ^ (m & 15)) << 4));
            hacc[t] = MM<T>::run(af, xf[ks], hacc[t]);
          }
        }
        frag hf;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) hf[4 * t + r] = (T)gelu_fast(hacc[t][r]);
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
          const int r = 16 * t + lp;
          const frag af = *reinterpret_cast<const frag*>(w2s + r * 64 + ((q ^ swz4(r)) << 4));
          x[t] = MM<T>::run(af, hf, x[t]);
        }
      }
    }
    __syncthreads();   // the last chunk's fc1 bias reads are done before the next block rewrites b1s
  }

  // ---- stage output
  if (live) {
    float* dst = a.out + ((size_t)a0 * 9 + p) * C + 4 * q;
#pragma unroll
    for (int t = 0; t < NT16; ++t)
      *reinterpret_cast<float4*>(dst + 16 * t) = make_float4(x[t][0], x[t][1], x[t][2], x[t][3]);
  }
}

template <typename T> int launch_stage2m_t(const Stage2Args& a, hipStream_t st) {
  auto kern = stage2m_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.B + GA - 1) / GA), dim3(512), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

// dst[(c * C + r) * 32 + k] = gamma[r] * w2[r][32 c + k]: the 32 hidden units a megakernel consumes
// per chunk, for all output channels, as one contiguous run (an LDS-DMA piece = 1 KiB of it)
template <typename T>
__global__ void pack_w2_chunks_kernel(const float* __restrict__ w2, const float* __restrict__ gamma,
                                      T* __restrict__ dst, int Cc, int H) {
  const long n = (long)Cc * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i & 31);
    const long rc = i >> 5;
    const int r = (int)(rc % Cc), c = (int)(rc / Cc);
    dst[i] = (T)(gamma[r] * w2[(size_t)r * H + 32 * c + k]);
  }
}

int launch_pack_w2_chunks(int prec, const float* w2, const float* gamma, void* dst, int Cc, int H,
                          hipStream_t st) {
  const long n = (long)Cc * H;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_w2_chunks_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w2, gamma,
                       reinterpret_cast<bf16_t*>(dst), Cc, H);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_w2_chunks_kernel<f16_t>, dim3(grid), dim3(256), 0, st, w2, gamma,
                       reinterpret_cast<f16_t*>(dst), Cc, H);
  else {
    btsbot_set_error("pack_w2_chunks: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int stage2m_max_depth() { return 8; }

// 16-bit modes, C = 256, 3x3 maps.  Every block needs w1 = plain [1024][256] and w2g = gamma-scaled
// fc2 filter in the operand type, chunk-major [32][256][32] (launch_pack_w2_chunks).
int launch_stage2m(int prec, const Stage2Args& a, hipStream_t st) {
  if (a.B <= 0 || a.depth <= 0) return BTSBOT_OK;
  if (a.depth > 8) {
    btsbot_set_error("stage2m: depth %d > 8", a.depth);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16) return launch_stage2m_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_stage2m_t<f16_t>(a, st);
  btsbot_set_error("stage2m: precision %d is not a 16-bit mode", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
