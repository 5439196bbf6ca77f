// MaxViT image branch: the non-GEMM kernels (gfx950).
//
// What they stand in for: the ATen ops timm's maxvit_tiny_rw_224 dispatches to when
// /root/reference/btsbot/architectures.py:51,97 calls the backbone -- F.interpolate (:44-50,:90-96),
// conv2d 3x3 (stem), BatchNorm2d, depthwise conv2d 3x3, SiLU, squeeze-excite, avg_pool2d, LayerNorm,
// windowed / grid multi-head attention with a learned relative position bias, final LayerNorm2d and the
// global average pool.  All 1x1 convolutions and nn.Linear layers go through launch_gemm (gemm.hip /
// gemm2.hip).  Activations are NHWC pixel rows; T is the staged activation type of the precision mode.
//
// These are the first-correct versions: HBM-streaming kernels with 8..16-byte accesses per lane, one
// thread per output vector.  The attention kernel keeps one query row per lane and reads keys / values
// as LDS broadcasts.
#include "maxvit.h"

namespace {

template <typename T> struct V4 { typedef T __attribute__((ext_vector_type(4))) type; };

inline unsigned nblk(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mv_resize_im2col_kernel(const float* __restrict__ img,
                                                               T* __restrict__ out, long total) {
  const long idx = blockIdx.x * 256L + threadIdx.x;   // one output pixel of the 112x112 stem map
  if (idx >= total) return;
  const int b = (int)(idx / 12544), p = (int)(idx % 12544), oy = p / 112, ox = p % 112;
  const float* im = img + (size_t)b * 3 * 3969;
  const float scale = 63.0f / 224.0f;     // torch: input_size / output_size, src = scale*(dst+0.5)-0.5
  typedef typename V4<T>::type T4;
  T4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = T4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
  T* e = reinterpret_cast<T*>(v);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
    if (iy < 0 || iy >= 224) continue;
    const float sy = fmaxf(scale * ((float)iy + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, 62), y1 = y0 + (y0 < 62 ? 1 : 0);
    const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      if (ix < 0 || ix >= 224) continue;
      const float sx = fmaxf(scale * ((float)ix + 0.5f) - 0.5f, 0.f);
      const int x0 = min((int)sx, 62), x1 = x0 + (x0 < 62 ? 1 : 0);
      const float lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pc = im + c * 3969;
        const float top = (1.f - lx) * pc[y0 * 63 + x0] + lx * pc[y0 * 63 + x1];
        const float bot = (1.f - lx) * pc[y1 * 63 + x0] + lx * pc[y1 * 63 + x1];
        e[(ky * 3 + kx) * 3 + c] = (T)((1.f - ly) * top + ly * bot);
      }
    }
  }
  T4* o = reinterpret_cast<T4*>(out + (size_t)idx * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = v[i];
}

// 16-bit modes: bilinear resize + stem conv 3x3 s2 (3 -> 32) + folded BN + SiLU as ONE direct kernel: a
// thread samples its output pixel's 27 inputs once and runs the 27 x 32 FMAs against the filter in LDS
// (broadcast reads).  Replaces the im2col matrix (64 B per pixel written + read) and a K = 27 GEMM whose
// MFMA tiles were 16 % full.  w is the packed [32][32] image of mv_pack_stem1 (k = (ky*3+kx)*3 + c).
template <typename T>
__global__ __launch_bounds__(256) void mv_stem1_kernel(const float* __restrict__ img,
                                                       const T* __restrict__ w,
                                                       const float* __restrict__ shift,
                                                       T* __restrict__ out, long total) {
  __shared__ __attribute__((aligned(8))) float ws[27][32];
  __shared__ float sh[32];
  for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i / 32][i % 32] = (float)w[(i % 32) * 32 + i / 32];
  if (threadIdx.x < 32) sh[threadIdx.x] = shift[threadIdx.x];
  __syncthreads();
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int b = (int)(idx / 12544), p = (int)(idx % 12544), oy = p / 112, ox = p % 112;
  const float* im = img + (size_t)b * 3 * 3969;
  const float scale = 63.0f / 224.0f;
  // (the 27 x 32 products as packed fp32 FMAs -- v_pk_fma_f32, two outputs per instruction: this kernel is bound by them)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = f2{sh[2 * o], sh[2 * o + 1]};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy - 1 + ky;
    if (iy < 0 || iy >= 224) continue;
    const float sy = fmaxf(scale * ((float)iy + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)sy, 62), y1 = y0 + (y0 < 62 ? 1 : 0);
    const float ly = fminf(fmaxf(sy - (float)y0, 0.f), 1.f);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox - 1 + kx;
      if (ix < 0 || ix >= 224) continue;
      const float sx = fmaxf(scale * ((float)ix + 0.5f) - 0.5f, 0.f);
      const int x0 = min((int)sx, 62), x1 = x0 + (x0 < 62 ? 1 : 0);
      const float lx = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pc = im + c * 3969;
        const float top = (1.f - lx) * pc[y0 * 63 + x0] + lx * pc[y0 * 63 + x1];
        const float bot = (1.f - lx) * pc[y1 * 63 + x0] + lx * pc[y1 * 63 + x1];
        // the GEMM path rounds the sample to T before the product: keep that (same numbers either way)
        const float v = (float)(T)((1.f - ly) * top + ly * bot);
        const f2* wr = reinterpret_cast<const f2*>(ws[(ky * 3 + kx) * 3 + c]);
        const f2 v2 = f2{v, v};
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = v2 * wr[o] + acc[o];
      }
    }
  }
  typedef T __attribute__((ext_vector_type(8))) T8;
  T8* op = reinterpret_cast<T8*>(out + (size_t)idx * 32);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    T8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (T)silu_fast(acc[q * 4 + e / 2][e & 1]);
    op[q] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mv_im2col3_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                         long total, int HW, int C) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int cpc = C / EPC;
  const int chunk = (int)(idx % cpc);
  const int tap = (int)((idx / cpc) % 9);
  const long pix = idx / (9L * cpc);
  const int x = (int)(pix % HW), y = (int)((pix / HW) % HW);
  const long b = pix / ((long)HW * HW);
  const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (iy >= 0 && iy < HW && ix >= 0 && ix < HW)
    v = *reinterpret_cast<const uint4*>(in + ((b * HW + iy) * HW + ix) * C + chunk * EPC);
  *reinterpret_cast<uint4*>(out + pix * 9 * C + (long)tap * C + chunk * EPC) = v;
}

template <typename T>
__global__ __launch_bounds__(256) void mv_bn_cast_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         T* __restrict__ out, long n4, int C) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
  const float4 s = *reinterpret_cast<const float4*>(scale + c);
  const float4 t = *reinterpret_cast<const float4*>(shift + c);
  typedef typename V4<T>::type T4;
  T4 o;
  o[0] = (T)(v.x * s.x + t.x);
  o[1] = (T)(v.y * s.y + t.y);
  o[2] = (T)(v.z * s.z + t.z);
  o[3] = (T)(v.w * s.w + t.w);
  *reinterpret_cast<T4*>(out + i * 4) = o;
}

template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void mv_dw3_kernel(const T* __restrict__ in,
                                                     const float* __restrict__ w9,
                                                     const float* __restrict__ bias,
                                                     T* __restrict__ out, long total, int H, int C) {
  const long idx = blockIdx.x * 256L + threadIdx.x;     // (output pixel, 4-channel group)
  if (idx >= total) return;
  const int c4 = C / 4, Ho = H / STRIDE;
  const int c = (int)(idx % c4) * 4;
  const long pix = idx / c4;
  const int ox = (int)(pix % Ho), oy = (int)((pix / Ho) % Ho);
  const long b = pix / ((long)Ho * Ho);
  typedef typename V4<T>::type T4;
  float4 acc = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * STRIDE - 1 + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * STRIDE - 1 + kx;
      if (ix < 0 || ix >= H) continue;
      const T4 v = *reinterpret_cast<const T4*>(in + ((b * H + iy) * H + ix) * C + c);
      const float4 w = *reinterpret_cast<const float4*>(w9 + (ky * 3 + kx) * C + c);
      acc.x = fmaf((float)v[0], w.x, acc.x);
      acc.y = fmaf((float)v[1], w.y, acc.y);
      acc.z = fmaf((float)v[2], w.z, acc.z);
      acc.w = fmaf((float)v[3], w.w, acc.w);
    }
  }
  T4 o;
  o[0] = (T)silu_for<T>(acc.x);
  o[1] = (T)silu_for<T>(acc.y);
  o[2] = (T)silu_for<T>(acc.z);
  o[3] = (T)silu_for<T>(acc.w);
  *reinterpret_cast<T4*>(out + pix * C + c) = o;
}

// 16-bit modes: one thread = a strip of 7 output pixels of one row x 8 channels (16-byte accesses, every
// input column loaded once per row and used for up to three outputs), and the squeeze-excite pool comes for
// free: the thread sums its 7 outputs, the workgroup (PL strips x all channel groups of ONE alert) reduces
// them in LDS and writes one partial row -> part [B][gridDim.x][C] f32, summed by mv_se_kernel in a fixed
// order (no atomics: bit-reproducible).
template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void mv_dw3s_kernel(const T* __restrict__ in,
                                                      const float* __restrict__ w9,
                                                      const float* __restrict__ bias,
                                                      T* __restrict__ out, float* __restrict__ part,
                                                      int H, int C) {
  typedef T __attribute__((ext_vector_type(8))) T8;
  constexpr int NIN = 7 * STRIDE + (STRIDE == 1 ? 2 : 1);   // input columns a strip touches
  extern __shared__ float red[];                              // [PL][C]
  const int Ho = H / STRIDE, SPR = Ho / 7, S = Ho * SPR;      // strips per row / per alert
  const int c8n = C / 8;
  const int CT = c8n < 256 ? c8n : 256, PL = 256 / CT;
  const int cl = threadIdx.x % CT, pl = threadIdx.x / CT;
  const long b = blockIdx.y;
  const int strip = blockIdx.x * PL + pl;
  const int c = cl * 8;
  float psum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) psum[e] = 0.f;
  if (strip < S) {
    const int oy = strip / SPR, ox0 = (strip % SPR) * 7;
    float acc[7][8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c);
      const float4 b1 = *reinterpret_cast<const float4*>(bias + c + 4);
#pragma unroll
      for (int p = 0; p < 7; ++p) {
        acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
        acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
      }
    }
    const int ix0 = ox0 * STRIDE - 1;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * STRIDE - 1 + ky;
      const bool rok = iy >= 0 && iy < H;
      // a row's input columns are requested as one batch: the address is clamped into the image and the value zeroed
      // by a select (a load under a lane-varying branch is waited for on the spot -- one L2 round trip per column)
      const T* rowp = in + ((b * H + min(max(iy, 0), H - 1)) * H) * C + c;
      i32x4 vi[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j)
        vi[j] = *reinterpret_cast<const i32x4*>(rowp + (long)min(max(ix0 + j, 0), H - 1) * C);
      float w[3][8];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w0 = *reinterpret_cast<const float4*>(w9 + (ky * 3 + kx) * C + c);
        const float4 w1 = *reinterpret_cast<const float4*>(w9 + (ky * 3 + kx) * C + c + 4);
        w[kx][0] = w0.x; w[kx][1] = w0.y; w[kx][2] = w0.z; w[kx][3] = w0.w;
        w[kx][4] = w1.x; w[kx][5] = w1.y; w[kx][6] = w1.z; w[kx][7] = w1.w;
      }
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        asm volatile("" : "+v"(vi[j]));               // (keeps the load unconditional)
        const int ix = ix0 + j;
        if (!(rok && ix >= 0 && ix < H)) vi[j] = i32x4{0, 0, 0, 0};
        const T8 v = __builtin_bit_cast(T8, vi[j]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          // input column j feeds output p when j == p*STRIDE + kx
          if ((j - kx) >= 0 && (j - kx) % STRIDE == 0 && (j - kx) / STRIDE < 7) {
            const int p = (j - kx) / STRIDE;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = fmaf((float)v[e], w[kx][e], acc[p][e]);
          }
        }
      }
    }
    T* orow = out + ((b * Ho + oy) * Ho + ox0) * C + c;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
      T8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (T)silu_fast(acc[p][e]);
        psum[e] += (float)o[e];          // the pool sees the rounded activations, like the separate pass
      }
      *reinterpret_cast<T8*>(orow + (long)p * C) = o;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[pl * C + c + e] = psum[e];
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[q * C + i];
    part[(b * gridDim.x + blockIdx.x) * C + i] = a;
  }
}

// squeeze-excite gate as three small, fully parallel launches (the one-workgroup-per-alert version spent
// 100+ us per block in serial loops on 64..256 workgroups):
//   mv_se_pool_kernel  mean[b][c] = inv * sum_r y[b][r][c]        one thread per (alert, 4 channels); y = the
//                                                                 map itself or mv_dw3s_kernel's partial rows
//   mv_se_fc1_kernel   s[b][r]   = silu(W1[r] . mean[b] + b1[r])  workgroup = 8 alerts x 16 outputs, one wave
//                                                                 per 4 outputs: weight rows coalesced, read
//                                                                 once per 8 alerts
//   mv_se_fc2_kernel   gate[b][c] = sigmoid(W2t[:,c] . s[b] + b2[c])  workgroup = 8 alerts x 256 channels over
//                                                                 the TRANSPOSED fc2 matrix [RD][C]
constexpr int SE_AG = 8;

template <typename T>
__global__ __launch_bounds__(256) void mv_se_pool_kernel(const T* __restrict__ y,
                                                         float* __restrict__ mean, long total, int HW,
                                                         int C, float inv) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int c4n = C / 4;
  const int cg = (int)(idx % c4n);
  const long b = idx / c4n;
  typedef typename V4<T>::type T4;
  const T* p = y + (size_t)b * HW * C + cg * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < HW; ++r) {
    const T4 v = *reinterpret_cast<const T4*>(p + (size_t)r * C);
    acc.x += (float)v[0];
    acc.y += (float)v[1];
    acc.z += (float)v[2];
    acc.w += (float)v[3];
  }
  *reinterpret_cast<float4*>(mean + b * C + cg * 4) =
      make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

__global__ __launch_bounds__(256) void mv_se_fc1_kernel(const float* __restrict__ mean,
                                                        const float* __restrict__ w1,
                                                        const float* __restrict__ b1,
                                                        float* __restrict__ s, int B, int C, int RD) {
  const int b0 = blockIdx.x * SE_AG, r0 = blockIdx.y * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4][SE_AG];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int a = 0; a < SE_AG; ++a) acc[i][a] = 0.f;
  for (int c = lane; c < C; c += 64) {
    float mv[SE_AG];
#pragma unroll
    for (int a = 0; a < SE_AG; ++a) mv[a] = mean[(size_t)min(b0 + a, B - 1) * C + c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(r0 + wave * 4 + i, RD - 1);
      const float w = w1[(size_t)r * C + c];
#pragma unroll
      for (int a = 0; a < SE_AG; ++a) acc[i][a] = fmaf(w, mv[a], acc[i][a]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + wave * 4 + i;
#pragma unroll
    for (int a = 0; a < SE_AG; ++a) {
      const float t = wave_sum(acc[i][a]);
      if (lane == 0 && r < RD && b0 + a < B) s[(size_t)(b0 + a) * RD + r] = silu_f(t + b1[r]);
    }
  }
}

__global__ __launch_bounds__(256) void mv_se_fc2_kernel(const float* __restrict__ s,
                                                        const float* __restrict__ w2t,
                                                        const float* __restrict__ b2,
                                                        float* __restrict__ gate, int B, int C, int RD) {
  extern __shared__ float sv[];   // [SE_AG][RD]
  const int b0 = blockIdx.x * SE_AG, c = blockIdx.y * 256 + threadIdx.x;
  for (int i = threadIdx.x; i < SE_AG * RD; i += 256)
    sv[i] = s[(size_t)min(b0 + i / RD, B - 1) * RD + i % RD];
  __syncthreads();
  if (c >= C) return;
  float acc[SE_AG];
#pragma unroll
  for (int a = 0; a < SE_AG; ++a) acc[a] = b2[c];
  for (int r = 0; r < RD; ++r) {
    const float w = w2t[(size_t)r * C + c];
#pragma unroll
    for (int a = 0; a < SE_AG; ++a) acc[a] = fmaf(w, sv[a * RD + r], acc[a]);
  }
#pragma unroll
  for (int a = 0; a < SE_AG; ++a)
    if (b0 + a < B) gate[(size_t)(b0 + a) * C + c] = 1.0f / (1.0f + __expf(-acc[a]));
}

template <typename OT>
__global__ __launch_bounds__(256) void mv_avgpool2_kernel(const float* __restrict__ x,
                                                          OT* __restrict__ out, long total, int H,
                                                          int C) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= total) return;
  const int c4 = C / 4, Ho = H / 2;
  const int c = (int)(idx % c4) * 4;
  const long pix = idx / c4;
  const int ox = (int)(pix % Ho), oy = (int)((pix / Ho) % Ho);
  const long b = pix / ((long)Ho * Ho);
  const float* p = x + ((b * H + 2 * oy) * H + 2 * ox) * C + c;
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 bb = *reinterpret_cast<const float4*>(p + C);
  const float4 cc = *reinterpret_cast<const float4*>(p + (size_t)H * C);
  const float4 d = *reinterpret_cast<const float4*>(p + (size_t)H * C + C);
  typedef typename V4<OT>::type O4;
  O4 o;
  o[0] = (OT)((((a.x + bb.x) + cc.x) + d.x) * 0.25f);
  o[1] = (OT)((((a.y + bb.y) + cc.y) + d.y) * 0.25f);
  o[2] = (OT)((((a.z + bb.z) + cc.z) + d.z) * 0.25f);
  o[3] = (OT)((((a.w + bb.w) + cc.w) + d.w) * 0.25f);
  *reinterpret_cast<O4*>(out + pix * C + c) = o;
}

// 16 lanes per row (a DPP row: the two reductions are four v_add_dpp each), 16 rows per workgroup; lane l
// holds the float4 chunks l, l+16, ... of its row, so every load / store instruction of a row is one
// contiguous 256-byte (f32) piece
template <typename T, int NCH>
__global__ __launch_bounds__(256) void mv_ln_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bsh,
                                                    T* __restrict__ out, long M) {
  constexpr int C = NCH * 64;
  const long row_raw = blockIdx.x * 16L + (threadIdx.x >> 4);
  const long row = row_raw < M ? row_raw : M - 1;     // keep every lane of the DPP row alive
  const int l = threadIdx.x & 15;
  float4 v[NCH];
  const float* p = x + row * C;
#pragma unroll
  for (int k = 0; k < NCH; ++k) v[k] = *reinterpret_cast<const float4*>(p + (k * 16 + l) * 4);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mean = group16_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    v[k].x -= mean; v[k].y -= mean; v[k].z -= mean; v[k].w -= mean;
    q = fmaf(v[k].x, v[k].x, q);
    q = fmaf(v[k].y, v[k].y, q);
    q = fmaf(v[k].z, v[k].z, q);
    q = fmaf(v[k].w, v[k].w, q);
  }
  const float rstd = rsqrtf(group16_sum(q) * (1.0f / C) + 1e-6f);
  if (row_raw >= M) return;
  typedef typename V4<T>::type T4;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = (k * 16 + l) * 4;
    const float4 g = *reinterpret_cast<const float4*>(w + c);
    const float4 bb = *reinterpret_cast<const float4*>(bsh + c);
    T4 o4;
    o4[0] = (T)(v[k].x * rstd * g.x + bb.x);
    o4[1] = (T)(v[k].y * rstd * g.y + bb.y);
    o4[2] = (T)(v[k].z * rstd * g.z + bb.z);
    o4[3] = (T)(v[k].w * rstd * g.w + bb.w);
    *reinterpret_cast<T4*>(out + row * C + c) = o4;
  }
}

// one wave per (alert, partition, head); lane = query token (49 active)
template <typename T>
__global__ __launch_bounds__(64) void mv_attn_kernel(const T* __restrict__ qkv,
                                                     const float* __restrict__ bias_t,
                                                     T* __restrict__ out, int H, int C,
                                                     int grid_mode) {
  __shared__ float ks[49][36];
  __shared__ float vs[49][36];
  const int heads = C / 32, G = H / 7, nW = G * G;
  int id = blockIdx.x;
  const int head = id % heads;
  id /= heads;
  const int w = id % nW;
  const long b = id / nW;
  const int wy = w / G, wx = w % G;
  const int t = threadIdx.x;
  const bool active = t < 49;
  const int ty = t / 7, tx = t % 7;
  const int py = grid_mode ? ty * G + wy : wy * 7 + ty;
  const int px = grid_mode ? tx * G + wx : wx * 7 + tx;
  const long row = (b * H + py) * H + px;
  float q[32];
  if (active) {
    const T* base = qkv + row * 3 * C + head * 96;
#pragma unroll
    for (int d = 0; d < 32; ++d) q[d] = (float)base[d] * 0.17677669529663687f;   // 32^-0.5
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      *reinterpret_cast<float4*>(&ks[t][d]) = make_float4((float)base[32 + d], (float)base[33 + d],
                                                          (float)base[34 + d], (float)base[35 + d]);
      *reinterpret_cast<float4*>(&vs[t][d]) = make_float4((float)base[64 + d], (float)base[65 + d],
                                                          (float)base[66 + d], (float)base[67 + d]);
    }
  }
  __syncthreads();
  if (!active) return;
  const float* bt = bias_t + (size_t)head * 2401 + t;
  float s[49];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 49; ++j) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(&ks[j][d]);
      a = fmaf(q[d], k4.x, a);
      a = fmaf(q[d + 1], k4.y, a);
      a = fmaf(q[d + 2], k4.z, a);
      a = fmaf(q[d + 3], k4.w, a);
    }
    s[j] = a + bt[j * 49];
    mx = fmaxf(mx, s[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 49; ++j) {
    s[j] = __expf(s[j] - mx);
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  float o[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 49; ++j) {
    const float p = s[j] * inv;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
      const float4 v4 = *reinterpret_cast<const float4*>(&vs[j][d]);
      o[d] = fmaf(p, v4.x, o[d]);
      o[d + 1] = fmaf(p, v4.y, o[d + 1]);
      o[d + 2] = fmaf(p, v4.z, o[d + 2]);
      o[d + 3] = fmaf(p, v4.w, o[d + 3]);
    }
  }
  T* dst = out + row * C + head * 32;
  typedef typename V4<T>::type T4;
#pragma unroll
  for (int d = 0; d < 32; d += 4) {
    T4 v;
    v[0] = (T)o[d];
    v[1] = (T)o[d + 1];
    v[2] = (T)o[d + 2];
    v[3] = (T)o[d + 3];
    *reinterpret_cast<T4*>(dst + d) = v;
  }
}

// one workgroup per alert: LayerNorm every pixel, average the normalised rows
template <int EPL>
__global__ __launch_bounds__(256) void mv_final_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bsh,
                                                       float* __restrict__ feat, int P) {
  constexpr int C = EPL * 64;
  __shared__ float part[4][C];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc[i] = 0.f;
  for (int p = wave; p < P; p += 4) {
    const float* px = x + ((size_t)b * P + p) * C + lane * EPL;
    float v[EPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      v[i] = px[i];
      s += v[i];
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      v[i] -= mean;
      q = fmaf(v[i], v[i], q);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + 1e-6f);
#pragma unroll
    for (int i = 0; i < EPL; ++i) acc[i] += v[i] * rstd * w[lane * EPL + i] + bsh[lane * EPL + i];
  }
#pragma unroll
  for (int i = 0; i < EPL; ++i) part[wave][lane * EPL + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    feat[(size_t)b * C + c] = (part[0][c] + part[1][c] + part[2][c] + part[3][c]) * (1.0f / (float)P);
}

// y[b][p][c] *= gate[b][c] in place (the narrow MBConv stages: the map is small, and the projection can then
// run on the LDS-DMA GEMM instead of the register-staged gated one)
template <typename T>
__global__ __launch_bounds__(256) void mv_gate_kernel(T* __restrict__ y, const float* __restrict__ gate,
                                                      long n8, int HW, int C) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n8) return;
  const int c8n = C / 8;
  const int c = (int)(i % c8n) * 8;
  const long b = i / ((long)c8n * HW);
  typedef T __attribute__((ext_vector_type(8))) T8;
  T8 v = *reinterpret_cast<T8*>(y + i * 8);
  const float4 g0 = *reinterpret_cast<const float4*>(gate + b * C + c);
  const float4 g1 = *reinterpret_cast<const float4*>(gate + b * C + c + 4);
  v[0] = (T)((float)v[0] * g0.x); v[1] = (T)((float)v[1] * g0.y);
  v[2] = (T)((float)v[2] * g0.z); v[3] = (T)((float)v[3] * g0.w);
  v[4] = (T)((float)v[4] * g1.x); v[5] = (T)((float)v[5] * g1.y);
  v[6] = (T)((float)v[6] * g1.z); v[7] = (T)((float)v[7] * g1.w);
  *reinterpret_cast<T8*>(y + i * 8) = v;
}

// wg[b][n][k] = w[n][k] * gate[b][k]: per-alert projection filters of the wide MBConv stages, where the
// filter (32..128 KB) is far smaller than the map it multiplies
template <typename T>
__global__ __launch_bounds__(256) void mv_scale_w_kernel(const float* __restrict__ w,
                                                         const float* __restrict__ gate,
                                                         T* __restrict__ wg, long total, int N, int K) {
  const long i = blockIdx.x * 256L + threadIdx.x;    // 4 k per thread
  if (i >= total) return;
  const int k4n = K / 4;
  const int k = (int)(i % k4n) * 4;
  const int n = (int)((i / k4n) % N);
  const long b = i / ((long)k4n * N);
  const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)n * K + k);
  const float4 gv = *reinterpret_cast<const float4*>(gate + b * K + k);
  typedef typename V4<T>::type T4;
  T4 o;
  o[0] = (T)(wv.x * gv.x); o[1] = (T)(wv.y * gv.y); o[2] = (T)(wv.z * gv.z); o[3] = (T)(wv.w * gv.w);
  *reinterpret_cast<T4*>(wg + (b * N + n) * K + k) = o;
}

// ---- packing ------------------------------------------------------------------------------------
template <typename T>
__global__ void mv_pack_stem1_kernel(const float* w, const float* scale, T* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // [32][32]
  if (i >= 1024) return;
  const int o = i >> 5, k = i & 31;
  float v = 0.f;
  if (k < 27) {
    const int tap = k / 3, c = k % 3;
    v = w[(o * 3 + c) * 9 + tap] * scale[o];
  }
  out[i] = (T)v;
}

template <typename T>
__global__ void mv_pack_conv3_kernel(const float* w, T* out, int O, int C) {
  const long i = blockIdx.x * 256L + threadIdx.x;   // [O][9][C]
  if (i >= (long)O * 9 * C) return;
  const int c = (int)(i % C), tap = (int)((i / C) % 9), o = (int)(i / (9L * C));
  out[i] = (T)w[((size_t)o * C + c) * 9 + tap];
}

__global__ void mv_fold_bias_kernel(const float* b, const float* s, const float* t, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (b ? b[i] : 0.f) * s[i] + t[i];
}

__global__ void mv_pack_dw_kernel(const float* w, const float* scale, float* out, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // [9][C]
  if (i >= 9 * C) return;
  const int c = i % C, tap = i / C;
  out[i] = w[c * 9 + tap] * scale[c];
}

__global__ void mv_pack_relbias_kernel(const float* table, float* out, int heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // [heads][key j][query i]
  if (i >= heads * 2401) return;
  const int qi = i % 49, kj = (i / 49) % 49, hd = i / 2401;
  const int dy = qi / 7 - kj / 7, dx = qi % 7 - kj % 7;
  out[i] = table[((dy + 6) * 13 + dx + 6) * heads + hd];
}

__global__ void mv_fill_kernel(float* out, float v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

#define MV_DISPATCH(prec, CALL)                                     \
  switch (prec) {                                                   \
    case BTSBOT_F32: { typedef float T; CALL; } break;              \
    case BTSBOT_BF16: { typedef bf16_t T; CALL; } break;            \
    case BTSBOT_F16: { typedef f16_t T; CALL; } break;              \
    default:                                                        \
      btsbot_set_error("maxvit op: bad precision %d", prec);        \
      return BTSBOT_ERR_INVALID_ARG;                                \
  }

}  // namespace

int launch_mv_resize_im2col(int prec, const float* img, void* out, int B, hipStream_t st) {
  const long total = (long)B * 12544;
  if (total <= 0) return BTSBOT_OK;
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_resize_im2col_kernel<T>, dim3(nblk(total)), dim3(256), 0, st,
                                       img, reinterpret_cast<T*>(out), total));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_stem1(int prec, const float* img, const void* w, const float* shift, void* out, int B,
                    hipStream_t st) {
  const long total = (long)B * 12544;
  if (total <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(mv_stem1_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, st, img,
                       reinterpret_cast<const bf16_t*>(w), shift, reinterpret_cast<bf16_t*>(out), total);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(mv_stem1_kernel<f16_t>, dim3(nblk(total)), dim3(256), 0, st, img,
                       reinterpret_cast<const f16_t*>(w), shift, reinterpret_cast<f16_t*>(out), total);
  else {
    btsbot_set_error("mv_stem1: 16-bit modes only");
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_im2col3(int prec, const void* in, void* out, int B, int HW, int C, hipStream_t st) {
  const int epc = prec == BTSBOT_F32 ? 4 : 8;
  if (C % epc != 0) {
    btsbot_set_error("mv_im2col3: C=%d must be a multiple of %d", C, epc);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long total = (long)B * HW * HW * 9 * (C / epc);
  if (total <= 0) return BTSBOT_OK;
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_im2col3_kernel<T>, dim3(nblk(total)), dim3(256), 0, st,
                                       reinterpret_cast<const T*>(in), reinterpret_cast<T*>(out), total,
                                       HW, C));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_bn_cast(int prec, const float* x, const float* scale, const float* shift, void* out,
                      long M, int C, hipStream_t st) {
  if (C % 4 != 0) {
    btsbot_set_error("mv_bn_cast: C=%d must be a multiple of 4", C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long n4 = M * C / 4;
  if (n4 <= 0) return BTSBOT_OK;
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_bn_cast_kernel<T>, dim3(nblk(n4)), dim3(256), 0, st, x, scale,
                                       shift, reinterpret_cast<T*>(out), n4, C));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_dw3(int prec, const void* in, const float* w9, const float* bias, void* out, int B,
                  int H, int C, int stride, hipStream_t st) {
  if (C % 4 != 0 || (stride != 1 && stride != 2) || H % stride != 0) {
    btsbot_set_error("mv_dw3: bad shape H=%d C=%d stride=%d", H, C, stride);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int Ho = H / stride;
  const long total = (long)B * Ho * Ho * (C / 4);
  if (total <= 0) return BTSBOT_OK;
  if (stride == 1) {
    MV_DISPATCH(prec, hipLaunchKernelGGL((mv_dw3_kernel<T, 1>), dim3(nblk(total)), dim3(256), 0, st,
                                         reinterpret_cast<const T*>(in), w9, bias,
                                         reinterpret_cast<T*>(out), total, H, C));
  } else {
    MV_DISPATCH(prec, hipLaunchKernelGGL((mv_dw3_kernel<T, 2>), dim3(nblk(total)), dim3(256), 0, st,
                                         reinterpret_cast<const T*>(in), w9, bias,
                                         reinterpret_cast<T*>(out), total, H, C));
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int mv_dw3s_groups(int H, int C, int stride) {
  const int Ho = H / stride, S = Ho * (Ho / 7), c8n = C / 8;
  const int PL = 256 / (c8n < 256 ? c8n : 256);
  return (S + PL - 1) / PL;
}

int launch_mv_dw3s(int prec, const void* in, const float* w9, const float* bias, void* out, float* part,
                   int B, int H, int C, int stride, hipStream_t st) {
  const int c8n = C / 8;
  if (prec == BTSBOT_F32 || C % 8 != 0 || (c8n < 256 && 256 % c8n != 0) || c8n > 256 ||
      (stride != 1 && stride != 2) || (H / stride) % 7 != 0) {
    btsbot_set_error("mv_dw3s: bad shape H=%d C=%d stride=%d prec=%d", H, C, stride, prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (B <= 0) return BTSBOT_OK;
  const int PL = 256 / (c8n < 256 ? c8n : 256);
  const dim3 grid(mv_dw3s_groups(H, C, stride), B);
  const size_t lds = (size_t)PL * C * sizeof(float);
#define MV_DWS(TT, SS)                                                                              \
  hipLaunchKernelGGL((mv_dw3s_kernel<TT, SS>), grid, dim3(256), lds, st,                            \
                     reinterpret_cast<const TT*>(in), w9, bias, reinterpret_cast<TT*>(out), part, H, C)
  if (prec == BTSBOT_BF16) {
    if (stride == 1) MV_DWS(bf16_t, 1); else MV_DWS(bf16_t, 2);
  } else {
    if (stride == 1) MV_DWS(f16_t, 1); else MV_DWS(f16_t, 2);
  }
#undef MV_DWS
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_se(int prec, const void* y, const float* w1, const float* b1, const float* w2t,
                 const float* b2, float* gate, float* scratch, int B, int HW, int C, int RD,
                 float inv_count, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (C % 4 != 0 || RD < 1 || RD > 512) {
    btsbot_set_error("mv_se: bad shape C=%d RD=%d", C, RD);
    return BTSBOT_ERR_INVALID_ARG;
  }
  float* mean = scratch;                  // [B][C]
  float* s = scratch + (size_t)B * C;     // [B][RD]
  const long total = (long)B * (C / 4);
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_se_pool_kernel<T>, dim3(nblk(total)), dim3(256), 0, st,
                                       reinterpret_cast<const T*>(y), mean, total, HW, C, inv_count));
  hipLaunchKernelGGL(mv_se_fc1_kernel, dim3((B + SE_AG - 1) / SE_AG, (RD + 15) / 16), dim3(256), 0, st,
                     mean, w1, b1, s, B, C, RD);
  hipLaunchKernelGGL(mv_se_fc2_kernel, dim3((B + SE_AG - 1) / SE_AG, (C + 255) / 256), dim3(256),
                     (size_t)SE_AG * RD * sizeof(float), st, s, w2t, b2, gate, B, C, RD);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_gate(int prec, void* y, const float* gate, int B, int HW, int C, hipStream_t st) {
  if (prec == BTSBOT_F32 || C % 8 != 0) {
    btsbot_set_error("mv_gate: 16-bit modes, C %% 8 == 0 (prec %d, C %d)", prec, C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long n8 = (long)B * HW * C / 8;
  if (n8 <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(mv_gate_kernel<bf16_t>, dim3(nblk(n8)), dim3(256), 0, st,
                       reinterpret_cast<bf16_t*>(y), gate, n8, HW, C);
  else
    hipLaunchKernelGGL(mv_gate_kernel<f16_t>, dim3(nblk(n8)), dim3(256), 0, st,
                       reinterpret_cast<f16_t*>(y), gate, n8, HW, C);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_scale_w(int prec, const float* w, const float* gate, void* wg, int B, int N, int K,
                      hipStream_t st) {
  const long total = (long)B * N * (K / 4);
  if (total <= 0) return BTSBOT_OK;
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_scale_w_kernel<T>, dim3(nblk(total)), dim3(256), 0, st, w, gate,
                                       reinterpret_cast<T*>(wg), total, N, K));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_avgpool2(int prec, const float* x, void* out, int to_t, int B, int H, int C,
                       hipStream_t st) {
  if (C % 4 != 0 || H % 2 != 0) {
    btsbot_set_error("mv_avgpool2: bad shape H=%d C=%d", H, C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long total = (long)B * (H / 2) * (H / 2) * (C / 4);
  if (total <= 0) return BTSBOT_OK;
  if (!to_t || prec == BTSBOT_F32) {
    hipLaunchKernelGGL(mv_avgpool2_kernel<float>, dim3(nblk(total)), dim3(256), 0, st, x,
                       reinterpret_cast<float*>(out), total, H, C);
  } else {
    MV_DISPATCH(prec, hipLaunchKernelGGL(mv_avgpool2_kernel<T>, dim3(nblk(total)), dim3(256), 0, st, x,
                                         reinterpret_cast<T*>(out), total, H, C));
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_ln(int prec, const float* x, const float* w, const float* b, void* out, long M, int C,
                 hipStream_t st) {
  if (M <= 0) return BTSBOT_OK;
  const dim3 grid(nblk(M, 16));
#define MV_LN_CASE(EPL)                                                                              \
  MV_DISPATCH(prec, hipLaunchKernelGGL((mv_ln_kernel<T, EPL>), grid, dim3(256), 0, st, x, w, b,      \
                                       reinterpret_cast<T*>(out), M))
  switch (C) {
    case 64: MV_LN_CASE(1); break;
    case 128: MV_LN_CASE(2); break;
    case 256: MV_LN_CASE(4); break;
    case 512: MV_LN_CASE(8); break;
    default:
      btsbot_set_error("mv_ln: C=%d not in {64,128,256,512}", C);
      return BTSBOT_ERR_INVALID_ARG;
  }
#undef MV_LN_CASE
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_attn(int prec, const void* qkv, const float* bias_t, void* out, int B, int H, int C,
                   int grid_mode, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (H % 7 != 0 || C % 32 != 0) {
    btsbot_set_error("mv_attn: bad shape H=%d C=%d", H, C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long blocks = (long)B * (H / 7) * (H / 7) * (C / 32);
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_attn_kernel<T>, dim3((unsigned)blocks), dim3(64), 0, st,
                                       reinterpret_cast<const T*>(qkv), bias_t,
                                       reinterpret_cast<T*>(out), H, C, grid_mode));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_final(const float* x, const float* w, const float* b, float* feat, int B, int P, int C,
                    hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (C != 512) {
    btsbot_set_error("mv_final: C=%d (only 512 is built)", C);
    return BTSBOT_ERR_INVALID_ARG;
  }
  hipLaunchKernelGGL(mv_final_kernel<8>, dim3(B), dim3(256), 0, st, x, w, b, feat, P);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_pack_stem1(int prec, const float* w, const float* scale, void* out, hipStream_t st) {
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_pack_stem1_kernel<T>, dim3(4), dim3(256), 0, st, w, scale,
                                       reinterpret_cast<T*>(out)));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_pack_conv3(int prec, const float* w, void* out, int O, int C, hipStream_t st) {
  const long n = (long)O * 9 * C;
  MV_DISPATCH(prec, hipLaunchKernelGGL(mv_pack_conv3_kernel<T>, dim3(nblk(n)), dim3(256), 0, st, w,
                                       reinterpret_cast<T*>(out), O, C));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_fold_bias(const float* b, const float* s, const float* t, float* out, int n,
                        hipStream_t st) {
  hipLaunchKernelGGL(mv_fold_bias_kernel, dim3(nblk(n)), dim3(256), 0, st, b, s, t, out, n);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_pack_dw(const float* w, const float* scale, float* out, int C, hipStream_t st) {
  hipLaunchKernelGGL(mv_pack_dw_kernel, dim3(nblk(9L * C)), dim3(256), 0, st, w, scale, out, C);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_pack_relbias(const float* table, float* out, int heads, hipStream_t st) {
  hipLaunchKernelGGL(mv_pack_relbias_kernel, dim3(nblk(heads * 2401L)), dim3(256), 0, st, table, out,
                     heads);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_fill(float* out, float v, int n, hipStream_t st) {
  hipLaunchKernelGGL(mv_fill_kernel, dim3(nblk(n)), dim3(256), 0, st, out, v, n);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
