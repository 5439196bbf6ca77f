// MaxViT partition attention at C = 64 as ONE kernel (gfx950, 16-bit modes):
//
//   x += proj( MHSA_7x7( qkv( xn ) ) ),   xn2 = LayerNorm(x)            xn = LN1(x) comes in, xn2 = LN2(x) goes out
//
// timm PartitionAttentionCl's first half (norm1 is already applied by the producer's epilogue; attn.qkv,
// rel-pos attention, attn.proj, the residual add and norm2), reached from
// /root/reference/btsbot/architectures.py:51,97.  Unfused, this is three launches that move 28 bytes per
// row-channel through HBM (qkv written and re-read is 12 of them); fused, a workgroup owns one 49-token
// partition at a time and moves 12: it reads the partition's xn rows and x rows and writes x and xn2.
//
//   phase A  qkv^T = Wqkv . xn^T + b      wave w -> token tile w (16 tokens x 192 channels, 24 MFMAs); filter
//                                         fragments from an LDS-resident image, token rows straight from HBM
//                                         (the partition is an index map on the row address); result -> LDS
//                                         image [64 tokens][192] (416-byte pitch: conflict-free tr reads)
//   phase B  attention                    wave w -> head w/2, query tiles 2(w&1), 2(w&1)+1: S^T = K Q^T (8 MFMAs),
//                                         softmax in registers (bias + key mask from a padded [64][64] image held
//                                         in registers), O^T = V^T P^T (8 MFMAs) with the S^T accumulator as B
//                                         operand and V^T by ds_read_b64_tr_b16; O overwrites the wave's own Q rows
//   phase C  x += Wproj . O + b; LN2      wave w -> token tile w (8 MFMAs); a token's 64 outputs sit on 4 lanes:
//                                         LayerNorm = 16 in-lane values + 2 cross-lane steps
// Workgroups are persistent over partitions (filters stay in LDS).  Tokens 49..63 of the padded tile repeat
// token 48 and are masked as keys / never stored as queries.
#include "maxvit.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <typename T> struct BM;
template <> struct BM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct BM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 64, C3 = 192;
constexpr int IPITCH = 416;                 // bytes per token row of the qkv image (192 x 2 + 32)
constexpr int WPITCH = C * 2 + 16;          // bytes per filter row in LDS (144)

template <typename T>
__global__ __launch_bounds__(256, 2) void mv_attn_block64_kernel(
    const T* __restrict__ xn, float* x, T* xn2, const T* __restrict__ wqkv,
    const float* __restrict__ bqkv, const T* __restrict__ wproj, const float* __restrict__ bproj,
    const float* __restrict__ bias64, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    int H, int grid_mode, int units) {
  using frag = typename BM<T>::frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img = smem;                              // [64][IPITCH]
  unsigned char* wq = smem + 64 * IPITCH;                 // [192][WPITCH]
  unsigned char* wp = wq + C3 * WPITCH;                   // [64][WPITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int G = H / 7, nW = G * G;

  // ---- filters -> LDS (once per workgroup)
  for (int i = tid; i < (C3 + C) * 8; i += 256) {
    const int row = i >> 3, ck = i & 7;
    const T* src = row < C3 ? wqkv + (size_t)row * C + ck * 8 : wproj + (size_t)(row - C3) * C + ck * 8;
    unsigned char* dst = row < C3 ? wq + row * WPITCH + ck * 16 : wp + (row - C3) * WPITCH + ck * 16;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
  }
  // ---- phase B constants of this wave: head, query tiles, bias / key mask
  const int head = wave >> 1, it0 = 2 * (wave & 1);
  float bias[4][2][4];   // [key tile][query tile][r]
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bias[jt][q][r] =
            bias64[((size_t)head * 64 + jt * 16 + 4 * g + r) * 64 + (it0 + q) * 16 + l15];
  // phase A / C constants: biases of this lane's channels
  __syncthreads();

  typedef T __attribute__((ext_vector_type(4))) T4;
  // row of this lane's phase-A / phase-C token (16*wave + l15, padded tokens repeat token 48) in unit u
  const int tokA = wave * 16 + l15;
  auto rowA_of = [&](int u) -> long {
    const int w = u % nW;
    const long b = u / nW;
    const int wy = w / G, wx = w % G;
    const int t = min(tokA, 48);
    const int ty = t / 7, tx = t - ty * 7;
    const int py = grid_mode ? ty * G + wy : wy * 7 + ty;
    const int px = grid_mode ? tx * G + wx : wx * 7 + tx;
    return (b * H + py) * H + px;
  };
  // the NEXT partition's token rows are requested during the current one's attention phase, this
  // partition's residual rows at its start (PMC: the waves were parked on s_waitcnt / barriers 62 % of the time)
  frag xf[2];
  if ((int)blockIdx.x < units) {
    const long r0 = rowA_of(blockIdx.x);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xf[ks] = *reinterpret_cast<const frag*>(xn + r0 * C + ks * 32 + g * 8);
  }
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const long rowA = rowA_of(u);
    float4 rres[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) rres[ct] = *reinterpret_cast<const float4*>(x + rowA * C + ct * 16 + 4 * g);
    // ---- phase A: qkv^T tile of this wave's 16 tokens
    {
#pragma unroll
      for (int ct = 0; ct < 12; ++ct) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const frag a = *reinterpret_cast<const frag*>(wq + (ct * 16 + l15) * WPITCH + ks * 64 + g * 16);
          acc = BM<T>::run(a, xf[ks], acc);
        }
        const float4 bv = *reinterpret_cast<const float4*>(bqkv + ct * 16 + 4 * g);
        T4 v;
        v[0] = (T)(acc[0] + bv.x);
        v[1] = (T)(acc[1] + bv.y);
        v[2] = (T)(acc[2] + bv.z);
        v[3] = (T)(acc[3] + bv.w);
        *reinterpret_cast<T4*>(img + tokA * IPITCH + (ct * 16 + 4 * g) * 2) = v;
      }
    }
    __syncthreads();
    {
      const int un = u + (int)gridDim.x < units ? u + (int)gridDim.x : u;
      const long rn = rowA_of(un);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xf[ks] = *reinterpret_cast<const frag*>(xn + rn * C + ks * 32 + g * 8);
    }
    // ---- phase B: attention of (head, two query tiles)
    {
      frag kf[4], qf[2];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
        kf[jt] = *reinterpret_cast<const frag*>(img + (jt * 16 + l15) * IPITCH + (head * 96 + 32 + g * 8) * 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        qf[q] = *reinterpret_cast<const frag*>(img + ((it0 + q) * 16 + l15) * IPITCH + (head * 96 + g * 8) * 2);
      f32x4 s[4][2];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 2; ++q) s[jt][q] = BM<T>::run(kf[jt], qf[q], f32x4{0.f, 0.f, 0.f, 0.f});
      float inv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float mx = -3.0e38f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[jt][q][r] = fmaf(s[jt][q][r], 0.17677669529663687f, bias[jt][q][r]);
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
      frag vf[2][2];   // [d tile][k step]
      {
        const int qq = l15 >> 2, p = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* a =
                img + (32 * ks + 4 * g + qq) * IPITCH + (head * 96 + 64 + dt * 16 + 4 * p) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 16 * IPITCH));
            union { short h[8]; frag f; } cv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cv.h[e] = lo[e];
              cv.h[4 + e] = hi[e];
            }
            vf[dt][ks] = cv.f;
          }
      }
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
          f32x4 acc = BM<T>::run(vf[dt][0], pf[0], f32x4{0.f, 0.f, 0.f, 0.f});
          acc = BM<T>::run(vf[dt][1], pf[1], acc);
          // O[token (it0+q)*16 + l15][head*32 + dt*16 + 4g .. +3] -> this head's (dead) Q columns
          T4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (T)(acc[r] * inv[q]);
          *reinterpret_cast<T4*>(img + ((it0 + q) * 16 + l15) * IPITCH + (head * 96 + dt * 16 + 4 * g) * 2) = v;
        }
      }
    }
    __syncthreads();
    // ---- phase C: x += Wproj O + b, LN2
    {
      frag of[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)   // k step = head: its 32 output columns live at head*96 .. +31
        of[ks] = *reinterpret_cast<const frag*>(img + tokA * IPITCH + (ks * 96 + g * 8) * 2);
      float v[4][4];
      float sum = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const frag a = *reinterpret_cast<const frag*>(wp + (ct * 16 + l15) * WPITCH + ks * 64 + g * 16);
          acc = BM<T>::run(a, of[ks], acc);
        }
        const int c = ct * 16 + 4 * g;
        const float4 bv = *reinterpret_cast<const float4*>(bproj + c);
        const float4 rv = rres[ct];
        v[ct][0] = rv.x + (acc[0] + bv.x);
        v[ct][1] = rv.y + (acc[1] + bv.y);
        v[ct][2] = rv.z + (acc[2] + bv.z);
        v[ct][3] = rv.w + (acc[3] + bv.w);
        sum += (v[ct][0] + v[ct][1]) + (v[ct][2] + v[ct][3]);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.0f / C);
      float qv = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[ct][r] - mean;
          qv = fmaf(d, d, qv);
        }
      qv += __shfl_xor(qv, 16);
      qv += __shfl_xor(qv, 32);
      const float rstd = rsqrtf(qv * (1.0f / C) + 1e-6f);
      if (tokA < 49) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int c = ct * 16 + 4 * g;
          *reinterpret_cast<float4*>(x + rowA * C + c) = make_float4(v[ct][0], v[ct][1], v[ct][2], v[ct][3]);
          const float4 gw = *reinterpret_cast<const float4*>(ln_w + c);
          const float4 gb = *reinterpret_cast<const float4*>(ln_b + c);
          T4 y;
          y[0] = (T)((v[ct][0] - mean) * rstd * gw.x + gb.x);
          y[1] = (T)((v[ct][1] - mean) * rstd * gw.y + gb.y);
          y[2] = (T)((v[ct][2] - mean) * rstd * gw.z + gb.z);
          y[3] = (T)((v[ct][3] - mean) * rstd * gw.w + gb.w);
          *reinterpret_cast<T4*>(xn2 + rowA * C + c) = y;
        }
      }
    }
    __syncthreads();   // the next partition's phase A rewrites the image
  }
}

}  // namespace

bool mv_attn_block_supported(int prec, int C_) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && C_ == 64;
}

// xn [B*H*H, 64] T (LN1 output) is read, x [B*H*H, 64] f32 updated in place, xn2 [B*H*H, 64] T = LN2(x) written
// (xn2 may alias xn: a partition's rows are read completely before any of them is written).
int launch_mv_attn_block(int prec, const void* xn, float* x, void* xn2, const void* wqkv,
                         const float* bqkv, const void* wproj, const float* bproj, const float* bias64,
                         const float* ln_w, const float* ln_b, int B, int H, int C_, int grid_mode,
                         hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (!mv_attn_block_supported(prec, C_) || H % 7 != 0) {
    btsbot_set_error("mv_attn_block: unsupported (prec %d, C %d, H %d)", prec, C_, H);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int units = B * (H / 7) * (H / 7);
  const size_t lds = 64 * IPITCH + (size_t)(C3 + C) * WPITCH;
  const int grid = units < 2048 ? units : 2048;
#define ABLK(TT)                                                                                        \
  do {                                                                                                  \
    auto kern = mv_attn_block64_kernel<TT>;                                                             \
    static DevOnce attr;                                                                           \
    if (attr.need()) {                                                                                        \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
      attr.done();                                                                                      \
    }                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, reinterpret_cast<const TT*>(xn), x,        \
                       reinterpret_cast<TT*>(xn2), reinterpret_cast<const TT*>(wqkv), bqkv,             \
                       reinterpret_cast<const TT*>(wproj), bproj, bias64, ln_w, ln_b, H, grid_mode,     \
                       units);                                                                          \
  } while (0)
  if (prec == BTSBOT_BF16) ABLK(bf16_t); else ABLK(f16_t);
#undef ABLK
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
