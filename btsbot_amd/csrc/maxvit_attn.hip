// MaxViT window / grid attention on MFMA for the 16-bit modes (gfx950).
//
// One wave owns one (alert, partition, head) unit at a time: 49 tokens, dim_head 32 -- timm AttentionCl
// inside PartitionAttentionCl, reached from /root/reference/btsbot/architectures.py:51,97.
//
//   S^T = K Q^T   16 x v_mfma_f32_16x16x32: the K rows are the A operand and the Q rows the B operand,
//                 both fetched as 16-byte pieces straight from the qkv rows in HBM/L2 (the partition is an
//                 index map on the row address; tokens 49..63 of the padded tile re-read token 48).
//                 The transposed product leaves a query's 64 logits on 4 lanes (16 each): the softmax
//                 reductions are 15 in-lane ops + 2 cross-lane steps.
//   P           = exp(S*scale + B - max): relative-position bias and the key padding mask (-1e30 for keys
//                 49..63) come from one padded [64 key][64 query] image per head, held in registers
//                 across the units a wave processes.
//   O^T = V^T P^T 16 x MFMA.  The accumulator of S^T is already the B operand (k slots = the lane's own
//                 keys, in the order jt*16 + 4g + r); V rows are staged once in a wave-private LDS image
//                 and read back with ds_read_b64_tr_b16 in that same key order; 1/sum is applied to O^T,
//                 whose columns (queries) sit on the lanes that own the sums.
// LDS: 64 rows x 96 B per wave (the 96-byte pitch makes the transposed reads conflict-free); rows 49..63
// are zeroed once so that padded keys contribute exact zeros.
#include "maxvit.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <typename T> struct AM;
template <> struct AM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct AM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int VPITCH = 96;

template <typename T>
__global__ __launch_bounds__(256) void mv_attn_mfma_kernel(const T* __restrict__ qkv,
                                                           const float* __restrict__ bias64,
                                                           T* __restrict__ out, int H, int C,
                                                           int grid_mode, int units, int upw) {
  using frag = typename AM<T>::frag;
  __shared__ __attribute__((aligned(16))) unsigned char vsm[4][64 * VPITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int head = blockIdx.y;
  const int G = H / 7, nW = G * G;
  unsigned char* vs = vsm[wave];
  for (int p = lane; p < (64 - 49) * VPITCH / 16; p += 64)
    *reinterpret_cast<uint4*>(vs + 49 * VPITCH + p * 16) = make_uint4(0, 0, 0, 0);

  float bias[4][4][4];   // [key tile][query tile][r]: key jt*16 + 4g + r, query it*16 + l15
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bias[jt][it][r] = bias64[((size_t)head * 64 + jt * 16 + 4 * g + r) * 64 + it * 16 + l15];

  const int u0 = (blockIdx.x * 4 + wave) * upw;
  for (int uu = 0; uu < upw; ++uu) {
    const int u = u0 + uu;
    if (u >= units) break;               // wave-uniform
    const int w = u % nW;
    const long b = u / nW;
    const int wy = w / G, wx = w % G;
    auto row_of = [&](int t) -> long {   // token of this partition -> row of the [B*H*H, .] maps
      const int ty = t / 7, tx = t - ty * 7;
      const int py = grid_mode ? ty * G + wy : wy * 7 + ty;
      const int px = grid_mode ? tx * G + wx : wx * 7 + tx;
      return (b * H + py) * H + px;
    };
    // ---- operand fetch: Q / K fragments from HBM, V rows into the LDS image
    frag qf[4], kf[4];
    long rows[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      rows[t] = row_of(min(t * 16 + l15, 48));
      const T* base = qkv + rows[t] * 3 * C + head * 96 + g * 8;
      qf[t] = *reinterpret_cast<const frag*>(base);
      kf[t] = *reinterpret_cast<const frag*>(base + 32);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = lane + 64 * k, r = p >> 2, part = p & 3;
      if (r < 49)
        *reinterpret_cast<uint4*>(vs + r * VPITCH + part * 16) =
            *reinterpret_cast<const uint4*>(qkv + row_of(r) * 3 * C + head * 96 + 64 + part * 8);
    }
    // ---- S^T = K Q^T
    f32x4 s[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int it = 0; it < 4; ++it)
        s[jt][it] = AM<T>::run(kf[jt], qf[it], f32x4{0.f, 0.f, 0.f, 0.f});
    // ---- softmax over the keys of each query (unnormalised; 1/sum goes onto O)
    float inv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float mx = -3.0e38f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[jt][it][r] = fmaf(s[jt][it][r], 0.17677669529663687f, bias[jt][it][r]);
          mx = fmaxf(mx, s[jt][it][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[jt][it][r] = __expf(s[jt][it][r] - mx);
          sum += s[jt][it][r];
        }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      inv[it] = 1.0f / sum;
    }
    // ---- O^T = V^T P^T
    frag vf[2][2];   // [d tile][k step]: V^T rows d = dt*16 + l15, keys {32ks + 4g + e, 32ks + 16 + 4g + e}
    {
      const int q = l15 >> 2, p = lane & 3;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* a = vs + (32 * ks + 4 * g + q) * VPITCH + (dt * 16 + 4 * p) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 16 * VPITCH));
          union { short h[8]; frag f; } cv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cv.h[e] = lo[e];
            cv.h[4 + e] = hi[e];
          }
          vf[dt][ks] = cv.f;
        }
    }
    f32x4 o[2][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      frag pf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pf[ks][e] = (T)s[2 * ks][it][e];
          pf[ks][4 + e] = (T)s[2 * ks + 1][it][e];
        }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 acc = AM<T>::run(vf[dt][0], pf[0], f32x4{0.f, 0.f, 0.f, 0.f});
        o[dt][it] = AM<T>::run(vf[dt][1], pf[1], acc);
      }
    }
    // ---- store: lane owns d = dt*16 + 4g .. +3 of query it*16 + l15
    typedef T __attribute__((ext_vector_type(4))) T4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it * 16 + l15 >= 49) continue;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        T4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (T)(o[dt][it][r] * inv[it]);
        *reinterpret_cast<T4*>(out + rows[it] * C + head * 32 + dt * 16 + 4 * g) = v;
      }
    }
  }
}

__global__ void mv_pack_relbias64_kernel(const float* table, float* out, int heads) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // [heads][64 key][64 query]
  if (i >= heads * 4096) return;
  const int qi = i & 63, kj = (i >> 6) & 63, hd = i >> 12;
  float v = 0.f;
  if (kj >= 49) {
    v = -1.0e30f;                                  // key padding mask
  } else if (qi < 49) {
    const int dy = qi / 7 - kj / 7, dx = qi % 7 - kj % 7;
    v = table[((dy + 6) * 13 + dx + 6) * heads + hd];
  }
  out[i] = v;
}

}  // namespace

int launch_mv_attn_mfma(int prec, const void* qkv, const float* bias64, void* out, int B, int H,
                        int C, int grid_mode, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (H % 7 != 0 || C % 32 != 0 || (prec != BTSBOT_BF16 && prec != BTSBOT_F16)) {
    btsbot_set_error("mv_attn_mfma: bad shape H=%d C=%d or precision %d", H, C, prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int units = B * (H / 7) * (H / 7), heads = C / 32;
  // enough workgroups to fill 256 CUs a few times over, as many units per wave as that leaves
  int upw = units * heads / (4 * 2048);
  upw = upw < 1 ? 1 : (upw > 8 ? 8 : upw);
  const dim3 grid((units + 4 * upw - 1) / (4 * upw), heads);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(mv_attn_mfma_kernel<bf16_t>, grid, dim3(256), 0, st,
                       reinterpret_cast<const bf16_t*>(qkv), bias64, reinterpret_cast<bf16_t*>(out), H,
                       C, grid_mode, units, upw);
  else
    hipLaunchKernelGGL(mv_attn_mfma_kernel<f16_t>, grid, dim3(256), 0, st,
                       reinterpret_cast<const f16_t*>(qkv), bias64, reinterpret_cast<f16_t*>(out), H, C,
                       grid_mode, units, upw);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_mv_pack_relbias64(const float* table, float* out, int heads, hipStream_t st) {
  hipLaunchKernelGGL(mv_pack_relbias64_kernel, dim3((heads * 4096 + 255) / 256), dim3(256), 0, st, table,
                     out, heads);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// ------------------------------------------------------------------------------------------------
// Stem conv 3x3 s1 p1 (32 -> 64) as an implicit GEMM without LDS (16-bit modes): every tap is one
// K = 32 step of v_mfma_f32_16x16x32.  The 64 x 288 filter lives in registers as 36 A fragments per wave
// (loaded once, reused over the wave's image rows); the B fragment of a tap is a 16-byte piece of each of
// 16 neighbouring input pixels, fetched straight from the NHWC map (L1/L2 serve the 9x re-reads), zero
// for taps that fall off the image.  Replaces mv_im2col3_kernel + GEMM, which moved a 7.2 MB/alert patch
// matrix through HBM twice.  in [B,112,112,32] T -> out [B,112,112,64] f32.
namespace {

template <typename T, bool POOL>
__global__ __launch_bounds__(256, 2) void mv_stem2_kernel(const T* __restrict__ in,
                                                       const T* __restrict__ w, float* __restrict__ out,
                                                       T* __restrict__ xn, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int pairs_total,
                                                       int pairs_per_wave) {
  // A wave walks image-row PAIRS (2y, 2y+1).  POOL: `out` receives the 2x2 average pool of the result
  // ([B,56,56,64] f32 -- the first MBConv block's shortcut is the only consumer of the fp32 map) instead of
  // the full map [B,112,112,64]: the horizontal neighbour comes from the adjacent lane (quad_perm [1,0,3,2]),
  // the vertical one from the pair's first row, summed in avg_pool2d's order ((a+b)+c)+d.
  using frag = typename AM<T>::frag;
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  frag wf[4][9];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wf[mt][tap] = *reinterpret_cast<const frag*>(w + (mt * 16 + l15) * 288 + tap * 32 + g * 8);
  typedef T __attribute__((ext_vector_type(4))) T4;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  // the pre-norm constants, staged once (a load from L2 inside the epilogue costs its whole latency per tile)
  __shared__ float4 ss[32];
  if (xn != nullptr && threadIdx.x < 32)
    ss[threadIdx.x] = threadIdx.x < 16 ? reinterpret_cast<const float4*>(scale)[threadIdx.x]
                                       : reinterpret_cast<const float4*>(shift)[threadIdx.x - 16];
  __syncthreads();
  for (int pp = 0; pp < pairs_per_wave; ++pp) {
    const int pair = wid * pairs_per_wave + pp;
    if (pair >= pairs_total) break;                    // wave-uniform
    const long b = pair / 56;
    const int yp = pair - (int)b * 56;
    for (int xt = 0; xt < 7; ++xt) {
      const int x = xt * 16 + l15;
      // the four input rows 2 yp - 1 .. 2 yp + 2 serve both output rows of the pair: all twelve B fragments are
      // requested in one batch (one wait per tile; rows and columns off the image are clamped on the address and
      // zeroed by a select -- a branch per row would serialise the requests, each paying the L2 round trip)
      frag xf[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int iy = 2 * yp - 1 + r;
        const bool rok = iy >= 0 && iy < 112;          // wave-uniform
        const T* rowp = in + ((b * 112 + (rok ? iy : 2 * yp)) * 112) * 32 + g * 8;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = x + kx - 1;
          const bool cok = ix >= 0 && ix < 112;
          i32x4 v = *reinterpret_cast<const i32x4*>(rowp + (long)(cok ? ix : x) * 32);
          if (!(rok && cok)) v = i32x4{0, 0, 0, 0};
          xf[r][kx] = __builtin_bit_cast(frag, v);
        }
      }
      float top[4][4];                                 // POOL: a + b of the pair's first row
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int y = 2 * yp + sub;
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = AM<T>::run(wf[mt][ky * 3 + kx], xf[sub + ky][kx], acc[mt]);
        const long po = ((b * 112 + y) * 112 + x) * 64 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (!POOL)
            *reinterpret_cast<float4*>(out + po + mt * 16) =
                make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
          if (xn != nullptr) {   // the first MBConv block's pre-norm BatchNorm + cast, while the values are here
            const float4 sc = ss[mt * 4 + g];
            const float4 sh = ss[16 + mt * 4 + g];
            T4 v;
            v[0] = (T)(acc[mt][0] * sc.x + sh.x);
            v[1] = (T)(acc[mt][1] * sc.y + sh.y);
            v[2] = (T)(acc[mt][2] * sc.z + sh.z);
            v[3] = (T)(acc[mt][3] * sc.w + sh.w);
            *reinterpret_cast<T4*>(xn + po + mt * 16) = v;
          }
        }
        if (POOL) {
          const long pq = ((b * 56 + yp) * 56 + (x >> 1)) * 64 + 4 * g;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            float res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float own = acc[mt][r];
              const float nb = __builtin_bit_cast(
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0xB1, 0xF, 0xF, true));
              if (sub == 0) top[mt][r] = own + nb;                           // a + b  (even lanes)
              else res[r] = ((top[mt][r] + own) + nb) * 0.25f;               // ((a+b)+c)+d
            }
            if (sub == 1 && (l15 & 1) == 0)
              *reinterpret_cast<float4*>(out + pq + mt * 16) = make_float4(res[0], res[1], res[2], res[3]);
          }
        }
      }
    }
  }
}

}  // namespace

int launch_mv_stem2(int prec, const void* in, const void* w, float* out, int pooled, void* xn,
                    const float* scale, const float* shift, int B, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  if (prec != BTSBOT_BF16 && prec != BTSBOT_F16) {
    btsbot_set_error("mv_stem2: 16-bit modes only (precision %d)", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const int pairs = B * 56, ppw = 2;
  const int waves = (pairs + ppw - 1) / ppw;
  const dim3 grid((waves + 3) / 4);
#define STEM2(TT, PP)                                                                                  \
  hipLaunchKernelGGL((mv_stem2_kernel<TT, PP>), grid, dim3(256), 0, st, reinterpret_cast<const TT*>(in), \
                     reinterpret_cast<const TT*>(w), out, reinterpret_cast<TT*>(xn), scale, shift, pairs, \
                     ppw)
  if (prec == BTSBOT_BF16) {
    if (pooled) STEM2(bf16_t, true); else STEM2(bf16_t, false);
  } else {
    if (pooled) STEM2(f16_t, true); else STEM2(f16_t, false);
  }
#undef STEM2
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
