// Fused ConvNeXt MLP for the wide-map stages (gfx950):
//
//     x[m][:] += gamma * ( W2 . gelu( W1 . xn[m][:] + b1 ) + b2 )
//
// i.e. timm ConvNeXtBlock's mlp.fc1 -> GELU -> mlp.fc2 -> layer-scale -> residual
// (/root/reference/btsbot/architectures.py:108,132 -> timm), with the 4C-wide hidden activation
// never leaving the CU.  Unfused, fc1/fc2 of stage 0 move 2 x 118 MB of hidden activations per
// 1024 alerts through HBM and are bandwidth-bound; fused, the kernel reads xn (16-bit) + x (f32)
// and writes x.
//
// Formulation ("transposed", register-chained):  per wave a 32-pixel column block.
//   GEMM1  Ht[32 hid x 32 px] = W1c[32 hid x C] . Xt[C x 32 px]      v_mfma_f32_32x32x16
//          A = filter rows (LDS), B = the wave's xn rows, resident in registers for the whole tile.
//   GELU   on the 16 accumulator registers (lane = pixel, register = hidden unit).
//   GEMM2  Yt[C x 32 px] += W2c[C x 32 hid] . Ht                      v_mfma_f32_32x32x16
//          The accumulator tile of GEMM1 IS the B operand of GEMM2 (the sum runs over its row
//          index), so it is only converted to 16 bit -- no LDS round trip, no lane movement.  The
//          k order inside a k-step is then permuted (position 8h+e of k-step s holds hidden unit
//          16s + 8(e>>2) + 4h + (e&3)); the packed W2 carries the same permutation.
// Filters are streamed L2 -> LDS in chunks of SUBS x 32 hidden units through a 2-deep ring (one
// barrier per chunk); for C = 64 the whole block's filters stay resident and the tile loop runs
// barrier-free.  The packed filter image in HBM is byte-identical to the LDS image (row padding
// included: 16 B per W1 row, 80-byte W2 rows), so staging is a flat 16-byte copy.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T> struct M32;
template <> struct M32<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct M32<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <int C> struct FusedGeom {
  static constexpr int CP = (C + 31) / 32 * 32;         // GEMM2's output rows: C rounded up to whole 32-row tiles
                                                        // (convnext_nano's 80 -> 96: the packed W2 carries zero rows)
  static constexpr int NSUB = 4 * C / 32;               // 32-hidden-unit sub-chunks per block
  static constexpr int SUBS = C <= 64 ? 4 : 2;          // sub-chunks per ring slot
  static constexpr int NCHUNK = NSUB / SUBS;
  static constexpr int W1ROW = 2 * C + 16;              // bytes per staged W1 row (padded)
  static constexpr int W2ROW = 80;                      // bytes per staged W2 row (32 x 2 B + 16)
  static constexpr int SUBBYTES = 32 * W1ROW + CP * W2ROW;
  static constexpr int CHUNKBYTES = SUBS * SUBBYTES;
  static constexpr bool RESIDENT = NCHUNK <= 2;
};

template <typename T, int C>
__global__ __launch_bounds__(512, C <= 64 ? 4 : 2) void fused_mlp_kernel(
    const T* __restrict__ xn, const unsigned char* __restrict__ wpk, const float* __restrict__ b1,
    const float* __restrict__ b2, const float* __restrict__ gamma, float* x, int M, int ntiles,
    T* __restrict__ post_out = nullptr, const float* __restrict__ pw = nullptr,
    const float* __restrict__ pb = nullptr, int post_mode = 0, const float* xres = nullptr) {
  // xres: where the residual rows are read (the training forward keeps a block's input); x itself when omitted
  using G = FusedGeom<C>;
  if (xres == nullptr) xres = x;
  using frag = typename M32<T>::frag;
  constexpr int KS1 = C / 16;   // k-steps of GEMM1
  constexpr int CT = G::CP / 32;   // 32-channel output tiles of GEMM2 (the last one ragged when C % 32 != 0)
  static_assert(C % 16 == 0 && (4 * C) % 32 == 0 && G::NSUB % G::SUBS == 0, "geometry");
  constexpr int PIECES = G::CHUNKBYTES / 16;      // 16-byte pieces per chunk
  constexpr int CPT = (PIECES + 511) / 512;       // ... per thread (last round ragged)
  static_assert(G::CHUNKBYTES % 16 == 0, "chunk must be a whole number of 16-byte pieces");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;                                  // 2 x CHUNKBYTES
  float* b1s = reinterpret_cast<float*>(smem + 2 * G::CHUNKBYTES);  // 4C floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;

  for (int i = tid; i < 4 * C; i += 512) b1s[i] = b1[i];

  uint4 stage[CPT];
#define GLOAD(chunk_)                                                                          \
  do {                                                                                         \
    const uint4* src_ = reinterpret_cast<const uint4*>(wpk + (size_t)(chunk_) * G::CHUNKBYTES); \
    _Pragma("unroll") for (int i_ = 0; i_ < CPT; ++i_) {                                       \
      const int p_ = tid + i_ * 512;                                                           \
      stage[i_] = src_[p_ < PIECES ? p_ : PIECES - 1]; /* ragged last round: clamped */       \
    }                                                                                          \
  } while (0)
#define SSTORE(buf_)                                                                           \
  do {                                                                                         \
    uint4* dst_ = reinterpret_cast<uint4*>(ring + (buf_) * G::CHUNKBYTES);                     \
    _Pragma("unroll") for (int i_ = 0; i_ < CPT; ++i_) {                                       \
      const int p_ = tid + i_ * 512;                                                           \
      if (i_ < CPT - 1 || p_ < PIECES) dst_[p_] = stage[i_];                                   \
    }                                                                                          \
  } while (0)

  if (G::RESIDENT) {
#pragma unroll
    for (int c = 0; c < G::NCHUNK; ++c) {
      GLOAD(c);
      SSTORE(c);
    }
  } else {
    GLOAD(0);
    SSTORE(0);
  }
  __syncthreads();

  long seq = 0;  // chunks consumed so far by this workgroup (ring position)
  // B operand of GEMM1: the pixel's C channels, k-step ks holds channels 16ks + 8h .. +7.  The fragments of
  // the NEXT tile and the residual rows of THIS tile are requested before the tile's MFMA work, so their HBM
  // latency hides under it (PMC: the waves of this kernel were parked on s_waitcnt / barriers 55 % of the time)
  frag xf[KS1], xnext[KS1];
  {
    const int m0 = (int)blockIdx.x * 256 + wave * 32 + lr;
    const int mc0 = m0 < M ? m0 : M - 1;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
      xf[ks] = *reinterpret_cast<const frag*>(xn + (size_t)mc0 * C + ks * 16 + h * 8);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m = tile * 256 + wave * 32 + lr;          // this lane's pixel row
    const int mc = m < M ? m : M - 1;
    // (C = 64 runs four waves per SIMD on 128 VGPRs, C = 160 holds 2 x 10 input fragments and 5 output tiles:
    //  no room to hold the residual rows across the tile)
    constexpr bool PRE_R = C > 64 && C <= 128;
    float4 rres[PRE_R ? CT : 1][4];                      // x[m][c .. c+3] for the epilogue
    if (PRE_R) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ct * 32 + 8 * q < C)
            rres[PRE_R ? ct : 0][q] =
                *reinterpret_cast<const float4*>(xres + (size_t)mc * C + ct * 32 + 8 * q + 4 * h);
    }
    {
      const int tn = tile + (int)gridDim.x;
      const int mn = (tn < ntiles ? tn : tile) * 256 + wave * 32 + lr;
      const int mcn = mn < M ? mn : M - 1;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
        xnext[ks] = *reinterpret_cast<const frag*>(xn + (size_t)mcn * C + ks * 16 + h * 8);
    }
    f32x16 yacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;

    for (int chunk = 0; chunk < G::NCHUNK; ++chunk, ++seq) {
      const int buf = G::RESIDENT ? chunk : (int)(seq & 1);
      // ring: fetch the next chunk (wrapping into the next tile) while this one is consumed; the
      // very last fetch of a workgroup is redundant but harmless (nobody reads that slot again)
      if (!G::RESIDENT) GLOAD((chunk + 1) % G::NCHUNK);
      const unsigned char* cb = ring + buf * G::CHUNKBYTES;
#pragma unroll
      for (int sub = 0; sub < G::SUBS; ++sub) {
        const unsigned char* w1s = cb + sub * G::SUBBYTES;
        const unsigned char* w2s = w1s + 32 * G::W1ROW;
        // ---- GEMM1
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const frag a = *reinterpret_cast<const frag*>(w1s + lr * G::W1ROW + ks * 32 + h * 16);
          hacc = M32<T>::run(a, xf[ks], hacc);
        }
        // ---- bias + GELU; register r <-> hidden unit (r&3) + 8(r>>2) + 4h of this sub-chunk
        const float* bp = b1s + (chunk * G::SUBS + sub) * 32 + 4 * h;
        frag hf[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q);
          const float v0 = gelu_for<T>(hacc[4 * q + 0] + bv.x);
          const float v1 = gelu_for<T>(hacc[4 * q + 1] + bv.y);
          const float v2 = gelu_for<T>(hacc[4 * q + 2] + bv.z);
          const float v3 = gelu_for<T>(hacc[4 * q + 3] + bv.w);
          hf[q >> 1][(q & 1) * 4 + 0] = (T)v0;
          hf[q >> 1][(q & 1) * 4 + 1] = (T)v1;
          hf[q >> 1][(q & 1) * 4 + 2] = (T)v2;
          hf[q >> 1][(q & 1) * 4 + 3] = (T)v3;
        }
        // ---- GEMM2
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const frag a = *reinterpret_cast<const frag*>(w2s + (ct * 32 + lr) * G::W2ROW +
                                                          s * 32 + h * 16);
            yacc[ct] = M32<T>::run(a, hf[s], yacc[ct]);
          }
        }
      }
      if (!G::RESIDENT) {
        SSTORE((int)((seq + 1) & 1));
        __syncthreads();
      }
    }
    // ---- epilogue: x[m][c] += gamma[c] * (y + b2[c]); lane owns 4 consecutive channels per q
    if (m < M) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (ct * 32 + 8 * q >= C) continue;   // rows C .. CP-1 of the ragged last tile
          const int c = ct * 32 + 8 * q + 4 * h;
          const float4 bv = *reinterpret_cast<const float4*>(b2 + c);
          const float4 gv = *reinterpret_cast<const float4*>(gamma + c);
          float4* px = reinterpret_cast<float4*>(x + (size_t)m * C + c);
          float4 r = PRE_R ? rres[PRE_R ? ct : 0][q] : *reinterpret_cast<const float4*>(xres + (size_t)m * C + c);
          r.x += gv.x * (yacc[ct][4 * q + 0] + bv.x);
          r.y += gv.y * (yacc[ct][4 * q + 1] + bv.y);
          r.z += gv.z * (yacc[ct][4 * q + 2] + bv.z);
          r.w += gv.w * (yacc[ct][4 * q + 3] + bv.w);
          *px = r;
          yacc[ct][4 * q + 0] = r.x;   // keep the new row for the optional post-op below
          yacc[ct][4 * q + 1] = r.y;
          yacc[ct][4 * q + 2] = r.z;
          yacc[ct][4 * q + 3] = r.w;
        }
      }
      // ---- optional second output for the next consumer of x (MaxViT schedule): post_mode 1 =
      // LayerNorm_C(x) * pw + pb (eps 1e-6), post_mode 2 = x * pw + pb (an eval-mode BatchNorm), in the operand
      // type.  A pixel's channels live on lanes lr and lr + 32: one cross-lane step per reduction.
      if (G::CP == C && post_mode != 0) {   // (whole tiles only: the MaxViT widths)
        float mean = 0.f, rstd = 1.f;
        if (post_mode == 1) {
          float s = 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += yacc[ct][r];
          s += __shfl_xor(s, 32);
          mean = s * (1.0f / C);
          float qv = 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = yacc[ct][r] - mean;
              qv = fmaf(d, d, qv);
            }
          qv += __shfl_xor(qv, 32);
          rstd = rsqrtf(qv * (1.0f / C) + 1e-6f);
        }
        typedef T __attribute__((ext_vector_type(4))) T4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = ct * 32 + 8 * q + 4 * h;
            const float4 wv = *reinterpret_cast<const float4*>(pw + c);
            const float4 sv = *reinterpret_cast<const float4*>(pb + c);
            T4 o;
            o[0] = (T)((yacc[ct][4 * q + 0] - mean) * rstd * wv.x + sv.x);
            o[1] = (T)((yacc[ct][4 * q + 1] - mean) * rstd * wv.y + sv.y);
            o[2] = (T)((yacc[ct][4 * q + 2] - mean) * rstd * wv.z + sv.z);
            o[3] = (T)((yacc[ct][4 * q + 3] - mean) * rstd * wv.w + sv.w);
            *reinterpret_cast<T4*>(post_out + (size_t)m * C + c) = o;
          }
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) xf[ks] = xnext[ks];
  }
}

#undef GLOAD
#undef SSTORE

// master fp32 W1 [4C][C], W2 [C][4C]  ->  padded, chunked, k-permuted 16-bit image
template <typename T, int C>
__global__ void pack_fused_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                  T* __restrict__ dst) {
  using G = FusedGeom<C>;
  constexpr int SUBEL = G::SUBBYTES / 2;
  constexpr int W1EL = G::W1ROW / 2, W2EL = G::W2ROW / 2;
  const int total = G::NSUB * SUBEL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int sub = i / SUBEL, o = i - sub * SUBEL;
    float v = 0.f;
    if (o < 32 * W1EL) {
      const int row = o / W1EL, k = o - row * W1EL;
      if (k < C) v = w1[(size_t)(sub * 32 + row) * C + k];
    } else {
      const int o2 = o - 32 * W1EL;
      const int c = o2 / W2EL, pos = o2 - c * W2EL;
      if (pos < 32 && c < C) {
        const int s = pos >> 4, hh = (pos >> 3) & 1, e = pos & 7;
        const int hid = sub * 32 + 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3);
        v = w2[(size_t)c * 4 * C + hid];
      }
    }
    dst[i] = (T)v;
  }
}

template <typename T, int C>
int launch_fused_cfg(const void* xn, const void* wpk, const float* b1, const float* b2,
                     const float* gamma, float* x, int M, hipStream_t st, void* post_out,
                     const float* pw, const float* pb, int post_mode, const float* xres = nullptr) {
  using G = FusedGeom<C>;
  const size_t lds = 2 * (size_t)G::CHUNKBYTES + 4 * C * sizeof(float);
  auto kern = fused_mlp_kernel<T, C>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.done();
  }
  const int ntiles = (M + 255) / 256;
  const int maxwg = (C <= 64 ? 2 : 1) * 256;   // workgroups resident on the chip
  const int grid = ntiles < maxwg ? ntiles : maxwg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, reinterpret_cast<const T*>(xn),
                     reinterpret_cast<const unsigned char*>(wpk), b1, b2, gamma, x, M, ntiles,
                     reinterpret_cast<T*>(post_out), pw, pb, post_out ? post_mode : 0, xres);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T, int C>
int launch_pack_cfg(const float* w1, const float* w2, void* dst, hipStream_t st) {
  using G = FusedGeom<C>;
  const int total = G::NSUB * G::SUBBYTES / 2;
  hipLaunchKernelGGL((pack_fused_kernel<T, C>), dim3((total + 255) / 256), dim3(256), 0, st, w1, w2,
                     reinterpret_cast<T*>(dst));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

bool fused_mlp_supported(int prec, int C) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (C == 64 || C == 128 || C == 80 || C == 160);
}

size_t fused_mlp_packed_bytes(int C) {
  if (C == 64) return (size_t)FusedGeom<64>::NSUB * FusedGeom<64>::SUBBYTES;
  if (C == 128) return (size_t)FusedGeom<128>::NSUB * FusedGeom<128>::SUBBYTES;
  if (C == 80) return (size_t)FusedGeom<80>::NSUB * FusedGeom<80>::SUBBYTES;
  if (C == 160) return (size_t)FusedGeom<160>::NSUB * FusedGeom<160>::SUBBYTES;
  return 0;
}

int launch_pack_fused_mlp(int prec, int C, const float* w1, const float* w2, void* dst,
                          hipStream_t st) {
  if (prec == BTSBOT_BF16 && C == 64) return launch_pack_cfg<bf16_t, 64>(w1, w2, dst, st);
  if (prec == BTSBOT_BF16 && C == 128) return launch_pack_cfg<bf16_t, 128>(w1, w2, dst, st);
  if (prec == BTSBOT_F16 && C == 64) return launch_pack_cfg<f16_t, 64>(w1, w2, dst, st);
  if (prec == BTSBOT_F16 && C == 128) return launch_pack_cfg<f16_t, 128>(w1, w2, dst, st);
  if (prec == BTSBOT_BF16 && C == 80) return launch_pack_cfg<bf16_t, 80>(w1, w2, dst, st);
  if (prec == BTSBOT_BF16 && C == 160) return launch_pack_cfg<bf16_t, 160>(w1, w2, dst, st);
  if (prec == BTSBOT_F16 && C == 80) return launch_pack_cfg<f16_t, 80>(w1, w2, dst, st);
  if (prec == BTSBOT_F16 && C == 160) return launch_pack_cfg<f16_t, 160>(w1, w2, dst, st);
  btsbot_set_error("pack_fused_mlp: unsupported (prec %d, C %d)", prec, C);
  return BTSBOT_ERR_INVALID_ARG;
}

int launch_fused_mlp(int prec, int C, const void* xn, const void* wpk, const float* b1,
                     const float* b2, const float* gamma, float* x, int M, hipStream_t st,
                     void* post_out, const float* pw, const float* pb, int post_mode, const float* xres) {
  if (M <= 0) return BTSBOT_OK;
  if (prec == BTSBOT_BF16 && C == 64)
    return launch_fused_cfg<bf16_t, 64>(xn, wpk, b1, b2, gamma, x, M, st, post_out, pw, pb, post_mode, xres);
  if (prec == BTSBOT_BF16 && C == 128)
    return launch_fused_cfg<bf16_t, 128>(xn, wpk, b1, b2, gamma, x, M, st, post_out, pw, pb, post_mode, xres);
  if (prec == BTSBOT_F16 && C == 64)
    return launch_fused_cfg<f16_t, 64>(xn, wpk, b1, b2, gamma, x, M, st, post_out, pw, pb, post_mode, xres);
  if (prec == BTSBOT_F16 && C == 128)
    return launch_fused_cfg<f16_t, 128>(xn, wpk, b1, b2, gamma, x, M, st, post_out, pw, pb, post_mode, xres);
  // convnext_nano's stages 0-1 (no second output: post_mode is a MaxViT feature)
  if (post_out == nullptr && prec == BTSBOT_BF16 && C == 80)
    return launch_fused_cfg<bf16_t, 80>(xn, wpk, b1, b2, gamma, x, M, st, nullptr, nullptr, nullptr, 0, xres);
  if (post_out == nullptr && prec == BTSBOT_BF16 && C == 160)
    return launch_fused_cfg<bf16_t, 160>(xn, wpk, b1, b2, gamma, x, M, st, nullptr, nullptr, nullptr, 0, xres);
  if (post_out == nullptr && prec == BTSBOT_F16 && C == 80)
    return launch_fused_cfg<f16_t, 80>(xn, wpk, b1, b2, gamma, x, M, st, nullptr, nullptr, nullptr, 0, xres);
  if (post_out == nullptr && prec == BTSBOT_F16 && C == 160)
    return launch_fused_cfg<f16_t, 160>(xn, wpk, b1, b2, gamma, x, M, st, nullptr, nullptr, nullptr, 0, xres);
  btsbot_set_error("fused_mlp: unsupported (prec %d, C %d)", prec, C);
  return BTSBOT_ERR_INVALID_ARG;
}
