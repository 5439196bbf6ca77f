// Persistent MFMA GEMM with the filter panel resident in LDS (gfx950, 16-bit modes):
//
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ),   K in {128, 256},  N % 128 == 0,  typed output
//
// The wide-N, short-K products of the MaxViT middle stages (conv1_1x1, attn.qkv, mlp.fc1 at C = 128 / 256,
// reached from /root/reference/btsbot/architectures.py:51,97): two to four k-tiles per output tile.  In
// gemm2.hip every 128x128 tile pays its own prologue (first loads exposed) and re-reads its 128 x K filter
// slice; those GEMMs ran at 2.1 TB/s and 17 % of the MFMA peak, bound by neither.  Here
//   * a workgroup owns ONE 128-column filter panel (all of K, up to 64 KB, LDS-DMA'd once) and walks over
//     many 128-row tiles of X (persistent: blockIdx.x strides over the row tiles);
//   * the X k-tiles stream through a 3-slot LDS-DMA ring that keeps running ACROSS row tiles: while a tile's
//     epilogue runs, the next tile's first two k-tiles are already in flight;
//   * the epilogue is staged through its own LDS tile (whole 256-byte rows to HBM) so it never collides
//     with the ring.
// Same swizzled linear LDS image, fragment reads and wave layout (2 x 2 waves, 64 x 64 each) as gemm2.hip.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct Mma4;
template <> struct Mma4<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma4<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int TM = 128, TN = 128, ROWB = 128, NSLOT = 3;
constexpr int XB = TM / 8 / 4;                 // LDS-DMA pieces per thread per X k-tile (4)
constexpr int TILEB = TM * ROWB;               // one k-tile image (16 KB)

template <typename T, int NK, int EPI>
__global__ __launch_bounds__(256) void gemm4_kernel(const T* __restrict__ X, const T* __restrict__ W,
                                                    const float* __restrict__ bias, T* __restrict__ out,
                                                    int M, int N, int mtiles) {
  using MM = Mma4<T>;
  using frag = typename MM::frag;
  constexpr int K = NK * 64;
  constexpr int OPITCH = TN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wp = smem;                         // [NK][TN rows][128 B]
  unsigned char* ring = smem + NK * TILEB;          // [NSLOT][TM rows][128 B]
  unsigned char* stage = ring + NSLOT * TILEB;      // [TM][OPITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.y * TN;
  const int prow = lane >> 3, ppos = lane & 7;
  const int lrow = lane & 15, lq = lane >> 4;

  // ---- filter panel: NK k-tile images, swizzled through the per-lane source address
#pragma unroll
  for (int kt = 0; kt < NK; ++kt)
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int row = (wave + 4 * i) * 8 + prow;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(W + (size_t)(n0 + row) * K) +
                                 ((ppos ^ ((row >> 1) & 7)) << 4) + kt * ROWB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wp + kt * TILEB + (wave + 4 * i) * 8 * ROWB),
                                       16, 0, 0);
    }

  const int my_tiles = (mtiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * NK;                  // k-tiles this workgroup consumes
  auto issue = [&](int gidx) {
    const int t = gidx / NK, kt = gidx - t * NK;
    const int m0 = ((int)blockIdx.x + t * (int)gridDim.x) * TM;
    unsigned char* dst = ring + (gidx % NSLOT) * TILEB;
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int row = (wave + 4 * i) * 8 + prow;
      const int gr = min(m0 + row, M - 1);
      const unsigned char* src = reinterpret_cast<const unsigned char*>(X + (size_t)gr * K) +
                                 ((ppos ^ ((row >> 1) & 7)) << 4) + kt * ROWB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (wave + 4 * i) * 8 * ROWB), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) issue(0);
  if (total > 1) issue(1);
  for (int gidx = 0; gidx < total; ++gidx) {
    const int t = gidx / NK, kt = gidx - t * NK;
    // k-tile gidx (and everything older: the filter panel, earlier stores) has landed once only the
    // one younger k-tile is outstanding
    if (gidx + 1 < total) wait_vm<XB>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const bool last_k = kt == NK - 1;
    if (!last_k && gidx + 2 < total) issue(gidx + 2);   // (after the epilogue on a tile's last k-tile)
    const unsigned char* xs = ring + (gidx % NSLOT) * TILEB;
    const unsigned char* ws = wp + kt * TILEB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag bfr[4], afr[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int r = wm * 64 + mi * 16 + lrow;
        bfr[mi] = *reinterpret_cast<const frag*>(xs + r * ROWB + (((ks * 4 + lq) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int r = wn * 64 + ni * 16 + lrow;
        afr[ni] = *reinterpret_cast<const frag*>(ws + r * ROWB + (((ks * 4 + lq) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = MM::run(afr[ni], bfr[mi], acc[ni][mi]);
    }
    if (!last_k) continue;
    // ---- epilogue of row tile t: registers -> staging tile -> whole rows to HBM
    const int m0 = ((int)blockIdx.x + t * (int)gridDim.x) * TM;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int nl = wn * 64 + ni * 16 + lq * 4;
      const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + nl);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int ml = wm * 64 + mi * 16 + lrow;
        const f32x4 a = acc[ni][mi];
        typedef T __attribute__((ext_vector_type(4))) T4;
        const float p[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
        T4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = (T)(EPI == EPI_SILU ? silu_fast(p[e]) : EPI == EPI_GELU ? gelu_fast(p[e]) : p[e]);
        *reinterpret_cast<T4*>(stage + ml * OPITCH + nl * 2) = v;
        acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (not __syncthreads: that would also drain the
    __builtin_amdgcn_s_barrier();                         //  X prefetch that is in flight)
    for (int i = tid; i < TM * 16; i += 256) {        // 16 x 16-byte chunks per 256-byte row
      const int ml = i >> 4, ch = i & 15;
      const int m = m0 + ml;
      if (m < M)
        *reinterpret_cast<uint4*>(out + (size_t)m * N + n0 + ch * 8) =
            *reinterpret_cast<const uint4*>(stage + ml * OPITCH + ch * 16);
    }
    if (gidx + 2 < total) issue(gidx + 2);
    // (the next tile's first barrier orders these staging reads before the next epilogue's writes)
  }
}

template <typename T, int NK, int EPI>
int launch4(const void* X, const void* W, const float* bias, void* out, int M, int N, hipStream_t st) {
  constexpr size_t lds = (size_t)(NK + NSLOT) * TILEB + (size_t)TM * (TN * 2 + 16);
  auto kern = gemm4_kernel<T, NK, EPI>;
  static bool attr = false;
  if (!attr) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  const int mtiles = (M + TM - 1) / TM, ntiles = N / TN;
  int gx = 256 / ntiles;                    // one workgroup per CU (the LDS footprint allows no more)
  gx = gx < 1 ? 1 : (gx > mtiles ? mtiles : gx);
  hipLaunchKernelGGL(kern, dim3(gx, ntiles), dim3(256), lds, st, reinterpret_cast<const T*>(X),
                     reinterpret_cast<const T*>(W), bias, reinterpret_cast<T*>(out), M, N, mtiles);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

template <typename T, int NK>
int launch4_epi(int epi, const void* X, const void* W, const float* bias, void* out, int M, int N,
                hipStream_t st) {
  switch (epi) {
    case EPI_SILU: return launch4<T, NK, EPI_SILU>(X, W, bias, out, M, N, st);
    case EPI_GELU: return launch4<T, NK, EPI_GELU>(X, W, bias, out, M, N, st);
    case EPI_BIAS_T: return launch4<T, NK, EPI_BIAS_T>(X, W, bias, out, M, N, st);
  }
  btsbot_set_error("launch_gemm4: bad epilogue %d", epi);
  return BTSBOT_ERR_INVALID_ARG;
}

}  // namespace

bool gemm4_supported(int prec, int epi, int M, int N, int K) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && (K == 128 || K == 256) && N % 128 == 0 &&
         N >= 256 && M >= 4096 && (epi == EPI_SILU || epi == EPI_GELU || epi == EPI_BIAS_T);
}

int launch_gemm4(int prec, int epi, const void* X, const void* W, const float* bias, void* out, int M,
                 int N, int K, hipStream_t st) {
  if (!gemm4_supported(prec, epi, M, N, K)) {
    btsbot_set_error("launch_gemm4: unsupported (prec %d, epi %d, M %d, N %d, K %d)", prec, epi, M, N, K);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (prec == BTSBOT_BF16)
    return K == 128 ? launch4_epi<bf16_t, 2>(epi, X, W, bias, out, M, N, st)
                    : launch4_epi<bf16_t, 4>(epi, X, W, bias, out, M, N, st);
  return K == 128 ? launch4_epi<f16_t, 2>(epi, X, W, bias, out, M, N, st)
                  : launch4_epi<f16_t, 4>(epi, X, W, bias, out, M, N, st);
}
