// Pipelined MFMA GEMM for the 16-bit modes (gfx950), same contract as gemm.hip:
//
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ),   K % 64 == 0
//
// What is different from gemm.hip (which stays as the fp32 / ragged-K path):
//   * operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers) into a
//     2- or 3-slot ring (128x128 tiles: 2 slots = 64 KB so two workgroups share a CU; 64x64 tiles:
//     3 slots, two k-tiles in flight while one is consumed), ONE raw s_barrier per k-tile, counted
//     s_waitcnt vmcnt;
//   * the LDS image is linear (what LDS-DMA requires); bank conflicts are removed by an XOR swizzle
//     applied to the per-lane SOURCE address and again on the fragment read:
//         16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7);
//   * the epilogue is staged through the (then idle) ring so that HBM sees whole rows: 16-byte
//     vectors, 256..512 contiguous bytes per output row instead of 8-byte pieces.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma2<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int TM, int TN, int WM, int WN, int EPI, int NSLOT, bool PRE>
__global__ __launch_bounds__(256) void gemm2_kernel(const T* __restrict__ X,
                                                    const T* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ gamma,
                                                    const float* resid, void* out, int M, int N,
                                                    int K, long bsX = 0, long bsW = 0, long bsO = 0,
                                                    const float* __restrict__ ln_w = nullptr,
                                                    const float* __restrict__ ln_b = nullptr,
                                                    T* __restrict__ ln_out = nullptr) {
  using MM = Mma2<T>;
  // batched form (gridDim.z > 1): problem z reads X + z*bsX, W + z*bsW and writes out/resid + z*bsO
  X += blockIdx.z * bsX;
  W += blockIdx.z * bsW;
  using frag = typename MM::frag;
  constexpr int ROWB = 128;                       // bytes per staged row (64 elements)
  constexpr int SLOT = (TM + TN) * ROWB;          // one ring slot
  constexpr int WTM = TM / WM, WTN = TN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int XB = TM / 8 / 4, WB = TN / 8 / 4; // 8-row blocks per wave per tile
  constexpr int LPT = XB + WB;                    // LDS-DMA instructions per thread per k-tile
  constexpr bool TOUT = EPI == EPI_GELU || EPI == EPI_SILU || EPI == EPI_BIAS_T ||
                        EPI == EPI_GELU_SAVE;   // operand-typed staging tile
  using OT = typename std::conditional<TOUT, T, float>::type;
  constexpr int OPITCH = TN * (int)sizeof(OT) + 16;  // epilogue staging row pitch
  static_assert(NSLOT >= 1 && NSLOT <= 3, "ring depth");   // NSLOT == 1: K == 64 only (one k-tile)
  static_assert(WM * WN == 4 && XB >= 1 && WB >= 1, "tile/wave layout");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;

  // ---- per-lane source pointers of this wave's LDS-DMA pieces (k advances by 128 B per tile)
  const int prow = lane >> 3, ppos = lane & 7;
  const unsigned char* src[LPT];
  int dsto[LPT];  // wave-uniform LDS byte offset of each 1-KiB piece inside a slot
#pragma unroll
  for (int i = 0; i < XB; ++i) {
    const int row = (wave + 4 * i) * 8 + prow;
    const int gr = min(m0 + row, M - 1);
    src[i] = reinterpret_cast<const unsigned char*>(X + (size_t)gr * K) +
             ((ppos ^ ((row >> 1) & 7)) << 4);
    dsto[i] = (wave + 4 * i) * 8 * ROWB;
  }
#pragma unroll
  for (int i = 0; i < WB; ++i) {
    const int row = (wave + 4 * i) * 8 + prow;
    const int gr = min(n0 + row, N - 1);
    src[XB + i] = reinterpret_cast<const unsigned char*>(W + (size_t)gr * K) +
                  ((ppos ^ ((row >> 1) & 7)) << 4);
    dsto[XB + i] = TM * ROWB + (wave + 4 * i) * 8 * ROWB;
  }
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < LPT; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (size_t)kt * ROWB),
                                       (lptr_t)(smem + slot * SLOT + dsto[i]), 16, 0, 0);
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The row-wise epilogue below reads one 16-byte (RESID: fp32 residual) or 8-byte (DGELU: pre-activation)
  // piece per staged chunk from HBM.  Inside its store loop those loads cannot move above the previous
  // iteration's store (out may alias resid), so each of the NIT iterations paid a full memory round trip;
  // requested here, before the first k-tile, they arrive under the main loop (vmcnt retires in order: they are
  // older than every LDS-DMA, so the counted waits below still mean what they say).
  constexpr int CPR = TN * (int)sizeof(OT) / 16;   // 16-byte chunks per staged row
  constexpr int EPC = 16 / (int)sizeof(OT);        // elements per chunk
  constexpr int NIT = TM * CPR / 256;              // epilogue iterations per thread
  static_assert((TM * CPR) % 256 == 0, "epilogue mapping");
  typedef T __attribute__((ext_vector_type(4))) T4p;
  constexpr bool PRE_R = PRE && EPI == EPI_RESID, PRE_D = PRE && EPI == EPI_DGELU;
  float4 rpre[PRE_R ? NIT : 1];
  T4p dpre[PRE_D ? NIT : 1];
  if (PRE_R || PRE_D) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int ml = i / CPR, ch = i - ml * CPR;
      const int m = min(m0 + ml, M - 1), n = min(n0 + ch * EPC, N - EPC);
      const size_t o = (size_t)blockIdx.z * bsO + (size_t)m * N + n;
      if (PRE_R) rpre[it] = *reinterpret_cast<const float4*>(resid + o);
      if (PRE_D) dpre[it] = *reinterpret_cast<const T4p*>(reinterpret_cast<const T*>(resid) + o);
    }
  }

  const int nk = K / 64;
  const int lrow = lane & 15, lq = lane >> 4;
  // NSLOT-1 k-tiles are in flight while one is consumed
  issue(0, 0);
  if (NSLOT == 3 && nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed for this wave once only the younger tiles are outstanding
    if (NSLOT == 3 && kt + 1 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // ... for every wave; and everyone is done reading tile kt-1
    if (NSLOT > 1 && kt + NSLOT - 1 < nk) issue(kt + NSLOT - 1, (kt + NSLOT - 1) % NSLOT);
    const unsigned char* xs = smem + (kt % NSLOT) * SLOT;
    const unsigned char* ws = xs + TM * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag bfr[MI], afr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int r = wm * WTM + mi * 16 + lrow;
        bfr[mi] = *reinterpret_cast<const frag*>(xs + r * ROWB +
                                                 (((ks * 4 + lq) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int r = wn * WTN + ni * 16 + lrow;
        afr[ni] = *reinterpret_cast<const frag*>(ws + r * ROWB +
                                                 (((ks * 4 + lq) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MM::run(afr[ni], bfr[mi], acc[ni][mi]);
    }
  }
  __syncthreads();  // ring is idle from here on (no LDS-DMA in flight: last wait was vmcnt(0))

  // ---- epilogue stage 1: registers -> LDS tile [TM][TN] of OT (lane owns 4 consecutive n)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int nl = wn * WTN + ni * 16 + lq * 4;
    const int n = min(n0 + nl, N - 4);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI != EPI_DGELU && EPI != EPI_PLAIN) bv = *reinterpret_cast<const float4*>(bias + n);
    float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == EPI_RESID) gv = *reinterpret_cast<const float4*>(gamma + n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int ml = wm * WTM + mi * 16 + lrow;
      const f32x4 a = acc[ni][mi];
      unsigned char* dst = smem + ml * OPITCH + nl * (int)sizeof(OT);
      if (EPI == EPI_GELU) {
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 v;
        v[0] = (T)gelu_for<T>(a[0] + bv.x);
        v[1] = (T)gelu_for<T>(a[1] + bv.y);
        v[2] = (T)gelu_for<T>(a[2] + bv.z);
        v[3] = (T)gelu_for<T>(a[3] + bv.w);
        *reinterpret_cast<T4*>(dst) = v;
      } else if (EPI == EPI_SILU || EPI == EPI_BIAS_T || EPI == EPI_GELU_SAVE) {
        // (GELU_SAVE stages the rounded pre-activation; stage 2 writes it and its GELU)
        typedef T __attribute__((ext_vector_type(4))) T4;
        T4 v;
        const float p[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (T)(EPI == EPI_SILU ? silu_for<T>(p[e]) : p[e]);
        *reinterpret_cast<T4*>(dst) = v;
      } else {
        *reinterpret_cast<float4*>(dst) =
            make_float4(gv.x * (a[0] + bv.x), gv.y * (a[1] + bv.y), gv.z * (a[2] + bv.z),
                        gv.w * (a[3] + bv.w));
      }
    }
  }
  __syncthreads();
  // ---- epilogue stage 2: whole rows LDS -> HBM, 16 bytes per lane
  if (EPI == EPI_RESID && ln_out != nullptr) {
    // RESID with the NEXT LayerNorm fused (host guarantees N == TN, so a row of the staged tile is a whole
    // row of the map): the row sits on CPR = TN/4 consecutive lanes -- one or two DPP rows
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int ml = i / CPR, ch = i - ml * CPR;
      const int m = min(m0 + ml, M - 1), n = ch * 4;
      const size_t o = (size_t)blockIdx.z * bsO + (size_t)m * N + n;
      float4 f = *reinterpret_cast<const float4*>(smem + ml * OPITCH + ch * 16);
      const float4 r = PRE_R ? rpre[PRE_R ? it : 0] : *reinterpret_cast<const float4*>(resid + o);
      f.x += r.x; f.y += r.y; f.z += r.z; f.w += r.w;
      float s = group16_sum((f.x + f.y) + (f.z + f.w));
      if (CPR == 32) s += __shfl_xor(s, 16);
      const float mean = s * (1.0f / TN);
      const float dx = f.x - mean, dy = f.y - mean, dz = f.z - mean, dw = f.w - mean;
      float q = group16_sum(fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, dw * dw))));
      if (CPR == 32) q += __shfl_xor(q, 16);
      const float rstd = rsqrtf(q * (1.0f / TN) + 1e-6f);
      if (m0 + ml >= M) continue;
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = f;
      const float4 gw = *reinterpret_cast<const float4*>(ln_w + n);
      const float4 gb = *reinterpret_cast<const float4*>(ln_b + n);
      typedef T __attribute__((ext_vector_type(4))) T4;
      T4 y;
      y[0] = (T)(dx * rstd * gw.x + gb.x);
      y[1] = (T)(dy * rstd * gw.y + gb.y);
      y[2] = (T)(dz * rstd * gw.z + gb.z);
      y[3] = (T)(dw * rstd * gw.w + gb.w);
      *reinterpret_cast<T4*>(ln_out + o) = y;
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * 256;
    const int ml = i / CPR, ch = i - ml * CPR;
    const int m = m0 + ml, n = n0 + ch * EPC;
    if (m >= M || n >= N) continue;
    uint4 v = *reinterpret_cast<const uint4*>(smem + ml * OPITCH + ch * 16);
    const size_t o = (size_t)blockIdx.z * bsO + (size_t)m * N + n;
    if (EPI == EPI_RESID) {
      const float4 r = PRE_R ? rpre[PRE_R ? it : 0] : *reinterpret_cast<const float4*>(resid + o);
      float4 f = *reinterpret_cast<float4*>(&v);
      f.x += r.x; f.y += r.y; f.z += r.z; f.w += r.w;
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + o) = f;
    } else if (EPI == EPI_GELU_SAVE) {
      // training forward: aux (passed through `resid`) keeps the pre-activation the backward
      // differentiates, out = GELU of that ROUNDED value
      *reinterpret_cast<uint4*>(reinterpret_cast<T*>(const_cast<float*>(resid)) + o) = v;
      T* pe = reinterpret_cast<T*>(&v);
#pragma unroll
      for (int e = 0; e < 8; ++e) pe[e] = (T)gelu_for<T>((float)pe[e]);
      *reinterpret_cast<uint4*>(reinterpret_cast<T*>(out) + o) = v;
    } else if (EPI == EPI_DGELU) {
      // backward: out = acc * gelu'(pre), pre read from aux (`resid`)
      typedef T __attribute__((ext_vector_type(4))) T4;
      const float4 f = *reinterpret_cast<float4*>(&v);
      const T4 pre = PRE_D ? dpre[PRE_D ? it : 0] : *reinterpret_cast<const T4*>(reinterpret_cast<const T*>(resid) + o);
      T4 r;
      r[0] = (T)(f.x * gelu_grad_for<T>((float)pre[0]));
      r[1] = (T)(f.y * gelu_grad_for<T>((float)pre[1]));
      r[2] = (T)(f.z * gelu_grad_for<T>((float)pre[2]));
      r[3] = (T)(f.w * gelu_grad_for<T>((float)pre[3]));
      *reinterpret_cast<T4*>(reinterpret_cast<T*>(out) + o) = r;
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<OT*>(out) + o) = v;
    }
  }
}

template <typename T, int TM, int TN, int WM, int WN, int EPI, int NSLOT, bool PRE>
int launch_tile2p(const T* x, const T* w, const float* bias, const float* gamma,
                 const float* resid, void* out, int M, int N, int K, hipStream_t st, int batch = 1,
                 long bsX = 0, long bsW = 0, long bsO = 0, const float* ln_w = nullptr,
                 const float* ln_b = nullptr, void* ln_out = nullptr) {
  constexpr size_t ring = NSLOT * (size_t)(TM + TN) * 128;
  constexpr size_t otile = (size_t)TM * (TN * ((EPI == EPI_GELU || EPI == EPI_SILU || EPI == EPI_BIAS_T || EPI == EPI_GELU_SAVE) ? sizeof(T) : sizeof(float)) + 16);
  constexpr size_t lds = ring > otile ? ring : otile;   // the epilogue tile reuses the ring
  auto kern = gemm2_kernel<T, TM, TN, WM, WN, EPI, NSLOT, PRE>;
  static DevOnce attr_set;
  if (attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.done();
  }
  dim3 grid((M + TM - 1) / TM, (N + TN - 1) / TN, batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, w, bias, gamma, resid, out, M, N, K, bsX, bsW,
                     bsO, ln_w, ln_b, reinterpret_cast<T*>(ln_out));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// BTSBOT_AMD_GEMM2_NO_PREFETCH=1: the epilogue's residual / pre-activation loads stay inside its store loop (A/B)
template <typename T, int TM, int TN, int WM, int WN, int EPI, int NSLOT>
int launch_tile2(const T* x, const T* w, const float* bias, const float* gamma,
                 const float* resid, void* out, int M, int N, int K, hipStream_t st, int batch = 1,
                 long bsX = 0, long bsW = 0, long bsO = 0, const float* ln_w = nullptr,
                 const float* ln_b = nullptr, void* ln_out = nullptr) {
  static const bool no_pre = [] {
    const char* e = getenv("BTSBOT_AMD_GEMM2_NO_PREFETCH");
    return e != nullptr && e[0] == '1';
  }();
  if constexpr (EPI == EPI_RESID || EPI == EPI_DGELU) {
    if (!no_pre)
      return launch_tile2p<T, TM, TN, WM, WN, EPI, NSLOT, true>(x, w, bias, gamma, resid, out, M, N, K, st, batch,
                                                                bsX, bsW, bsO, ln_w, ln_b, ln_out);
  }
  return launch_tile2p<T, TM, TN, WM, WN, EPI, NSLOT, false>(x, w, bias, gamma, resid, out, M, N, K, st, batch, bsX,
                                                             bsW, bsO, ln_w, ln_b, ln_out);
}

template <typename T, int EPI>
int launch_typed2(const void* X, const void* W, const float* bias, const float* gamma,
                  const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  const T* x = reinterpret_cast<const T*>(X);
  const T* w = reinterpret_cast<const T*>(W);
  const long wg128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  // K == 64 is a single k-tile: a one-slot ring halves (thirds) the LDS per workgroup, so twice (three
  // times) as many workgroups share a CU and cover each other's load -> MFMA -> store chain
  static const bool one_slot = [] {
    const char* e = getenv("BTSBOT_AMD_GEMM2_NO_1SLOT");
    return !(e != nullptr && e[0] == '1');
  }();
  static const bool tm64 = [] {
    const char* e = getenv("BTSBOT_AMD_GEMM2_TM64");   // A/B: 64x128 tiles, 3-slot ring (72 KB: two workgroups per CU, two k-tiles in flight each)
    return e != nullptr && e[0] == '1';
  }();
  // long reductions on big problems (MaxViT fc2 / conv3 at C >= 256: K >= 1024) run 7 % faster on 64x128 tiles
  // with a 3-slot ring (two workgroups per CU, two k-tiles in flight each); the short-K shapes lose 15 % there
  if (N >= 128 && wg128 >= 512 && (K >= 1024 || (tm64 && K >= 128)))
    return launch_tile2<T, 64, 128, 1, 4, EPI, 3>(x, w, bias, gamma, resid, out, M, N, K, st);
  if (N >= 128 && wg128 >= 256) {
    if (K == 64 && one_slot)
      return launch_tile2<T, 128, 128, 2, 2, EPI, 1>(x, w, bias, gamma, resid, out, M, N, K, st);
    return launch_tile2<T, 128, 128, 2, 2, EPI, 2>(x, w, bias, gamma, resid, out, M, N, K, st);
  }
  if (K == 64 && one_slot)
    return launch_tile2<T, 64, 64, 2, 2, EPI, 1>(x, w, bias, gamma, resid, out, M, N, K, st);
  return launch_tile2<T, 64, 64, 2, 2, EPI, 3>(x, w, bias, gamma, resid, out, M, N, K, st);
}

template <typename T>
int launch_epi2(int epi, const void* X, const void* W, const float* bias, const float* gamma,
                const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  switch (epi) {
    case EPI_GELU: return launch_typed2<T, EPI_GELU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_RESID: return launch_typed2<T, EPI_RESID>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS: return launch_typed2<T, EPI_BIAS>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_SILU: return launch_typed2<T, EPI_SILU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_BIAS_T: return launch_typed2<T, EPI_BIAS_T>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_GELU_SAVE: return launch_typed2<T, EPI_GELU_SAVE>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_DGELU: return launch_typed2<T, EPI_DGELU>(X, W, bias, gamma, resid, out, M, N, K, st);
    case EPI_PLAIN: return launch_typed2<T, EPI_PLAIN>(X, W, bias, gamma, resid, out, M, N, K, st);
  }
  btsbot_set_error("launch_gemm2: bad epilogue %d", epi);
  return BTSBOT_ERR_INVALID_ARG;
}

}  // namespace

// out_b (f32) = resid_b + X_b . W_b^T for `batch` independent problems of M rows each (per-alert filters)
// (ln_out != NULL: also ln_out [rows][N] T = LayerNorm_N(out row) * ln_w + ln_b, eps 1e-6; needs N in {64,128}.
//  bsW == 0 with batch == 1 is the plain, un-batched case: `bias` / `gamma` are then real vectors)
int launch_gemm2_batched_resid(int prec, const void* X, const void* W, const float* zero_bias,
                               const float* one_gamma, const float* resid, float* out, int batch, int M,
                               int N, int K, hipStream_t st, const float* ln_w, const float* ln_b,
                               void* ln_out) {
  if (!gemm2_supported(prec, M, N, K) || batch < 1) {
    btsbot_set_error("gemm2_batched: unsupported (prec %d, M %d, N %d, K %d)", prec, M, N, K);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long bsX = (long)M * K, bsW = (long)N * K, bsO = (long)M * N;
  if (ln_out != nullptr && N != 64 && N != 128) {
    btsbot_set_error("gemm2 resid+LN: N=%d must be 64 or 128 (one tile per row)", N);
    return BTSBOT_ERR_INVALID_ARG;
  }
  const bool big = ln_out != nullptr ? N == 128
                                     : N >= 128 && (long)((M + 127) / 128) * ((N + 127) / 128) * batch >= 256;
#define G2B(TT)                                                                                         \
  (big ? launch_tile2<TT, 128, 128, 2, 2, EPI_RESID, 2>(reinterpret_cast<const TT*>(X),                \
                                                        reinterpret_cast<const TT*>(W), zero_bias,      \
                                                        one_gamma, resid, out, M, N, K, st, batch, bsX,  \
                                                        bsW, bsO, ln_w, ln_b, ln_out)                    \
       : launch_tile2<TT, 64, 64, 2, 2, EPI_RESID, 3>(reinterpret_cast<const TT*>(X),                  \
                                                      reinterpret_cast<const TT*>(W), zero_bias,        \
                                                      one_gamma, resid, out, M, N, K, st, batch, bsX,    \
                                                      bsW, bsO, ln_w, ln_b, ln_out))
  return prec == BTSBOT_BF16 ? G2B(bf16_t) : G2B(f16_t);
#undef G2B
}

bool gemm2_supported(int prec, int M, int N, int K) {
  return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && K % 64 == 0 && N % 64 == 0 && N >= 64 &&
         M >= 1;
}

int launch_gemm2(int prec, int epi, const void* X, const void* W, const float* bias,
                 const float* gamma, const float* resid, void* out, int M, int N, int K,
                 hipStream_t st) {
  if (prec == BTSBOT_BF16)
    return launch_epi2<bf16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
  return launch_epi2<f16_t>(epi, X, W, bias, gamma, resid, out, M, N, K, st);
}
