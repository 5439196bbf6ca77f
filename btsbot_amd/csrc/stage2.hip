// Stage-2 (3x3 maps, C = 256) front half of a ConvNeXt block in ONE launch (gfx950, 16-bit modes):
//
//   h = GELU( LN( dwconv7x7(x) + b_dw ) W1^T + b1 )          x: [B][9][256] f32 -> h: [B*9][1024]
//
// (timm ConvNeXtBlock.conv_dw / norm / mlp.fc1 / mlp.act, reached from
// /root/reference/btsbot/architectures.py:108,132).  Replaces dw3_ln_kernel + the fc1 GEMM launch:
// the LayerNorm output never goes to HBM, and the GEMM gets a tile shape that fits the problem.
//
// Workgroup = 16 alerts (144 pixel rows) x 256 hidden units, one workgroup per CU: at B = 1024
// that is exactly 64 x 4 = 256 workgroups.  The four workgroups that share a row tile are given
// the same blockIdx % 8, i.e. the same XCD / L2 (speed only).
// 16 waves (1024 threads): two thirds of this kernel is VALU work (depthwise taps, LayerNorm,
// GELU), and one wave can issue a VALU instruction only every ~4.7 cycles while the SIMD accepts
// one from ANOTHER wave every ~1.2 (tools/unit/valu_rate.hip) -- so 4 waves per SIMD, 128 VGPRs.
//   prologue  wave = one alert, lane = 4 channels (16-byte loads); the central 5x5 taps (all a 3x3
//             map can touch) sit in LDS; LayerNorm sums are wave-local (DPP + permlane swaps); the
//             16-bit result is the MFMA operand image ([144][256], 16-byte chunk c of row r at
//             position c ^ (r & 15));
//   main      K = 256 in 8 slabs of 32 through a 4-slot LDS-DMA ring (64-byte rows, chunk c of row r
//             at position c ^ F[(r >> 2) & 3], F = {0,3,2,1}: conflict-free ds_read_b128); a wave
//             multiplies 144 pixels x ITS 16 filter rows -- the rows it fetched itself, so it only
//             waits for its own DMA and there is no barrier in the loop; v_mfma_f32_16x16x32 with
//             the filters as the A operand: a lane ends up with 4 consecutive hidden units of a pixel;
//   epilogue  + bias, GELU (packed f32 math), 16-bit, staged through LDS, whole 512-byte rows out.
#include "common.h"

struct S2Fc1Args {
  const float* x;        // [B][9][256]
  const float* dw_w;     // [49][256] tap-major
  const float* dw_b;
  const float* ln_w;
  const float* ln_b;
  const void* w1;        // [1024][256] 16-bit
  const float* b1;       // [1024]
  void* h;               // [B*9][1024] 16-bit
  int B;
  unsigned long long* stamps;   // optional: workgroup 0 / thread 0 stores the shader clock per phase
};

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct M2;
template <> struct M2<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct M2<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int C = 256, HID = 1024, GA = 16, ROWS = GA * 9;       // 144
constexpr int TN = 256;                                            // hidden units per workgroup
constexpr int NT = 1024;                                           // threads: 16 waves
constexpr int AROW = C * 2;                                        // 512 B per operand-image row
constexpr int ATILE = ROWS * AROW;                                 // 73728
constexpr int KS = 32, NSLAB = C / KS, NSLOT = 4;
constexpr int SLABB = TN * KS * 2;                                 // 16384
constexpr int OPITCH = TN * 2 + 16;                                // epilogue staging pitch
constexpr int MI = ROWS / 16;
constexpr float LN_EPS = 1e-6f;
#define S2STAMP(i)                                                                         \
  do {                                                                                     \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)
constexpr size_t LDS_BYTES = (size_t)ATILE + NSLOT * SLABB;       // 139264
static_assert((size_t)ROWS * OPITCH <= LDS_BYTES, "epilogue staging must fit");
static_assert(25 * C * 4 <= 2 * SLABB, "the tap table borrows ring slots 2..3");

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swz4(int row) {   // F[(row >> 2) & 3], F = {0,3,2,1}
  const int g = (row >> 2) & 3;
  return (4 - g) & 3;
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T>
__global__ __launch_bounds__(NT) void s2_fc1_kernel(S2Fc1Args a) {
  using frag = typename M2<T>::frag;
  typedef T T4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* atile = smem;
  unsigned char* ring = smem + ATILE;
  float* taps = reinterpret_cast<float*>(ring + 2 * SLABB);   // [25][256], dead before slab 2 lands

  // ---- which tile: blocks with equal (blockIdx % 8) share an XCD; keep a row tile's 4 column
  //      tiles there so the stage input is fetched into one L2 only
  const int MT = (a.B + GA - 1) / GA;
  const int xcd = blockIdx.x & 7, id = blockIdx.x >> 3;
  const int nt = id & 3, mt = (id >> 2) * 8 + xcd;
  if (mt >= MT) return;                       // whole workgroup, before any barrier
  const int n0 = nt * TN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;
  S2STAMP(0);
  if (a.stamps != nullptr && tid == 0) a.stamps[64 + 2 * blockIdx.x] = wall_clock64();

  // ---- this wave's alert (x) and the shared tap table are requested first
  const int c4 = 4 * lane;
  f32x2 in[9][2];
  {
    const int ag = min(mt * GA + wave, a.B - 1);
    const float* src = a.x + (size_t)ag * 9 * C + c4;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(src + p * C);
      in[p][0] = f32x2{v.x, v.y};
      in[p][1] = f32x2{v.z, v.w};
    }
  }
  for (int i = tid; i < 25 * C / 4; i += NT) {   // tap t of the 5x5 = (dy+2)*5 + dx+2
    const int t = i >> 6, cc = (i & 63) * 4;
    const int dy = t / 5, dx = t - dy * 5;
    *reinterpret_cast<float4*>(taps + t * C + cc) =
        *reinterpret_cast<const float4*>(a.dw_w + ((dy + 1) * 7 + dx + 1) * C + cc);
  }
  // ---- W1 slabs: this wave fetches (and later multiplies) filter rows n0 + 16*wave .. +15
  const unsigned char* wsrc;
  {
    const int row = wave * 16 + (lane >> 2);
    wsrc = reinterpret_cast<const unsigned char*>(a.w1) + (size_t)(n0 + row) * AROW +
           (((lane & 3) ^ swz4(row)) << 4);
  }
  auto issue = [&](int s) {
    __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + s * (KS * 2)),
                                     (lptr_t)(ring + (s % NSLOT) * SLABB + wave * 1024), 16, 0, 0);
  };
  issue(0);
  issue(1);
  S2STAMP(1);
  __syncthreads();   // tap table complete
  S2STAMP(2);

  // ---- prologue: depthwise 7x7 (only the central 5x5 taps can touch a 3x3 map) + LN, one alert
  {
    const float4 b4 = *reinterpret_cast<const float4*>(a.dw_b + c4);
    f32x2 acc[9][2];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      acc[p][0] = f32x2{b4.x, b4.y};
      acc[p][1] = f32x2{b4.z, b4.w};
    }
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const float4 w4 = *reinterpret_cast<const float4*>(taps + ((dy + 2) * 5 + dx + 2) * C + c4);
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
    for (int p = 0; p < 9; ++p) {
      const f32x2 t = acc[p][0] + acc[p][1];
      mean[p] = wave_sum(t[0] + t[1]) * (1.0f / C);
    }
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      acc[p][0] -= (f32x2)(mean[p]);
      acc[p][1] -= (f32x2)(mean[p]);
      const f32x2 q = __builtin_elementwise_fma(acc[p][0], acc[p][0], acc[p][1] * acc[p][1]);
      var[p] = wave_sum(q[0] + q[1]) * (1.0f / C);
    }
    const float4 g4 = *reinterpret_cast<const float4*>(a.ln_w + c4);
    const float4 bb4 = *reinterpret_cast<const float4*>(a.ln_b + c4);
    const f32x2 g0 = {g4.x, g4.y}, g1 = {g4.z, g4.w}, bb0 = {bb4.x, bb4.y}, bb1 = {bb4.z, bb4.w};
#pragma unroll
    for (int p = 0; p < 9; ++p) {
      const float rstd = rsqrtf(var[p] + LN_EPS);
      const f32x2 o0 = __builtin_elementwise_fma(acc[p][0] * (f32x2)(rstd), g0, bb0);
      const f32x2 o1 = __builtin_elementwise_fma(acc[p][1] * (f32x2)(rstd), g1, bb1);
      T4 o;
      o[0] = (T)o0[0];
      o[1] = (T)o0[1];
      o[2] = (T)o1[0];
      o[3] = (T)o1[1];
      const int r = wave * 9 + p;
      *reinterpret_cast<T4*>(atile + r * AROW + (((lane >> 1) ^ (r & 15)) << 4) + (lane & 1) * 8) = o;
    }
  }
  S2STAMP(3);
  if (a.stamps != nullptr && blockIdx.x == 0 && lane == 0) a.stamps[16 + wave] = clock64();
  __syncthreads();   // operand image complete; the tap table is dead
  issue(2);
  S2STAMP(4);

  // ---- main loop: one 16x16x32 k-step per slab, no barriers (see the header)
  f32x4 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    // slab s has landed once only the younger slabs' pieces are outstanding
    if (s + 2 < NSLAB) wait_vm<2>();
    else if (s + 1 < NSLAB) wait_vm<1>();
    else wait_vm<0>();
    const int row = wave * 16 + lrow;
    const frag afr = *reinterpret_cast<const frag*>(ring + (s % NSLOT) * SLABB + row * (KS * 2) +
                                                    ((lq ^ swz4(row)) << 4));
    if (s + 3 < NSLAB) issue(s + 3);   // into the slot of slab s-1, which this wave has consumed
    frag bfr[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      bfr[mi] = *reinterpret_cast<const frag*>(atile + (mi * 16 + lrow) * AROW +
                                               (((s * 4 + lq) ^ lrow) << 4));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = M2<T>::run(afr, bfr[mi], acc[mi]);
  }
  S2STAMP(5);
  if (a.stamps != nullptr && blockIdx.x == 0 && lane == 0) a.stamps[32 + wave] = clock64();
  __syncthreads();   // operand image and ring are idle (no LDS-DMA in flight: last wait was vmcnt(0))

  // ---- epilogue 1: bias + GELU -> staging tile [144][256] (lane owns 4 consecutive hidden units)
  {
    const int nl = wave * 16 + lq * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.b1 + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      T4 o;
      o[0] = (T)gelu_for<T>(acc[mi][0] + bv.x);
      o[1] = (T)gelu_for<T>(acc[mi][1] + bv.y);
      o[2] = (T)gelu_for<T>(acc[mi][2] + bv.z);
      o[3] = (T)gelu_for<T>(acc[mi][3] + bv.w);
      *reinterpret_cast<T4*>(smem + (mi * 16 + lrow) * OPITCH + nl * 2) = o;
    }
  }
  S2STAMP(6);
  if (a.stamps != nullptr && blockIdx.x == 0 && lane == 0) a.stamps[48 + wave] = clock64();
  __syncthreads();
  // ---- epilogue 2: whole 512-byte rows to HBM
  const long M = (long)a.B * 9;
  unsigned char* hb = reinterpret_cast<unsigned char*>(a.h);
  for (int i = tid; i < ROWS * (TN * 2 / 16); i += NT) {
    const int ml = i >> 5, ch = i & 31;
    const long m = (long)mt * ROWS + ml;
    if (m < M)
      *reinterpret_cast<uint4*>(hb + ((size_t)m * HID + n0) * 2 + ch * 16) =
          *reinterpret_cast<const uint4*>(smem + ml * OPITCH + ch * 16);
  }
  S2STAMP(7);
  if (a.stamps != nullptr && tid == 0) a.stamps[64 + 2 * blockIdx.x + 1] = wall_clock64();
}

template <typename T> int launch_s2_fc1_t(const S2Fc1Args& a, hipStream_t st) {
  auto kern = s2_fc1_kernel<T>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  const int MT = (a.B + GA - 1) / GA;
  const int grid = ((MT + 7) / 8) * 8 * 4;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

}  // namespace

// 16-bit modes, C = 256 (ConvNeXt pico / atto-like stage 2) only; the caller falls back to
// dwconv_ln + GEMM launches for other widths.
int launch_s2_fc1(int prec, const float* x, const float* dw_w, const float* dw_b, const float* ln_w,
                  const float* ln_b, const void* w1, const float* b1, void* h, int B,
                  unsigned long long* stamps, hipStream_t st) {
  if (B <= 0) return BTSBOT_OK;
  S2Fc1Args a{x, dw_w, dw_b, ln_w, ln_b, w1, b1, h, B, stamps};
  if (prec == BTSBOT_BF16) return launch_s2_fc1_t<bf16_t>(a, st);
  if (prec == BTSBOT_F16) return launch_s2_fc1_t<f16_t>(a, st);
  btsbot_set_error("s2_fc1: precision %d is not a 16-bit mode", prec);
  return BTSBOT_ERR_INVALID_ARG;
}
