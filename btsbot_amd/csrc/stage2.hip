// Stage-2 (3x3 maps, C = 256) front half of a ConvNeXt block in ONE launch (gfx950, 16-bit modes):
//
//   h = GELU( LN( dwconv7x7(x) + b_dw ) W1^T + b1 )          x: [B][9][256] f32 -> h: [B*9][1024]
//
// (timm ConvNeXtBlock.conv_dw / norm / mlp.fc1 / mlp.act, reached from
// /root/reference/btsbot/architectures.py:108,132).  Replaces dw3_ln_kernel + the fc1 GEMM launch:
// the LayerNorm output never goes to HBM, and the GEMM gets a tile shape that fits the problem.
//
// Workgroup = 16 alerts (144 pixel rows) x 256 hidden units, 4 waves, one workgroup per CU:
// at B = 1024 that is exactly 64 x 4 = 256 workgroups.  The four workgroups that share a row tile
// are given the same blockIdx % 8, i.e. the same XCD / L2 (speed only).
//   prologue  W1 slabs 0..3 start by LDS-DMA; each wave takes 4 of the alerts: lane = 4 channels
//             (16-byte loads), the 3x3 map of a channel lives in registers, LayerNorm sums are
//             wave-local; the 16-bit result is the MFMA operand image in LDS
//             ([144][256], 16-byte chunk c of row r at position c ^ (r & 15));
//   main      K = 256 in 8 slabs of 32 through a 4-slot LDS-DMA ring (64-byte rows, chunk c of row r
//             at position c ^ F[(r >> 2) & 3], F = {0,3,2,1}: conflict-free ds_read_b128);
//             v_mfma_f32_16x16x32, wave = 144 rows x 64 columns = 36 accumulator tiles, filters
//             as the A operand so a lane ends up with 4 consecutive hidden units of one pixel;
//   epilogue  + bias, GELU, 16-bit, staged through LDS, whole 512-byte rows to HBM.
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
constexpr int AROW = C * 2;                                        // 512 B per operand-image row
constexpr int ATILE = ROWS * AROW;                                 // 73728
constexpr int KS = 32, NSLAB = C / KS, NSLOT = 4;
constexpr int SLABB = TN * KS * 2;                                 // 16384
constexpr int OPITCH = TN * 2 + 16;                                // epilogue staging pitch
constexpr int MI = ROWS / 16, NI = 4;
constexpr float LN_EPS = 1e-6f;
#define S2STAMP(i)                                                                         \
  do {                                                                                     \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)
constexpr size_t LDS_BYTES = (size_t)ATILE + NSLOT * SLABB;       // 139264
static_assert((size_t)ROWS * OPITCH <= LDS_BYTES, "epilogue staging must fit");

__device__ __forceinline__ int swz4(int row) {   // F[(row >> 2) & 3], F = {0,3,2,1}
  const int g = (row >> 2) & 3;
  return (4 - g) & 3;
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z),
                     fmaf(a.w, b.w, c.w));
}

template <typename T>
__global__ __launch_bounds__(256) void s2_fc1_kernel(S2Fc1Args a) {
  using frag = typename M2<T>::frag;
  typedef T T4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* atile = smem;
  unsigned char* ring = smem + ATILE;

  // ---- which tile: blocks with equal (blockIdx % 8) share an XCD; keep a row tile's 4 column
  //      tiles there so the stage input is fetched into one L2 only
  const int MT = (a.B + GA - 1) / GA;
  const int xcd = blockIdx.x & 7, id = blockIdx.x >> 3;
  const int nt = id & 3, mt = (id >> 2) * 8 + xcd;
  if (mt >= MT) return;                       // whole workgroup, before any barrier
  const int n0 = nt * TN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lq = lane >> 4;

  // ---- W1 slabs: per-lane source pointers (slab s adds 64 bytes), wave-uniform LDS offsets
  const unsigned char* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + (lane >> 2);
    wsrc[i] = reinterpret_cast<const unsigned char*>(a.w1) + (size_t)(n0 + row) * AROW +
              (((lane & 3) ^ swz4(row)) << 4);
  }
  auto issue = [&](int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + s * (KS * 2)),
                                       (lptr_t)(ring + (s % NSLOT) * SLABB + (wave * 4 + i) * 1024),
                                       16, 0, 0);
  };
  S2STAMP(0);

  // ---- prologue: depthwise 7x7 (only the central 5x5 taps can touch a 3x3 map) + LN
  {
    const int c4 = 4 * lane;
    float4 in[4][9];   // all four alerts of this wave in flight at once (36 x 16-byte loads per lane)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ag = min(mt * GA + wave + 4 * it, a.B - 1);
      const float* src = a.x + (size_t)ag * 9 * C + c4;
#pragma unroll
      for (int p = 0; p < 9; ++p) in[it][p] = *reinterpret_cast<const float4*>(src + p * C);
    }
    float4 wq[25];
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx)
        wq[(dy + 2) * 5 + dx + 2] =
            *reinterpret_cast<const float4*>(a.dw_w + ((dy + 3) * 7 + dx + 3) * C + c4);
    const float4 bias = *reinterpret_cast<const float4*>(a.dw_b + c4);
    const float4 g = *reinterpret_cast<const float4*>(a.ln_w + c4);
    const float4 bb = *reinterpret_cast<const float4*>(a.ln_b + c4);
    S2STAMP(1);
    // the filter slabs queue up behind the prologue's own loads (vmcnt retires in order)
    issue(0);
    issue(1);
    issue(2);
    S2STAMP(2);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int al = wave + 4 * it;
      const float4* v = in[it];
      float4 acc[9];
#pragma unroll
      for (int p = 0; p < 9; ++p) acc[p] = bias;
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx)
#pragma unroll
          for (int oy = 0; oy < 3; ++oy)
#pragma unroll
            for (int ox = 0; ox < 3; ++ox) {
              const int iy = oy + dy, ix = ox + dx;
              if (iy >= 0 && iy < 3 && ix >= 0 && ix < 3)
                acc[oy * 3 + ox] = f4fma(v[iy * 3 + ix], wq[(dy + 2) * 5 + dx + 2], acc[oy * 3 + ox]);
            }
      float mean[9], var[9];
#pragma unroll
      for (int p = 0; p < 9; ++p)
        mean[p] = wave_sum((acc[p].x + acc[p].y) + (acc[p].z + acc[p].w)) * (1.0f / C);
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        acc[p] = make_float4(acc[p].x - mean[p], acc[p].y - mean[p], acc[p].z - mean[p],
                             acc[p].w - mean[p]);
        var[p] = wave_sum((acc[p].x * acc[p].x + acc[p].y * acc[p].y) +
                          (acc[p].z * acc[p].z + acc[p].w * acc[p].w)) * (1.0f / C);
      }
#pragma unroll
      for (int p = 0; p < 9; ++p) {
        const float4 d = acc[p];
        const float rstd = rsqrtf(var[p] + LN_EPS);
        T4 o;
        o[0] = (T)(d.x * rstd * g.x + bb.x);
        o[1] = (T)(d.y * rstd * g.y + bb.y);
        o[2] = (T)(d.z * rstd * g.z + bb.z);
        o[3] = (T)(d.w * rstd * g.w + bb.w);
        const int r = al * 9 + p;
        *reinterpret_cast<T4*>(atile + r * AROW + (((lane >> 1) ^ (r & 15)) << 4) + (lane & 1) * 8) = o;
      }
      S2STAMP(3 + it);
    }
  }

  // ---- main loop: one 16x16x32 k-step per slab
  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // the operand image is complete
  // A wave multiplies only the 64 filter rows it fetched itself, so inside the loop it waits for
  // its own LDS-DMA only: no barriers, the waves drift apart.  The pixel operand of slab s+1 is
  // read from the (static) image while the MFMAs of slab s run.
  frag bfr[2][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    bfr[0][mi] = *reinterpret_cast<const frag*>(atile + (mi * 16 + lrow) * AROW + ((lq ^ lrow) << 4));
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    // slab s has landed once only the younger slabs' pieces are outstanding
    if (s + 2 < NSLAB) wait_vm<8>();
    else if (s + 1 < NSLAB) wait_vm<4>();
    else wait_vm<0>();
    const unsigned char* ws = ring + (s % NSLOT) * SLABB;
    frag afr[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = wave * 64 + ni * 16 + lrow;
      afr[ni] = *reinterpret_cast<const frag*>(ws + row * (KS * 2) + ((lq ^ swz4(row)) << 4));
    }
    if (s + 3 < NSLAB) issue(s + 3);   // into the slot of slab s-1, which this wave has consumed
    if (s + 1 < NSLAB) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        bfr[(s + 1) & 1][mi] = *reinterpret_cast<const frag*>(
            atile + (mi * 16 + lrow) * AROW + ((((s + 1) * 4 + lq) ^ lrow) << 4));
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni][mi] = M2<T>::run(afr[ni], bfr[s & 1][mi], acc[ni][mi]);
  }
  S2STAMP(7);
  __syncthreads();   // operand image and ring are idle (no LDS-DMA in flight: last wait was vmcnt(0))

  // ---- epilogue 1: bias + GELU -> staging tile [144][256] (lane owns 4 consecutive hidden units)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int nl = wave * 64 + ni * 16 + lq * 4;
    const float4 bv = *reinterpret_cast<const float4*>(a.b1 + n0 + nl);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const f32x4 v = acc[ni][mi];
      T4 o;
      o[0] = (T)gelu_fast(v[0] + bv.x);
      o[1] = (T)gelu_fast(v[1] + bv.y);
      o[2] = (T)gelu_fast(v[2] + bv.z);
      o[3] = (T)gelu_fast(v[3] + bv.w);
      *reinterpret_cast<T4*>(smem + (mi * 16 + lrow) * OPITCH + nl * 2) = o;
    }
  }
  S2STAMP(8);
  __syncthreads();
  // ---- epilogue 2: whole 512-byte rows to HBM
  const long M = (long)a.B * 9;
  unsigned char* hb = reinterpret_cast<unsigned char*>(a.h);
  for (int i = tid; i < ROWS * (TN * 2 / 16); i += 256) {
    const int ml = i >> 5, ch = i & 31;
    const long m = (long)mt * ROWS + ml;
    if (m < M)
      *reinterpret_cast<uint4*>(hb + ((size_t)m * HID + n0) * 2 + ch * 16) =
          *reinterpret_cast<const uint4*>(smem + ml * OPITCH + ch * 16);
  }
  S2STAMP(9);
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
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, st, a);
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
