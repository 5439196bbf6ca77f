// Input-gradient half of a 256-channel block's MLP backward as ONE launch in stage2p.hip's shape (gfx950, bf16 / f16):
//
//     da  [R][1024] = ((gamma W2)^T dy) * gelu'(a)          operand type: the filter-gradient GEMMs read it (wgrad.hip)
//     dxn [R][256]  = W1^T da                               fp32: the LayerNorm / depthwise backward reads it
//
// (timm ConvNeXtBlock.mlp of stages[2], reached from /root/reference/btsbot/architectures.py:108,132; what autograd
// does for it at /root/reference/btsbot/train.py:526).  Was two launches of the tiled GEMM (gemm2.hip, DGELU and PLAIN
// epilogues): 9216 rows are 576 tiles of 128 x 128 resp. 64 x 64 -- 1.1 rounds of the chip's 512 workgroup slots, the
// second round a ninth full -- and da made a round trip through HBM between them: 26-28 + 17-28 us per block, six
// blocks per step in the backward's chain.  Here a workgroup keeps RW = 45 pixel rows (three 16-column MFMA blocks)
// resident, streams both filters past them as packed MFMA A fragments (1 KiB contiguous per wave instruction,
// straight into registers: stage2p.hip) and never leaves the CU between the two products:
//   step ch (128 hidden units):  t = (gamma W2)^T[chunk] . dy          (hidden tile 8 ch + wave, K = 256)
//                                da = t * gelu'(a[chunk])               a: 8-byte loads requested at the step's start
//                                dxn += W1^T[:, chunk - 1] . da[chunk - 1]   (one step behind, through a double-buffered
//                                                                           [pixel][hidden] LDS image: one barrier per step)
// 205 workgroups for 9216 rows: every CU's share is the same and one round.
// 30-35 us per launch against 44 for the two GEMMs; 15 us of a 2.57 ms step (the step is bound by the
// chip's total work, not by this chain).  BTSBOT_AMD_NO_S2MLP=1: the two GEMMs (tests/test_gpu_train.py::
// test_full_backward_16bit[*-stage2_two_gemms]; the default cases run this kernel in both 16-bit modes).
#include <stdlib.h>

#include "common.h"

namespace {

template <typename T> struct SM;
template <> struct SM<bf16_t> {
  typedef bf16_t frag __attribute__((ext_vector_type(8)));
  typedef bf16_t quad __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct SM<f16_t> {
  typedef f16_t frag __attribute__((ext_vector_type(8)));
  typedef f16_t quad __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// da leaves through buffer stores (scalar descriptor, 32-bit lane offset, scalar chunk offset), as whole 256-byte row pieces
// out of its LDS image one step behind: stage2p.hip's keeping form says why (scattered quads: 64 separate 8-byte writes per
// wave instruction; conditional copy loops: vmcnt(0) in front of the fragment streams)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t s2_rsrc(void* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ void s2_st16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, uint4 q) {
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, q), r, (int)voff, (int)soff, 0);
}

constexpr int C = 256, HID = 1024, NT = 512, NW = 8;
constexpr int CHUNK = 128, NCHUNK = HID / CHUNK;
constexpr int KS1 = C / 32, KS2 = CHUNK / 32, KH = HID / 32;   // k-steps: first product (K = 256), second per chunk, second in all
constexpr int NB = 3, NCOL = 16 * NB;                          // 48 columns, RW of them live
// operand images [pixel][k]: rows 32 bytes (mod 256) apart (stage2p.hip: every 16-lane group of a ds_read_b128 then
// covers the 16 slots of a 256-byte bank row)
constexpr int YP = C * 2 + 32;        // dy image: bytes per pixel row (544)
constexpr int HP = CHUNK * 2 + 32;    // da image: bytes per pixel row (288)
constexpr int OFF_Y = 0, OFF_H = NCOL * YP, H_IMG = NCOL * HP, LDS_BYTES = OFF_H + 2 * H_IMG;   // 26112 + 2 x 13824 = 53760

struct S2MlpBwdArgs {
  const void* dy;     // [R][256] operand type
  const void* a;      // [R][1024] operand type: the fc1 pre-activation the forward kept
  const void* w2tp;   // (gamma W2)^T as A fragments [hidden tile 64][k-step 8][lane 64][8]
  const void* w1tp;   // W1^T as A fragments [channel tile 16][k-step 32][lane 64][8]
  void* da;           // [R + 48][1024] operand type (the last workgroup stores its dead rows -- zeros -- too)
  float* dxn;         // [R][256] fp32
  int R, rw;          // rows; rows per workgroup (<= 48)
};

template <typename T>
__global__ __launch_bounds__(NT, 2) void s2mlp_bwd_kernel(S2MlpBwdArgs a) {
  using frag = typename SM<T>::frag;
  using quad = typename SM<T>::quad;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* yi = smem + OFF_Y;
  unsigned char* hb = smem + OFF_H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int r0 = blockIdx.x * a.rw;
  const int nlive = min(a.rw, a.R - r0);
  const frag* w2f = reinterpret_cast<const frag*>(a.w2tp) + lane;
  const frag* w1f = reinterpret_cast<const frag*>(a.w1tp) + lane;

  // chunk 0 of both filters (the second product runs one step behind: its chunk 0 is first used in step 1)
  frag a1[KS1], a2[2][KS2];
#pragma unroll
  for (int s = 0; s < KS1; ++s) a1[s] = w2f[(size_t)(wave * KS1 + s) * 64];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int s = 0; s < KS2; ++s) a2[m][s] = w1f[(size_t)((2 * wave + m) * KH + s) * 64];

  // dy rows -> LDS image (16-byte pieces: 32 per row); rows beyond the live ones read as zero
  {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.dy) + (size_t)r0 * C * 2;
    for (int i = tid; i < NCOL * 32; i += NT) {
      const int p = i >> 5, c16 = i & 31;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (p < nlive) v = *reinterpret_cast<const uint4*>(src + (size_t)p * C * 2 + 16 * c16);
      *reinterpret_cast<uint4*>(yi + p * YP + 16 * c16) = v;
    }
  }
  // this lane's pre-activation / da pieces: pixel 16 n + col, hidden 128 ch + 16 wave + 4 kg + 0..3 (8 bytes of 16 different
  // rows per wave instruction.  Routed through LDS images as whole 256-byte row pieces instead -- pre-activations parked a
  // step ahead, da copied out of its image a step behind -- the kernel took 36-40 us instead of 30-34: measured, not kept)
  const T* ap = reinterpret_cast<const T*>(a.a) + 16 * wave + 4 * kg;
  const __amdgpu_buffer_rsrc_t rda = s2_rsrc(a.da);
  // chunk kc's da image -> columns kc of da [R][1024]: fixed trips per thread, the last pieces written twice
  auto da_out = [&](int kc, int img) {
    constexpr int PPR = CHUNK * 2 / 16, TRIPS = (NCOL * PPR + NT - 1) / NT;   // (rw <= 48 rows x 16 pieces: two trips)
    const int npc = a.rw * PPR;
    const unsigned so = (unsigned)(((size_t)r0 * HID + (size_t)kc * CHUNK) * 2);
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = min(tid + it * NT, npc - 1);
      const int pr = i / PPR, c = i - pr * PPR;
      s2_st16(rda, (unsigned)(pr * HID * 2 + 16 * c), so, *reinterpret_cast<const uint4*>(hb + img * H_IMG + pr * HP + 16 * c));
    }
  };
  size_t rowoff[NB];
  bool rlive[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    rlive[n] = 16 * n + col < nlive;
    rowoff[n] = (size_t)(r0 + (rlive[n] ? 16 * n + col : 0)) * HID;
  }
  f32x4 acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the pre-activations are requested a whole step ahead of their use: a wait for them then never includes the da
  // stores of the step in between (vmcnt retires in order: a load issued BEHIND a store waits for the store's round trip)
  quad aqn[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) aqn[n] = *reinterpret_cast<const quad*>(ap + rowoff[n]);
  __syncthreads();

  auto step = [&](int p, bool first, int ch) {
    quad aq[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      aq[n] = aqn[n];
      aqn[n] = *reinterpret_cast<const quad*>(ap + rowoff[n] + (ch + 1 < NCHUNK ? ch + 1 : 0) * CHUNK);
    }
    // t = (gamma W2)^T[hidden tile 8 ch + wave] . dy: B = the dy image, two k-steps' fragments in flight
    f32x4 t[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) t[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nch = ch + 1 < NCHUNK ? ch + 1 : 0;   // (behind the last chunk: an unconditional reload of chunk 0)
    frag yb[2][NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) yb[0][n] = *reinterpret_cast<const frag*>(yi + (16 * n + col) * YP + (8 * kg) * 2);
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      if (s + 1 < KS1) {
#pragma unroll
        for (int n = 0; n < NB; ++n)
          yb[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(yi + (16 * n + col) * YP + (32 * (s + 1) + 8 * kg) * 2);
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) t[n] = SM<T>::run(a1[s], yb[s & 1][n], t[n]);
      a1[s] = w2f[(size_t)((nch * NW + wave) * KS1 + s) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!first) da_out(ch - 1, 1 - p);   // (the previous chunk's image is complete and lives through this step)
    // second product of the previous chunk (dxn += W1^T[:, chunk - 1] . da[chunk - 1]) between the GELU' pieces of this one
    unsigned char* hcur = hb + p * H_IMG;
    const unsigned char* hprev = hb + (1 - p) * H_IMG;
    frag hbf[2][NB];
    if (!first) {
#pragma unroll
      for (int n = 0; n < NB; ++n) hbf[0][n] = *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (8 * kg) * 2);
    }
    constexpr int NS2 = KS2 > NB ? KS2 : NB;
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
      if (s < KS2 && !first) {
        if (s + 1 < KS2) {
#pragma unroll
          for (int n = 0; n < NB; ++n)
            hbf[(s + 1) & 1][n] = *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (32 * (s + 1) + 8 * kg) * 2);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = SM<T>::run(a2[m][s], hbf[s & 1][n], acc[m][n]);
      }
      if (s < KS2) {   // refill with THIS chunk's fragments (used in the next step)
        a2[0][s] = w1f[(size_t)((2 * wave) * KH + ch * KS2 + s) * 64];
        a2[1][s] = w1f[(size_t)((2 * wave + 1) * KH + ch * KS2 + s) * 64];
      }
      if (s < NB) {
        quad dv;
#pragma unroll
        for (int r = 0; r < 4; ++r) dv[r] = (T)(t[s][r] * gelu_grad_for<T>((float)aq[s][r]));
        *reinterpret_cast<quad*>(hcur + (16 * s + col) * HP + (16 * wave + 4 * kg) * 2) = dv;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // image hb[p] complete; hb[1 - p] is read out
  };
  step(0, true, 0);
#pragma unroll 1
  for (int ch = 1; ch < NCHUNK; ++ch) step(ch & 1, false, ch);
  {   // the last chunk's second product (chunk NCHUNK - 1 is odd: image 1)
    static_assert(NCHUNK % 2 == 0, "the last chunk writes hidden image 1");
    const unsigned char* hprev = hb + 1 * H_IMG;
    da_out(NCHUNK - 1, 1);
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      frag hbf[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) hbf[n] = *reinterpret_cast<const frag*>(hprev + (16 * n + col) * HP + (32 * s + 8 * kg) * 2);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = SM<T>::run(a2[m][s], hbf[n], acc[m][n]);
    }
  }
  // dxn: lane holds channels 32 wave + 16 m + 4 kg + 0..3 of pixel 16 n + col
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (16 * n + col < nlive)
        *reinterpret_cast<f32x4*>(a.dxn + (size_t)(r0 + 16 * n + col) * C + 32 * wave + 16 * m + 4 * kg) = acc[m][n];
}

// 16-bit row-major [rows][K] -> MFMA A fragments [row tile][k-step][lane][8] (stage2p.hip's pack_frag_kernel on values
// that are already in the operand type: the dgrad transposes the training re-pack writes anyway)
__global__ void pack_frag16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ out, int rows, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * K) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = K / 32;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 16 * tile + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
  out[i] = src[(long)row * K + k];
}

}  // namespace

bool s2mlp_bwd_supported(int prec, int C_) { return (prec == BTSBOT_BF16 || prec == BTSBOT_F16) && C_ == C; }

int launch_pack_frag16(const void* src, void* dst, int rows, int K, hipStream_t st) {
  const long total = (long)rows * K;
  hipLaunchKernelGGL(pack_frag16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const unsigned short*>(src), reinterpret_cast<unsigned short*>(dst), rows, K);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// dy [R][256], a [R][1024] in the operand type; w2tp / w1tp: launch_pack_frag16 of (diag(gamma) W2)^T [1024][256] and
// W1^T [256][1024]; writes da [R][1024] (operand type) and dxn [R][256] (fp32)
int launch_s2mlp_bwd(int prec, const void* dy, const void* a, const void* w2tp, const void* w1tp, void* da, float* dxn, int R,
                     hipStream_t st) {
  if (R <= 0) return BTSBOT_OK;
  if (!s2mlp_bwd_supported(prec, C)) {
    btsbot_set_error("s2mlp_bwd: precision %d is not bf16 / f16", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  static const int rw = [] {
    const char* e = getenv("BTSBOT_AMD_S2MLP_ROWS");   // tuning knob: pixel rows per workgroup (<= 48)
    const int v = e ? atoi(e) : 0;
    return v >= 16 && v <= NCOL ? v : 45;
  }();
  S2MlpBwdArgs g{dy, a, w2tp, w1tp, da, dxn, R, rw};
  const int grid = (R + rw - 1) / rw;
  static DevOnce attr;
  if (attr.need()) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(s2mlp_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(s2mlp_bwd_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES));
    attr.done();
  }
  if (prec == BTSBOT_BF16) hipLaunchKernelGGL(s2mlp_bwd_kernel<bf16_t>, dim3(grid), dim3(NT), LDS_BYTES, st, g);
  else hipLaunchKernelGGL(s2mlp_bwd_kernel<f16_t>, dim3(grid), dim3(NT), LDS_BYTES, st, g);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
