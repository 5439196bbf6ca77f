// Fused classifier heads on the matrix pipe (gfx950, the 16-bit modes): optional head LayerNorm on the 1x1 image
// feature, metadata branch (BatchNorm1d folded to scale/shift -> Linear -> act -> Linear [-> act]), concat (image
// first, then metadata), fusion MLP, logits + sigmoid scores -- one launch, 16 alerts per workgroup, nothing but the
// logits leaves the CU.  Same wirings as head.hip (which stays the fp32 mode's head):
// /root/reference/btsbot/architectures.py:146-171 (mm_ConvNeXt, GELU), :109-122 (ConvNeXt head), :282-293 (um_nn,
// ReLU), :299-313,358-372 (frozen_fusion); sigmoid: inference_example.py:91.
//
// The head is a chain of small dependent layers -- latency, not FLOPs -- so it keeps fp32-class accuracy at three
// MFMAs per k-step: activations and filters are split into a 16-bit head and a 16-bit remainder,
//     a w  ~  a_hi w_hi + a_lo w_hi + a_hi w_lo          (what is dropped is below 2^-16 of the product, bf16),
// which keeps the head out of the precision mode's error budget.  Layout per layer: filters packed as
// 16x16x32 A fragments [hi | lo][row tile of 16][k-step][lane][8] (zero padded), activations [alert][k] 16-bit rows in
// LDS as the B operand; a wave owns a row tile (and, where a layer has fewer tiles than the workgroup has waves, a
// slice of K), its accumulator holds 4 consecutive outputs of the lane's alert, written straight into the next
// layer's rows.  16 alerts per workgroup rather than 32: the chain's length is instructions per wave, not FLOPs.
// (Single f16 filters with split activations were tried: frozen_fusion's unnormalised features put the bf16 mode's
//  score error at 3.3e-3 against 2e-3 with the filters' remainder kept.)
#include "common.h"
#include "head16.h"

namespace {


template <typename T> struct HM;
template <> struct HM<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct HM<f16_t> {
  using frag = f16x8;
  static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

#define H_STAMP(i)                                                                         \
  do {                                                                                     \
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = clock64(); \
  } while (0)

constexpr int HA = 16;            // alerts per workgroup (the column block of the 16x16x32 MFMA)
constexpr int HNT = 512, HNW = HNT / 64;
constexpr float HN_EPS = 1e-6f;

__host__ __device__ constexpr int rup(int v, int m) { return (v + m - 1) / m * m; }
constexpr int KGRAN = 2;          // every layer's K is padded to a multiple of 2 k-steps of 32 (64 inputs)
__host__ __device__ constexpr int kpad(int K) { return rup(K, 32 * KGRAN); }
__host__ __device__ constexpr size_t head16_bytes(int N, int K) { return (size_t)2 * rup(N, 16) * kpad(K) * 2; }

// an activation buffer in LDS: hi rows then lo rows, `pitch` bytes per alert row
struct Rows {
  unsigned char* hi;
  int pitch;
  __device__ __forceinline__ unsigned char* lo() const { return hi + HA * pitch; }
};

template <typename T> __device__ __forceinline__ void split_store(unsigned char* hi, unsigned char* lo, float v) {
  const T h = (T)v;
  *reinterpret_cast<T*>(hi) = h;
  *reinterpret_cast<T*>(lo) = (T)(v - (float)h);
}

// GELU here = gelu_poly<5> (|error| < 5e-7, common.h): erff costs ~45 instructions per value, ~3k cycles per layer
__device__ __forceinline__ float head_act(float x, int act) {
  if (act == ACT_GELU) return gelu_poly<5>(x);
  if (act == ACT_RELU) return relu_f(x);
  return x;
}

// out[alert][col0 + n] = act(bias[n] + sum_k in[alert][k] W[n][k]),  n < Np (rows past N come out as act(0) = 0).
// Jobs: (row tile of 16 outputs, K part).  With fewer row tiles than waves the k-steps of a tile are split over 2, 4
// or 8 waves (a layer is a chain of dependent MFMAs: 3 per k-step on one accumulator) and the parts meet in LDS.
// K is padded to KGRAN k-steps with zero filters / zero activations.
template <typename T>
__device__ __forceinline__ void dense16(const Rows& in, const H16Layer& L, const Rows& out, int col0, float* red, int red_slots,
                                        float* logits, float* scores, int b0, int B, unsigned long long* stamps) {
#define D_STAMP(i) do { if (stamps != nullptr && col0 != 0 && blockIdx.x == 0 && threadIdx.x == 0) stamps[i] = clock64(); } while (0)
  D_STAMP(10);
  using frag = typename HM<T>::frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, q = lane >> 4;
  const int KS = kpad(L.K) / 32, tiles = rup(L.N, 16) / 16;
  int split = 1;
  while (split * 2 * tiles <= HNW && KS % (split * 2 * KGRAN) == 0 && tiles * (split * 2 - 1) <= red_slots) split *= 2;
  const int rt = wave % tiles, part = wave / tiles;      // (tiles <= HNW: head16_supported)
  const bool job = part < split;
  const int ksp = KS / split, k0 = (job ? part : 0) * ksp;
  f32x4 acc;
  if (job) {
    // B fragment of k-step ks: lane (alert lc, quarter q) holds k = 32 ks + 8 q + 0..7
    const unsigned char* bh = in.hi + lc * in.pitch + q * 16;
    const unsigned char* bl = bh + HA * in.pitch;
    const frag* whi = reinterpret_cast<const frag*>(L.w) + (size_t)rt * KS * 64 + lane;
    const frag* wlo = whi + (size_t)tiles * KS * 64;
    // A part is G groups of KGRAN k-steps; the next group's filter fragments are requested before this group's
    // products (a loop with one group in flight: a head layer is short, and the code has to stay short too -- this
    // kernel runs every instruction once per workgroup, so a fully unrolled version spent its time in instruction
    // fetch: 58 KB of code streamed at ~1 byte per cycle; a ring of four groups was no faster and its clamped
    // re-reads cost the short layers ~1.4k cycles each).  Three independent accumulators (hi.hi, lo.hi, hi.lo) and
    // the B fragments of the group's k-steps requested at once: on one accumulator with the LDS reads in line a
    // k-step is a chain of two LDS round trips and three dependent MFMAs.
    const int G = ksp / KGRAN;
    frag ah[KGRAN], al[KGRAN];
#pragma unroll
    for (int j = 0; j < KGRAN; ++j) {
      ah[j] = whi[(k0 + j) * 64];
      al[j] = wlo[(k0 + j) * 64];
    }
    // register r of a lane = output 16 rt + 4 q + r of alert lc
    const int n0 = 16 * rt + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = L.bias[n0 + r < L.N ? n0 + r : L.N - 1];   // (unconditional load; masked below)
    __builtin_amdgcn_sched_barrier(0);   // one round trip for the first group and the bias
    f32x4 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (part != 0 || n0 + r >= L.N) acc[r] = 0.f;
      acc1[r] = acc2[r] = 0.f;
    }
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      const int kg = k0 + KGRAN * g, kn = g + 1 < G ? kg + KGRAN : kg;   // (the last refill re-reads its own group)
      frag nh[KGRAN], nl[KGRAN], xh[KGRAN], xl[KGRAN];
#pragma unroll
      for (int j = 0; j < KGRAN; ++j) {
        nh[j] = whi[(kn + j) * 64];
        nl[j] = wlo[(kn + j) * 64];
      }
#pragma unroll
      for (int j = 0; j < KGRAN; ++j) {
        xh[j] = *reinterpret_cast<const frag*>(bh + (kg + j) * 64);
        xl[j] = *reinterpret_cast<const frag*>(bl + (kg + j) * 64);
      }
#pragma unroll
      for (int j = 0; j < KGRAN; ++j) {
        acc = HM<T>::run(ah[j], xh[j], acc);
        acc1 = HM<T>::run(al[j], xh[j], acc1);
        acc2 = HM<T>::run(ah[j], xl[j], acc2);
      }
#pragma unroll
      for (int j = 0; j < KGRAN; ++j) {
        ah[j] = nh[j];
        al[j] = nl[j];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc1[r] + acc2[r];
    D_STAMP(12);
    if (part != 0) {
      float* dst = red + ((size_t)(rt * (split - 1) + part - 1) * 4) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * 64] = acc[r];
    }
  }
  if (split > 1) __syncthreads();
  D_STAMP(13);
  if (job && part == 0) {
    for (int p = 1; p < split; ++p) {
      const float* src = red + ((size_t)(rt * (split - 1) + p - 1) * 4) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += src[r * 64];
    }
    if (logits != nullptr) {   // the last layer: output 0 of tile 0 = register 0 of the lanes with q = 0
      if (rt == 0 && q == 0 && b0 + lc < B) {
        const float zz = acc[0];
        logits[b0 + lc] = zz;
        if (scores != nullptr) scores[b0 + lc] = 1.0f / (1.0f + expf(-zz));
      }
    } else {
      typedef T __attribute__((ext_vector_type(4))) T4;
      T4 oh, ol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = head_act(acc[r], L.act);
        const T hv = (T)v;
        oh[r] = hv;
        ol[r] = (T)(v - (float)hv);
      }
      const int off = lc * out.pitch + (col0 + 16 * rt + 4 * q) * 2;
      *reinterpret_cast<T4*>(out.hi + off) = oh;
      *reinterpret_cast<T4*>(out.lo() + off) = ol;
    }
  }
  D_STAMP(14);
  // columns between the padded row tiles and the next layer's padded K read as zero
  if (logits == nullptr) {
    const int c0 = col0 + 16 * tiles, c1 = col0 + kpad(L.N);
    for (int i = threadIdx.x; i < HA * (c1 - c0); i += HNT) {
      const int g = i / (c1 - c0), c = c0 + i % (c1 - c0);
      split_store<T>(out.hi + g * out.pitch + c * 2, out.lo() + g * out.pitch + c * 2, 0.f);
    }
  }
}

template <typename T> __global__ __launch_bounds__(HNT) void head16_kernel(Head16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * HA;
  const Rows z{smem, a.pitch_z};
  const Rows t0{smem + 2 * HA * a.pitch_z, a.pitch_t};
  const Rows t1{t0.hi + 2 * HA * a.pitch_t, a.pitch_t};
  // K-split partial tiles (4 KB each) go to whichever of the three buffers the layer neither reads nor writes
  const int zcols = (a.pitch_z - 16) / 2;
  // ---- image feature (+ head LayerNorm) -> z[:, 0:feat_dim]; wave = 4 of the 32 rows, lane = channels 64 i + lane.
  H_STAMP(0);
  //      The row loop is not unrolled (code size, see dense16); the next row is requested before this one is normalised.
  constexpr int RPW = HA / HNW;
  if (a.feat_dim > 0) {
    float nx[12];   // feat_dim <= 768
    auto request = [&](int u) {
      const int b = b0 + wave + HNW * u;
      const float* src = a.feat + (size_t)(b < a.B ? b : a.B - 1) * a.feat_dim;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int c = lane + 64 * i;
        nx[i] = src[c < a.feat_dim ? c : a.feat_dim - 1];   // (unconditional load, see dense16; masked below)
      }
    };
    request(0);
    float hw[12], hb[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int c = lane + 64 * i, cc = c < a.feat_dim ? c : a.feat_dim - 1;
      hw[i] = a.hn_w != nullptr ? a.hn_w[cc] : 1.f;
      hb[i] = a.hn_b != nullptr ? a.hn_b[cc] : 0.f;
    }
    H_STAMP(1);
#pragma unroll 1
    for (int u = 0; u < RPW; ++u) {
      const int g = wave + HNW * u;
      float v[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] = lane + 64 * i < a.feat_dim ? nx[i] : 0.f;
      request(u + 1 < RPW ? u + 1 : u);
      if (a.hn_w != nullptr) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) sum += v[i];
        const float mean = wave_sum(sum) / a.feat_dim;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const float d = lane + 64 * i < a.feat_dim ? v[i] - mean : 0.f;
          sq += d * d;
        }
        const float rstd = rsqrtf(wave_sum(sq) / a.feat_dim + HN_EPS);
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = (v[i] - mean) * rstd * hw[i] + hb[i];
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int c = lane + 64 * i;
        if (c < a.feat_dim) split_store<T>(z.hi + g * z.pitch + c * 2, z.lo() + g * z.pitch + c * 2, v[i]);
      }
    }
  }
  H_STAMP(2);
  // (columns of z past the concat width are the fusion layer's zero padding)
  for (int i = tid; i < HA * (zcols - a.zwidth); i += HNT) {
    const int g = i / (zcols - a.zwidth), c = a.zwidth + i % (zcols - a.zwidth);
    split_store<T>(z.hi + g * z.pitch + c * 2, z.lo() + g * z.pitch + c * 2, 0.f);
  }
  // ---- metadata branch -> z[:, feat_dim : feat_dim + f2]
  if (a.n_meta > 0) {
    // (n_meta <= 32: thread e is column e % 32 of alert e / 32)
    const int mk = kpad(a.n_meta);
    constexpr int EPT = HA * 32 / HNT;
    float mv[EPT], sc[EPT], sh[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = tid + HNT * i, g = e >> 5, j = e & 31;
      const int b = b0 + g, jc = j < a.n_meta ? j : a.n_meta - 1;
      mv[i] = a.meta[(size_t)(b < a.B ? b : a.B - 1) * a.n_meta + jc];
      sc[i] = a.bn_scale[jc];
      sh[i] = a.bn_shift[jc];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = tid + HNT * i, g = e >> 5, j = e & 31;
      split_store<T>(t0.hi + g * t0.pitch + j * 2, t0.lo() + g * t0.pitch + j * 2, j < a.n_meta ? fmaf(mv[i], sc[i], sh[i]) : 0.f);
    }
    for (int i = tid; i < HA * (mk - 32); i += HNT) {
      const int g = i / (mk - 32), j = 32 + i % (mk - 32);
      split_store<T>(t0.hi + g * t0.pitch + j * 2, t0.lo() + g * t0.pitch + j * 2, 0.f);
    }
  }
  __syncthreads();
  H_STAMP(3);
  // ---- the layers, one after the other through ONE copy of the layer code (a loop, not five inlined copies and not
  //      calls: this kernel runs every instruction once per workgroup, so unrolled code is paid in instruction fetch,
  //      and a real call spills its callee-saved registers to scratch memory on both sides)
  auto buf = [&](int id) { return Rows{id == 0 ? z.hi : id == 1 ? t0.hi : t1.hi, id == 0 ? z.pitch : t0.pitch}; };
#pragma unroll 1
  for (int si = 0; si < a.n_steps; ++si) {
    const H16Step st = a.steps[si];
    const bool last = si + 1 == a.n_steps;
    // K-split partial tiles (1 KB each) go to whichever of the three buffers the layer neither reads nor writes
    const Rows rb = buf(st.red_buf);
    dense16<T>(buf(st.in_buf), st.L, buf(st.out_buf), st.col0, reinterpret_cast<float*>(rb.hi),
               st.red_buf >= 0 ? 2 * HA * rb.pitch / 1024 : 0, last ? a.logits : nullptr, a.scores, b0, a.B, a.stamps);
    __syncthreads();
    H_STAMP(4 + si);
  }
}

// fp32 [N][K] -> [hi | lo][row tile of 16][k-step of 32][lane][8] A fragments of the 16x16x32 MFMA (lane l: row l & 15,
// k = 8 (l >> 4) + 0..7), zero padded
template <typename T>
__global__ void pack_h16_kernel(const float* __restrict__ w, T* __restrict__ out, int N, int K) {
  const int Kp = kpad(K), Np = rup(N, 16);
  const long half = (long)Np * Kp;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
  const long fs = i >> 9;
  const int ksteps = Kp / 32;
  const int s = (int)(fs % ksteps), tile = (int)(fs / ksteps);
  const int row = 16 * tile + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
  const float v = (row < N && k < K) ? w[(long)row * K + k] : 0.f;
  const T hv = (T)v;
  out[i] = hv;
  out[half + i] = (T)(v - (float)hv);
}

}  // namespace

size_t head16_packed_bytes(int N, int K) { return head16_bytes(N, K); }

bool head16_supported(int prec, int feat_dim, int n_meta, int f1, int f2, int n_layers, const int* dims) {
  if (prec != BTSBOT_BF16 && prec != BTSBOT_F16) return false;
  if (feat_dim > 768 || (feat_dim & 7) != 0 || n_layers < 1 || n_layers > 3) return false;
  if (n_meta > 0 && (f1 > 16 * HNW || f2 > 16 * HNW || n_meta > 32)) return false;
  for (int i = 1; i <= n_layers; ++i)
    if (dims[i] > 16 * HNW) return false;
  return dims[n_layers] == 1;
}

int launch_pack_h16(int prec, const float* w, void* dst, int N, int K, hipStream_t st) {
  const long half = (long)rup(N, 16) * kpad(K);
  const dim3 grid((unsigned)((half + 255) / 256)), blk(256);
  if (prec == BTSBOT_BF16)
    hipLaunchKernelGGL(pack_h16_kernel<bf16_t>, grid, blk, 0, st, w, reinterpret_cast<bf16_t*>(dst), N, K);
  else if (prec == BTSBOT_F16)
    hipLaunchKernelGGL(pack_h16_kernel<f16_t>, grid, blk, 0, st, w, reinterpret_cast<f16_t*>(dst), N, K);
  else {
    btsbot_set_error("pack_h16: precision %d is not a 16-bit mode", prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_head16(int prec, const Head16Args& a0, hipStream_t st) {
  if (a0.B <= 0) return BTSBOT_OK;
  Head16Args a = a0;
  // LDS rows: z = concat width (what the first fusion layer reads, its k padding included), t0 / t1 = the widest of
  // the other layers; every row 16 bytes longer than its data (8 rows then cover the 32 banks)
  a.zwidth = a.feat_dim + (a.n_meta > 0 ? a.m2.N : 0);
  int zc = kpad(a.zwidth);
  if (a.n_meta > 0 && a.feat_dim + kpad(a.m2.N) > zc) zc = a.feat_dim + kpad(a.m2.N);
  int tw = a.n_meta > 0 ? kpad(a.n_meta) : 0;
  if (a.n_meta > 0) tw = tw > kpad(a.m1.N) ? tw : kpad(a.m1.N);
  for (int i = 0; i < a.n_layers; ++i) tw = tw > kpad(a.comb[i].N) ? tw : kpad(a.comb[i].N);
  a.n_steps = 0;
  if (a.n_meta > 0) {
    a.steps[a.n_steps++] = H16Step{a.m1, 1, 2, 0, -1};
    a.steps[a.n_steps++] = H16Step{a.m2, 2, 0, a.feat_dim, 1};
  }
  for (int i = 0; i < a.n_layers; ++i)
    a.steps[a.n_steps++] = H16Step{a.comb[i], i == 0 ? 0 : ((i - 1) & 1) ? 2 : 1, (i & 1) ? 2 : 1, 0, i == 0 ? 2 : 0};
  a.pitch_z = zc * 2 + 16;
  a.pitch_t = tw * 2 + 16;
  const size_t lds = (size_t)2 * HA * a.pitch_z + (size_t)4 * HA * a.pitch_t;
  if (lds > 150 * 1024) {
    btsbot_set_error("head16: layer widths too large for one workgroup (%zu bytes of LDS)", lds);
    return BTSBOT_ERR_INVALID_ARG;
  }
  static size_t lds_attr[2] = {0, 0};
  const int ti = prec == BTSBOT_BF16 ? 0 : 1;
  const void* kern = ti == 0 ? reinterpret_cast<const void*>(head16_kernel<bf16_t>)
                             : reinterpret_cast<const void*>(head16_kernel<f16_t>);
  if (lds > lds_attr[ti]) {
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr[ti] = lds;
  }
  const dim3 grid((a.B + HA - 1) / HA), blk(HNT);
  if (ti == 0)
    hipLaunchKernelGGL(head16_kernel<bf16_t>, grid, blk, lds, st, a);
  else
    hipLaunchKernelGGL(head16_kernel<f16_t>, grid, blk, lds, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
