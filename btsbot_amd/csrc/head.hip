// Fused classifier heads (fp32): optional head LayerNorm on the 1x1 image feature, metadata branch
// (BatchNorm1d folded to scale/shift -> Linear -> act -> Linear [-> act]), concat (image first, then
// metadata), fusion MLP, logits + sigmoid scores -- one kernel, nothing but the logits leaves the CU.
//
// Reference wiring: /root/reference/btsbot/architectures.py:146-171 (mm_ConvNeXt, GELU),
// :109-122 (ConvNeXt head), :282-293 (um_nn, ReLU), :299-313,358-372 (frozen_fusion, ReLU, metadata
// branch without its trailing activation); sigmoid: inference_example.py:91.
// Also the small parameter (re)packing kernels.
#include "common.h"

namespace {

constexpr int HG = 8;     // alerts per workgroup
constexpr int HNT = 512;   // threads (1024 would cap VGPRs at 128 and spill)
constexpr float HN_EPS = 1e-6f;
constexpr int PARTN = 2048;   // cross-slice scratch: KS * N <= PARTN rows of HG floats (64 KB)

// Activations of the workgroup's HG alerts live in LDS k-major: v[k][g] (8 alerts = two float4),
// so one thread reads all alerts of a k with two ds_read_b128 broadcasts.
// out[n][g] = act(bias[n] + sum_k in[k][g] * wt[k][n]); wt is K-major.  The head is a chain of small
// dependent layers, i.e. latency-bound, so the layout aims at few dependent memory round trips:
//   * a thread owns NPT = 4 consecutive neurons (one float4 weight load per k) of one K slice;
//     the HNT threads cover N/4 quads x KS slices, KS <= 16 so the cross-slice sum stays short;
//   * two groups of 8 k are in flight per thread (a slice of <= 16 k issues every load up front);
//   * partial sums meet in LDS (`part`), biases are fetched before the K loop.
// N not a multiple of 4 (the final N = 1 layer) takes the scalar path with the same structure.
template <int NPT>
__device__ __forceinline__ void dense_t(const float* in, int K, const float* __restrict__ wt,
                                        const float* __restrict__ bias, int N, int act, float* outp,
                                        float* part) {
  constexpr int GK = 8;
  typedef float __attribute__((ext_vector_type(NPT))) wvec;
  const int nq = N / NPT;                              // neuron groups
  int ks = 1;
  while (ks * 2 * nq <= HNT && ks * 2 * N <= PARTN && ks < 16 && K / (ks * 2) >= 8) ks *= 2;
  const int kchunk = (K + ks - 1) / ks;
  for (int q0 = 0; q0 < nq; q0 += HNT) {               // nq > HNT: several passes (ks == 1)
    const int q = q0 + (threadIdx.x % (nq < HNT ? nq : HNT));
    const int slice = nq < HNT ? threadIdx.x / nq : 0;
    const bool live = q < nq && slice < ks;
    const int fin = threadIdx.x;                       // element of [N][HG] finalised by this thread
    const float bfin = fin < HG * N ? bias[fin / HG] : 0.f;
    float acc[NPT][HG];
#pragma unroll
    for (int j = 0; j < NPT; ++j)
#pragma unroll
      for (int g = 0; g < HG; ++g) acc[j][g] = 0.f;
    if (live) {
      const int k0 = slice * kchunk, k1 = min(K, k0 + kchunk);
      const int ngrp = k1 > k0 ? (k1 - k0 + GK - 1) / GK : 0;
      const float* wp = wt + (size_t)q * NPT;
      wvec wa[GK], wb[GK];
      auto ld = [&](int kk) {
        return kk < k1 ? *reinterpret_cast<const wvec*>(wp + (size_t)kk * N) : wvec(0.f);
      };
      if (ngrp > 0) {
#pragma unroll
        for (int u = 0; u < GK; ++u) wa[u] = ld(k0 + u);
      }
      if (ngrp > 1) {
#pragma unroll
        for (int u = 0; u < GK; ++u) wb[u] = ld(k0 + GK + u);
      }
      for (int gi = 0; gi < ngrp; ++gi) {
        const int kb = k0 + gi * GK;
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int kk = min(kb + u, k1 - 1);          // padded taps carry weight 0
          const float4 a0 = *reinterpret_cast<const float4*>(in + kk * HG);
          const float4 a1 = *reinterpret_cast<const float4*>(in + kk * HG + 4);
#pragma unroll
          for (int j = 0; j < NPT; ++j) {
            const float w = wa[u][j];
            acc[j][0] = fmaf(a0.x, w, acc[j][0]); acc[j][1] = fmaf(a0.y, w, acc[j][1]);
            acc[j][2] = fmaf(a0.z, w, acc[j][2]); acc[j][3] = fmaf(a0.w, w, acc[j][3]);
            acc[j][4] = fmaf(a1.x, w, acc[j][4]); acc[j][5] = fmaf(a1.y, w, acc[j][5]);
            acc[j][6] = fmaf(a1.z, w, acc[j][6]); acc[j][7] = fmaf(a1.w, w, acc[j][7]);
          }
        }
#pragma unroll
        for (int u = 0; u < GK; ++u) wa[u] = wb[u];
        if (gi + 2 < ngrp) {
          const int kn = k0 + (gi + 2) * GK;
#pragma unroll
          for (int u = 0; u < GK; ++u) wb[u] = ld(kn + u);
        }
      }
#pragma unroll
      for (int j = 0; j < NPT; ++j)
#pragma unroll
        for (int g = 0; g < HG; ++g) part[((size_t)slice * N + q * NPT + j) * HG + g] = acc[j][g];
    }
    __syncthreads();
    for (int i = fin; i < HG * N; i += HNT) {
      float t = i == fin ? bfin : bias[i / HG];
      for (int s2 = 0; s2 < ks; ++s2) t += part[s2 * N * HG + i];
      outp[i] = apply_act(t, act);
    }
  }
}

__device__ __forceinline__ void dense(const float* in, int K, const float* __restrict__ wt,
                                      const float* __restrict__ bias, int N, int act, float* outp,
                                      float* part, bool skipk = false) {
  if (skipk) K = 0;
  if ((N & 3) == 0)
    dense_t<4>(in, K, wt, bias, N, act, outp, part);
  else
    dense_t<1>(in, K, wt, bias, N, act, outp, part);
}

__global__ __launch_bounds__(HNT) void head_kernel(HeadArgs a) {
  static_assert(HG == 8, "dense() reads the 8 alerts of a k as two float4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int zd = a.dims[0];
  int maxw = a.f1 > a.n_meta ? a.f1 : a.n_meta;
  for (int i = 1; i <= a.n_layers; ++i) maxw = a.dims[i] > maxw ? a.dims[i] : maxw;
  float* z = smem;               // [zd][HG]
  float* t0 = z + HG * zd;       // [maxw][HG]
  float* t1 = t0 + HG * maxw;    // [maxw][HG]
  float* part = t1 + HG * maxw;  // [KS][N][HG], KS * N <= PARTN
  const int b0 = blockIdx.x * HG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- image feature (+ head LayerNorm) -> z[0:feat_dim][g]
  if (a.feat_dim > 0 && !(a.diag & 1)) {
    for (int g = wave; g < HG; g += HNT / 64) {
      const int b = b0 + g;
      const float* src = a.feat + (size_t)(b < a.B ? b : a.B - 1) * a.feat_dim;
      if (a.hn_w != nullptr) {
        float v[12];                                   // feat_dim <= 768: one load per value
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int c = lane + 64 * i;
          v[i] = c < a.feat_dim ? src[c] : 0.f;
          sum += v[i];
        }
        const float mean = wave_sum(sum) / a.feat_dim;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const float d = lane + 64 * i < a.feat_dim ? v[i] - mean : 0.f;
          sq += d * d;
        }
        const float rstd = rsqrtf(wave_sum(sq) / a.feat_dim + HN_EPS);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int c = lane + 64 * i;
          if (c < a.feat_dim) z[c * HG + g] = (v[i] - mean) * rstd * a.hn_w[c] + a.hn_b[c];
        }
      } else {
        for (int c = lane; c < a.feat_dim; c += 64) z[c * HG + g] = src[c];
      }
    }
  }
  // ---- metadata branch -> z[feat_dim : feat_dim + f2][g]
  if (a.n_meta > 0 && !(a.diag & 2)) {
    for (int i = tid; i < HG * a.n_meta; i += HNT) {
      const int g = i / a.n_meta, j = i - g * a.n_meta;
      const int b = b0 + g;
      const float v = a.meta[(size_t)(b < a.B ? b : a.B - 1) * a.n_meta + j];
      t0[j * HG + g] = fmaf(v, a.bn_scale[j], a.bn_shift[j]);
    }
    __syncthreads();
    dense(t0, a.n_meta, a.m1_wt, a.m1_b, a.f1, a.meta_act, t1, part, a.diag & 16);
    __syncthreads();
    dense(t1, a.f1, a.m2_wt, a.m2_b, a.f2, a.meta_trailing_act ? a.meta_act : ACT_NONE,
          z + a.feat_dim * HG, part, a.diag & 16);
  }
  __syncthreads();
  // ---- fusion MLP
  const float* in = z;
  float* bufs[2] = {t0, t1};
  for (int i = 0; i < a.n_layers; ++i) {
    float* o = bufs[i & 1];
    if ((i == 0 && (a.diag & 4)) || (i > 0 && (a.diag & 8))) { in = o; continue; }
    dense(in, a.dims[i], a.wt[i], a.b[i], a.dims[i + 1],
          i + 1 < a.n_layers ? a.comb_act : ACT_NONE, o, part, a.diag & 16);
    __syncthreads();
    in = o;
  }
  if (tid < HG && b0 + tid < a.B) {
    const float zz = in[tid];
    a.logits[b0 + tid] = zz;
    if (a.scores != nullptr) a.scores[b0 + tid] = 1.0f / (1.0f + expf(-zz));
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ s, T* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    d[i] = (T)s[i];
}

// four values per thread (16-byte loads, 8-byte stores): the scalar form moved 2 TB/s on the 10-100 MB operands the
// 16-bit MaxViT training casts per GEMM
template <typename T>
__global__ void cast4_kernel(const float4* __restrict__ s, T* __restrict__ d, int64_t n4) {
  typedef __attribute__((ext_vector_type(4))) T t4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = s[i];
    *reinterpret_cast<t4*>(d + 4 * i) = t4{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
  }
}

__global__ void transpose_kernel(const float* __restrict__ s, float* __restrict__ d, int R, int Cc) {
  const int64_t n = (int64_t)R * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);  // d[c][r]
    d[i] = s[(int64_t)r * Cc + c];
  }
}

template <typename T>
__global__ void pack_down_kernel(const float* __restrict__ s, T* __restrict__ d, int Cout, int Cin) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    // d[co][q][ci] <- s[co][ci][q],  q = ky*2 + kx
    const int ci = (int)(i % Cin);
    const int q = (int)((i / Cin) & 3);
    const int co = (int)(i / (4 * (int64_t)Cin));
    d[i] = (T)s[((int64_t)co * Cin + ci) * 4 + q];
  }
}

// split mode: the same order as pack_down_kernel, f16 heads in dh and f16 remainders in dl
__global__ void pack_down_split_kernel(const float* __restrict__ s, f16_t* __restrict__ dh, f16_t* __restrict__ dl,
                                       int Cout, int Cin) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int q = (int)((i / Cin) & 3);
    const int co = (int)(i / (4 * (int64_t)Cin));
    f16_t hi, lo;
    split_f16(s[((int64_t)co * Cin + ci) * 4 + q], hi, lo);
    dh[i] = hi;
    dl[i] = lo;
  }
}

template <typename T>
__global__ void transpose_cast_kernel(const float* __restrict__ s, const float* __restrict__ rowscale,
                                      T* __restrict__ d, int R, int Cc) {
  const int64_t n = (int64_t)R * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);  // d[c][r]
    d[i] = (T)(s[(int64_t)r * Cc + c] * (rowscale != nullptr ? rowscale[r] : 1.f));
  }
}

// downsample filter [Cout][Cin][2][2] fp32 -> [(q*Cin + ci)][Cout] (the dgrad GEMM's "W" operand)
template <typename T>
__global__ void pack_down_t_kernel(const float* __restrict__ s, T* __restrict__ d, int Cout, int Cin) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const int k = (int)(i / Cout);          // q*Cin + ci
    const int q = k / Cin, ci = k - q * Cin;
    d[i] = (T)s[((int64_t)co * Cin + ci) * 4 + q];
  }
}

// (g is the accumulator of the filter-gradient GEMM in front: read exactly once here and left zero for its next user)
__global__ void unpack_down_grad_kernel(float* __restrict__ g, float* __restrict__ d, int Cout,
                                        int Cin) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);
    const int ci = (int)((i >> 2) % Cin);
    const int co = (int)(i / (4 * (int64_t)Cin));
    const int64_t j = ((int64_t)co * 4 + q) * Cin + ci;
    d[i] = g[j];                                     // d[co][ci][q] = g[co][q][ci]
    g[j] = 0.f;
  }
}

__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv,
                               float* scale, float* shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = w[i] / sqrtf(rv[i] + 1e-5f);
    scale[i] = s;
    shift[i] = b[i] - rm[i] * s;
  }
}

// the five element maps above behind one launch: see PackJob in common.h
template <typename T>
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {   // last job whose first block is <= this block (uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackJob jb = jobs[lo];
  const int nblk = (lo + 1 < njobs ? jobs[lo + 1].blk0 : (int)gridDim.x) - jb.blk0;
  const float* __restrict__ s = jb.src;
  const int R = jb.R, Cc = jb.Cc;
  const int64_t n = jb.op == PACK_DOWN || jb.op == PACK_DOWN_T || jb.op == PACK_FRAG_DOWN ? (int64_t)R * Cc * 4 : (int64_t)R * Cc;
  const int64_t i0 = (int64_t)((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x, step = (int64_t)nblk * 256;
  switch (jb.op) {
    case PACK_CAST: {
      T* d = reinterpret_cast<T*>(jb.dst);
      for (int64_t i = i0; i < n; i += step) d[i] = (T)s[i];
      break;
    }
    case PACK_TRANSPOSE_F32: {
      float* d = reinterpret_cast<float*>(jb.dst);
      for (int64_t i = i0; i < n; i += step) {
        const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);
        d[i] = s[(int64_t)r * Cc + c];
      }
      break;
    }
    case PACK_TRANSPOSE_CAST: {
      T* d = reinterpret_cast<T*>(jb.dst);
      for (int64_t i = i0; i < n; i += step) {
        const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);
        d[i] = (T)(s[(int64_t)r * Cc + c] * (jb.scale != nullptr ? jb.scale[r] : 1.f));
      }
      break;
    }
    case PACK_TFRAG: {   // d = fragments of t[c][r] = s[r][c] * scale[r]: lane l of fragment (tile, k-step) holds t[16 tile + (l & 15)][32 k-step + 8 (l >> 4) + 0..7]
      T* d = reinterpret_cast<T*>(jb.dst);
      const int ksteps = R / 32;
      for (int64_t i = i0; i < n; i += step) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        const int64_t fs = i >> 9;
        const int ks = (int)(fs % ksteps), tile = (int)(fs / ksteps);
        const int c = 16 * tile + (l & 15), r = 32 * ks + 8 * (l >> 4) + j;
        d[i] = (T)(s[(int64_t)r * Cc + c] * (jb.scale != nullptr ? jb.scale[r] : 1.f));
      }
      break;
    }
    case PACK_FRAG:
    case PACK_FRAG_DOWN: {   // lane l of fragment (tile, k-step) holds w[16 tile + (l & 15)][32 k-step + 8 (l >> 4) + 0..7] (stage2p.hip: pack_frag_kernel)
      T* d = reinterpret_cast<T*>(jb.dst);
      const bool down = jb.op == PACK_FRAG_DOWN;
      const int K = down ? 4 * Cc : Cc, ksteps = K / 32;
      for (int64_t i = i0; i < n; i += step) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        const int64_t fs = i >> 9;
        const int ks = (int)(fs % ksteps), tile = (int)(fs / ksteps);
        const int row = 16 * tile + (l & 15), k = 32 * ks + 8 * (l >> 4) + j;
        float v;
        if (down) {
          const int q = k / Cc, c = k - q * Cc;
          v = s[((int64_t)row * Cc + c) * 4 + q];
        } else {
          v = s[(int64_t)row * K + k];
        }
        // (the product is rounded to fp32 BEFORE the conversion, as in stage2p.hip's pack_frag_kernel, which writes the same
        //  image in the full pack: left alone hipcc fuses the two into v_fma_mixlo_f16 -- one rounding instead of two, other
        //  bits in a few entries, and the first training step after a full pack would differ from the ones behind a re-pack)
        float pr = v * (jb.scale != nullptr ? jb.scale[row] : 1.f);
        asm volatile("" : "+v"(pr));
        d[i] = (T)pr;
      }
      break;
    }
    case PACK_DOWN: {   // d[co][q][ci] <- s[co][ci][q]
      T* d = reinterpret_cast<T*>(jb.dst);
      for (int64_t i = i0; i < n; i += step) {
        const int ci = (int)(i % Cc);
        const int q = (int)((i / Cc) & 3);
        const int co = (int)(i / (4 * (int64_t)Cc));
        d[i] = (T)s[((int64_t)co * Cc + ci) * 4 + q];
      }
      break;
    }
    default: {          // PACK_DOWN_T: d[q*Cin + ci][co] <- s[co][ci][q]
      T* d = reinterpret_cast<T*>(jb.dst);
      for (int64_t i = i0; i < n; i += step) {
        const int co = (int)(i % R);
        const int k = (int)(i / R);
        const int q = k / Cc, ci = k - q * Cc;
        d[i] = (T)s[((int64_t)co * Cc + ci) * 4 + q];
      }
    }
  }
}

inline int nblocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int launch_head(const HeadArgs& a, hipStream_t st) {
  if (a.B <= 0) return BTSBOT_OK;
  int maxw = a.f1 > a.n_meta ? a.f1 : a.n_meta;
  for (int i = 1; i <= a.n_layers; ++i) maxw = a.dims[i] > maxw ? a.dims[i] : maxw;
  const size_t lds = (size_t)HG * (a.dims[0] + 2 * maxw + PARTN) * sizeof(float);
  if (a.feat_dim > 768) {
    btsbot_set_error("head: feature width %d above 768", a.feat_dim);
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (lds > 150 * 1024) {
    btsbot_set_error("head: layer widths too large for one workgroup (%zu bytes of LDS)", lds);
    return BTSBOT_ERR_INVALID_ARG;
  }
  static size_t lds_attr = 0;
  if (lds > lds_attr) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr = lds;
  }
  hipLaunchKernelGGL(head_kernel, dim3((a.B + HG - 1) / HG), dim3(HNT), lds, st, a);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int pack_job_blocks(const PackJob& j) {
  const int64_t n = (j.op == PACK_DOWN || j.op == PACK_DOWN_T || j.op == PACK_FRAG_DOWN ? 4 : 1) * (int64_t)j.R * j.Cc;
  const int64_t b = (n + 1023) / 1024;   // four elements per thread
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int launch_pack_jobs(int prec, const PackJob* dev_jobs, int njobs, int total_blocks, hipStream_t st) {
  if (njobs <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(pack_jobs_kernel<float>, dim3(total_blocks), dim3(256), 0, st, dev_jobs, njobs);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(pack_jobs_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, dev_jobs, njobs);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(pack_jobs_kernel<f16_t>, dim3(total_blocks), dim3(256), 0, st, dev_jobs, njobs);
      break;
    default:
      btsbot_set_error("pack_jobs: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_cast(int prec, const float* src, void* dst, int64_t n, hipStream_t st) {
  if (n <= 0) return BTSBOT_OK;
  switch (prec) {
    case BTSBOT_F32:
      HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
      return BTSBOT_OK;
    case BTSBOT_BF16:
      if (n >= 4096 && n % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0)
        hipLaunchKernelGGL(cast4_kernel<bf16_t>, dim3(nblocks(n / 4)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(src), reinterpret_cast<bf16_t*>(dst), n / 4);
      else
      hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<bf16_t*>(dst), n);
      break;
    case BTSBOT_F16:
      if (n >= 4096 && n % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0)
        hipLaunchKernelGGL(cast4_kernel<f16_t>, dim3(nblocks(n / 4)), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(src), reinterpret_cast<f16_t*>(dst), n / 4);
      else
      hipLaunchKernelGGL(cast_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<f16_t*>(dst), n);
      break;
    default:
      btsbot_set_error("cast: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_transpose_f32(const float* src, float* dst, int R, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(nblocks((int64_t)R * Cc)), dim3(256), 0, st, src, dst,
                     R, Cc);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_pack_down(int prec, const float* src, void* dst, int Cout, int Cin, hipStream_t st) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(pack_down_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<float*>(dst), Cout, Cin);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(pack_down_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<bf16_t*>(dst), Cout, Cin);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(pack_down_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<f16_t*>(dst), Cout, Cin);
      break;
    default:
      btsbot_set_error("pack_down: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_pack_down_split(const float* src, void* hi, void* lo, int Cout, int Cin, hipStream_t st) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  hipLaunchKernelGGL(pack_down_split_kernel, dim3(nblocks(n)), dim3(256), 0, st, src, reinterpret_cast<f16_t*>(hi),
                     reinterpret_cast<f16_t*>(lo), Cout, Cin);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                   float* shift, int n, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, b, rm, rv, scale,
                     shift, n);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// dst[c][r] = src[r][c] * (rowscale ? rowscale[r] : 1)
int launch_transpose_cast(int prec, const float* src, const float* rowscale, void* dst, int R, int Cc,
                          hipStream_t st) {
  const int64_t n = (int64_t)R * Cc;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(transpose_cast_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         rowscale, reinterpret_cast<float*>(dst), R, Cc);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         rowscale, reinterpret_cast<bf16_t*>(dst), R, Cc);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(transpose_cast_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         rowscale, reinterpret_cast<f16_t*>(dst), R, Cc);
      break;
    default:
      btsbot_set_error("transpose_cast: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_unpack_down_grad(float* Gd, float* dst, int Cout, int Cin, hipStream_t st) {
  hipLaunchKernelGGL(unpack_down_grad_kernel, dim3(nblocks((int64_t)Cout * Cin * 4)), dim3(256), 0,
                     st, Gd, dst, Cout, Cin);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

int launch_pack_down_t(int prec, const float* src, void* dst, int Cout, int Cin, hipStream_t st) {
  const int64_t n = (int64_t)Cout * Cin * 4;
  switch (prec) {
    case BTSBOT_F32:
      hipLaunchKernelGGL(pack_down_t_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<float*>(dst), Cout, Cin);
      break;
    case BTSBOT_BF16:
      hipLaunchKernelGGL(pack_down_t_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<bf16_t*>(dst), Cout, Cin);
      break;
    case BTSBOT_F16:
      hipLaunchKernelGGL(pack_down_t_kernel<f16_t>, dim3(nblocks(n)), dim3(256), 0, st, src,
                         reinterpret_cast<f16_t*>(dst), Cout, Cin);
      break;
    default:
      btsbot_set_error("pack_down_t: bad precision %d", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
