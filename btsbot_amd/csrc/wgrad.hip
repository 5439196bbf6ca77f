// Filter-gradient GEMM of the 1x1 convolutions, 16-bit operands (gfx950):
//
//   out[n][k] += sum_m D[m][n] * A[m][k]        colsum[n] += sum_m D[m][n]   (optional)
//
// D = gradient w.r.t. the layer output [M pixels][N], A = the layer input [M][K]; both are stored
// pixel-major, i.e. with the REDUCTION index outermost -- the opposite of what an MFMA operand
// wants.  The tiles are therefore staged row-major (16-byte global loads -> ds_write_b128) and
// read back with ds_read_b64_tr_b16, gfx950's transposing LDS read: two reads hand a lane its
// eight reduction-consecutive values of one output row/column (cdna_hip_programming.md T10).
// Rows are padded by 64 bytes so the four rows a 32-lane half touches fall on disjoint banks.
//
// Workgroup = TN x TK output tile x one slice of M, 4 waves in 2 x 2, v_mfma_f32_32x32x16.
// Global loads of tile t+1 are in flight while tile t is multiplied (register-staged double
// buffer, one barrier per 64-row tile: with 32-row tiles a tile's 8 MFMAs per wave hid a quarter of the load
// latency, 64 rows took the isolated backward from 3.68 to 3.58 ms).  Slices leave as dense partial tiles which
// wgrad_reduce_kernel adds in a fixed order (or, without scratch, meet in the output through fp32 atomics).
// Replaces (with backward.hip) what autograd does for nn.Linear / 1x1 Conv2d weight gradients at
// /root/reference/btsbot/train.py:526.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <typename T> struct W2M;
template <> struct W2M<bf16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float tofloat(unsigned short v) {
    return __builtin_bit_cast(float, (unsigned)v << 16);
  }
};
template <> struct W2M<f16_t> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float tofloat(unsigned short v) {
    return (float)__builtin_bit_cast(f16_t, v);
  }
};

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// lane's 8 reduction-consecutive elements (rows r0 + 8h .. +7) of column c0 + (lane & 31)
__device__ __forceinline__ s16x8 tr_frag(const unsigned char* tile, int pitchb, int r0, int c0,
                                         int lane) {
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const unsigned char* a = tile + (r0 + 8 * h + q) * pitchb + (c0 + 16 * g1 + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 4 * pitchb));
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int TM = 64;   // reduction rows per LDS tile

// (bx, by, bz) = the workgroup's output tile and slice; (gx, gy) = tiles along n and k
template <typename T, int TN, int TK>
__device__ __forceinline__ void wgrad2_body(const T* __restrict__ D, const T* __restrict__ A, float* __restrict__ out,
                                            float* __restrict__ colsum, int M, int N, int K, int ldo, int mslice,
                                            float* __restrict__ part, int bx, int by, int bz, int gx, int gy) {
  constexpr int PN = TN * 2 + 64, PK = TK * 2 + 64;          // row pitch in bytes
  constexpr int DB = TM * PN, AB = TM * PK;                   // bytes per tile
  constexpr int FN = TN / 64, FK = TK / 64;                   // 32x32 fragments per wave
  constexpr int LN = TN * TM / 2048, LK = TK * TM / 2048;     // 16-byte chunks per thread per tile
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * (DB + AB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = bx * TN, k0 = by * TK;
  const int mbeg = bz * mslice, mend = min(M, mbeg + mslice);
  const int nt = (mend - mbeg + TM - 1) / TM;
  const bool do_sum = colsum != nullptr && by == 0;

  f32x16 acc[FN][FK];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float csum = 0.f;

  u32x4 rd[LN], ra[LK];
  auto fetch = [&](int t) {
    const int m0 = mbeg + t * TM;
#pragma unroll
    for (int s = 0; s < LN; ++s) {
      const int q = tid + 256 * s, r = q / (TN / 8), cc = q % (TN / 8);
      const int m = m0 + r, n = n0 + 8 * cc;
      const bool ok = m < mend && n < N;
      const u32x4* src = reinterpret_cast<const u32x4*>(D + (size_t)(ok ? m : mbeg) * N + (ok ? n : 0));
      const u32x4 v = *src;
      rd[s] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int s = 0; s < LK; ++s) {
      const int q = tid + 256 * s, r = q / (TK / 8), cc = q % (TK / 8);
      const int m = m0 + r, k = k0 + 8 * cc;
      const bool ok = m < mend && k < K;
      const u32x4* src = reinterpret_cast<const u32x4*>(A + (size_t)(ok ? m : mbeg) * K + (ok ? k : 0));
      const u32x4 v = *src;
      ra[s] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&](int buf) {
    unsigned char* ds = sm + buf * (DB + AB);
    unsigned char* as = ds + DB;
#pragma unroll
    for (int s = 0; s < LN; ++s) {
      const int q = tid + 256 * s, r = q / (TN / 8), cc = q % (TN / 8);
      *reinterpret_cast<u32x4*>(ds + r * PN + cc * 16) = rd[s];
    }
#pragma unroll
    for (int s = 0; s < LK; ++s) {
      const int q = tid + 256 * s, r = q / (TK / 8), cc = q % (TK / 8);
      *reinterpret_cast<u32x4*>(as + r * PK + cc * 16) = ra[s];
    }
  };

  if (nt > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;   // workgroup-uniform
    if (more) fetch(t + 1);
    const unsigned char* ds = sm + (t & 1) * (DB + AB);
    const unsigned char* as = ds + DB;
#pragma unroll
    for (int ms = 0; ms < TM; ms += 16) {
      s16x8 af[FN], bf[FK];
#pragma unroll
      for (int i = 0; i < FN; ++i) af[i] = tr_frag(ds, PN, ms, wn * (TN / 2) + 32 * i, lane);
#pragma unroll
      for (int j = 0; j < FK; ++j) bf[j] = tr_frag(as, PK, ms, wk * (TK / 2) + 32 * j, lane);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FK; ++j) acc[i][j] = W2M<T>::run(af[i], bf[j], acc[i][j]);
    }
    if (do_sum) {   // thread = (column, row group); rows rg, rg + G, ...
      constexpr int G = 256 / TN;
      const int col = tid % TN, rg = tid / TN;
#pragma unroll
      for (int r = 0; r < TM / G; ++r)
        csum += W2M<T>::tofloat(
            *reinterpret_cast<const unsigned short*>(ds + (rg + G * r) * PN + col * 2));
    }
    if (more) stash((t + 1) & 1);
    __syncthreads();
  }

  // C/D layout of 32x32: column = lane & 31 -> k, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n
  const int lc = lane & 31, lh = lane >> 5;
  if (part != nullptr) {
    // slice partial as a dense TN x TK tile (plain coalesced stores; wgrad_reduce_kernel adds the slices in a
    // fixed order): an fp32 atomic per element per slice was the larger half of this kernel's time
    float* pt = part + ((size_t)(bz * gy + by) * gx + bx) * (TN * TK);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nl = wn * (TN / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int kl = wk * (TK / 2) + 32 * j + lc;
          pt[nl * TK + kl] = acc[i][j][r];
        }
  } else
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (TN / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = k0 + wk * (TK / 2) + 32 * j + lc;
        if (n < N && k < K) atomicAdd(out + (size_t)n * ldo + k, acc[i][j][r]);
      }
  if (do_sum) {
    constexpr int G = 256 / TN;
    float* red = reinterpret_cast<float*>(sm);   // all tile reads are behind the loop's last barrier
    red[tid] = csum;
    __syncthreads();
    if (tid < TN && n0 + tid < N) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) s += red[tid + TN * g];
      atomicAdd(colsum + n0 + tid, s);
    }
  }
}

template <typename T, int TN, int TK>
__global__ __launch_bounds__(256) void wgrad2_kernel(const T* __restrict__ D, const T* __restrict__ A, float* __restrict__ out,
                                                     float* __restrict__ colsum, int M, int N, int K, int ldo, int mslice,
                                                     float* __restrict__ part) {
  wgrad2_body<T, TN, TK>(D, A, out, colsum, M, N, K, ldo, mslice, part, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x,
                         gridDim.y);
}

// Several filter-gradient GEMMs as ONE launch: a stage's blocks each bring two (fc2: dy^T h, fc1: da^T xn) whose
// operands all exist once the stage's input-gradient chain has passed.  Per GEMM they are 16 output tiles of 128 x 128
// over 9216 rows (stage 2): launched one by one each needs ~24 reduction slices to fill the chip (24 MB of partial tiles
// written and read again per GEMM, ~26 us per launch, twelve launches per step); together 192 tiles x 3 slices do.
constexpr int WB_MAX = 16;
struct WgradBatch {
  int njobs;
  int wg0[WB_MAX + 1];   // first workgroup of job i (prefix sums)
  struct Job {
    const void* D;
    const void* A;
    float* out;
    float* colsum;
    float* part;
    int M, N, K, ldo, gx, gy, mslice;
  } j[WB_MAX];
};
template <typename T, int TN, int TK> __global__ __launch_bounds__(256) void wgrad2_batched_kernel(WgradBatch a) {
  int i = 0;
  while (i + 1 < a.njobs && (int)blockIdx.x >= a.wg0[i + 1]) ++i;
  const WgradBatch::Job& J = a.j[i];
  const int w = (int)blockIdx.x - a.wg0[i];
  const int bx = w % J.gx, by = (w / J.gx) % J.gy, bz = w / (J.gx * J.gy);
  wgrad2_body<T, TN, TK>(reinterpret_cast<const T*>(J.D), reinterpret_cast<const T*>(J.A), J.out, J.colsum, J.M, J.N, J.K,
                         J.ldo, J.mslice, J.part, bx, by, bz, J.gx, J.gy);
}

// out[n][k] += sum over slices of the partial tiles (fixed order: bit-reproducible).  One launch serves up to two
// GEMMs (a block's fc2 and fc1 filter gradients): workgroups [0, nblk0) belong to job 0, the rest to job 1.
struct ReduceJobs {
  WgradReduceJob j[2];
  int nblk0;
};

// SG = slice groups per output: 1 = a thread walks all slices of its output; 4 = small outputs with many slices (the
// stem's 64 x 48 filter over ~400 slices: 12 workgroups walked 400 dependent-latency loads each) -- a workgroup is
// 64 outputs x 4 interleaved slice groups which meet in LDS in a fixed order
template <int SG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceJobs a) {
  constexpr int OPB = 256 / SG;   // outputs per workgroup
  const bool second = (int)blockIdx.x >= a.nblk0;
  const WgradReduceJob& J = a.j[second ? 1 : 0];
  const int o = threadIdx.x % OPB, sg = threadIdx.x / OPB;
  const int idx = ((int)blockIdx.x - (second ? a.nblk0 : 0)) * OPB + o;
  const bool live = idx < J.N * J.K;
  const int n = live ? idx / J.K : 0, k = live ? idx - n * J.K : 0;
  const int tx = n / J.TN, ty = k / J.TK;
  const float* p = J.part + ((size_t)ty * J.gx + tx) * (J.TN * J.TK) + (n - tx * J.TN) * J.TK + (k - ty * J.TK);
  const size_t sstride = (size_t)J.gx * J.gy * J.TN * J.TK;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    int s = sg;
#pragma unroll 2
    for (; s + 3 * SG < J.nsl; s += 4 * SG) {
      s0 += p[(size_t)s * sstride];
      s1 += p[(size_t)(s + SG) * sstride];
      s2 += p[(size_t)(s + 2 * SG) * sstride];
      s3 += p[(size_t)(s + 3 * SG) * sstride];
    }
    for (; s < J.nsl; s += SG) s0 += p[(size_t)s * sstride];
  }
  float tot = (s0 + s1) + (s2 + s3);
  if (SG > 1) {
    __shared__ float sh[SG][OPB];
    sh[sg][o] = tot;
    __syncthreads();
    if (sg != 0) return;
    tot = 0.f;
#pragma unroll
    for (int g = 0; g < SG; ++g) tot += sh[g][o];
  }
  if (live) J.out[(size_t)n * J.ldo + k] += tot;
}

// The same sum, laid out for the memory system: a workgroup owns 128 consecutive floats of the partial-tile layout
// (512 bytes of every slice) and its eight 32-lane groups walk the slices s = group, group + 8, ... with 16-byte loads,
// four independent accumulators each, and meet in LDS in a fixed order.  (The one-thread-per-output walk above moved
// 0.5-1.8 TB/s over 33-134 MB of partials per launch -- 73-110 us behind stage 0's fused MLP backward.)
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(ReduceJobs a) {
  const bool second = (int)blockIdx.x >= a.nblk0;
  const WgradReduceJob& J = a.j[second ? 1 : 0];
  const int q = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const size_t tile = (size_t)J.TN * J.TK, per_slice = (size_t)J.gx * J.gy * tile;
  const size_t e = ((size_t)((int)blockIdx.x - (second ? a.nblk0 : 0)) * 32 + q) * 4;   // first of this thread's 4 floats
  const bool inside = e < per_slice;
  const int t = inside ? (int)(e / tile) : 0;
  const int r = inside ? (int)(e - (size_t)t * tile) : 0;
  const int nl = r / J.TK, kl = r - nl * J.TK;
  const int tx = t % J.gx, ty = t / J.gx;
  const int n = tx * J.TN + nl, k = ty * J.TK + kl;
  const bool live = inside && n < J.N && k < J.K;    // (K is a multiple of 4: the four floats are live together)
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto add = [](float4& d, const float4 v) {
    d.x += v.x;
    d.y += v.y;
    d.z += v.z;
    d.w += v.w;
  };
  if (live) {
    const float* p = J.part + e;
    int s = sg;
    for (; s + 24 < J.nsl; s += 32) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)s * per_slice);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (size_t)(s + 8) * per_slice);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (size_t)(s + 16) * per_slice);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (size_t)(s + 24) * per_slice);
      add(s0, v0);
      add(s1, v1);
      add(s2, v2);
      add(s3, v3);
    }
    for (; s < J.nsl; s += 8) add(s0, *reinterpret_cast<const float4*>(p + (size_t)s * per_slice));
  }
  add(s0, s1);
  add(s2, s3);
  add(s0, s2);
  __shared__ float4 sh[8][32];
  sh[sg][q] = s0;
  __syncthreads();
  if (sg != 0 || !live) return;
  float4 tot = sh[0][q];
#pragma unroll
  for (int g = 1; g < 8; ++g) add(tot, sh[g][q]);
  float4* o = reinterpret_cast<float4*>(J.out + (size_t)n * J.ldo + k);
  float4 cur = *o;
  add(cur, tot);
  *o = cur;
}

// the slice reduction of up to WB_MAX GEMMs as one launch (wgrad_reduce4_kernel's walk, per job)
struct ReduceJobsN {
  int n;
  int blk0[WB_MAX + 1];
  WgradReduceJob j[WB_MAX];
};
__global__ __launch_bounds__(256) void wgrad_reduce4n_kernel(ReduceJobsN a) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.blk0[i + 1]) ++i;
  const WgradReduceJob& J = a.j[i];
  const int q = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const size_t tile = (size_t)J.TN * J.TK, per_slice = (size_t)J.gx * J.gy * tile;
  const size_t e = ((size_t)((int)blockIdx.x - a.blk0[i]) * 32 + q) * 4;
  const bool inside = e < per_slice;
  const int t = inside ? (int)(e / tile) : 0;
  const int r = inside ? (int)(e - (size_t)t * tile) : 0;
  const int nl = r / J.TK, kl = r - nl * J.TK;
  const int tx = t % J.gx, ty = t / J.gx;
  const int n = tx * J.TN + nl, k = ty * J.TK + kl;
  const bool live = inside && n < J.N && k < J.K;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* p = J.part + e;
    for (int s = sg; s < J.nsl; s += 8) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)s * per_slice);
      s0.x += v.x;
      s0.y += v.y;
      s0.z += v.z;
      s0.w += v.w;
    }
  }
  __shared__ float4 sh[8][32];
  sh[sg][q] = s0;
  __syncthreads();
  if (sg != 0 || !live) return;
  float4 tot = sh[0][q];
#pragma unroll
  for (int g = 1; g < 8; ++g) {
    tot.x += sh[g][q].x;
    tot.y += sh[g][q].y;
    tot.z += sh[g][q].z;
    tot.w += sh[g][q].w;
  }
  float4* o = reinterpret_cast<float4*>(J.out + (size_t)n * J.ldo + k);
  float4 cur = *o;
  cur.x += tot.x;
  cur.y += tot.y;
  cur.z += tot.z;
  cur.w += tot.w;
  *o = cur;
}

template <typename T, int TN, int TK>
int wgrad2_launch(const void* D, const void* A, float* out, float* colsum, int M, int N, int K,
                  int ldo, hipStream_t st, float* part, size_t part_floats, WgradReduceJob* defer) {
  const int gx = (N + TN - 1) / TN, gy = (K + TK - 1) / TK;
  // slices of the reduction.  Two-pass form (partial tiles + wgrad_reduce_kernel, when the caller lends scratch):
  // ~384 workgroups, at least 256 rows each (re-swept with 64-row LDS tiles and the second stream: 256-512 workgroups
  // x 256-512 rows all land within 1.5 % of each other); atomic form (every slice ends in TN x TK fp32 atomics: fewer, longer
  // slices win): ~384 workgroups, at least 512 rows.  Measured per 1024-alert step: 4.59 ms vs 4.67 ms.
  static const int env_rows = [] {
    const char* e = getenv("BTSBOT_AMD_WGRAD_MIN_ROWS");   // tuning knob (overrides both defaults)
    const int v = e ? atoi(e) : 0;
    return v >= 32 ? v : 0;
  }();
  static const int env_wg = [] {
    const char* e = getenv("BTSBOT_AMD_WGRAD_WGS");        // tuning knob (overrides both defaults)
    const int v = e ? atoi(e) : 0;
    return v >= 1 ? v : 0;
  }();
  static const bool atomic_only = [] {
    const char* e = getenv("BTSBOT_AMD_WGRAD_ATOMIC");     // 1: every slice adds its tile with fp32 atomics (A/B)
    return e != nullptr && e[0] == '1';
  }();
  int nsl = 1, mslice = M;
  auto slices = [&](int min_rows, int target_wg) {
    nsl = (target_wg + gx * gy - 1) / (gx * gy);
    if (nsl > (M + min_rows - 1) / min_rows) nsl = (M + min_rows - 1) / min_rows;
    if (nsl < 1) nsl = 1;
    mslice = ((M + nsl - 1) / nsl + TM - 1) / TM * TM;
    nsl = (M + mslice - 1) / mslice;
  };
  bool two_pass = !atomic_only && part != nullptr;
  if (two_pass) {
    slices(env_rows ? env_rows : 256, env_wg ? env_wg : 384);
    two_pass = nsl > 1 && (size_t)nsl * gx * gy * TN * TK <= part_floats;
  }
  if (!two_pass) slices(env_rows ? env_rows : 512, env_wg ? env_wg : 384);
  hipLaunchKernelGGL((wgrad2_kernel<T, TN, TK>), dim3(gx, gy, nsl), dim3(256), 0, st,
                     reinterpret_cast<const T*>(D), reinterpret_cast<const T*>(A), out, colsum, M,
                     N, K, ldo, mslice, two_pass ? part : nullptr);
  LAUNCH_CHECK();
  WgradReduceJob job = {part, out, N, K, ldo, gx, gy, two_pass ? nsl : 0, TN, TK};
  if (defer != nullptr) {
    *defer = job;   // (nsl == 0: the slices met through atomics, nothing left to add)
    return BTSBOT_OK;
  }
  return launch_wgrad_reduce(&job, 1, st);
}

template <typename T>
int wgrad2_t(const void* D, const void* A, float* out, float* colsum, int M, int N, int K, int ldo,
             hipStream_t st, float* part, size_t pf, WgradReduceJob* defer) {
  if (N > 64 && K > 64) return wgrad2_launch<T, 128, 128>(D, A, out, colsum, M, N, K, ldo, st, part, pf, defer);
  if (N > 64) return wgrad2_launch<T, 128, 64>(D, A, out, colsum, M, N, K, ldo, st, part, pf, defer);
  if (K > 64) return wgrad2_launch<T, 64, 128>(D, A, out, colsum, M, N, K, ldo, st, part, pf, defer);
  return wgrad2_launch<T, 64, 64>(D, A, out, colsum, M, N, K, ldo, st, part, pf, defer);
}

}  // namespace

// the slice reduction of up to two filter-gradient GEMMs (jobs with nsl == 0 are skipped) as one launch
int launch_wgrad_reduce(const WgradReduceJob* jobs, int njobs, hipStream_t st) {
  ReduceJobs a;
  int n = 0, nblk[2] = {0, 0};
  for (int i = 0; i < njobs && n < 2; ++i)
    if (jobs[i].nsl > 0) {
      a.j[n] = jobs[i];
      nblk[n] = (jobs[i].N * jobs[i].K + 255) / 256;
      ++n;
    }
  if (n == 0) return BTSBOT_OK;
  if (n == 1) a.j[1] = a.j[0];
  {
    // the 16-byte form wherever the outputs allow it (every arena tensor does: 16-byte aligned, K a multiple of 4)
    static const bool old_form = [] {
      const char* e = getenv("BTSBOT_AMD_WGRAD_REDUCE1");   // 1: the one-thread-per-output kernels (A/B)
      return e != nullptr && e[0] == '1';
    }();
    bool ok = !old_form;
    for (int i = 0; i < n; ++i)
      ok = ok && (a.j[i].K % 4 == 0) && (a.j[i].TK % 4 == 0) && (a.j[i].ldo % 4 == 0) &&
           (((uintptr_t)a.j[i].out & 15) == 0) && (((uintptr_t)a.j[i].part & 15) == 0);
    if (ok) {
      auto nb = [](const WgradReduceJob& j) { return (int)(((size_t)j.gx * j.gy * j.TN * j.TK + 127) / 128); };
      a.nblk0 = nb(a.j[0]);
      const int total = a.nblk0 + (n > 1 ? nb(a.j[1]) : 0);
      hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(total), dim3(256), 0, st, a);
      LAUNCH_CHECK();
      return BTSBOT_OK;
    }
  }
  // many slices per output (the stem's 64 x 48 filter over ~400 slices; mlp_bwd_kernel's 256 workgroup partials of a
  // block's two 64 x 256 filter gradients: one thread per output walked 256 dependent-latency loads, 76-118 us next to
  // the chain's kernels): four or eight slice groups per output
  int minsl = a.j[0].nsl;
  for (int i = 1; i < n; ++i) minsl = a.j[i].nsl < minsl ? a.j[i].nsl : minsl;
  const bool small = n == 1 ? a.j[0].N * a.j[0].K <= 16384 : a.j[0].N * a.j[0].K + a.j[1].N * a.j[1].K <= 65536;
  auto blocks = [&](int opb) {
    nblk[0] = (a.j[0].N * a.j[0].K + opb - 1) / opb;
    nblk[1] = n > 1 ? (a.j[1].N * a.j[1].K + opb - 1) / opb : 0;
    a.nblk0 = nblk[0];
    return nblk[0] + nblk[1];
  };
  if (small && minsl >= 128 && n == 2) {
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(blocks(32)), dim3(256), 0, st, a);
  } else if (small && minsl >= 64) {
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(blocks(64)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(blocks(256)), dim3(256), 0, st, a);
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// 16-bit modes only; N and K must be multiples of 8 and the operands 16-byte aligned.
// part (optional): scratch for the slices' partial tiles -> two-pass reduction instead of atomics into `out`
// defer (optional): the reduction is not launched here; *defer describes it for launch_wgrad_reduce()
int launch_wgrad16(int prec, const void* D, const void* A, float* out, float* colsum, int M, int N,
                   int K, int ldo, hipStream_t st, float* part, size_t part_floats, WgradReduceJob* defer) {
  if (defer != nullptr) defer->nsl = 0;
  if (M <= 0) return BTSBOT_OK;
  if ((N & 7) || (K & 7) || ((uintptr_t)D & 15) || ((uintptr_t)A & 15)) {
    btsbot_set_error("wgrad16: N=%d K=%d / operand alignment not supported", N, K);
    return BTSBOT_ERR_INVALID_ARG;
  }
  switch (prec) {
    case BTSBOT_BF16: return wgrad2_t<bf16_t>(D, A, out, colsum, M, N, K, ldo, st, part, part_floats, defer);
    case BTSBOT_F16: return wgrad2_t<f16_t>(D, A, out, colsum, M, N, K, ldo, st, part, part_floats, defer);
    default:
      btsbot_set_error("wgrad16: precision %d is not a 16-bit mode", prec);
      return BTSBOT_ERR_INVALID_ARG;
  }
}

// ---- batched form (backbone_train.hip): up to 16 GEMMs with N, K multiples of 128, one launch + one reduction.
// part: scratch of part_floats floats shared by the jobs; target_wg: workgroups the launch should have (slices are
// chosen per job so that the jobs together come to about that many; >= 256 rows per slice).
int launch_wgrad16_batched(int prec, const WgradBatchJob* jobs, int njobs, float* part, size_t part_floats, int target_wg,
                           hipStream_t st) {
  if (njobs <= 0) return BTSBOT_OK;
  if (njobs > WB_MAX || (prec != BTSBOT_BF16 && prec != BTSBOT_F16)) {
    btsbot_set_error("wgrad16_batched: %d jobs / precision %d not supported", njobs, prec);
    return BTSBOT_ERR_INVALID_ARG;
  }
  WgradBatch a;
  ReduceJobsN r;
  a.njobs = njobs;
  r.n = 0;
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    if ((jobs[i].N & 127) || (jobs[i].K & 127) || jobs[i].M <= 0) {
      btsbot_set_error("wgrad16_batched: job %d: N=%d K=%d must be multiples of 128", i, jobs[i].N, jobs[i].K);
      return BTSBOT_ERR_INVALID_ARG;
    }
    tiles += (jobs[i].N / 128) * (jobs[i].K / 128);
  }
  int wg = 0;
  size_t pused = 0;
  r.blk0[0] = 0;
  for (int i = 0; i < njobs; ++i) {
    const WgradBatchJob& b = jobs[i];
    const int gx = b.N / 128, gy = b.K / 128;
    int nsl = (target_wg + tiles - 1) / tiles;
    if (nsl > (b.M + 255) / 256) nsl = (b.M + 255) / 256;
    if (nsl < 1) nsl = 1;
    int mslice = ((b.M + nsl - 1) / nsl + TM - 1) / TM * TM;
    nsl = (b.M + mslice - 1) / mslice;
    const size_t need = (size_t)nsl * gx * gy * 128 * 128;
    const bool two_pass = nsl > 1 && pused + need <= part_floats;
    if (nsl > 1 && !two_pass) {   // no room for the partial tiles: one slice (the out += path has a single adder then)
      nsl = 1;
      mslice = (b.M + TM - 1) / TM * TM;
    }
    a.wg0[i] = wg;
    a.j[i] = WgradBatch::Job{b.D, b.A, b.out, b.colsum, two_pass ? part + pused : nullptr, b.M, b.N, b.K, b.ldo, gx, gy, mslice};
    wg += gx * gy * nsl;
    if (two_pass) {
      r.j[r.n] = WgradReduceJob{part + pused, b.out, b.N, b.K, b.ldo, gx, gy, nsl, 128, 128};
      r.blk0[r.n + 1] = r.blk0[r.n] + (int)(((size_t)gx * gy * 128 * 128 + 127) / 128);
      ++r.n;
      pused += need;
    }
  }
  a.wg0[njobs] = wg;
  if (prec == BTSBOT_BF16) hipLaunchKernelGGL((wgrad2_batched_kernel<bf16_t, 128, 128>), dim3(wg), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad2_batched_kernel<f16_t, 128, 128>), dim3(wg), dim3(256), 0, st, a);
  LAUNCH_CHECK();
  if (r.n > 0) {
    hipLaunchKernelGGL(wgrad_reduce4n_kernel, dim3(r.blk0[r.n]), dim3(256), 0, st, r);
    LAUNCH_CHECK();
  }
  return BTSBOT_OK;
}
