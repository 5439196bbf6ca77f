// Shared device helpers and launcher prototypes for libbtsbot_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/btsbot_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
struct fp8_t { unsigned char v; };   // tag of the fp8 (OCP e4m3) operand mode of stage2p.hip / stage3.hip
struct f16x2_t { _Float16 v; };      // tag of the split-operand mode (BTSBOT_F16X2): x = hi + lo, both f16
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4v;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
// split operands: 8 / 4 / 2 values as f16 head + f16 remainder (the remainder of a value below ~2^-13 is an f16
// subnormal: the matrix pipe takes subnormal f16 operands as they are, kernels are built with the default
// denormal mode)
struct h2x8 { f16x8 hi, lo; };
struct h2x4 { f16x4v hi, lo; };
struct h2x2 { f16x2v hi, lo; };
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
__device__ __forceinline__ h2x8 split8(const float (&v)[8]) {
  h2x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split_f16(v[j], a, b);
    o.hi[j] = a;
    o.lo[j] = b;
  }
  return o;
}
__device__ __forceinline__ h2x4 split4(const float (&v)[4]) {
  h2x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 a, b;
    split_f16(v[j], a, b);
    o.hi[j] = a;
    o.lo[j] = b;
  }
  return o;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds for ONE device: a launcher keeps one flag per (kernel, device),
// so a handle on cuda:1 after one on cuda:0 sets it again (a process-wide bool skipped it: launch failure on the second
// GPU for every kernel above 64 KB of LDS).  Racing threads may both set the attribute: harmless.
#include <atomic>
struct DevOnce {
  std::atomic<unsigned long long> mask{0};
  static int dev() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
  }
  bool need() const { return ((mask.load(std::memory_order_relaxed) >> dev()) & 1ULL) == 0; }
  void done() { mask.fetch_or(1ULL << dev(), std::memory_order_relaxed); }
};

// ---------------------------------------------------------------------------------------
// error plumbing (no exception crosses the ABI)
// ---------------------------------------------------------------------------------------
void btsbot_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      btsbot_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                       \
      return BTSBOT_ERR_HIP;                                                            \
    }                                                                                   \
  } while (0)

#define LAUNCH_CHECK()                                                                  \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) {                                                             \
      btsbot_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),       \
                       __FILE__, __LINE__);                                             \
      return BTSBOT_ERR_HIP;                                                            \
    }                                                                                   \
  } while (0)

// ---------------------------------------------------------------------------------------
// device math
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) {
  // nn.GELU() default (exact erf form), architectures.py:35,149
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for the 16-bit MFMA modes, one transcendental instead of erff's ~40 instructions:
//     gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),      Phi(-a) = 2^q(a)  for a >= 0,
// q = minimax polynomial of log2 Phi(-a) fitted so that |a 2^q(a) - a Phi(-a)| is smallest (tools/fit_gelu.py).
// The leading coefficients are negative, so q -> -inf and the tail is exactly 0 for any |x|: no clamp.
//   DEG 3: max |error| 5.8e-5 (below half an ulp of bf16 for |gelu| > 0.015), 3 fma + v_exp + v_max + fma;
//   DEG 5: max |error| 4.7e-7 (f32-class), 2 more fma.
// On gfx950 a v_exp_f32 issues in ~8 cycles, a plain VALU op in ~3: 24 / 30 cycles per wave-GELU against 42
// for the x * sigmoid(quintic) form this replaces (v_exp + v_rcp + v_med3 + 6).  The fp32 parity mode keeps gelu_erf.
template <int DEG> __device__ __forceinline__ float gelu_q(float a);
template <> __device__ __forceinline__ float gelu_q<3>(float a) {
  float t = fmaf(a, -0.024772998623334343f, -0.49926576060257244f);
  t = fmaf(a, t, -1.1287482669759885f);
  return fmaf(a, t, -1.0036805164077327f);
}
template <> __device__ __forceinline__ float gelu_q<5>(float a) {
  float t = fmaf(a, -0.0004726569791655389f, 0.007079169600613161f);
  t = fmaf(a, t, -0.05181158088529847f);
  t = fmaf(a, t, -0.46001256950698816f);
  t = fmaf(a, t, -1.1507770495088248f);
  return fmaf(a, t, -1.000039487932206f);
}
// max(x, 0) as ONE instruction: v_max_i32 on the bits (a positive float is a positive integer, anything with the
// sign bit set is negative).  fmaxf costs two (hipcc canonicalises the operand first for its NaN rule), and an
// inline-asm v_max_f32 is invisible to the compiler's hazard recogniser: scheduled right behind the MFMA that
// produces x it read the register before the matrix pipe had written it (stage2p.hip, first cut).
__device__ __forceinline__ float relu_f(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}
template <int DEG> __device__ __forceinline__ float gelu_poly(float x) {
  const float a = __builtin_fabsf(x);
  const float e = __builtin_amdgcn_exp2f(gelu_q<DEG>(a));
  return fmaf(-a, e, relu_f(x));
}
// d/dx of gelu_poly: with E = 2^q(a), D = E (1 + a ln2 q'(a)):  x > 0: 1 - D,  x < 0: D  (x = 0: 1/2 either way
// up to the fit error).  The 16-bit training forward applies gelu_poly, so its backward differentiates gelu_poly.
template <int DEG> __device__ __forceinline__ float gelu_dq(float a);
template <> __device__ __forceinline__ float gelu_dq<3>(float a) {
  float t = fmaf(a, 3.0f * -0.024772998623334343f, 2.0f * -0.49926576060257244f);
  return fmaf(a, t, -1.1287482669759885f);
}
template <> __device__ __forceinline__ float gelu_dq<5>(float a) {
  float t = fmaf(a, 5.0f * -0.0004726569791655389f, 4.0f * 0.007079169600613161f);
  t = fmaf(a, t, 3.0f * -0.05181158088529847f);
  t = fmaf(a, t, 2.0f * -0.46001256950698816f);
  return fmaf(a, t, -1.1507770495088248f);
}
template <int DEG> __device__ __forceinline__ float gelu_poly_grad(float x) {
  const float a = __builtin_fabsf(x);
  const float e = __builtin_amdgcn_exp2f(gelu_q<DEG>(a));
  const float d = e * fmaf(a * 0.6931471805599453f, gelu_dq<DEG>(a), 1.0f);
  return x > 0.0f ? 1.0f - d : d;
}
template <typename T> struct GeluDeg { static constexpr int value = 5; };     // f16 operands: f32-class GELU
template <> struct GeluDeg<bf16_t> { static constexpr int value = 3; };       // bf16 operands
template <typename T> __device__ __forceinline__ float gelu_for(float x) { return gelu_poly<GeluDeg<T>::value>(x); }
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }

// d/dx of the exact erf GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// SiLU for the 16-bit modes: v_exp + v_rcp instead of the IEEE division sequence (relative error ~1e-6,
// far below half an ulp of f16 / bf16); the fp32 parity mode keeps silu_f
__device__ __forceinline__ float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
template <typename T> __device__ __forceinline__ float silu_for(float x) { return silu_fast(x); }
template <> __device__ __forceinline__ float silu_for<float>(float x) { return silu_f(x); }

template <typename T> __device__ __forceinline__ float gelu_grad_for(float x) {
  return gelu_poly_grad<GeluDeg<T>::value>(x);
}
template <> __device__ __forceinline__ float gelu_grad_for<float>(float x) { return gelu_grad(x); }

template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }
template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

// Sum within aligned groups of 16 lanes (a DPP "row"); every lane ends with its row's total.
// quad_perm xor 1, xor 2, then row_half_mirror / row_mirror (the quads / halves are uniform by then):
// four v_add_f32_dpp, no LDS crossbar (ds_bpermute) round trips.
__device__ __forceinline__ float group16_sum(float v) {
#define BTS_DPP_ADD(ctrl)                                                                      \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, \
                                                             0xF, 0xF, true))
  BTS_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  BTS_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  BTS_DPP_ADD(0x141);   // row_half_mirror
  BTS_DPP_ADD(0x140);   // row_mirror
#undef BTS_DPP_ADD
  return v;
}

// 64-lane sum; every lane ends with the total.  Rows meet through v_permlane16_swap /
// v_permlane32_swap (gfx950): swap(v, v) hands every lane its own and its partner row's value.
__device__ __forceinline__ float wave_sum(float v) {
  v = group16_sum(v);
  // (inline asm: given the same value for both operands of __builtin_amdgcn_permlane16_swap,
  //  hipcc 7.2 folds its two results into one and adds a register to itself; s_nop covers the
  //  VALU-write -> permlane-read hazard the compiler would otherwise pad)
  float w = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  v += w;
  w = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  return v + w;
}

// ---------------------------------------------------------------------------------------
// Deterministic batch reductions (btsbot_set_option(h, "deterministic", 1) / BTSBOT_AMD_DETERMINISTIC=1): the kernels of
// the ConvNeXt training step that meet across workgroups through fp32 atomics -- LayerNorm / depthwise parameter
// gradients, column sums, the fc1-bias gradient of the fused MLP backward -- write one partial row per workgroup instead
// (`part`, from the scratch of the running btsbot_backward() call) and launch_det_reduce() adds the rows in a fixed order
// behind them on the same stream: two identical backward passes then agree bit for bit.  part == nullptr: atomics.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void det_add(float* out, float v, float* part, size_t slot) {
  if (part != nullptr) part[slot] = v;
  else atomicAdd(out, v);
}
struct DetOut { float* ptr; int stride; };   // column c of output o goes to ptr[c * stride]
void det_begin(float* base, size_t floats);  // the scratch of this thread's btsbot_backward() call (base == nullptr: mode off)
void det_end();
bool det_fell_short();                       // (and clears the flag) a det_alloc() since the last call found the scratch full
float* det_alloc(size_t floats);             // nullptr when the mode is off (or the scratch is exhausted: atomics then)
// outs[o].ptr[c * stride] += sum over rows r (in order) of part[r * nout * C + o * C + c]
int launch_det_reduce(const float* part, int nrows, int C, int nout, const DetOut* outs, hipStream_t st);

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == ACT_GELU) return gelu_erf(x);
  if (act == ACT_RELU) return x > 0.f ? x : 0.f;
  return x;
}

// ---------------------------------------------------------------------------------------
// launchers (one per kernel family); all return btsbot_status
// ---------------------------------------------------------------------------------------
enum { EPI_GELU = 0, EPI_RESID = 1, EPI_BIAS = 2, EPI_GELU_SAVE = 3, EPI_DGELU = 4, EPI_PLAIN = 5,
       EPI_SILU = 6, EPI_BIAS_T = 7 };

// out = epi(X[M,K] . W[N,K]^T + bias[N]);  X, W are `prec`-typed, bias/gamma/resid fp32.
//   EPI_GELU : out (prec-typed) [M,N] = gelu(acc + bias)
//   EPI_RESID: out (fp32)       [M,N] = resid + gamma[n] * (acc + bias)   (in place allowed)
//   EPI_BIAS : out (fp32)       [M,N] = acc + bias
//   EPI_SILU : out (prec-typed) [M,N] = silu(acc + bias)     (MaxViT MBConv: BatchNorm folded into W / bias)
//   EPI_BIAS_T: out (prec-typed) [M,N] = acc + bias           (MaxViT qkv projection)
// training-only epilogues (register-staged kernel; `resid` carries a prec-typed aux pointer):
//   EPI_GELU_SAVE: aux [M,N] = acc + bias (pre-activation, written), out = gelu(aux)
//   EPI_DGELU    : out (prec-typed) = acc * gelu'(aux[m][n])            (aux read, no bias)
//   EPI_PLAIN    : out (fp32) = acc
int launch_gemm(int prec, int epi, const void* X, const void* W, const float* bias,
                const float* gamma, const float* resid, void* out, int M, int N, int K,
                hipStream_t st);

// out (fp32) [M,N] = resid + (X[m][k] * gate[m / rows_per_alert][k]) . W[n][k]^T: the squeeze-excite gate of a
// MaxViT MBConv block applied to the A operand on its way to LDS (register-staged kernel); in place allowed
int launch_gemm_gated(int prec, const void* X, const float* gate, int rows_per_alert, const void* W,
                      const float* resid, float* out, int M, int N, int K, hipStream_t st);

// pipelined LDS-DMA variant for the 16-bit modes (gemm2.hip); launch_gemm dispatches to it
bool gemm2_supported(int prec, int M, int N, int K);
int launch_gemm2(int prec, int epi, const void* X, const void* W, const float* bias,
                 const float* gamma, const float* resid, void* out, int M, int N, int K,
                 hipStream_t st);

int launch_gemm2_batched_resid(int prec, const void* X, const void* W, const float* zero_bias,
                               const float* one_gamma, const float* resid, float* out, int batch, int M,
                               int N, int K, hipStream_t st, const float* ln_w = nullptr,
                               const float* ln_b = nullptr, void* ln_out = nullptr);

// stem: conv 4x4 s4 (+bias) + LayerNorm over C0.  img [B,3,63,63] fp32 -> out [B,225,C0] fp32.
// (pre_out, optional: the convolution's output before the LayerNorm, [B,225,C0] fp32 -- what the training backward needs)
int launch_stem(const float* img, const float* w48xC, const float* bias, const float* lnw,
                const float* lnb, float* out, int B, int C0, hipStream_t st, float* pre_out = nullptr);

// the same on the matrix pipe for the 16-bit modes (stem16.hip): w is the fp32 filter [C0][48], converted in registers
bool stem16_supported(int prec, int C0);
// depthwise 7x7 + LayerNorm of a 15x15 map on the matrix pipe (dw15.hip): 16-bit modes, C in {64, 80}; the map enters
// the products rounded to the operand type (launch_dwconv_ln routes the inference forward there; BTSBOT_AMD_NO_DW15=1: A/B)
bool dw15_supported(int prec, int C);
int launch_dw15_ln(int prec, const float* x, const float* wdw, const float* bdw, const float* lnw, const float* lnb,
                   void* xn, int B, int C, hipStream_t st);
int launch_stem16(int prec, const float* img, const float* w, const float* bias, const float* lnw, const float* lnb,
                  float* out, int B, int C0, hipStream_t st, float* pre_out = nullptr);

// depthwise 7x7 p3 (+bias) + LayerNorm over C.  x [B,HW*HW,C] fp32 -> xn [B,HW*HW,C] prec-typed.
// wdw is tap-major [49][C] fp32.
// dsave != NULL: also the pre-LayerNorm map d = dwconv(x) + bias, [B,HW*HW,C] fp32 (kept for the backward)
int launch_dwconv_ln(int prec, const float* x, const float* wdw, const float* bdw,
                     const float* lnw, const float* lnb, void* xn, int B, int HW, int C,
                     hipStream_t st, float* dsave = nullptr);

// downsample prologue: LayerNorm over Cin per pixel, then 2x2/s2 patch gather.
// x [B,HW,HW,Cin] fp32 -> patches [B*(HW/2)^2, 4*Cin] prec-typed, k = (ky*2+kx)*Cin + c.
int launch_ln_patch(int prec, const float* x, const float* lnw, const float* lnb, void* patches,
                    int B, int HW, int Cin, hipStream_t st);

// fused fc1 -> GELU -> fc2 -> layer-scale -> residual (fused_mlp.hip); 16-bit modes, C in {64,128} and nano's {80,160}
bool fused_mlp_supported(int prec, int C);
size_t fused_mlp_packed_bytes(int C);
int launch_pack_fused_mlp(int prec, int C, const float* w1, const float* w2, void* dst,
                          hipStream_t st);
// post_out != NULL: also post_out [M,C] (prec-typed) = LayerNorm_C(x_new) * pw + pb (post_mode 1, eps 1e-6) or
// x_new * pw + pb (post_mode 2) -- the next consumer's normalised input (MaxViT schedule)
int launch_fused_mlp(int prec, int C, const void* xn, const void* wpk, const float* b1,
                     const float* b2, const float* gamma, float* x, int M, hipStream_t st,
                     void* post_out = nullptr, const float* pw = nullptr, const float* pb = nullptr,
                     int post_mode = 0, const float* xres = nullptr);   // xres: residual source when it is not x itself

// stage-0 megakernel (stage0b.hip): stem + 2 blocks + downsample in one launch, 16-bit modes, C0 = 64
struct Stage0Args;
bool stage0_supported(int prec, int c0);

// backward kernels (backward.hip)
int launch_wgrad(int prec, const void* D, const void* A, float* out, int M, int N, int K, int ldo,
                 hipStream_t st);                                  // out[n][k] += sum_m D[m][n] A[m][k]
struct Stage0Args;
int launch_stage0b(int prec, const Stage0Args& a, hipStream_t st);   // stage0b.hip
struct Stage1Args;
int launch_stage1b(int prec, const Stage1Args& a, hipStream_t st);   // stage1b.hip
// persistent stage 2 (+ stages[3].downsample) for C = 256 -> 512, 16-bit modes (stage2p.hip)
struct Stage2pArgs;
bool stage2p_supported(int prec, int c2, int c3, int depth);
int launch_stage2p(int prec, const Stage2pArgs& a, hipStream_t st);
// the kernel's row-tile form (stage2p.hip, ROWS): x [rows][256] fp32 += W2 gelu(W1 LN(x) + b1) + b2 in place -- the MLP half
// of a MaxViT partition-attention layer at 256 channels.  blk: ln_w / ln_b = norm2, b1 / b2 the Linear biases, gamma = ones,
// w1p / w2p = launch_pack_s2p fragments of fc1 [1024][256] / fc2 [256][1024]; dw_* unused
struct Stage2pBlk;
bool stage2p_rows_supported(int prec, int c);
int launch_stage2p_rows(int prec, float* x, long rows, const Stage2pBlk& blk, hipStream_t st);
// fp32 [rows][K] (reorder_down: a [Cout][Cin][2][2] downsample filter, K = 4 Cin) -> 16x16x32 A fragments
// [row tile][k-step][lane][8], optionally scaled per row
int launch_pack_s2p(int prec, const float* src, const float* rowscale, void* dst, int rows, int K, int reorder_down,
                    int cin, float* scale, hipStream_t st);   // scale: fp8 mode only ({S, 1/S}, device)
size_t s1par_bytes();
int launch_pack_frag32(int prec, const float* src, void* dst, int cout, int cin, hipStream_t st);   // stage1b.hip
int launch_pack_s1par(int prec, const float* taps, const float* dw_b, const float* ln_w,
                      const float* ln_b, void* out, hipStream_t st);
size_t s0par_bytes();
int launch_pack_s0par(int prec, const float* taps, const float* dw_b, const float* ln_w, const float* ln_b,
                      const float* b1, const float* b2, const float* gamma, void* out,
                      hipStream_t st);
// slice reduction of a filter-gradient GEMM (wgrad.hip), described so that two of them can share a launch
struct WgradReduceJob {
  const float* part;   // partial tiles [nsl][gy][gx][TN][TK]
  float* out;          // out[n][k] += sum over the slices
  int N, K, ldo, gx, gy, nsl, TN, TK;   // nsl == 0: nothing to reduce
};

// backward of a block's MLP as one kernel (mlp_bwd.hip): 16-bit modes, C in {64, 128}
bool mlp_bwd_supported(int prec, int C);
int mlp_bwd_slices(int C, int R);
size_t mlp_bwd_part_floats(int C, int R);   // scratch for the partial filter-gradient tiles of one block
// dxn leaves as mlp_bwd_planes(C) addend planes, R * C floats apart (C = 128: one per hidden slice; the reader adds them:
// launch_dwln_bwd's nplanes)
int mlp_bwd_planes(int C);
int launch_mlp_bwd(int prec, int C, const void* xn, const void* dy, const void* w1, const void* w2g, const float* b1,
                   float* dxn, float* part, float* Gacc, float* Ssum, float* dW1, float* db1, int R, hipStream_t st,
                   WgradReduceJob* jobs);
int launch_wgrad16(int prec, const void* D, const void* A, float* out, float* colsum, int M, int N,
                   int K, int ldo, hipStream_t st, float* part = nullptr, size_t part_floats = 0,
                   WgradReduceJob* defer = nullptr);   // wgrad.hip: 16-bit modes, colsum optional
int launch_wgrad_reduce(const WgradReduceJob* jobs, int njobs, hipStream_t st);
// several filter-gradient GEMMs (N, K multiples of 128; out[n][k] += ..., colsum[n] += ... optional) as one launch + one
// slice reduction (wgrad.hip); out must be 16-byte aligned with ldo a multiple of 4
struct WgradBatchJob {
  const void* D;
  const void* A;
  float* out;
  float* colsum;
  int M, N, K, ldo;
};
int launch_wgrad16_batched(int prec, const WgradBatchJob* jobs, int njobs, float* part, size_t part_floats, int target_wg,
                           hipStream_t st);
int launch_colsum(int prec, const void* in, float* out, int M, int N, hipStream_t st);  // out[n] += ...
// fp32 mode: out[n][k] += sum_m D[m][n] A[m][k] and (cs != nullptr) cs[n] += sum_m D[m][n] (backward.hip)
int launch_wgrad_cs_f32(const float* D, const float* A, float* out, float* cs, int M, int N, int K, int ldo, hipStream_t st);
int launch_rowscale_cast(int prec, const float* in, const float* rowscale, void* out, int rows,
                         int cols, hipStream_t st);   // out[r][c] = in[r][c] * rowscale[r]
// f16 remainder image of launch_rowscale_cast(BTSBOT_F16, ...): out[r][c] = f16(v - f16(v)), v = in[r][c] * rowscale[r] (rowscale
// may be NULL: v = in[r][c]) -- the second half of a split operand (BTSBOT_F16X2)
int launch_rowscale_cast_lo(const float* in, const float* rowscale, void* out, int rows, int cols, hipStream_t st);
int launch_scale_cast(int prec, const float* in, const float* scale, void* out, long n, int C,
                      hipStream_t st);
int launch_fc2_grads(float* G, float* S, const float* w2, const float* b2,
                     const float* gamma, float* dW2, float* db2, float* dgamma, int C, int H,
                     hipStream_t st);
int launch_ln_bwd(const float* d, const float* dxn, const float* g, float* dd, float* dg,
                  float* dbeta, long rows, int C, hipStream_t st, void* out16 = nullptr, int prec16 = 0,
                  int patch_hw = 0);
// (out16: also dd in the 16-bit operand type prec16 -- the operand of the GEMM that consumes it;
//  patch_hw != 0: dxn holds the gradient of a 2x2 / stride-2 downsample's patch matrix [B (hw/2)^2][4 C] for input
//  maps of side patch_hw, gathered per input pixel -- the rows count input pixels)
int launch_dw_plain(const float* x, const float* w, int flip, const float* bias,
                    const float* addend, float* out, int B, int HW, int C, hipStream_t st,
                    void* out16 = nullptr, int prec16 = 0);
// (out16: also the result in the 16-bit operand type prec16 -- the next block's GEMM operand)
int launch_dw_wgrad(const float* x, const float* dd, float* dw, float* dbias, float* partials,
                    int B, int HW, int C, hipStream_t st);   // partials: >= 256 * 50 * C floats
// dwln_bwd.hip: the three launches above (LayerNorm backward, depthwise filter gradient, depthwise input gradient)
// as one kernel, dd never leaving LDS; dy is updated in place (dy += conv(dd, flipped taps))
// 1x1 maps: the same three steps (plus the filter gradient's column sum) as one per-(alert, channel) kernel, backward.hip
int launch_ln_dw1_bwd(const float* d, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                      void* out16, int prec16, float* dg, float* dbeta, float* dw, float* dbias, long rows, int C,
                      hipStream_t st);
bool dwln_bwd_supported(int HW, int C);
int dwln_bwd_rows(int HW, int C, int B);   // partial rows (52 * C floats each) one launch writes
// (d == nullptr: the depthwise output is recomputed from x_in, the taps and the depthwise bias dwb)
int launch_dwln_bwd(const float* d, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                    void* out16, int prec16, float* partials, int B, int HW, int C, hipStream_t st, int nplanes = 1,
                    size_t pstride = 0);
// 3x3 maps of 256 channels: their own kernel (d recomputed from x_in, compact partial rows) and its row reduction
bool dw3_bwd_active(int HW, int C);
int dw3_rows(int B);
int launch_dw3ln_bwd(const float* dwb, const float* dxn, const float* g, const float* xin, const float* w, float* dy,
                     void* out16, int prec16, float* partials, int B, hipStream_t st, int nplanes = 1, size_t pstride = 0);
int launch_dw3_rows(const float* partials, float* out, int nrows, hipStream_t st);
// the input-gradient half of a 256-channel block's MLP backward as one launch (s2mlp_bwd.hip): da = ((gamma W2)^T dy) *
// gelu'(a) [R][1024] (operand type) and dxn = W1^T da [R][256] (fp32); w2tp / w1tp = launch_pack_frag16 of the 16-bit
// dgrad transposes (diag(gamma) W2)^T [1024][256] and W1^T [256][1024]
bool s2mlp_bwd_supported(int prec, int C);
int launch_pack_frag16(const void* src, void* dst, int rows, int K, hipStream_t st);
int launch_s2mlp_bwd(int prec, const void* dy, const void* a, const void* w2tp, const void* w1tp, void* da, float* dxn, int R,
                     hipStream_t st);
int launch_stem_im2col(int prec, const float* img, void* patches, int B, hipStream_t st);
// src fp32 [R][Cc] -> dst prec-typed [Cc][R]
int launch_transpose_cast(int prec, const float* src, const float* rowscale, void* dst, int R, int Cc,
                          hipStream_t st);
int launch_pack_down_t(int prec, const float* src, void* dst, int Cout, int Cin, hipStream_t st);
// Gd [Cout][4][Cin] (q-major patches order) accumulated into dst [Cout][Cin][4] (master layout)
int launch_unpack_down_grad(float* Gd, float* dst, int Cout, int Cin, hipStream_t st);   // leaves Gd zero

bool stage1_supported(int prec, int c1, int c2);   // stage1b.hip

struct HeadArgs {
  // image feature part
  const float* feat;  // [B, feat_dim] fp32 (final 1x1 map), or nullptr
  int feat_dim;
  const float* hn_w;  // head LayerNorm (nullptr -> none)
  const float* hn_b;
  // metadata part (BatchNorm folded to scale/shift; training: scale/shift of the batch stats)
  const float* meta;  // [B, n_meta] or nullptr
  int n_meta, f1, f2;
  const float* bn_scale;
  const float* bn_shift;
  const float* m1_wt;  // [n_meta][f1]  (K-major)
  const float* m1_b;
  const float* m2_wt;  // [f1][f2]
  const float* m2_b;
  int meta_act, meta_trailing_act;
  // combined MLP: up to 3 linear layers, act between them
  int n_layers;
  int dims[4];         // dims[0] = feat_dim + f2 (or f2 / feat_dim), ..., dims[n_layers] = 1
  const float* wt[3];  // K-major [dims[i]][dims[i+1]]
  const float* b[3];
  int comb_act;
  float* logits;
  float* scores;  // may be nullptr
  int B;
  int diag;       // timing diagnostics (BTSBOT_AMD_HEAD_DIAG): 1 skip feature LN, 2 skip metadata
                  // branch, 4 skip fusion layer 0, 8 skip fusion layers 1.., 16 skip all K loops
};
int launch_head(const HeadArgs& a, hipStream_t st);

// parameter packing helpers
int launch_cast(int prec, const float* src, void* dst, int64_t n, hipStream_t st);
// fills / copies of the training step as plain kernels (train.hip says why not hipMemsetAsync / hipMemcpyAsync)
int launch_fill0(float* p, size_t n, hipStream_t st);
int launch_copy_f32(float* dst, const float* src, size_t n, hipStream_t st);
// One launch for a table of operand-packing jobs (the per-step re-pack of the training loop is ~75 of these,
// each a 2-5 us kernel: launch-floor bound one by one).  The table lives in device memory; job j owns blocks
// [blk0_j, blk0_{j+1}).  ops: the element maps of cast / transpose_f32 / transpose_cast / pack_down / pack_down_t.
// PACK_TFRAG: the transpose-cast's result [Cc][R] as 16x16x32 MFMA A fragments [row tile][k-step][lane][8] (s2mlp_bwd.hip)
// PACK_FRAG: src [R][Cc] (x scale[row]) as 16x16x32 MFMA A fragments [row tile][k-step][lane][8] (stage2p.hip's filters in the
// training re-pack); PACK_FRAG_DOWN: a downsample filter [R = Cout][Cc = Cin][2][2] likewise with k = (2 ky + kx) * Cin + c
enum { PACK_CAST = 0, PACK_TRANSPOSE_F32 = 1, PACK_TRANSPOSE_CAST = 2, PACK_DOWN = 3, PACK_DOWN_T = 4, PACK_TFRAG = 5, PACK_FRAG = 6,
       PACK_FRAG_DOWN = 7 };
struct PackJob {
  const float* src;
  const float* scale;   // PACK_TRANSPOSE_CAST: optional per-source-row scale
  void* dst;
  int R, Cc;            // CAST: R = element count, Cc = 1; DOWN / DOWN_T: R = Cout, Cc = Cin
  int op, blk0;
};
int pack_job_blocks(const PackJob& j);
int launch_pack_jobs(int prec, const PackJob* dev_jobs, int njobs, int total_blocks, hipStream_t st);
// src [R][Cc] fp32 -> dst [Cc][R] fp32
int launch_transpose_f32(const float* src, float* dst, int R, int Cc, hipStream_t st);
// downsample filter [Cout][Cin][2][2] fp32 -> [Cout][(ky*2+kx)*Cin + cin] prec-typed
int launch_pack_down(int prec, const float* src, void* dst, int Cout, int Cin, hipStream_t st);
// split mode (BTSBOT_F16X2): the same element order, f16 heads into `hi`, f16 remainders into `lo`
int launch_pack_down_split(const float* src, void* hi, void* lo, int Cout, int Cin, hipStream_t st);
// BatchNorm1d eval fold: scale = w / sqrt(rv + eps), shift = b - rm * scale
int launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                   float* shift, int n, hipStream_t st);
