// Loss and optimiser kernels of the training step.
//   BCEWithLogitsLoss(pos_weight) + its gradient   /root/reference/btsbot/train.py:211-212,525-526
//   torch.optim.AdamW.step over a flat arena       /root/reference/btsbot/train.py:242-246,527
#include "common.h"

namespace {

// l_i = -[ w*y*log(sigmoid(z)) + (1-y)*log(1-sigmoid(z)) ],  stable log-sigmoid:
// log(sigmoid(z)) = min(z,0) - log1p(exp(-|z|)).
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ logits,
                                                  const float* __restrict__ labels, float pw,
                                                  int n, float inv_n_global, float* loss_sum,
                                                  float* dlogits) {
  float part = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float z = logits[i], y = labels[i];
    const float sp = log1pf(expf(-fabsf(z)));
    const float ls_pos = fminf(z, 0.f) - sp;   // log sigmoid(z)
    const float ls_neg = fminf(-z, 0.f) - sp;  // log sigmoid(-z)
    part += -(pw * y * ls_pos + (1.f - y) * ls_neg);
    const float s = 1.f / (1.f + expf(-z));
    if (dlogits != nullptr) dlogits[i] = (((1.f - y) + pw * y) * s - pw * y) * inv_n_global;
  }
  part = wave_sum(part);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum != nullptr)
    atomicAdd(loss_sum, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p,
                                                    const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1,
                                                    float rsqrt_bc2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// Plain fills and copies of the training step as kernels of our own: the runtime's hipMemsetAsync / hipMemcpyAsync put
// barrier packets around their blit kernels, which cost the stream 4-6 us of idle time each (five of them per step).
__global__ __launch_bounds__(256) void fill0_kernel(float4* __restrict__ p, size_t n16, float* __restrict__ tail, int ntail) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void copy16_kernel(float4* __restrict__ d, const float4* __restrict__ s, size_t n16,
                                                     float* __restrict__ dt, const float* __restrict__ st, int ntail) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dt[threadIdx.x] = st[threadIdx.x];
}
__global__ __launch_bounds__(256) void fill0_scalar_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
__global__ __launch_bounds__(256) void copy_scalar_kernel(float* __restrict__ d, const float* __restrict__ s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
inline unsigned grid_for(size_t n) {
  const size_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}

}  // namespace

// n floats at p := 0 (p 4-byte aligned)
int launch_fill0(float* p, size_t n, hipStream_t st) {
  if (n == 0) return BTSBOT_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) {
    hipLaunchKernelGGL(fill0_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, n);
  } else {
    const size_t n16 = n / 4;
    hipLaunchKernelGGL(fill0_kernel, dim3(grid_for(n16)), dim3(256), 0, st, reinterpret_cast<float4*>(p), n16, p + 4 * n16,
                       (int)(n - 4 * n16));
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

// n floats from src to dst (non-overlapping, 4-byte aligned)
int launch_copy_f32(float* dst, const float* src, size_t n, hipStream_t st) {
  if (n == 0) return BTSBOT_OK;
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) != 0) {
    hipLaunchKernelGGL(copy_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, src, n);
  } else {
    const size_t n16 = n / 4;
    hipLaunchKernelGGL(copy16_kernel, dim3(grid_for(n16)), dim3(256), 0, st, reinterpret_cast<float4*>(dst),
                       reinterpret_cast<const float4*>(src), n16, dst + 4 * n16, src + 4 * n16, (int)(n - 4 * n16));
  }
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

extern "C" int btsbot_bce_fwd_bwd(const float* logits, const float* labels, float pos_weight,
                                  int batch, int n_global, float* loss_sum, float* dlogits,
                                  void* stream) {
  if (logits == nullptr || labels == nullptr || batch < 0 || n_global <= 0) {
    btsbot_set_error("bce_fwd_bwd: invalid argument");
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (batch == 0) return BTSBOT_OK;
  const int blocks = batch < 256 * 64 ? (batch + 255) / 256 : 64;
  hipLaunchKernelGGL(bce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     pos_weight, batch, 1.0f / (float)n_global, loss_sum, dlogits);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

extern "C" int btsbot_adamw_step(float* params, const float* grads, float* exp_avg,
                                 float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, void* stream) {
  if (params == nullptr || grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr ||
      n < 0 || step < 1) {
    btsbot_set_error("adamw_step: invalid argument");
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (n == 0) return BTSBOT_OK;
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)(1.0 / sqrt(bc2)));
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
