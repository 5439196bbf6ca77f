// The callers either side of the classifier path:
//   btsbot_augment       on-device counterpart of FlexibleDataset.__getitem__ + the torchvision transforms
//                        of /root/reference/btsbot/train.py:178-199 and utils.py:44-48: a batch gather by
//                        index plus, per alert, RandomHorizontalFlip, RandomVerticalFlip and a right-angle
//                        rotation -- all index permutations of the 63x63 cutouts, applied in one pass
//   btsbot_eval_metrics  the scalar metrics of val.py:159-168 / train.py:550-558 (BCEWithLogitsLoss with
//                        pos_weight over ALL logits, accuracy of sigmoid(z) > 0.5) without a host round trip
#include "common.h"

namespace {

constexpr int S = 63, PLANE = S * S;

// dst[b][c][y][x] = src[index[b]][c][ys][xs] with (ys, xs) the pre-image of (y, x) under
//   rot90^k( vflip?( hflip?( img ) ) )      ops bit0 = hflip, bit1 = vflip, bits 2..3 = k (counter-clockwise)
// torch.rot90(img, 1, (-2,-1))[y][x] = img[x][S-1-y]
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ src,
                                                      const int64_t* __restrict__ index,
                                                      const uint8_t* __restrict__ ops,
                                                      float* __restrict__ dst, long total) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % S), y = (int)((i / S) % S);
  const long bc = i / PLANE;             // b*3 + c
  const long b = bc / 3;
  const int c = (int)(bc - b * 3);
  const int op = ops ? ops[b] : 0;
  int ys = y, xs = x;
  switch ((op >> 2) & 3) {               // undo the rotation
    case 1: ys = x; xs = S - 1 - y; break;
    case 2: ys = S - 1 - y; xs = S - 1 - x; break;
    case 3: ys = S - 1 - x; xs = y; break;
    default: break;
  }
  if (op & 2) ys = S - 1 - ys;
  if (op & 1) xs = S - 1 - xs;
  const long sb = index ? index[b] : b;
  dst[i] = src[(sb * 3 + c) * PLANE + ys * S + xs];
}

// out[0] += sum_i l_i (BCE with pos_weight, same stable form as bce_kernel), out[1] += #correct
__global__ __launch_bounds__(256) void eval_metrics_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ labels, float pw,
                                                           long n, float* __restrict__ out) {
  float loss = 0.f, correct = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float z = logits[i], y = labels[i];
    const float sp = log1pf(expf(-fabsf(z)));
    loss += -(pw * y * (fminf(z, 0.f) - sp) + (1.f - y) * (fminf(-z, 0.f) - sp));
    const float score = 1.f / (1.f + expf(-z));
    correct += ((score > 0.5f) == (y > 0.5f)) ? 1.f : 0.f;
  }
  loss = wave_sum(loss);
  correct = wave_sum(correct);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, loss);
    atomicAdd(out + 1, correct);
  }
}

}  // namespace

extern "C" int btsbot_augment(const float* src, const int64_t* index, const uint8_t* ops, float* dst,
                              int batch, void* stream) {
  if (src == nullptr || dst == nullptr || batch < 0) {
    btsbot_set_error("augment: NULL src/dst or negative batch");
    return BTSBOT_ERR_INVALID_ARG;
  }
  const long total = (long)batch * 3 * PLANE;
  if (total == 0) return BTSBOT_OK;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, index, ops, dst, total);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}

extern "C" int btsbot_eval_metrics(const float* logits, const float* labels, float pos_weight,
                                   int64_t n, float* out2, void* stream) {
  if (logits == nullptr || labels == nullptr || out2 == nullptr || n < 0) {
    btsbot_set_error("eval_metrics: invalid argument");
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (n == 0) return BTSBOT_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(eval_metrics_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     logits, labels, pos_weight, (long)n, out2);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
