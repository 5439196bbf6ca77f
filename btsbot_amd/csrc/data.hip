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

// ------------------------------------------------------------------------------------------------
// btsbot_prep_triplets: the arithmetic of make_triplet (/root/reference/btsbot/alert_utils.py:110-196) after
// the host has gunzipped and FITS-decoded the three stamps of every alert -- per cutout, in the order
// science, template, difference:
//   median test   drop the alert if nanmedian(data) is +-inf                         (:152-161)
//   nan_to_num    NaN -> 0, +-inf -> +-FLT_MAX                                        (:164)
//   L2 normalise  data /= ||data||_2, skipped once the alert is flagged              (:167-168)
//   zero test     drop if every value is 0                                            (:171-177)
//   pad           to 63x63 at the bottom / right with 1e-9                            (:180-192)
// and the float32 NCHW layout inference_example.py:62-64 hands to the model.  One workgroup per alert
// (the drop flag carries from one cutout to the next, exactly as in the reference's loop).
namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void prep_triplets_kernel(const float* __restrict__ raw,
                                                            const int* __restrict__ shapes,
                                                            float* __restrict__ out,
                                                            uint8_t* __restrict__ drop, int normalize) {
  __shared__ float sh[4];
  const long b = blockIdx.x;
  bool dropped = false;
  for (int c = 0; c < 3; ++c) {
    const int h = shapes ? shapes[(b * 3 + c) * 2] : S, w = shapes ? shapes[(b * 3 + c) * 2 + 1] : S;
    const float* src = raw + (b * 3 + c) * PLANE;
    float n = 0.f, pinf = 0.f, ninf = 0.f, ssq = 0.f;
    for (int i = threadIdx.x; i < PLANE; i += 256) {
      const int y = i / S, x = i - y * S;
      if (y >= h || x >= w) continue;
      float v = src[i];
      if (v != v) continue;                        // NaN: ignored by nanmedian, 0 after nan_to_num
      n += 1.f;
      if (v == INFINITY) { pinf += 1.f; v = 3.4028234663852886e38f; }
      else if (v == -INFINITY) { ninf += 1.f; v = -3.4028234663852886e38f; }
      ssq += v * v;
    }
    n = block_sum(n, sh);
    pinf = block_sum(pinf, sh);
    ninf = block_sum(ninf, sh);
    ssq = block_sum(ssq, sh);
    // nanmedian == +-inf  <=>  the (upper) middle order statistic is infinite and the pair does not cancel
    const int ni = (int)n, lo_cnt = ni - ni / 2;     // elements from the upper-middle one to the top
    const bool med_pinf = ni > 0 && (int)pinf >= lo_cnt && !((ni % 2 == 0) && (int)ninf >= ni / 2);
    const bool med_ninf = ni > 0 && (int)ninf >= lo_cnt && !((ni % 2 == 0) && (int)pinf >= ni / 2);
    if (med_pinf || med_ninf) dropped = true;
    const bool do_norm = normalize && !dropped;
    const float norm = sqrtf(ssq);
    float nz = 0.f;
    float* dst = out + (b * 3 + c) * PLANE;
    for (int i = threadIdx.x; i < PLANE; i += 256) {
      const int y = i / S, x = i - y * S;
      float v = 1e-9f;                               // padding value
      if (y < h && x < w) {
        v = src[i];
        if (v != v) v = 0.f;
        else if (v == INFINITY) v = 3.4028234663852886e38f;
        else if (v == -INFINITY) v = -3.4028234663852886e38f;
        if (do_norm) v = v / norm;                   // 0/0 -> NaN for an all-zero stamp, as numpy does
        if (v != 0.f) nz += 1.f;
      }
      dst[i] = v;
    }
    nz = block_sum(nz, sh);
    if (nz == 0.f) dropped = true;
  }
  if (threadIdx.x == 0 && drop != nullptr) drop[b] = dropped ? 1 : 0;
}

}  // namespace

extern "C" int btsbot_prep_triplets(const float* raw, const int* shapes, float* triplets, uint8_t* drop,
                                    int batch, int normalize, void* stream) {
  if (raw == nullptr || triplets == nullptr || batch < 0) {
    btsbot_set_error("prep_triplets: NULL raw/triplets or negative batch");
    return BTSBOT_ERR_INVALID_ARG;
  }
  if (batch == 0) return BTSBOT_OK;
  hipLaunchKernelGGL(prep_triplets_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, raw, shapes,
                     triplets, drop, normalize);
  LAUNCH_CHECK();
  return BTSBOT_OK;
}
