// Unit check of common.h's cross-lane reductions on the device.
#include "../../btsbot_amd/csrc/common.h"
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out_w, float* out_g) {
  const float v = in[threadIdx.x];
  out_w[threadIdx.x] = wave_sum(v);
  out_g[threadIdx.x] = group16_sum(v);
}
__global__ void k9(const float* in, float* out) {   // nine back-to-back reductions, as dw3_ln does
  float acc[9];
  for (int p = 0; p < 9; ++p) acc[p] = in[threadIdx.x] * (p + 1);
  float r = 0.f;
#pragma unroll
  for (int p = 0; p < 9; ++p) {
    const float s = wave_sum(acc[p]);
    if ((threadIdx.x & 63) == 0) r += s;
  }
  out[threadIdx.x] = r;
}
int main() {
  const int n = 256;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)((i * 37) % 101) - 50.f;
  float *d, *w, *g, *o9;
  hipMalloc(&d, n * 4); hipMalloc(&w, n * 4); hipMalloc(&g, n * 4); hipMalloc(&o9, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<1, n>>>(d, w, g);
  k9<<<1, n>>>(d, o9);
  std::vector<float> hw(n), hg(n), h9(n);
  hipMemcpy(hw.data(), w, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hg.data(), g, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h9.data(), o9, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float sw = 0, sg = 0;
    for (int j = 0; j < 64; ++j) sw += h[(i / 64) * 64 + j];
    for (int j = 0; j < 16; ++j) sg += h[(i / 16) * 16 + j];
    if (hw[i] != sw || hg[i] != sg) { if (bad < 8) printf("lane %d: wave %g (want %g) group %g (want %g)\n", i, hw[i], sw, hg[i], sg); ++bad; }
    if ((i & 63) == 0 && h9[i] != 45.f * sw) { printf("k9 wave %d: %g want %g\n", i / 64, h9[i], 45.f * sw); ++bad; }
  }
  printf("%s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
  return bad != 0;
}
