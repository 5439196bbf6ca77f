// Dependent-load latency on gfx950: one wave chases a random cycle through a buffer of the given size.
//   pass 0 = first touch after kernel start (the L2s start a kernel empty), pass 1 = the same lines again.
// Build: hipcc -O3 --offload-arch=gfx950 -o memlat memlat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const unsigned* next, int steps, unsigned long long* out) {
  unsigned p = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned long long t0 = clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    const unsigned long long t1 = clock64();
    out[pass] = t1 - t0;
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    out[2 + pass] = wall_clock64() - w0;
  }
  out[4] = p;
}

int main() {
  unsigned long long* out;
  hipMalloc(&out, 64);
  for (size_t bytes : {size_t(64) << 10, size_t(1) << 20, size_t(16) << 20, size_t(256) << 20, size_t(1) << 30}) {
    const size_t lines = bytes / 128;
    const int steps = (int)std::min<size_t>(lines, 2048);
    std::vector<unsigned> perm(lines);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin() + 1, perm.end(), rng);
    std::vector<unsigned> next(bytes / 4, 0u);
    for (size_t i = 0; i < lines; ++i) next[(size_t)perm[i] * 32] = perm[(i + 1) % lines] * 32;
    unsigned* d;
    hipMalloc(&d, bytes);
    hipMemcpy(d, next.data(), bytes, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, steps, out);
    unsigned long long h[5];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%8zu KiB, %d dependent loads: first pass %.0f cycles / load, (wall %.0f ns), second pass %.0f cycles (wall %.0f ns)\n",
           bytes >> 10, steps, (double)h[0] / steps, (double)h[2] * 10.0 / steps, (double)h[1] / steps, (double)h[3] * 10.0 / steps);
    fflush(stdout);
    hipFree(d);
  }
  return 0;
}
