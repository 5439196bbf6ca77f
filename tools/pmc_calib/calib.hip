// FETCH_SIZE / WRITE_SIZE calibration on known byte counts, in the access patterns this library uses
// (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 calib.hip -o calib ; rocprofv3 --pmc FETCH_SIZE ... -- ./calib
// Every kernel reads exactly `alerts * 3*63*63*4` bytes (or the stated subset) once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct f4u { float v[4]; } __attribute__((packed, aligned(4)));

__global__ void read_dword(const float* __restrict__ p, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void read_dwordx4(const float4* __restrict__ p, size_t n4, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}
// stage0_kernel's stem gather: workgroup = alert, 8 waves, lane -> (pixel = wave*32 + lane&31, h = lane>>5),
// per input channel two unaligned 16-byte reads (rows 4py+2h, +1) at column 4px.  Reads 60x60 of 63x63.
__global__ __launch_bounds__(512) void read_stem(const float* __restrict__ img, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int p = wave * 32 + lr;
  const int pc = p < 225 ? p : 0;
  const int py = pc / 15, px = pc - py * 15;
  const float* src = img + (size_t)blockIdx.x * 3 * 63 * 63;
  float s = 0.f;
  for (int ci = 0; ci < 3; ++ci) {
    const float* r0 = src + (ci * 63 + 4 * py + 2 * h) * 63 + 4 * px;
    const f4u v0 = *reinterpret_cast<const f4u*>(r0);
    const f4u v1 = *reinterpret_cast<const f4u*>(r0 + 63);
    for (int e = 0; e < 4; ++e) s += v0.v[e] + v1.v[e];
  }
  if (s == 12345.678f) out[0] = s;
}
// 16-byte-per-lane streaming store / 4-byte-per-lane store with a 512-byte row stride (MFMA accumulator layout)
__global__ void write_x4(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write_strided4(float* p, size_t rows) {   // row = 128 floats; lane = row, 128 passes over columns
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x)
    for (int c = 0; c < 128; ++c) p[r * 128 + c] = 1.f;
}

int main(int argc, char** argv) {
  const size_t alerts = argc > 1 ? atoll(argv[1]) : 1024;
  const size_t n = alerts * 3 * 63 * 63;
  float *buf, *out;
  CK(hipMalloc(&buf, n * 4 + 64));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, n * 4 + 64));
  // flush the caches between kernels with a 1 GiB memset
  void* flush; CK(hipMalloc(&flush, 1ull << 30));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(flush, rep, 1ull << 30));
    read_dword<<<2048, 256>>>(buf, n, out);
    CK(hipMemset(flush, rep, 1ull << 30));
    read_dwordx4<<<2048, 256>>>(reinterpret_cast<const float4*>(buf), n / 4, out);
    CK(hipMemset(flush, rep, 1ull << 30));
    read_stem<<<alerts, 512>>>(buf, out);
    CK(hipMemset(flush, rep, 1ull << 30));
    write_x4<<<2048, 256>>>(reinterpret_cast<float4*>(buf), n / 4);
    CK(hipMemset(flush, rep, 1ull << 30));
    write_strided4<<<2048, 256>>>(buf, n / 128);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per kernel: %zu (stem pattern touches %zu)\n", n * 4, alerts * 3 * 60 * 60 * 4);
  return 0;
}
