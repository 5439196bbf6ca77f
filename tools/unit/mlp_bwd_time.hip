// Development harness: times mlp_bwd_kernel alone (random operands) with phases left out (-DMLP_BWD_ABL=mask).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DMLP_BWD_ABL=0 tools/unit/mlp_bwd_time.hip -o mlp_bwd_time
#include <stdarg.h>
#include <vector>
#include "../../btsbot_amd/csrc/mlp_bwd.hip"
void btsbot_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int launch_wgrad_reduce(const WgradReduceJob*, int, hipStream_t) { return 0; }
int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 64, R = argc > 2 ? atoi(argv[2]) : 230400, H = 4 * C;
  std::vector<unsigned short> hx((size_t)R * C), hw((size_t)H * C);
  for (auto& v : hx) v = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);   // ~ +-0.01 .. 0.03 as bf16
  for (auto& v : hw) v = 0x3d00 + (rand() & 0xff) + ((rand() & 1) << 15);
  void *xn, *dy, *w1, *w2; float *b1, *dxn, *part, *G, *dW, *db;
  hipMalloc(&xn, hx.size() * 2); hipMalloc(&dy, hx.size() * 2); hipMalloc(&w1, hw.size() * 2); hipMalloc(&w2, hw.size() * 2);
  hipMalloc(&b1, H * 4); hipMalloc(&dxn, (size_t)R * C * 4 * mlp_bwd_planes(C)); hipMalloc(&part, mlp_bwd_part_floats(C, R) * 4);
  hipMalloc(&G, (size_t)C * H * 4); hipMalloc(&dW, (size_t)C * H * 4); hipMalloc(&db, H * 4);
  hipMemcpy(xn, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dy, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(b1, 0, H * 4); hipMemset(db, 0, H * 4);
  WgradReduceJob jobs[2];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_mlp_bwd(BTSBOT_BF16, C, xn, dy, w1, w2, b1, dxn, part, G, db, dW, db, R, 0, jobs);
  hipEventRecord(e0, 0);
  const int N = 20;
  for (int i = 0; i < N; ++i) launch_mlp_bwd(BTSBOT_BF16, C, xn, dy, w1, w2, b1, dxn, part, G, db, dW, db, R, 0, jobs);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
#if MLP_BWD_STAMP
  {
    unsigned long long hs[8 * 16];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(mlp_bwd_stamps), sizeof(hs));
    const char* names[] = {"dxn(t-1)", "A(0)", "unit 0: A(1) + GELU", "unit 1: C(0) + GELU", "C(1)", "colsum(dy) [+ dxn(t-1) in waves 0-3]", "stash", "barrier"};
    for (int w = 0; w < 8; w += (C == 64 ? 1 : 2)) {
      printf("wave %d (shader clocks):", w);
      for (int i = 0; i < 8; ++i) printf("  %s %lld", names[i], (long long)(hs[w * 16 + i + 1] - hs[w * 16 + i]));
      printf("  | tile %lld\n", (long long)(hs[w * 16 + 8] - hs[w * 16 + 0]));
    }
  }
#endif
  printf("C=%d R=%d abl=%d: %.1f us per launch (%s)\n", C, R, MLP_BWD_ABL, ms * 1000 / N, hipGetErrorString(hipGetLastError()));
  return 0;
}
