// Which lane's scale byte multiplies which part of the A / B operand of the block-scaled fp8 MFMAs (gfx950)?
// The operand of lane l is 8 dwords; a "slot" is (lane group g = l / M, dword half hv = first / last four dwords).
// One slot of A is set to ones (B all ones, or the other way round), every lane group gets its own scale 2^g': the
// result, divided by the number of ones, names the lane group whose scale the hardware applied to that slot.
//   hipcc --offload-arch=gfx950 -O2 tools/unit/mx_probe2.hip -o tools/unit/mx_probe2 && tools/unit/mx_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: slot of A probed with lane-group scales on A; mode 1: the same for B.  opsel selects the scale byte.
template <int M> __global__ void probe(int mode, int g, int hv, int opsel, float* out) {
  const int l = threadIdx.x;
  const int ONE = 0x38383838;   // four e4m3 1.0
  v8i full, slot;
  for (int w = 0; w < 8; ++w) {
    full[w] = ONE;
    slot[w] = (l / M == g && w / 4 == hv) ? ONE : 0;
  }
  // scale register: byte `opsel` = 127 + lane group, the other bytes 127 + 7 (a wrong byte shows up as 128x)
  const int sc_group = 127 + l / M;
  int sreg = 0;
  for (int b = 0; b < 4; ++b) sreg |= (b == opsel ? sc_group : 134) << (8 * b);
  const int one = 0x7f7f7f7f;
  const v8i a = mode == 0 ? slot : full, b = mode == 0 ? full : slot;
  const int sa = mode == 0 ? sreg : one, sb = mode == 0 ? one : sreg;
  float r;
  if (M == 16) {
    f32x4 acc = {0, 0, 0, 0};
    if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    else if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 1, sa, 0, sb);
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 1, sb);
    r = acc[0];
  } else {
    f32x16 acc = {};
    if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    else if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 1, sa, 0, sb);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 1, sb);
    r = acc[0];
  }
  if (l == 0) *out = r;
}

int main() {
  float* out;
  hipMallocManaged(&out, 4);
  for (int M : {16, 32}) {
    const int groups = 64 / M;
    printf("%s\n", M == 16 ? "v_mfma_scale_f32_16x16x128_f8f6f4" : "v_mfma_scale_f32_32x32x64_f8f6f4");
    for (int opsel = 0; opsel < 2; ++opsel)
      for (int mode = 0; mode < 2; ++mode)
        for (int g = 0; g < groups; ++g)
          for (int hv = 0; hv < 2; ++hv) {
            if (M == 16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, mode, g, hv, opsel, out);
            else hipLaunchKernelGGL(probe<32>, dim3(1), dim3(64), 0, 0, mode, g, hv, opsel, out);
            hipDeviceSynchronize();
            printf("  opsel %d, %c slot (lane group %d, dwords %d..%d): D[0][0] = %g = 16 x %g\n", opsel, mode == 0 ? 'A' : 'B', g,
                   4 * hv, 4 * hv + 3, *out, *out / 16.0);
          }
  }
  return 0;
}
