// Argument block of the stage-3 kernels (stage3.hip): 1x1 maps, two launches per ConvNeXt block.
#pragma once

constexpr int S3_MAX_DEPTH = 4;

struct Stage3Blk {
  const float* dw_c;     // [C] centre taps of the 7x7 depthwise filter (all a 1x1 map ever sees)
  const float* dw_b;     // [C]
  const float* ln_w;     // [C]
  const float* ln_b;     // [C]
  const void* w1p;       // fc1 filter as 32x32x16 A fragments: [hidden tile][k-step][lane 64][8], rows bit-2/3 swapped
  const float* b1;       // [4C]
  const void* w2p;       // gamma * fc2 filter as A fragments: [channel tile][k-step][lane 64][8]
  const float* b2;       // [C]
  const float* gamma;    // [C]
  const float* scales;   // fp8 mode: {S1, 1/S1, S2, 1/S2} (device), powers of two the two filters were packed with; else nullptr
};

struct Stage3Args {
  float* x;              // [B][C] f32 residual stream, updated in place
  Stage3Blk blk[S3_MAX_DEPTH];
  int depth;
  void* hfrag;           // scratch: GELU(fc1) as fc2's B fragments [alert block of 32][k-step 4C/16][lane 64][8]
  int B;
  unsigned long long* stamps;   // optional: workgroup 0 / thread 0 stores the shader clock per phase
};

bool stage3_supported(int prec, int c3, int depth);
size_t stage3_hfrag_bytes(int prec, int c3, int B);
// phase 0: centre tap + LN + fc1 + GELU of block `block` (x -> hfrag); phase 1: fc2 + layer scale + residual (in place)
int launch_stage3(int prec, int c3, const Stage3Args& a, int block, int phase, hipStream_t st);
// fp32 [rows][K] (x rowscale[row]) -> A fragments of 32 rows x 16 k; swap23: tile row r holds source row swap23(r)
// fp8: `scale` (device, 2 floats) receives {S, 1/S}, S = the power of two that puts max |w| just below 240
int launch_pack_s3(int prec, const float* src, const float* rowscale, void* dst, int rows, int K, int swap23,
                   float* scale, hipStream_t st);
// fp8 packing helper shared with stage2p.hip: scale[0..1] <- {S, 1/S}, S = the power of two that puts
// max |src[i] * rowscale[i / K]| just below 240 (OCP e4m3's largest finite value is 448)
int launch_fp8_scale(const float* src, const float* rowscale, long n, int K, float* scale, hipStream_t st);
