// Argument blocks of the stage-0 / stage-1 megakernels (stage0b.hip, stage1b.hip).
#pragma once

struct Stage0Blk {
  const float* dw_w;   // [49][64] tap-major
  const float* dw_b;
  const float* ln_w;
  const float* ln_b;
  const float* b1;
  const float* b2;
  const float* gamma;
  const unsigned char* w1;    // plain 16-bit fc1 filter [4C][C]              (stage0b.hip / stage1b.hip)
  const unsigned char* par;   // fp32 parameter image of the block (stage0b.hip: launch_pack_s0par)
  const unsigned char* w2g;   // gamma-scaled 16-bit fc2 filter [C][4C]       (stage0b.hip / stage1b.hip)
  const unsigned char* w1_lo;   // split mode (BTSBOT_F16X2): the f16 remainders of w1 / w2g, same layouts; else unused
  const unsigned char* w2g_lo;
};
struct Stage0Args {
  const float* img;       // [B][3][63][63]
  const void* stem_w;     // [64][48] 16-bit
  const void* stem_w_lo;  // split mode: its f16 remainders
  const float* stem_b;
  const float* stem_lnw;
  const float* stem_lnb;
  Stage0Blk blk[2];
  const float* ds_lnw;
  const float* ds_lnb;
  const void* ds_w;       // [128][256] 16-bit, k = (ky*2+kx)*64 + c
  const void* ds_w_lo;    // split mode (BTSBOT_F16X2): the f16 remainders of ds_w, same layout; else unused
  const float* ds_b;
  float* out;             // [B][49][128] f32
  float* tap_stem;        // optional [B][225][64] f32 copies (validation)
  float* tap_stage;
  int B;
  unsigned long long* stamps;   // optional: workgroup 0 / thread 0 writes s_memtime at phase ends
  unsigned long long* wgt;      // optional: every workgroup's start / end (100 MHz wall clock), [grid][2]
  int diag;               // timing diagnostics only (BTSBOT_AMD_S0_DIAG): bit0 skip depthwise FMAs,
                          // bit1 skip fc1/GELU/fc2, bit2 skip LDS-DMA of the filters, bit3 skip GELU
  // Training forward (keep_xn[0] != nullptr; bf16 / f16): what the backward reads is written on the way --
  // backbone_train.hip's buffers: the stem convolution's output before its LayerNorm, block 0's input = tap_stem,
  // block 1's input, the stage output = tap_stage, per block the depthwise output before the LayerNorm (fp32) and
  // the LayerNorm output (operand type), and the downsample's LayerNorm'd patch rows [B][49][q = 2 ky + kx][64].
  float* keep_stem_pre;   // [B][225][64] f32
  float* keep_xin1;       // [B][225][64] f32
  float* keep_d[2];       // [B][225][64] f32, or both nullptr: the backward recomputes it (dwln_bwd.hip)
  void* keep_xn[2];       // [B][225][64] operand type
  void* keep_patches;     // [B][49][256] operand type
};


// Stage-1 megakernel (stage1b.hip): two alerts per workgroup, C = 128.
struct Stage1Args {
  const float* x_in;      // [B][49][128] f32 (stage-0 output after its downsample)
  Stage0Blk blk[2];       // dw_w is [49][128]
  const float* ds_lnw;
  const float* ds_lnb;
  const void* ds_w;       // 32x32x16 A fragments of the [256][512] filter, k = (ky*2+kx)*128 + c (launch_pack_frag32);
                          // split mode (BTSBOT_F16X2): 2 KiB per fragment, heads then remainders
  const float* ds_b;
  float* out;             // [B][9][256] f32
  float* scratch;         // [B][49][128] f32: the residual stream between the two blocks (stage1b.hip)
  float* tap_stage;       // optional [B][49][128] f32 copy of the stage output (validation)
  int B;
  unsigned long long* stamps;   // optional phase timestamps (workgroup 0, thread 0)
  unsigned long long* wgt;      // optional per-workgroup start / end, [grid][2]
  int diag;               // timing diagnostics, same bits as Stage0Args::diag
  // Training forward (keep_xn[0] != nullptr; bf16 / f16): block 0's input is x_in itself, block 1's input is what the
  // kernel parks in `scratch` anyway (point it at that buffer), the stage output is tap_stage; per block the depthwise
  // output before the LayerNorm (fp32) and the LayerNorm output (operand type), and the downsample's LayerNorm'd patch
  // rows [B][9][q = 2 ky + kx][128]
  float* keep_d[2];       // [B][49][128] f32, or both nullptr (as above)
  void* keep_xn[2];       // [B][49][128] operand type
  void* keep_patches;     // [B][9][512] operand type
};
