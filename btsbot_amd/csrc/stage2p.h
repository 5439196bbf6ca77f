// Argument block of the persistent stage-2 kernel (stage2p.hip).
#pragma once

constexpr int S2P_ALERTS = 4;      // alerts resident per workgroup
constexpr int S2P_MAX_DEPTH = 8;   // blocks of the stage (pico: 6 at 256 channels, nano: 8 at 320)

struct Stage2pBlk {
  const float* dw_w;     // [49][256] tap-major fp32
  const float* dw_b;
  const float* ln_w;
  const float* ln_b;
  const float* b1;       // [1024]
  const float* b2;       // [256]
  const float* gamma;    // [256]
  const void* w1p;       // fc1 filter as MFMA A fragments: [hidden tile 64][k-step 8][lane 64][8]
  const void* w2p;       // gamma * fc2 filter as A fragments: [channel tile 16][k-step 32][lane 64][8]
  const float* scales;   // fp8 mode: {S1, 1/S1, S2, 1/S2} (device): the powers of two the two filters were packed with
};
// Training forward (stage2p_kernel<T, G, 1>): what the backward reads of block j, written on the way -- backbone_train.hip's
// BlkBuf rows [(B + 9) * 9][...] (room for 9 alerts behind the batch: the kernel stores every row unconditionally): the
// block's input (block 0: any buffer of that size, its input is x_in itself), the LayerNorm output, the rounded fc1
// pre-activation and its GELU (operand type).  The depthwise output is not kept (dw3ln_bwd_kernel recomputes it).
struct Stage2pKeep {
  float* xin;
  void* xn;
  void* a;
  void* hh;
};
struct Stage2pArgs {
  const float* x_in;     // [B][9][256] f32 (stage-1 output after its downsample)
  Stage2pBlk blk[S2P_MAX_DEPTH];
  int depth;
  const float* ds_lnw;   // stages[3].downsample: LayerNorm2d(256) + Conv2d(256, 512, 2, 2)
  const float* ds_lnb;
  const void* ds_wp;     // (16-bit in the fp8 mode too) A fragments: [output tile 32][k-step 32][lane 64][8], k = (2 ky + kx) * 256 + c
  const float* ds_b;
  float* out;            // [B][512] f32
  float* tap_stage;      // optional [B][9][256] f32 copy of the stage output (validation)
  int B;
  int cw;                // the stage's width: 256 (pico: dims above) or 320 (convnext_nano: [B][9][320] -> [B][640], 16-bit inference)
  unsigned long long* stamps;   // optional: workgroup 0 / thread 0 stores the shader clock per phase (64 entries)
  int diag;              // developer switches (BTSBOT_AMD_S2P_DIAG): 1 barrier at every chunk start, 2 drain loads there
  int alerts_hint;       // 0: the library picks 5 or 7 alerts per workgroup by rounds; 4 / 5 / 7: the caller's choice
                         // (btsbot_set_option "stage2p_alerts")
  int train;             // 1: the training forward (16-bit modes, 256 channels): keep[] / ds_patches are written; tap_stage =
                         // the stage output the backward keeps
  Stage2pKeep keep[S2P_MAX_DEPTH];
  void* ds_patches;      // [B][4 * 256] operand type: the downsample's LayerNorm'd patch rows, k = (2 ky + kx) * 256 + c
};

