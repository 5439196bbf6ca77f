// Argument block of the stage-2 megakernel (stage2m.hip).
#pragma once

struct Stage2Blk {
  const float* dw_w;          // [49][256] tap-major
  const float* dw_b;
  const float* ln_w;
  const float* ln_b;
  const unsigned char* w1;    // plain 16-bit [1024][256]
  const unsigned char* w2g;   // gamma-scaled 16-bit fc2 filter, chunk-major [32][256][32]
  const float* b1;            // [1024]
  const float* b2;            // [256]
  const float* gamma;         // [256]
};
struct Stage2Args {
  const float* x_in;          // [B][9][256] f32
  float* out;                 // [B][9][256] f32 (may alias x_in)
  Stage2Blk blk[8];
  int depth;
  int B;
  int diag;                   // timing diagnostics (BTSBOT_AMD_S2_DIAG): 1 skip the chunk math, 2 skip the DMA
};

