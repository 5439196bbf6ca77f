// Internal: parameter / operand tables of the MaxViT image branch (maxvit.hip builds them; maxvit_train.hip reads them).
#pragma once
#include <stdint.h>
#include <stddef.h>

#include <utility>
#include <vector>

struct BnPk {
  int64_t w, b, rm, rv;      // master offsets
  size_t p_scale, p_shift;   // folded scale / shift (fp32) in `extra`
};
struct AttnPk {
  int64_t n1w, n1b, qkv_w, qkv_b, rel, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b;
  size_t p_qkv, p_proj, p_fc1, p_fc2, p_bias, p_bias64, p_fused;
  size_t p_w1p = 0, p_w2p = 0;   // C = 256, 16-bit modes: fc1 / fc2 as stage2p.hip's MFMA A fragments (the streamed MLP)
  bool fused;
  bool smlp = false;             // the MLP runs stage2p_kernel's row-tile form (launch_stage2p_rows)
  size_t p_qkvp = 0, p_projp = 0, p_biasl = 0;   // C = 128 / 256, 16-bit modes: attn.qkv / attn.proj as MFMA A fragments, the bias in lane order (maxvit_part.hip)
  bool part = false;             // the partition block (attention half + MLP half) runs as one launch
};
struct MvBlock {
  int cin, c, mid, rd, stride, hin, hout;
  int64_t sc_w = -1;
  BnPk pre, n1, n2;
  int64_t c1_w, c1_b, c2_w, c2_b, se1_w, se1_b, se2_w, se2_b, c3_w;
  size_t p_sc, p_c1, p_c1b, p_dw, p_dwb, p_c3, p_se2t;
  AttnPk attn[2];   // [0] windows ("attn_block"), [1] grid ("attn_grid")
};

struct MaxVit {
  int64_t stem1_w, stem2_w, norm_w, norm_b;
  BnPk stem_bn;
  size_t p_stem1, p_stem2, p_zero, p_one;
  std::vector<MvBlock> blocks;
  // workspace offsets (bytes) for the current reservation
  size_t o_x, o_x2, o_a, o_b, o_c, o_d, o_e, o_gate, o_feat, o_part, o_sescr, o_wg;
  // training, 16-bit modes: (fp32 GEMM input of the forward, its kept 16-bit copy) in call order (maxvit_train.hip)
  std::vector<std::pair<const float*, void*>> xkept;
  bool no_part = false;     // BTSBOT_AMD_MV_NO_PART=1: the partition blocks of C = 64 / 128 / 256 launch by launch (A/B, parity tests)
  bool no_smlp = false;     // BTSBOT_AMD_MV_NO_SMLP=1: the 256-channel MLPs as LayerNorm + two GEMMs (A/B, parity tests)
  bool mlp_unfused = false; // BTSBOT_AMD_MV_MLP_UNFUSED=1: fc1 / fc2 GEMM pair also where the fused MLP kernel applies
  bool stem_im2col = false; // BTSBOT_AMD_MV_STEM_IM2COL=1: im2col + GEMM for the second stem conv in the 16-bit modes too
                            // (measured slower than gemm2 on these shapes: opt-in, kept as the record)
  bool gated_gemm = false;  // BTSBOT_AMD_MV_GATED_GEMM=1: register-staged gated GEMM for every conv3 (f32 mode's path)
  bool no_front = false;    // BTSBOT_AMD_MV_NO_FRONT=1: conv1 GEMM + depthwise kernel instead of the fused MBConv front
  bool no_ln_fuse = false;  // BTSBOT_AMD_MV_NO_LN_FUSE=1: separate LayerNorm launches everywhere
  bool no_attn_block = false;  // BTSBOT_AMD_MV_NO_ATTN_BLOCK=1: qkv GEMM + attention + proj GEMM at C = 64 too
                            // for the K = 128 / 256 wide-N shapes (measured 20-25 % slower than gemm2: opt-in)
  bool dw_plain = false;    // BTSBOT_AMD_MV_DW_PLAIN=1: per-pixel depthwise kernel + separate pool pass
  bool attn_valu = false;   // BTSBOT_AMD_MV_ATTN_VALU=1: the one-query-per-lane kernel in the 16-bit modes too
};

