// Internal: MaxViT image branch (timm maxvit_tiny_rw_224 reached from
// /root/reference/btsbot/architectures.py:31,62) -- launchers of maxvit_ops.hip and the schedule
// entry points of maxvit.hip.
#pragma once
#include "common.h"

struct btsbot_ctx;

// ---- schedule (maxvit.hip), called from api.hip
int maxvit_build_tables(btsbot_ctx* h, size_t* extra_cursor);            // parameter table + operand images
int maxvit_pack(btsbot_ctx* h, hipStream_t st);                          // mirror -> operand images
size_t maxvit_ws_bytes(const btsbot_ctx* h, int chunk);
int maxvit_chunk(btsbot_ctx* h, const float* img, int nb, hipStream_t st, float** feat_out);
void maxvit_free(btsbot_ctx* h);
// ---- training of the branch (maxvit_train.hip): BatchNorm2d batch statistics forward, backward of every layer; an
//      fp32 engine whatever the handle's operand mode.  cache = h->bbcache, sized by maxvit_train_cache_bytes(B)
size_t maxvit_train_cache_bytes(const btsbot_ctx* h, int B);
int maxvit_train_forward(btsbot_ctx* h, const float* img, int B, float* master_arena, hipStream_t st, float** feat_out);
int maxvit_train_backward(btsbot_ctx* h, const float* img, const float* dfeat, float* grads, int B, hipStream_t st);
constexpr int MV_MAX_CHUNK = 1024;  // alerts per workspace chunk the host asks for (about 22 MB of activations each in bf16)

// ---- kernels (maxvit_ops.hip).  `prec` selects the staged activation type T (float / bf16 / f16).
// bilinear 63 -> 224 (align_corners=False, architectures.py:44-50) fused with the im2col of the stem's
// 3x3 s2 p1 convolution: img [B,3,63,63] f32 -> out [B*112*112, 32] T, k = (ky*3+kx)*3 + c, 27..31 zero
int launch_mv_resize_im2col(int prec, const float* img, void* out, int B, hipStream_t st);
// 16-bit modes: resize + stem conv 3x3 s2 + BN + SiLU directly (no im2col): img -> out [B*112*112, 32] T;
// w = the [32][32] image of launch_mv_pack_stem1, shift = folded BatchNorm shift
int launch_mv_stem1(int prec, const float* img, const void* w, const float* shift, void* out, int B,
                    hipStream_t st);
// im2col of a 3x3 s1 p1 convolution on an NHWC map: in [B,HW,HW,C] T -> out [B*HW*HW, 9*C] T
int launch_mv_im2col3(int prec, const void* in, void* out, int B, int HW, int C, hipStream_t st);
// eval-mode BatchNorm2d as scale/shift + cast: x [M,C] f32 -> out [M,C] T
int launch_mv_bn_cast(int prec, const float* x, const float* scale, const float* shift, void* out,
                      long M, int C, hipStream_t st);
// depthwise 3x3 p1 (stride 1 or 2) + folded BatchNorm + SiLU: in [B,H,H,C] T -> out [B,H/s,H/s,C] T;
// w9 is tap-major [9][C] f32 (BatchNorm scale folded in), bias [C] f32 (conv bias and BN shift folded)
int launch_mv_dw3(int prec, const void* in, const float* w9, const float* bias, void* out, int B,
                  int H, int C, int stride, hipStream_t st);
// 16-bit modes: the same in strips of 7 outputs x 8 channels per thread, plus the squeeze-excite pool as
// per-workgroup partial sums: part [B][mv_dw3s_groups(H,C,stride)][C] f32 (feed to launch_mv_se as f32 rows)
int mv_dw3s_groups(int H, int C, int stride);
int launch_mv_dw3s(int prec, const void* in, const float* w9, const float* bias, void* out, float* part,
                   int B, int H, int C, int stride, hipStream_t st);
// conv1 1x1 + BN + SiLU + depthwise 3x3 + BN + SiLU + SE-pool partials in one kernel (maxvit_mbconv.hip;
// 16-bit modes, C_in 64 / 128, output maps >= 28x28): xn [B,H,H,CIN] T -> m2 [B,H/s,H/s,MID] T,
// part [B][mv_mbconv_front_tiles(H,stride)][MID] f32
bool mv_mbconv_front_supported(int prec, int H, int CIN, int MID, int stride);
int mv_mbconv_front_tiles(int H, int stride);
int launch_mv_mbconv_front(int prec, const void* xn, const void* w1, const float* b1, const float* w9,
                           const float* b2, void* m2, float* part, int B, int H, int CIN, int MID,
                           int stride, hipStream_t st);
// squeeze-excite gate: y [B,HW,C] T -> gate [B,C] f32 = sigmoid(fc2(silu(fc1(inv_count * sum_hw y))))
// (w2t = fc2 weight transposed to [RD][C]; scratch = B * (C + RD) floats)
int launch_mv_se(int prec, const void* y, const float* w1, const float* b1, const float* w2t,
                 const float* b2, float* gate, float* scratch, int B, int HW, int C, int RD,
                 float inv_count, hipStream_t st);
// y [B,HW,C] T *= gate [B,C] in place (16-bit modes)
int launch_mv_gate(int prec, void* y, const float* gate, int B, int HW, int C, hipStream_t st);
// wg [B][N][K] T = w [N][K] f32 * gate [B][K]
int launch_mv_scale_w(int prec, const float* w, const float* gate, void* wg, int B, int N, int K,
                      hipStream_t st);
// 2x2 average pool of the fp32 residual map: x [B,H,H,C] -> out [B,H/2,H/2,C] (T when to_t, else f32)
int launch_mv_avgpool2(int prec, const float* x, void* out, int to_t, int B, int H, int C,
                       hipStream_t st);
// LayerNorm over C (eps 1e-6): x [M,C] f32 -> out [M,C] T;  C in {64,128,256,512}
int launch_mv_ln(int prec, const float* x, const float* w, const float* b, void* out, long M, int C,
                 hipStream_t st);
// multi-head self-attention inside 7x7 windows (grid_mode 0) or on the 7x7 dilated grid (grid_mode 1):
// qkv [B*H*H, 3C] T with channel order [head][q|k|v][32], bias_t [heads][49 (key)][49 (query)] f32 ->
// out [B*H*H, C] T (channel = head*32 + d); rows stay in image order, the partition is an index map
int launch_mv_attn(int prec, const void* qkv, const float* bias_t, void* out, int B, int H, int C,
                   int grid_mode, hipStream_t st);
// the same on MFMA for the 16-bit modes (maxvit_attn.hip); bias64 is the padded [heads][64 key][64 query]
// image of launch_mv_pack_relbias64 (key padding mask folded in)
int launch_mv_attn_mfma(int prec, const void* qkv, const float* bias64, void* out, int B, int H, int C,
                        int grid_mode, hipStream_t st);
int launch_mv_pack_relbias64(const float* table, float* out, int heads, hipStream_t st);
// stem conv 3x3 s1 p1 32 -> 64 as an LDS-free implicit GEMM (16-bit modes): in [B,112,112,32] T,
// w = the [64][288] image of launch_mv_pack_conv3, out [B,112,112,64] f32
// (xn != NULL: also xn [B,112,112,64] T = result * scale[c] + shift[c], the next block's pre-norm;
//  pooled != 0: out is the 2x2 average pool [B,56,56,64] f32 of the result instead of the full map)
int launch_mv_stem2(int prec, const void* in, const void* w, float* out, int pooled, void* xn,
                    const float* scale, const float* shift, int B, hipStream_t st);
// qkv + window/grid attention + proj + residual + LN2 in one kernel (maxvit_attnblock.hip; 16-bit modes,
// C = 64): xn [B*H*H,64] T = LN1(x) in, x f32 updated in place, xn2 [B*H*H,64] T = LN2(x) out (may alias xn)
bool mv_attn_block_supported(int prec, int C);
int launch_mv_attn_block(int prec, const void* xn, float* x, void* xn2, const void* wqkv,
                         const float* bqkv, const void* wproj, const float* bproj, const float* bias64,
                         const float* ln_w, const float* ln_b, int B, int H, int C, int grid_mode,
                         hipStream_t st);
// A partition block in one kernel at C = 128 / 256 (maxvit_part.hip; 16-bit modes): x f32 updated in place by the attention
// half (norm1, qkv, window/grid attention, proj, residual) and, with w1p set, the MLP half (norm2, fc1, GELU, fc2, residual).
// The filters are launch_pack_s2p fragments of attn.qkv.weight [3C][C], attn.proj.weight [C][C], mlp.fc1.weight [4C][C],
// mlp.fc2.weight [C][4C]; biasl is launch_mv_pack_relbias_lanes' image.
struct MvPartW {
  const float *ln1w, *ln1b, *bqkv, *bproj, *biasl;
  const void *wqkvp, *wprojp;
  const float *ln2w, *ln2b, *b1, *b2;
  const void *w1p, *w2p;
  const float *post_s, *post_b;   // with post_out (a 16-bit map shaped like x): also out = x * post_s[c] + post_b[c], the next
  void* post_out;                 // block's pre-norm BatchNorm copy
  unsigned long long* stamps;   // developer diagnostic: 32 phase clocks of workgroup 0 (nullptr: none)
};
bool mv_part_supported(int prec, int C);
int launch_mv_part(int prec, float* x, const MvPartW& p, int B, int H, int C, int grid_mode, hipStream_t st);
// biasl [heads][4][4][64][4] f32: the relative-position bias (+ key mask) in the order the attention wave's lanes add it
int launch_mv_pack_relbias_lanes(const float* table, float* out, int heads, hipStream_t st);
// final LayerNorm2d + global average pool: x [B,49,C] f32 -> feat [B,C] f32
int launch_mv_final(const float* x, const float* w, const float* b, float* feat, int B, int P, int C,
                    hipStream_t st);

// ---- packing kernels
// stem conv1 [32][3][3][3] * BN scale -> [32][32] T (k = (ky*3+kx)*3 + c, zero padded)
int launch_mv_pack_stem1(int prec, const float* w, const float* scale, void* out, hipStream_t st);
// conv [O][C][3][3] f32 -> [O][(ky*3+kx)*C + c] T
int launch_mv_pack_conv3(int prec, const float* w, void* out, int O, int C, hipStream_t st);
// out[i] = b[i] * s[i] + t[i]  (b may be NULL = 0)
int launch_mv_fold_bias(const float* b, const float* s, const float* t, float* out, int n,
                        hipStream_t st);
// depthwise [C][1][3][3] -> tap-major [9][C] * scale[c]
int launch_mv_pack_dw(const float* w, const float* scale, float* out, int C, hipStream_t st);
// relative position table [169][heads] -> bias_t [heads][49][49] (key-major, see launch_mv_attn)
int launch_mv_pack_relbias(const float* table, float* out, int heads, hipStream_t st);
int launch_mv_fill(float* out, float v, int n, hipStream_t st);
