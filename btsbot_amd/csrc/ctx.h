// Internal: the handle behind btsbot_handle (shared by api.hip and head_train.hip).
#pragma once
#include <string>
#include <vector>

#include "common.h"

struct ParamRec {
  std::string name;
  int64_t off, numel;
  int ndim;
  int shape[4];
  int is_buffer;
};

struct BlockPk {  // per ConvNeXt block: master offsets + packed offsets (bytes into `extra`)
  int64_t gamma, dw_w, dw_b, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b;
  size_t p_dw, p_fc1, p_fc2, p_fused;
  size_t p_s0par;          // stage-0 / stage-1 blocks: parameter image for stage0b.hip / stage1b.hip
  size_t p_s0par_t = 0;    // ... for their keeping forms (training forward): the same image with f16 taps in every mode
  size_t p_fc2g;           // diag(gamma) W2 in the operand type (megakernels fold the layer scale)
  size_t p_w1p = 0, p_w2p = 0;   // stage2p.hip / stage3.hip: fc1 / gamma * fc2 filters as MFMA A fragments
  size_t p_scales = 0;           // fp8 mode: {S1, 1/S1, S2, 1/S2} of those two
  size_t p_x2_w1 = 0, p_x2_w2g = 0;   // split mode, stages 0-1: fc1 / gamma * fc2 filters, f16 heads (stage0b / stage1b)
  size_t p_x2_w1lo = 0, p_x2_w2glo = 0;   // ... and their f16 remainders, same layouts
  size_t p_fc1t, p_fc2t;   // for the dgrad GEMMs: W1^T [C][4C], (diag(gamma) W2)^T [4C][C]
  size_t p_w1tp = 0, p_w2tp = 0;   // 256-channel blocks, training: the same two as MFMA A fragments (s2mlp_bwd.hip)
  bool fused;
};
struct DownPk {
  int64_t ln_w, ln_b, w, b;
  size_t p_w, p_wt;        // p_wt: [4*Cin][Cout] transpose of the packed filter (dgrad)
  size_t p_wp = 0;         // stage2p.hip: the filter as MFMA A fragments
  size_t p_scale = 0;      // fp8 mode: {S, 1/S} of it
  size_t p_x2_w = 0, p_x2_wlo = 0;   // split mode, stage0b's downsample: the filter's f16 heads / remainders, [Cout][q][Cin]
};

constexpr int STAGE_HW[4] = {15, 7, 3, 1};

enum { CAT_STEM = 0, CAT_DWLN, CAT_FC1, CAT_FC2, CAT_LNPATCH, CAT_DOWN, CAT_HEAD, CAT_FUSED, CAT_STAGE0, CAT_STAGE1, CAT_STAGE2, CAT_S3FC1, CAT_S3FC2, CAT_HEAD16,
       CAT_MV_STEM, CAT_MV_G_STEM, CAT_MV_G_CONV1, CAT_MV_G_CONV3, CAT_MV_G_SC, CAT_MV_G_QKV, CAT_MV_G_PROJ, CAT_MV_G_FC1,
       CAT_MV_G_FC2, CAT_MV_FUSED, CAT_MV_FRONT, CAT_MV_ABLK, CAT_MV_ELT, CAT_MV_DW, CAT_MV_SE, CAT_MV_LN, CAT_MV_ATTN, CAT_MV_SMLP, CAT_MV_PART, NCAT };
const char* const CAT_NAMES[NCAT] = {"stem_kernel",       "dwconv_ln_kernel", "gemm_kernel<fc1,GELU>",
                                     "gemm_kernel<fc2,RESID>", "ln_patch_kernel", "gemm_kernel<down,BIAS>",
                                     "head_kernel", "fused_mlp_kernel", "stage0b_kernel", "stage1b_kernel",
                                     "stage2p_kernel", "s3_fc1_kernel", "s3_fc2_kernel", "head16_kernel",
                                     "mv_stem_im2col", "mv_gemm<stem>", "mv_gemm<conv1,SILU>", "mv_gemm<conv3,gated>",
                                     "mv_gemm<shortcut>", "mv_gemm<qkv>", "mv_gemm<proj,RESID>", "mv_gemm<fc1,GELU>",
                                     "mv_gemm<fc2,RESID>", "mv_fused_mlp", "mv_mbconv_front", "mv_attn_block", "mv_elementwise", "mv_dw3_kernel",
                                     "mv_se_kernel", "mv_ln_kernel", "mv_attn_kernel", "mv_streamed_mlp", "mv_partition"};
constexpr size_t PROF_MAX_LAUNCHES = 16384;

struct MaxVit;   // maxvit.hip
struct SidePick {
  hipStream_t caller, side;
  bool apart;
};

struct btsbot_ctx {
  btsbot_config cfg;
  bool has_image, has_meta;
  bool is_maxvit = false;   // image branch = timm maxvit_tiny_rw_224 (maxvit.hip) instead of ConvNeXt
  MaxVit* mv = nullptr;
  int n_comb;        // linear layers of the fusion MLP
  int comb_dims[4];
  int act;           // ACT_GELU / ACT_RELU of the heads
  int meta_trailing_act;
  std::vector<ParamRec> params;
  int64_t total_floats = 0;

  // master offsets
  int64_t stem_w, stem_b, stem_lnw, stem_lnb, hn_w = -1, hn_b = -1;
  std::vector<std::vector<BlockPk>> blocks;  // [stage][block]
  DownPk down[4];
  int64_t bn_w, bn_b, bn_rm, bn_rv, m1_w, m1_b, m2_w, m2_b;
  int64_t comb_w[3], comb_b[3];
  size_t p_m1, p_m2, p_comb[3], p_bn_scale, p_bn_shift, p_stem16 = 0;
  size_t p_x2_stem = 0, p_x2_stemlo = 0;    // split mode: the stem filter's f16 heads / remainders (stage0b)
  int prec_head() const { return x2 ? BTSBOT_F16 : cfg.precision; }   // head16.hip splits its operands in every mode
  int prec_s01() const { return x2 ? BTSBOT_F16X2 : cfg.precision; }   // operand mode of stage0b.hip / stage1b.hip
  size_t p_m1h = 0, p_m2h = 0, p_combh[3] = {0, 0, 0};   // head16.hip: the Linear filters as split A fragments
  bool head16 = false;     // the 16-bit modes run the head on the matrix pipe (head16.hip)
  bool use_head16 = true;  // BTSBOT_AMD_NO_HEAD16=1: the fp32 VALU head (head.hip) instead
  bool stage0 = false;     // stem + stage 0 + first downsample as one kernel

  // device memory
  float* mirror = nullptr;          // fp32 copy of the master arena (same offsets)
  unsigned char* extra = nullptr;   // transformed operands
  void* pack_jobs[3] = {nullptr, nullptr, nullptr};   // device tables of PackJob: [0] full pack, [1] training re-pack,
  int pack_njobs[3] = {0, 0, 0}, pack_blocks[3] = {0, 0, 0};   // [2] the part of [1] the stage-0 megakernel reads (s0_train)
  hipEvent_t pack_early_ev = nullptr;   // recorded behind that part and the stage-0 parameter images (pack_sync_early)
  bool pack_early = false;              // the running re-pack recorded it
  size_t extra_bytes = 0;
  bool packed = false;
  bool packed_full = false;   // false after btsbot_pack_params_train(): inference-only operand images are stale

  unsigned char* ws = nullptr;
  bool ws_owned = true;             // false: the caller's memory (btsbot_use_workspace), never freed here
  size_t ws_bytes = 0;
  int max_chunk = 0;
  size_t o_x, o_x2, o_xn, o_h;      // workspace offsets
  // per-kernel-family timing with HIP events on the launch stream (btsbot_set_profile)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;   // pairs: [2i] before, [2i+1] after launch i
  std::vector<int> prof_cat;
  size_t prof_used = 0;

  bool use_s2 = true;      // BTSBOT_AMD_NO_STAGE2=1: the per-op launches (dwconv_ln + fc1 / fc2 GEMMs) for stage 2
  int s2p_alerts_hint = 0; // btsbot_set_option("stage2p_alerts"): 0 = by rounds, 4 / 7 forced
  bool use_s2p = true;     // (= use_s2: stage2p.hip is the stage-2 kernel)
  bool stage2p = false;    // stage 2 + the last downsample as one persistent kernel
  bool fp8 = false;        // created with BTSBOT_FP8: cfg.precision reads BTSBOT_BF16, stages 2-3 run fp8 operands
  bool x2 = false;         // created with BTSBOT_F16X2: cfg.precision reads BTSBOT_F32 (the schedule of every kernel without
                           // a split-operand form), the kernels that have one run it
  // (x2_tail_plain: developer experiment BTSBOT_AMD_X2_TAIL_F16=1 -- the split mode with stages 2-3 on plain f16 operands:
  //  what does the mode's error owe to which stage?)
  bool x2_tail_plain = false;
  int prec_tail() const { return fp8 ? BTSBOT_FP8 : x2 ? (x2_tail_plain ? BTSBOT_F16 : BTSBOT_F16X2) : cfg.precision; }   // operand mode of stage2p.hip / stage3.hip
  int prec_down3() const { return x2 ? (x2_tail_plain ? BTSBOT_F16 : BTSBOT_F16X2) : cfg.precision; }   // ... of the last downsample inside stage2p.hip
  bool stage3 = false;     // stage3.hip: the 1x1 stage as two fragment-streaming launches per block
  bool use_s3 = true;      // BTSBOT_AMD_NO_S3=1: dwconv_ln + the generic GEMMs instead
  bool use_fused = true;   // BTSBOT_AMD_NO_FUSED_MLP=1 keeps the two-GEMM path (A/B timing)
  bool stage1 = false;     // stage 1 + second downsample as one kernel
  bool use_stage0 = true;
  bool use_stage1 = true;  // BTSBOT_AMD_NO_STAGE1=1 likewise for stage 1  // BTSBOT_AMD_NO_STAGE0=1 keeps the per-op schedule for stage 0
  // image-branch training cache (backbone_train.hip)
  unsigned char* bbcache = nullptr;
  int bbcache_batch = 0;
  int64_t img_floats = 0;     // master-arena floats [0, img_floats) belong to the image branch
  const float* t_img = nullptr;   // triplets of the last training forward (stem backward re-reads them)
  bool train_packs = false;   // also pack the dgrad transposes (set by btsbot_reserve_train)
  bool bb_saved = false;      // the last training forward kept the image-branch activations
  // training cache (head_train.hip): activations of the last training-mode forward
  float* tcache = nullptr;
  int tcache_batch = 0, train_batch = 0;
  const uint8_t* t_meta_mask = nullptr;
  const uint8_t* t_comb_mask = nullptr;

  // gradient buckets in the order btsbot_backward() completes them (btsbot_grad_buckets / btsbot_wait_grad_bucket)
  int n_buckets = 0;
  int64_t bucket_lo[3] = {0, 0, 0}, bucket_hi[3] = {0, 0, 0};
  hipEvent_t bucket_ev[3] = {nullptr, nullptr, nullptr};
  bool bucket_recorded = false;
  // The per-stage bucket events cost the backward's chain a fork each (an event record between two kernels: ~6 us), so they
  // are recorded only once somebody has waited for a bucket (btsbot_wait_grad_bucket / btsbot_allreduce_grads: a
  // multi-GPU run, from its first step on).  bucket_fine: the LAST backward recorded bucket_ev[]; otherwise a waiter gets
  // an event recorded on that backward's stream at the time it asks (everything the backward queued is in front of it).
  bool bucket_waits_seen = false, bucket_fine = false;
  bool meta_join_pending = false;   // the metadata branch's training forward sits on the side stream and `st` has not joined it yet
  hipStream_t xchg = nullptr;        // btsbot_allreduce_grads: the stream its collectives run on
  // Training forward of stage 2 + the last downsample as ONE launch of stage2p_kernel's keeping form (16-bit modes, 256
  // channels; needs the 3x3 LayerNorm / depthwise backward kernel, which recomputes the depthwise output the form does
  // not keep).  BTSBOT_AMD_NO_S2P_TRAIN=1: the per-op launches (A/B timing, parity tests).
  bool s2p_train = false;
  // Training forward of stem + stage 0 + first downsample as ONE launch of the inference megakernel's keeping form
  // (stage0b.hip, KEEP) instead of stem16 + 2 x (dwconv_ln + fused_mlp) + ln_patch + GEMM (16-bit modes).
  // BTSBOT_AMD_NO_S0_TRAIN=1: the per-op launches (A/B timing, parity tests).
  bool s0_train = false;
  bool s1_train = false;             // likewise stage 1 + the second downsample (stage1b.hip, KEEP): default in the f16 mode,
                                     // BTSBOT_AMD_S1_TRAIN=1 in bf16 (api.hip says why), BTSBOT_AMD_NO_S1_TRAIN=1 switches it off
  bool use_stem16 = true;            // BTSBOT_AMD_NO_STEM16=1: the fp32 VALU stem in the 16-bit modes too (A/B, parity)
  bool deterministic = false;        // btsbot_set_option("deterministic") / BTSBOT_AMD_DETERMINISTIC=1: fixed-order batch reductions
  float* det_scratch = nullptr;      // ... their partial rows (sized at btsbot_reserve_train)
  size_t det_floats = 0;
  int exchange_mode = 0;             // btsbot_set_option("exchange"): 0 all-reduce per span, 1 reduce-scatter + all-gather
  const float* last_grad_arena = nullptr;   // what the last btsbot_backward() wrote (the bucket events belong to it)
  hipEvent_t xchg_done = nullptr;
  // second stream of the image-branch backward (backbone_train.hip): filter-gradient GEMMs trail the dX chain on it
  hipStream_t side = nullptr;
  hipStream_t side_for = nullptr;    // the caller's stream h->side was chosen against (create_side_stream)
  bool side_apart = true;            // ... and was measured to run on another hardware pipe than it (false: none of 8 did)
  std::vector<SidePick> side_cache;  // one chosen side stream per caller stream this handle has seen (never re-probed)
  hipStream_t xchg_for[2] = {nullptr, nullptr};   // the (caller, side) pair h->xchg was placed against
  bool xchg_apart = true;
  hipEvent_t bwd_done = nullptr;     // recorded at the end of a backward that did not record the per-bucket events
  std::vector<hipEvent_t> side_ev;   // pool, side_used of them taken by the current btsbot_backward()
  size_t side_used = 0;
  // btsbot_pack_params_train() queues its packing launches on `side` behind the mirror copy: they then overlap the
  // first kernel of the training forward (the stem reads the fp32 mirror only).  Every consumer of the operand images
  // calls pack_sync() first.
  bool pack_on_side = false;
  bool use_dwln = true;    // BTSBOT_AMD_NO_DWLN=1: LayerNorm / depthwise backward as three launches (A/B timing)
  // widths whose block MLP runs fused in the training step (fused_mlp forward that keeps nothing 4C-wide + mlp_bwd_kernel):
  // 64 and 128.  BTSBOT_AMD_MLP_BWD_C=64 / =128 restricts it to one width, BTSBOT_AMD_NO_MLP_BWD=1 switches it off (A/B
  // runs and tests).  The 128-channel form hands dxn over as four addend planes which only dwln_bwd_kernel reads, so it
  // is tied to that kernel (BTSBOT_AMD_NO_DWLN keeps stage 1 unfused).
  int mlp_bwd_only = 0;
  bool mlp_fused(int ch) const {
    return mlp_bwd_only >= 0 && (mlp_bwd_only == 0 || mlp_bwd_only == ch) && use_fused && (ch == 64 || use_dwln) &&
           mlp_bwd_supported(cfg.precision, ch) && fused_mlp_supported(cfg.precision, ch);
  }
  bool s2mlp = true;       // 256-channel blocks: da and dxn of the MLP backward as one launch (s2mlp_bwd.hip) instead of two tiled
                           // GEMMs; BTSBOT_AMD_NO_S2MLP=1: the GEMMs (A/B timing, parity tests)
  bool fork_per_block = false;   // BTSBOT_AMD_FORK_PER_BLOCK=1: the blocks of a batched stage fork the side stream one by one, as
                                 // before the batch existed (A/B timing)
  bool wgrad_batch = true; // stages whose blocks run the unfused MLP backward (256 / 512 channels): their 2 x depth filter-gradient
                           // GEMMs as ONE launch + one slice reduction at the end of the stage's chain (wgrad.hip);
                           // BTSBOT_AMD_NO_WGRAD_BATCH=1: one launch per GEMM behind each block (A/B timing, parity tests)
  bool use_side = true;    // BTSBOT_AMD_NO_SIDE_STREAM=1: the whole backward on the caller's stream (A/B timing)

  unsigned long long* stamps = nullptr;   // 32 phase timestamps: [0..15] stage 0, [16..31] stage 1
  bool debug = false;
  float* taps[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int last_chunk = 0;

  int esz() const { return cfg.precision == BTSBOT_F32 ? 4 : 2; }
};

// Fork / join of the backward's second stream.  side_fork: work queued on *sd afterwards sees everything queued on
// `st` so far (*sd = st when the second stream is off); side_join: `st` waits for everything queued on the side.
int create_side_stream(btsbot_ctx* h, hipStream_t caller);   // api.hip: h->side, on another hardware queue than `caller`
// api.hip: a new stream measured to run beside every stream of busy[] (else the last candidate, *apart = false + a warning)
int pick_apart_stream(btsbot_ctx* h, const hipStream_t* busy, int nbusy, const char* role, hipStream_t* out, bool* apart);
int side_fork(btsbot_ctx* h, hipStream_t st, hipStream_t* sd);
int side_join(btsbot_ctx* h, hipStream_t st);

int pack_sync(btsbot_ctx* h, hipStream_t st);   // api.hip
int pack_sync_early(btsbot_ctx* h, hipStream_t st);   // api.hip: only what stage0b_kernel reads (else = pack_sync)
