/*
 * btsbot_hip.h -- C ABI of libbtsbot_hip.so: the MI355X (gfx950) implementation of BTSbot's
 * classifier forward/backward hot path.
 *
 * The reference (nabeelre/BTSbot, /root/reference) has no FFI: the path sits behind Python
 * nn.Modules (btsbot/architectures.py) driven by btsbot/train.py and btsbot/inference_example.py.
 * Each entry point below names the reference code it stands in for; the Python host in
 * btsbot_amd/ (same class names, kwargs and state-dict keys as the reference) binds these
 * symbols with ctypes.  INTEGRATION.md shows the stub a BTSbot maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  int status: 0 = OK, <0 = error
 *     (enum below); btsbot_last_error() returns a thread-local message.  No C++ exception
 *     crosses the ABI.
 *   - Every pointer is a BORROWED DEVICE pointer (hipMalloc'd / torch CUDA tensor) that must stay
 *     valid until the work enqueued on `stream` has completed.  `stream` is a hipStream_t passed
 *     as void* (NULL = the legacy default stream).  All work is enqueued asynchronously.  Device
 *     memory is allocated only by the first btsbot_pack_params() on a handle and by
 *     btsbot_reserve(); forward / loss / optimiser calls never allocate, synchronise or copy to
 *     the host -- so a caller may capture them into a hipGraph.
 *   - One handle per GPU and per model replica; a handle is thread-compatible (one thread at a
 *     time), matching "a single Python thread drives the model" (SURVEY.md section 8b).
 *   - Tensors: triplets are [B,3,63,63] fp32 NCHW contiguous exactly as
 *     inference_example.py:62-64 prepares them; metadata is [B,n_meta] fp32; logits/scores are
 *     [B] fp32 (the reference's [B,1]).
 */
#ifndef BTSBOT_HIP_H
#define BTSBOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTSBOT_ABI_VERSION 1

enum btsbot_status {
  BTSBOT_OK = 0,
  BTSBOT_ERR_INVALID_ARG = -1,   /* bad config / NULL pointer / shape the kernels do not cover  */
  BTSBOT_ERR_HIP = -2,           /* a HIP runtime call failed (message has hipGetErrorString)    */
  BTSBOT_ERR_WORKSPACE = -3,     /* batch larger than the reserved workspace                     */
  BTSBOT_ERR_UNKNOWN_PARAM = -4, /* parameter name not in the handle's table                     */
  BTSBOT_ERR_STATE = -5          /* call order violated (e.g. backward without a training fwd)   */
};

/* which nn.Module of btsbot/architectures.py the handle reproduces */
enum btsbot_wiring {
  BTSBOT_MM_CONVNEXT = 0,   /* mm_ConvNeXt     architectures.py:125-171 (GELU heads)            */
  BTSBOT_CONVNEXT = 1,      /* ConvNeXt        architectures.py:104-122 (image only)            */
  BTSBOT_FROZEN_FUSION = 2, /* frozen_fusion   architectures.py:296-372 (ConvNeXt + um_nn, ReLU)*/
  BTSBOT_UM_NN = 3,         /* um_nn           architectures.py:277-293 (metadata only)         */
  BTSBOT_MM_MAXVIT = 4,     /* mm_MaxViT       architectures.py:58-101 (maxvit_tiny_rw_224 + GELU heads);
                               the image branch trains (BatchNorm2d batch statistics + the backward of every
                               layer, reserve_train(with_image_grads = 1)) or stays a frozen eval-mode branch
                               under trainable heads / metadata branch (with_image_grads = 0); the two mixed
                               regimes (frozen branch on batch statistics, trainable branch on running
                               statistics) return BTSBOT_ERR_STATE                                        */
  BTSBOT_MAXVIT = 5,        /* MaxViT          architectures.py:25-55  (image only)             */
  BTSBOT_FROZEN_FUSION_MAXVIT = 6 /* frozen_fusion with a MaxViT image branch (head stripped to its global
                               pool, architectures.py:304-308) + um_nn metadata branch, ReLU fusion head:
                               what the published maxvit "-metadata" checkpoints instantiate      */
};

/* arithmetic type of the MFMA operands / staged activations (accumulation is always fp32;
 * the residual stream, LayerNorm, GELU and the heads are fp32 in every mode) */
enum btsbot_precision {
  BTSBOT_F32 = 0,  /* v_mfma_f32_16x16x4_f32: exact fp32 fma chains -- the parity mode          */
  BTSBOT_BF16 = 1, /* v_mfma_f32_16x16x32_bf16                                                   */
  BTSBOT_F16 = 2,  /* v_mfma_f32_16x16x32_f16 (same rate as bf16, 3 more mantissa bits)          */
  BTSBOT_FP8 = 3,  /* inference only: the bf16 schedule with the pointwise convolutions of stages 2-3
                      (55 % of the FLOPs, the filter-streaming-bound part) on the block-scaled fp8 MFMA
                      (v_mfma_scale_f32_16x16x128_f8f6f4 / 32x32x64), OCP e4m3 operands; scaling: see
                      DESIGN.md section 2; training entry points behave as BTSBOT_BF16 */
  BTSBOT_F16X2 = 4 /* split operands: every MFMA operand of the pointwise / downsample convolutions is an f16
                      head plus an f16 remainder (x = hi + lo, 22 significant bits), a product is three
                      v_mfma_f32_*_f16 (hi*hi + hi*lo + lo*hi) -- scores within 1e-4 of the fp32 reference at
                      the 16-bit matrix rate; kernels without a split form run the fp32 schedule.  Inference
                      only: the training entry points behave as BTSBOT_F32 */
};

typedef struct btsbot_config {
  int32_t abi_version;   /* = BTSBOT_ABI_VERSION                                                 */
  int32_t wiring;        /* enum btsbot_wiring                                                   */
  int32_t precision;     /* enum btsbot_precision                                                */
  /* timm ConvNeXt table (pico: depths 2,2,6,2 dims 64,128,256,512; nano: 2,2,8,2 / 80..640);
   * MaxViT wirings: maxvit_tiny_rw_224 = depths 2,2,5,2 dims 64,128,256,512 (the 63x63 cutouts are
   * resized to 224x224 inside the library, architectures.py:44-50)                                */
  int32_t depths[4];
  int32_t dims[4];
  int32_t image_size;    /* 63 (the only size the kernels are specialised for)                   */
  int32_t head_norm;     /* 1: pool + LayerNorm2d before flatten (ConvNeXt, frozen_fusion,
                               mm_ConvNeXt on "LS" data); 0: flatten only (architectures.py:142) */
  int32_t n_meta;        /* len(config["metadata_cols"]), 25 in prod_config.json:15-41           */
  int32_t meta_fc1, meta_fc2;           /* metadata branch widths                                */
  int32_t comb_fc1, comb_fc2;           /* fusion head widths (ConvNeXt: fc1_neurons/fc2_neurons)*/
  float meta_dropout, comb_dropout;     /* training-mode dropout probabilities                   */
} btsbot_config;

typedef struct btsbot_ctx* btsbot_handle;

/* One row of the handle's parameter table.  `name` is the canonical (prefix-free) name, e.g.
 * "stages.1.blocks.0.mlp.fc1.weight", "head_norm.weight", "meta.0.running_mean",
 * "comb.2.bias"; the Python host maps the reference's state-dict keys onto these
 * (btsbot_amd/architectures.py).  `offset` is in floats into the master arena. */
typedef struct btsbot_param_info {
  char name[96];
  int64_t offset;
  int64_t numel;
  int32_t ndim;
  int32_t shape[4];
  int32_t is_buffer;     /* 1 for BatchNorm running_mean / running_var                           */
} btsbot_param_info;

const char* btsbot_last_error(void);
int btsbot_abi_version(void);

/* Replaces: model_type(config)   (from_HF.py:71-73, train.py:218-222).  Builds the parameter table
 * only -- no HIP call, so it also works on a host without a GPU (the reference constructs on the
 * CPU and then calls .to(device)). */
int btsbot_create(const btsbot_config* cfg, btsbot_handle* out);
int btsbot_destroy(btsbot_handle h);

/* Parameter table: the fp32 "master arena" layout the caller allocates (one flat device buffer of
 * btsbot_param_floats() floats; the Python host makes every nn.Parameter a view into it so that
 * state_dict()/load_state_dict() keep the reference's keys, from_HF.py:74-79). */
int btsbot_param_count(btsbot_handle h);
int64_t btsbot_param_floats(btsbot_handle h);
int btsbot_param_info_at(btsbot_handle h, int index, btsbot_param_info* out);

/* Re-pack the master arena into the kernels' operand layouts (cast to the MFMA type, K-major
 * 1x1 / 2x2 / 4x4 filters, tap-major depthwise filters, BatchNorm folded for eval).
 * Replaces: load_state_dict (from_HF.py:74) / the implicit "weights are where cuDNN wants them".
 * Must be called after every change of the master arena (load, optimiser step). */
int btsbot_pack_params(btsbot_handle h, const float* master_arena, void* stream);
/* The same for a training loop that differentiates the image branch (btsbot_forward_train with
 * keep_image_activations != 0): skips the operand images only the fused inference kernels read; an inference
 * btsbot_forward() afterwards needs a full btsbot_pack_params() first (it returns BTSBOT_ERR_STATE otherwise).
 * Stream rule for both: the forward / forward_train that consumes a pack must be queued on the SAME stream as the pack
 * (or on one the caller has ordered behind it): part of the re-pack runs on a stream of the handle's own, which the
 * consumer joins on its stream, but the arena -> mirror copy is ordered by the pack's stream alone. */
int btsbot_pack_params_train(btsbot_handle h, const float* master_arena, void* stream);

/* Workspace: activations of one chunk of alerts.  reserve() (re)allocates for chunks of up to
 * `max_chunk` alerts; forward() splits larger batches into chunks internally. */
int64_t btsbot_workspace_bytes(btsbot_handle h, int max_chunk);
int btsbot_reserve(btsbot_handle h, int max_chunk);
/* The caller-sized form (no allocation inside the library): `workspace` = at least
 * btsbot_workspace_bytes(h, max_chunk) bytes of 256-byte-aligned device memory that stays the caller's -- borrowed
 * until the next btsbot_reserve() / btsbot_use_workspace() / btsbot_destroy(), never freed here. */
int btsbot_use_workspace(btsbot_handle h, int max_chunk, void* workspace, int64_t bytes);

/* Replaces: model(image_input=..., metadata_input=...) / model(input_data=...) followed by
 * torch.sigmoid (architectures.py:166-171,121-122,292-293,367-372; inference_example.py:84-91).
 * `triplets` may be NULL for BTSBOT_UM_NN, `meta` NULL for BTSBOT_CONVNEXT; `scores` may be NULL.
 * training != 0 selects BatchNorm batch statistics + dropout (seeded by dropout_seed) and keeps
 * the activations needed by btsbot_backward(). */
int btsbot_forward(btsbot_handle h, const float* triplets_nchw, const float* meta,
                   float* logits, float* scores, int batch, int training,
                   uint64_t dropout_seed, void* stream);

/* Training-mode forward: replaces model(...) under model.train() (train.py:510).  The image branch
 * is identical to inference (ConvNeXt has no BatchNorm / dropout, drop-path 0); the metadata
 * BatchNorm1d uses the statistics of THIS batch and, when master_arena is non-NULL, updates
 * running_mean / running_var there (momentum 0.1, unbiased variance), exactly as nn.BatchNorm1d --
 * also when the branch is frozen (frozen_fusion keeps its branches in train mode).  Dropout is
 * applied with caller-supplied keep-masks (uint8 [batch][meta_fc1] and [batch][comb_fc2]; 1 = keep,
 * scaled by 1/(1-p)); the masks must stay valid until btsbot_backward() has run.  Activations are
 * kept in a cache sized by btsbot_reserve_train(max_batch, with_image_grads) (whole batch, no
 * chunking, because of the batch statistics).  keep_image_activations != 0 runs the image branch
 * through the per-op training schedule that keeps, per block, x_in / LN output / fc1 pre-activation /
 * hidden activation (about 1.2 MB per alert; 2.8 MB with the backward's own per-block buffers) for a later
 * btsbot_backward(need_image_grads=1);
 * otherwise the image branch runs the fused inference kernels.  The MaxViT wirings take only that second form:
 * their BatchNorm2d layers use the running statistics (a frozen, eval-mode branch under trainable heads).
 * 16-bit modes, pico: stem + stage 0 (+ stage 1 in the f16 mode) run the inference megakernels' keeping forms, whose
 * depthwise phase reads its map and taps as F16 operands in EVERY mode (bf16 taps moved the 50-step loss curve ten
 * times further from the fp32 recipe): a bf16 handle's stage-0/1 residual stream therefore has f16 RANGE during
 * training -- values beyond +-65504 are saturated on their way into that phase (no inf / NaN), and the backward
 * differentiates the convolution as written.  BTSBOT_AMD_NO_S0_TRAIN=1 (and NO_S1_TRAIN=1) at btsbot_create restores
 * the per-op forward with its fp32 depthwise operands. */
int btsbot_reserve_train(btsbot_handle h, int max_batch, int with_image_grads);
int btsbot_forward_train(btsbot_handle h, const float* triplets_nchw, const float* meta,
                         float* logits, float* scores, int batch, const uint8_t* meta_keep_mask,
                         const uint8_t* comb_keep_mask, float* master_arena,
                         int keep_image_activations, void* stream);

/* Replaces loss.backward() (train.py:526) for the parameters of the fusion head (always) and the
 * metadata branch (need_meta_grads): writes d(loss)/d(param) into grad_arena, which has the master
 * arena's layout (other entries are left untouched).  dlogits = d(loss)/d(logits) [batch], e.g. from
 * btsbot_bce_fwd_bwd.  need_image_grads != 0 also differentiates the ConvNeXt image branch (stem,
 * every block, downsamples, head LayerNorm); some of those gradients are reduced over the batch with fp32
 * atomics, so their last bits vary from run to run.
 * Stream semantics are the caller's: everything is ordered after the work already queued on `stream`, and
 * work queued on `stream` afterwards sees every gradient.  Inside, the weight-gradient kernels run on a second
 * stream the handle owns (forked from and joined back into `stream` with events; BTSBOT_AMD_NO_SIDE_STREAM=1
 * keeps every launch on `stream`). */
int btsbot_backward(btsbot_handle h, const float* dlogits, float* grad_arena, int need_meta_grads,
                    int need_image_grads, void* stream);

/* Replaces the gradient reduction of torch.nn.DataParallel (train.py:238-240; SURVEY.md section 8b's
 * btsbot_allreduce_grads) -- split in two, because the communicator belongs to the host's process group
 * (torch.distributed over RCCL), not to this library: the library says WHICH parts of the gradient arena
 * btsbot_backward() finishes WHEN, the host issues one all-reduce per part on a side stream.
 * btsbot_grad_buckets: up to `capacity` arena ranges [lo[i], hi[i]) (floats) in the order btsbot_backward()
 * completes them -- ConvNeXt wirings: {last image stage + head LayerNorm + metadata branch + fusion head},
 * {stage 2}, {stem + stages 0-1}; every other wiring: one range over the whole arena.  Returns the count.
 * btsbot_wait_grad_bucket: makes `stream` wait (hipStreamWaitEvent) until the kernels of the LAST
 * btsbot_backward() call that write bucket `bucket` have finished; no host synchronisation.  (The per-bucket events
 * cost the backward a fork of its side stream each, so a handle records them only once it has seen a waiter -- this call
 * or btsbot_allreduce_grads.  A backward that ran before any waiter records ONE event at its end instead, and the FIRST
 * wait on a handle waits on that: `stream` is ordered behind everything that backward queued -- correct, but without
 * the overlap; from the next btsbot_backward() on the wait ends with the bucket.  No stream handle is kept: any thread,
 * any time after btsbot_backward() has returned.) */
int btsbot_grad_buckets(btsbot_handle h, int capacity, int64_t* lo, int64_t* hi);
int btsbot_wait_grad_bucket(btsbot_handle h, int bucket, void* stream);

/* The exchange step of data-parallel training (replaces torch.nn.parallel.DataParallel's reduce_add_coalesced,
 * /root/reference/btsbot/train.py:238-240, 526): all-reduce (SUM) over the ranks of `nccl_comm` (an RCCL ncclComm_t) of
 * `nspans` spans [lo[i], hi[i]) (floats) of the gradient arena `grads`, span i belonging to gradient bucket bucket[i]
 * of btsbot_grad_buckets().  Every collective runs on a stream of the library's own and starts as soon as the LAST
 * btsbot_backward() has written its bucket (the rest of the backward pass keeps `stream`: pass the spans in bucket
 * order); `stream` then waits for all of them, so the btsbot_adamw_step() queued behind this call sees the sums.
 * Local gradients are already scaled by 1 / n_global (btsbot_bce_fwd_bwd), so SUM is the global-batch mean's
 * gradient.  No host synchronisation.  RCCL is resolved at the first call (dlopen of librccl.so.1: the copy already
 * in the process, e.g. PyTorch's, wins) -- the communicator must come from that library; BTSBOT_ERR_STATE if there is
 * none.  `grads` must be the arena the last btsbot_backward() wrote (the bucket events belong to it;
 * BTSBOT_ERR_INVALID_ARG otherwise).  btsbot_set_option(h, "exchange", 1) switches every span from one ncclAllReduce
 * to the direct form for xGMI's point-to-point links: ncclReduceScatter (each rank owns the sum of its 1 / N slice)
 * + ncclAllGather, in place, plus a small ncclAllReduce for what is left after N equal slices.
 * The exchange stream is placed like the backward's side stream: measured (once per caller stream) to run on another
 * hardware pipe than `stream` AND the side stream, so that a long collective kernel does not take turns with the
 * backward's kernels; btsbot_set_option(h, "query_side_apart", 0) tells whether that succeeded. */
int btsbot_allreduce_grads(btsbot_handle h, void* nccl_comm, float* grads, int nspans, const int* bucket,
                           const int64_t* lo, const int64_t* hi, void* stream);

/* Scheduling hints (host-side state read at launch time; results do not depend on them).
 *   "stage2p_alerts": alerts resident per workgroup of the stage-2 kernel -- 0 (default): 5, or 7 where that takes fewer
 *   rounds of one workgroup per CU; 4 / 5: always (36 / 45 of the same 48 matrix columns); 7: always (a scoring loop with several forwards in flight on different streams:
 *   the kernel then leaves ~40 % of the CUs to the other stream at 1024 alerts); 4: always.
 *   "exchange": form of btsbot_allreduce_grads' collectives -- 0 (default) all-reduce, 1 reduce-scatter + all-gather.
 *   "deterministic" (also BTSBOT_AMD_DETERMINISTIC=1 at btsbot_create; set before btsbot_reserve_train): 1 = the batch
 *   reductions of the ConvNeXt training step that meet through fp32 atomics (LayerNorm / depthwise parameter gradients,
 *   column sums, the fused MLP backward's bias gradient) write partial rows and add them in a fixed order instead: two
 *   identical btsbot_backward() calls give bit-identical gradients (16-bit modes; ~25 small extra launches per step).
 *   btsbot_backward() checks the scratch against the batch BEFORE it launches anything (BTSBOT_ERR_STATE, no state
 *   change); ConvNeXt wirings only (refused for MaxViT, ignored there when it comes from the environment).
 *   "query_side_apart" (a query; `value` ignored): BTSBOT_OK when the second stream of btsbot_backward() and the
 *   exchange stream of btsbot_allreduce_grads() were each measured on a hardware pipe of their own, BTSBOT_ERR_STATE
 *   (and a warning on stderr at the time) when none of eight candidates was: two queues of one pipe take turns of ~50 us,
 *   a 1024-alert step then takes 5-7 ms instead of 2.6. */
int btsbot_set_option(btsbot_handle h, const char* key, int value);

/* Validation aid with no reference counterpart: when on, forward() keeps fp32 copies of the stem and
 * stage outputs (call before btsbot_reserve()). */
int btsbot_set_debug(btsbot_handle h, int on);

/* Developer aid: when non-NULL, workgroup 0 of the stage megakernels stores the shader clock at its
 * phase boundaries into device_buffer32[0..31] (uint64), and every workgroup its start / end on the
 * 100 MHz wall clock into [32 + 2*wg] (stage 0) and [32 + 8192 + 2*wg] (stage 1), and the stage-2
 * front kernel its phase clocks into [32 + 16384 ..+63] and per-workgroup wall clocks behind them: the
 * buffer holds 32 + 16384 + 64 + 2048 entries and the batch must not exceed 4096 alerts while it is installed. */
int btsbot_debug_stamps(btsbot_handle h, unsigned long long* device_buffer32);

/* Debug/validation tap: copy an intermediate of the LAST forward chunk to `dst` (fp32).
 * name: "stem", "stage0".."stage3" (NHWC [chunk, P, C]).  Returns the element count or <0. */
int64_t btsbot_read_tap(btsbot_handle h, const char* name, float* dst, int64_t capacity,
                        void* stream);

/* ---- op-level entry points (the ATen ops of SURVEY.md section 2.2 K1-K6, one kernel family each);
 * forward() is a schedule of these.  `prec` is enum btsbot_precision: the type of the staged
 * activations / filters (float, bf16 or f16 device arrays); fp32 everywhere else. ---- */

/* K4/K5/K6: out = epi(X[M,K] . W[N,K]^T + bias[N]) on MFMA.  Replaces conv2d 1x1 (+gelu) /
 * conv2d 1x1 + layer-scale + residual / conv2d 2x2 s2 on pre-gathered patches.
 *   epi 0: out (prec) = gelu(acc + bias)         epi 1: out (f32) = resid + gamma * (acc + bias)
 *   epi 2: out (f32) = acc + bias.   K % (16/sizeof(prec)) == 0, N % 4 == 0.                    */
int btsbot_op_gemm(int prec, int epi, const void* X, const void* W, const float* bias,
                   const float* gamma, const float* resid, void* out, int M, int N, int K,
                   void* stream);
/* K2+K3: depthwise 7x7 p3 + bias + LayerNorm(C, eps 1e-6).  x [B,HW,HW,C] f32 NHWC ->
 * xn [B,HW,HW,C] (prec).  w_tap_major is [49][C] f32.  (C,HW) in {(64,15),(128,7),(256,3),(512,1),
 * (80,15),(160,7),(320,3),(640,1)}.  Arithmetic: fp32 FMAs on the fp32 map, two-pass variance -- EXCEPT the 15x15 maps
 * in the bf16 / f16 modes, which run the convolution on the matrix pipe: map and taps are rounded to `prec` before
 * the 49 products (fp32 accumulation) and the variance is single-pass (E[d^2] - mean^2, clamped at 0), so an output
 * map whose per-pixel mean dwarfs its spread loses bits there; BTSBOT_AMD_NO_DW15=1 (process-wide) keeps the fp32
 * per-tap kernel. */
int btsbot_op_dwconv_ln(int prec, const float* x, const float* w_tap_major, const float* bias,
                        const float* ln_w, const float* ln_b, void* xn, int B, int HW, int C,
                        void* stream);
/* K1: conv2d 4x4 s4 + bias + LayerNorm(C0).  img [B,3,63,63] f32 -> out [B,225,C0] f32 NHWC;
 * w is [C0][3][4][4] as PyTorch stores it.  C0 in {64, 80}. */
int btsbot_op_stem(const float* img, const float* w, const float* bias, const float* ln_w,
                   const float* ln_b, float* out, int B, int C0, void* stream);
/* K6 prologue: LayerNorm(Cin) + 2x2/s2 patch gather.  x [B,HW,HW,Cin] f32 ->
 * patches [B*(HW/2)^2, 4*Cin] (prec), k = (ky*2+kx)*Cin + c. */
int btsbot_op_ln_patch(int prec, const float* x, const float* ln_w, const float* ln_b,
                       void* patches, int B, int HW, int Cin, void* stream);

/* Measurement aid with no reference counterpart (bench.py's roofline leg): when on, every kernel
 * launch of forward() is bracketed by two HIP events recorded on the launch stream;
 * profile_collect() waits for them and returns, per kernel family (profile_category_name), the
 * summed device time in ms and the number of launches since the last collect. */
int btsbot_set_profile(btsbot_handle h, int on);
int btsbot_profile_categories(void);
const char* btsbot_profile_category_name(int category);
int btsbot_profile_collect(btsbot_handle h, int n_categories, double* ms_sum, int64_t* launches);

/* Replaces: BCEWithLogitsLoss(pos_weight)(logits, labels) and its autograd
 * (train.py:211-212,525-526).  labels are fp32 0/1.  loss_sum (1 float, caller-zeroed)
 * accumulates sum_i l_i (divide by n_global for the mean); dlogits = d(mean loss)/dz over
 * n_global alerts (n_global = global batch across ranks). */
int btsbot_bce_fwd_bwd(const float* logits, const float* labels, float pos_weight,
                       int batch, int n_global, float* loss_sum, float* dlogits, void* stream);

/* Replaces: torch.optim.AdamW.step (train.py:242-246,527): decoupled weight decay, bias
 * correction, eps outside the sqrt-correction exactly as torch (amsgrad off).  Flat arenas of
 * n floats; `step` is 1-based. */
int btsbot_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                      int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, void* stream);

/* ---- the callers either side of the path ---- */

/* Replaces: FlexibleDataset.__getitem__ + DataLoader collation + the torchvision transforms of
 * train.py:178-199 / utils.py:12-48 for a training set resident in HBM.  dst[b] = T_b(src[index[b]]),
 * T_b = rot90^k(vflip?(hflip?(.))) with ops[b] = hflip | vflip<<1 | k<<2 (k quarter turns counter-clockwise,
 * utils.py:44-48); src [N,3,63,63] f32, dst [batch,3,63,63] f32, index int64 [batch] (NULL = identity),
 * ops uint8 [batch] (NULL = no transform).  Pure index permutation: bit-exact. */
int btsbot_augment(const float* src, const int64_t* index, const uint8_t* ops, float* dst, int batch,
                   void* stream);

/* Replaces: the arithmetic of make_triplet (alert_utils.py:110-196) once the host has gunzipped and
 * FITS-decoded the stamps: per cutout (science, template, difference) nanmedian +-inf test, nan_to_num,
 * L2 normalisation (skipped once the alert is flagged), all-zero test, padding to 63x63 with 1e-9 at the
 * bottom / right; output in the float32 NCHW layout of inference_example.py:62-64.
 * raw [batch,3,63,63] f32 with each stamp in the top-left h x w corner; shapes int32 [batch,3,2] = (h, w)
 * per stamp (NULL = all 63x63); triplets [batch,3,63,63] f32; drop uint8 [batch] (NULL = not wanted). */
int btsbot_prep_triplets(const float* raw, const int* shapes, float* triplets, uint8_t* drop, int batch,
                         int normalize, void* stream);

/* Replaces: the epoch / validation metrics of val.py:159-168 and train.py:550-558 -- out2[0] += sum_i of
 * BCEWithLogitsLoss(pos_weight) terms over n logits, out2[1] += number of alerts whose sigmoid(z) > 0.5
 * agrees with the label (caller zeroes out2 and divides by n). */
int btsbot_eval_metrics(const float* logits, const float* labels, float pos_weight, int64_t n,
                        float* out2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BTSBOT_HIP_H */
