// Training-mode schedule of the ConvNeXt image branch: a forward that keeps what the backward
// needs, and the backward itself (kernels in backward.hip / dwln_bwd.hip / wgrad.hip / gemm2.hip / convnext.hip).
//
// Replaces, for the timm backbone reached at /root/reference/btsbot/architectures.py:108,132, what
// torch autograd does between model(...) (train.py:510) and loss.backward() (train.py:526).
// The image branch has no BatchNorm and no dropout (drop-path 0): its training forward computes the
// same function as inference, only unfused (per-op kernels) with per-block buffers:
//     kept per block : x_in (fp32), d = dwconv(x_in) (fp32), xn = LN(d) (operand type),
//                      a = fc1 pre-activation, h = gelu(a)
//     kept per stage : the stage output (input of the next downsample), the 2x2 patch matrix
//     kept once      : the stem convolution's output before its LayerNorm (fp32)
// Everything else (LN statistics) is recomputed in the backward.
#include <string.h>

#include "ctx.h"
#include "stage0.h"
#include "stage2p.h"

namespace {

struct BlkBuf {
  float* xin;
  float* d;      // depthwise output before the LayerNorm (fp32)
  void* xn;
  void* a;
  void* h;
  // backward only: the gradients the filter-gradient GEMMs read, one pair per block so that those GEMMs can trail
  // the input-gradient chain on a second stream without a block overwriting what the previous one still reads
  void* dyT;     // d(loss)/d(block output), operand type [rows][C]
  void* da;      // d(loss)/d(fc1 pre-activation), operand type [rows][4C]
  float* dwpart; // dwln_bwd_kernel's partial rows [dwrows][52 C] (nullptr: the block runs the three-launch form)
  int dwrows;
  float* fS;     // ... and its colsum(dy) [C] (the side stream's fc2_grads_kernel reads and clears it)
  float* Gb;     // batched filter gradients (ctx.h: wgrad_batch): this block's own G = dy^T h [C][4C] and colsum(dy) [C]
  float* Sb;     // (nullptr: the block's filter-gradient GEMMs are launched behind it, into the shared G / S)
  float* fpart;  // mlp_bwd_kernel's partial filter-gradient tiles (nullptr: the block runs the unfused MLP backward);
                 // one region per block: the side stream reduces block j's while the chain writes block j-1's
};

constexpr size_t WPART_FLOATS = (size_t)16 << 20;   // 64 MB: every shape of the pico / nano schedule fits (else atomics)

struct BBCache {
  float* xs[4];             // stage outputs [B*P_i*C_i] fp32
  void* patches[4];         // [B*P_i][4*C_{i-1}] operand type (i >= 1)
  std::vector<std::vector<BlkBuf>> blk;
  // backward scratch
  float *dyA, *dyB, *dC;    // fp32 [max rows*C]
  void* dyT_down[4];        // operand type [rows_i][C_i]: d(loss)/d(downsample i output) (i >= 1)
  void* dyT_stem;           // operand type [B*225][C0]: gradient behind the stem LayerNorm
  float *G, *S;             // fp32 [max C*4C], [max 4C]
  float* gb0;               // the batched blocks' G / S buffers: one region, cleared with one memset
  size_t gb_floats;
  float* fS0;               // first of the fused blocks' colsum buffers (they sit directly in front of S), or nullptr
  size_t g_floats;
  float* dpat;              // fp32 [max rows*4Cin]
  float* dwpart;            // fp32 [256][50*Cmax] per-workgroup partials of the depthwise wgrad
  float* wpart;             // fp32 2 x WPART_FLOATS: slice partials of the filter-gradient GEMMs (wgrad.hip)
  void* stem_patches;       // [B*225][48] operand type
  float* stem_pre;          // [B*225][C0] fp32
  size_t total;
};

size_t al(size_t n) { return (n + 255) / 256 * 256; }

// carve the cache for a batch of B alerts; base == nullptr only measures
BBCache carve_bb(const btsbot_ctx* h, unsigned char* base, int B) {
  const btsbot_config& c = h->cfg;
  const size_t esz = h->esz();
  BBCache k;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    unsigned char* p = base ? base + cur : nullptr;
    cur += al(bytes);
    return p;
  };
  size_t maxrc = 0, maxc4c = 0, maxpat = 0, maxplanes = 0;
  k.blk.resize(4);
  for (int i = 0; i < 4; ++i) {
    const size_t rows = (size_t)B * STAGE_HW[i] * STAGE_HW[i], ch = c.dims[i];
    k.xs[i] = reinterpret_cast<float*>(take(rows * ch * 4));
    k.patches[i] = i > 0 ? take(rows * 4 * c.dims[i - 1] * esz) : nullptr;
    if (i > 0 && rows * 4 * c.dims[i - 1] > maxpat) maxpat = rows * 4 * c.dims[i - 1];
    // (stage 2 under stage2p_kernel's keeping form: room for 9 more alerts behind the batch -- the kernel stores the rows
    //  of a ragged last workgroup and of its pad columns unconditionally, stage2p.hip)
    const size_t prows = i == 2 && h->s2p_train ? rows + 9 * 9 : rows;
    for (int j = 0; j < c.depths[i]; ++j) {
      BlkBuf b;
      b.xin = reinterpret_cast<float*>(take(prows * ch * 4));
      b.d = reinterpret_cast<float*>(take(prows * ch * 4));
      b.xn = take(prows * ch * esz);
      const bool keep4c = !h->mlp_fused((int)ch);   // fused blocks keep nothing 4C-wide (mlp_bwd.hip recomputes fc1)
      b.a = keep4c ? take(prows * 4 * ch * esz) : nullptr;
      b.h = keep4c ? take(prows * 4 * ch * esz) : nullptr;
      b.dyT = take(rows * ch * esz);
      b.da = keep4c ? take((rows + 48) * 4 * ch * esz) : nullptr;   // (+ 48 rows: s2mlp_bwd_kernel's last workgroup stores its dead rows too)
      b.dwrows = h->use_dwln && dwln_bwd_supported(STAGE_HW[i], (int)ch) ? dwln_bwd_rows(STAGE_HW[i], (int)ch, B) : 0;
      b.dwpart = b.dwrows ? reinterpret_cast<float*>(take((size_t)b.dwrows * 52 * ch * 4)) : nullptr;
      b.fpart = h->mlp_fused((int)ch) ? reinterpret_cast<float*>(take(mlp_bwd_part_floats((int)ch, (int)rows) * 4))
                                      : nullptr;
      b.fS = nullptr;
      b.Gb = b.Sb = nullptr;
      k.blk[i].push_back(b);
    }
    k.dyT_down[i] = i > 0 ? take(rows * ch * esz) : nullptr;
    if (rows * ch > maxrc) maxrc = rows * ch;
    if (h->mlp_fused((int)ch) && rows * ch * mlp_bwd_planes((int)ch) > maxplanes) maxplanes = rows * ch * mlp_bwd_planes((int)ch);
    if (4 * ch * ch > maxc4c) maxc4c = 4 * ch * ch;
    if (i > 0 && 4 * ch * c.dims[i - 1] > maxc4c) maxc4c = 4 * ch * c.dims[i - 1];
  }
  k.dyA = reinterpret_cast<float*>(take(maxrc * 4));
  k.dyB = reinterpret_cast<float*>(take((maxplanes > maxrc ? maxplanes : maxrc) * 4));   // dxn, possibly as addend planes
  k.dC = reinterpret_cast<float*>(take(maxrc * 4));
  k.dyT_stem = take((size_t)B * 225 * c.dims[0] * esz);
  for (int i = 0; i < 4; ++i)                                          // the fused blocks' colsum(dy), S, G: one memset
    for (auto& b : k.blk[i])
      if (b.fpart != nullptr) b.fS = reinterpret_cast<float*>(take((size_t)c.dims[i] * 4));
  k.fS0 = nullptr;
  for (int i = 0; i < 4 && k.fS0 == nullptr; ++i)
    for (auto& b : k.blk[i])
      if (b.fS != nullptr) {
        k.fS0 = b.fS;
        break;
      }
  k.S = reinterpret_cast<float*>(take((size_t)4 * c.dims[3] * 4));   // S directly in front of G:
  k.G = reinterpret_cast<float*>(take(maxc4c * 4));                   // one memset clears both
  k.g_floats = maxc4c;
  // per-block G / S of the stages whose filter-gradient GEMMs are batched (unfused blocks, widths that tile by 128)
  k.gb0 = reinterpret_cast<float*>(take(0));
  {
    const size_t start = cur;
    for (int i = 0; i < 4; ++i) {
      const size_t ch = c.dims[i];
      if (!h->wgrad_batch || h->mlp_fused((int)ch) || (ch & 127) != 0 || c.precision == BTSBOT_F32) continue;
      for (auto& b : k.blk[i]) {
        b.Gb = reinterpret_cast<float*>(take(4 * ch * ch * 4));
        b.Sb = reinterpret_cast<float*>(take(ch * 4));
      }
    }
    k.gb_floats = (cur - start) / 4;
  }
  k.dpat = reinterpret_cast<float*>(take(maxpat * 4));
  k.dwpart = reinterpret_cast<float*>(take((size_t)256 * 50 * c.dims[3] * 4));
  k.wpart = reinterpret_cast<float*>(take(2 * WPART_FLOATS * 4));   // two GEMMs' partial tiles at a time
  k.stem_patches = take((size_t)B * 225 * 48 * esz);
  k.stem_pre = reinterpret_cast<float*>(take((size_t)B * 225 * c.dims[0] * 4));
  k.total = cur;
  return k;
}

#define TRYB(call)                  \
  do {                              \
    int _s = (call);                \
    if (_s != BTSBOT_OK) return _s; \
  } while (0)

}  // namespace

size_t bb_cache_bytes(const btsbot_ctx* h, int B) { return carve_bb(h, nullptr, B).total; }

// training forward of the image branch for the whole batch; *feat_out = [B][dims[3]] fp32
int backbone_train_forward(btsbot_ctx* h, const float* img, int B, hipStream_t st,
                           float** feat_out) {
  const btsbot_config& c = h->cfg;
  const float* m = h->mirror;
  BBCache k = carve_bb(h, h->bbcache, B);
  // A block's input is kept for its backward.  Instead of copying it aside, every producer writes
  // straight into the buffer its consumer keeps: the stem / downsample into block 0's xin, block j's
  // fc2 (+ residual) into block j+1's xin, the stage's last block into xs[i].
  auto stage_in = [&](int i) { return h->blocks[i].empty() ? k.xs[i] : k.blk[i][0].xin; };
  // (the pre-LayerNorm output is kept for the backward; 16-bit modes: the matrix-pipe stem, which like stem_kernel reads
  //  nothing packed -- the operand re-pack of this step runs on the side stream meanwhile)
  const bool s0t = h->s0_train && h->blocks[0].size() == 2 && h->mlp_fused(c.dims[0]) && h->use_dwln;
  if (s0t) {
    // stem + stage 0 + the first downsample as one launch of the inference megakernel's keeping form: every buffer the
    // backward reads of them is written on its way (stage0.h).  It needs this step's operand images at once: the
    // re-pack queued behind the previous optimiser step runs under the mask draws and whatever precedes this call
    TRYB(pack_sync_early(h, st));
    Stage0Args a;
    memset(&a, 0, sizeof(a));
    a.img = img;
    a.stem_w = h->extra + h->p_stem16;
    a.stem_b = m + h->stem_b;
    a.stem_lnw = m + h->stem_lnw;
    a.stem_lnb = m + h->stem_lnb;
    for (int j = 0; j < 2; ++j) {
      const BlockPk& b = h->blocks[0][j];
      a.blk[j].dw_w = reinterpret_cast<const float*>(h->extra + b.p_dw);
      a.blk[j].dw_b = m + b.dw_b;
      a.blk[j].ln_w = m + b.ln_w;
      a.blk[j].ln_b = m + b.ln_b;
      a.blk[j].b1 = m + b.fc1_b;
      a.blk[j].b2 = m + b.fc2_b;
      a.blk[j].gamma = m + b.gamma;
      a.blk[j].w1 = h->extra + b.p_fc1;
      a.blk[j].w2g = h->extra + b.p_fc2g;
      a.blk[j].par = h->extra + b.p_s0par_t;
      a.keep_d[j] = k.blk[0][j].d;
      a.keep_xn[j] = k.blk[0][j].xn;
    }
    a.ds_lnw = m + h->down[1].ln_w;
    a.ds_lnb = m + h->down[1].ln_b;
    a.ds_w = h->extra + h->down[1].p_w;
    a.ds_b = m + h->down[1].b;
    a.out = stage_in(1);
    a.keep_stem_pre = k.stem_pre;
    a.tap_stem = k.blk[0][0].xin;
    a.keep_xin1 = k.blk[0][1].xin;
    a.tap_stage = k.xs[0];
    a.keep_patches = k.patches[1];
    a.B = B;
    TRYB(launch_stage0b(c.precision, a, st));
  } else if (h->use_stem16 && stem16_supported(c.precision, c.dims[0]))
    TRYB(launch_stem16(c.precision, img, m + h->stem_w, m + h->stem_b, m + h->stem_lnw, m + h->stem_lnb, stage_in(0), B,
                       c.dims[0], st, k.stem_pre));
  else
    TRYB(launch_stem(img, m + h->stem_w, m + h->stem_b, m + h->stem_lnw, m + h->stem_lnb, stage_in(0),
                     B, c.dims[0], st, k.stem_pre));
  TRYB(pack_sync(h, st));   // the operand images of this step (packed on the side stream while the stem ran)
  const bool s1t = h->s1_train && h->blocks[1].size() == 2 && h->mlp_fused(c.dims[1]) && h->use_dwln;
  // stage 2 through stage2p_kernel's keeping form up to two rounds of one workgroup (5 alerts) per CU: 2.50 against 2.57 ms
  // per 1024-alert step; at 4096 alerts the per-op GEMMs (36 864 rows: full tiles, full rounds) are as fast or faster
  // (7.95 against 8.00 ms), so large batches keep them
  const bool s2t = h->s2p_train && B <= 2560;
  for (int i = s0t ? 1 : 0; i < 4; ++i) {
    const int ch = c.dims[i], hw = STAGE_HW[i], rows = B * hw * hw;
    if (i > 0 && !(i == 3 && s2t) && !(i == 1 && s0t) && !(i == 2 && s1t)) {
      const int cin = c.dims[i - 1];
      TRYB(launch_ln_patch(c.precision, k.xs[i - 1], m + h->down[i].ln_w, m + h->down[i].ln_b,
                           k.patches[i], B, STAGE_HW[i - 1], cin, st));
      TRYB(launch_gemm(c.precision, EPI_BIAS, k.patches[i], h->extra + h->down[i].p_w,
                       m + h->down[i].b, nullptr, nullptr, stage_in(i), rows, ch, 4 * cin, st));
    }
    const size_t nblk = h->blocks[i].size();
    if (i == 1 && s1t) {
      // stage 1 + the second downsample as one launch of stage1b's keeping form: block 0's input is its x_in, block 1's
      // input the residual copy the kernel parks between the blocks anyway
      Stage1Args a;
      memset(&a, 0, sizeof(a));
      a.x_in = stage_in(1);
      for (int j = 0; j < 2; ++j) {
        const BlockPk& b = h->blocks[1][j];
        a.blk[j].dw_w = reinterpret_cast<const float*>(h->extra + b.p_dw);
        a.blk[j].dw_b = m + b.dw_b;
        a.blk[j].ln_w = m + b.ln_w;
        a.blk[j].ln_b = m + b.ln_b;
        a.blk[j].b1 = m + b.fc1_b;
        a.blk[j].b2 = m + b.fc2_b;
        a.blk[j].gamma = m + b.gamma;
        a.blk[j].w1 = h->extra + b.p_fc1;
        a.blk[j].w2g = h->extra + b.p_fc2g;
        a.blk[j].par = h->extra + b.p_s0par_t;
        a.keep_d[j] = k.blk[1][j].d;
        a.keep_xn[j] = k.blk[1][j].xn;
      }
      a.ds_lnw = m + h->down[2].ln_w;
      a.ds_lnb = m + h->down[2].ln_b;
      a.ds_w = h->extra + h->down[2].p_wp;
      a.ds_b = m + h->down[2].b;
      a.out = stage_in(2);
      a.scratch = k.blk[1][1].xin;
      a.tap_stage = k.xs[1];
      a.keep_patches = k.patches[2];
      a.B = B;
      TRYB(launch_stage1b(c.precision, a, st));
      continue;
    }
    if (i == 2 && s2t) {
      // Stage 2 and the last downsample as ONE launch: the inference kernel's keeping form (stage2p.hip, TRAIN) writes
      // what the backward reads -- every block's input, LayerNorm output, fc1 pre-activation and GELU, the stage output,
      // the downsample's patch rows -- on its way.  Replaces 6 x (dw3_ln + two GEMMs) + ln_patch + GEMM.  The depthwise
      // output is not kept: dw3ln_bwd_kernel recomputes it (block 0's buffer for it takes the copy of the stage input the
      // kernel writes for every block alike).
      Stage2pArgs a;
      memset(&a, 0, sizeof(a));
      a.cw = c.dims[2];
      a.x_in = stage_in(2);
      a.depth = (int)nblk;
      for (size_t j = 0; j < nblk; ++j) {
        const BlockPk& b = h->blocks[2][j];
        const BlkBuf& sb = k.blk[2][j];
        a.blk[j].dw_w = reinterpret_cast<const float*>(h->extra + b.p_dw);
        a.blk[j].dw_b = m + b.dw_b;
        a.blk[j].ln_w = m + b.ln_w;
        a.blk[j].ln_b = m + b.ln_b;
        a.blk[j].b1 = m + b.fc1_b;
        a.blk[j].b2 = m + b.fc2_b;
        a.blk[j].gamma = m + b.gamma;
        a.blk[j].w1p = h->extra + b.p_w1p;
        a.blk[j].w2p = h->extra + b.p_w2p;
        a.keep[j].xin = j == 0 ? sb.d : sb.xin;
        a.keep[j].xn = sb.xn;
        a.keep[j].a = sb.a;
        a.keep[j].hh = sb.h;
      }
      a.ds_lnw = m + h->down[3].ln_w;
      a.ds_lnb = m + h->down[3].ln_b;
      a.ds_wp = h->extra + h->down[3].p_wp;
      a.ds_b = m + h->down[3].b;
      a.out = stage_in(3);
      a.tap_stage = k.xs[2];
      a.ds_patches = k.patches[3];
      a.B = B;
      a.train = 1;
      a.alerts_hint = h->s2p_alerts_hint;
      a.stamps = h->stamps ? h->stamps + 32 + 16384 : nullptr;   // (tools/stamps_train.py)
      TRYB(launch_stage2p(c.precision, a, st));
      continue;
    }
    for (size_t j = 0; j < nblk; ++j) {
      const BlockPk& b = h->blocks[i][j];
      const BlkBuf& s = k.blk[i][j];
      float* xout = j + 1 < nblk ? k.blk[i][j + 1].xin : k.xs[i];
      TRYB(launch_dwconv_ln(c.precision, s.xin, reinterpret_cast<const float*>(h->extra + b.p_dw),
                            m + b.dw_b, m + b.ln_w, m + b.ln_b, s.xn, B, hw, ch, st, s.d));
      if (s.fpart != nullptr) {
        // the block's backward recomputes fc1 (mlp_bwd.hip): neither the pre-activation nor the GELU output is kept
        TRYB(launch_fused_mlp(c.precision, ch, s.xn, h->extra + b.p_fused, m + b.fc1_b, m + b.fc2_b, m + b.gamma, xout,
                              rows, st, nullptr, nullptr, nullptr, 0, s.xin));
        continue;
      }
      TRYB(launch_gemm(c.precision, EPI_GELU_SAVE, s.xn, h->extra + b.p_fc1, m + b.fc1_b, nullptr,
                       reinterpret_cast<const float*>(s.a), s.h, rows, 4 * ch, ch, st));
      TRYB(launch_gemm(c.precision, EPI_RESID, s.h, h->extra + b.p_fc2, m + b.fc2_b, m + b.gamma,
                       s.xin, xout, rows, ch, 4 * ch, st));
    }
  }
  *feat_out = k.xs[3];
  h->bb_saved = true;
  return BTSBOT_OK;
}

// backward of the image branch.  dfeat: [B][dims[3]] fp32 gradient w.r.t. the branch output
// (before the head LayerNorm, which the caller has already differentiated).  Gradients are
// ACCUMULATED (atomics) into `grads` (master-arena layout): the caller zeroes the image-branch
// range first.
// out[n][k] += sum_m D[m][n] A[m][k] and cs[n] += sum_m D[m][n]; D and A in the mode's operand type
static int wgrad_cs(int prec, const void* D, const void* A, float* out, float* cs, int M, int N,
                    int K, int ldo, hipStream_t st, float* part = nullptr, WgradReduceJob* defer = nullptr) {
  if (defer != nullptr) defer->nsl = 0;
  if (prec != BTSBOT_F32) {
    // (deterministic mode: the column sum as a launch of its own -- inside the filter-gradient GEMM its slices meet
    //  through atomics; launch_colsum writes partial rows and adds them in a fixed order)
    float* probe = det_alloc(0);
    if (probe != nullptr && cs != nullptr) {
      const int rc = launch_wgrad16(prec, D, A, out, nullptr, M, N, K, ldo, st, part, WPART_FLOATS, defer);
      return rc != BTSBOT_OK ? rc : launch_colsum(prec, D, cs, M, N, st);
    }
    return launch_wgrad16(prec, D, A, out, cs, M, N, K, ldo, st, part, WPART_FLOATS, defer);
  }
  return launch_wgrad_cs_f32(reinterpret_cast<const float*>(D), reinterpret_cast<const float*>(A), out, cs, M, N, K, ldo, st);
}

int backbone_train_backward(btsbot_ctx* h, const float* img, const float* dfeat, float* grads,
                            int B, hipStream_t st) {
  const btsbot_config& c = h->cfg;
  const float* m = h->mirror;
  const int prec = c.precision;
  BBCache k = carve_bb(h, h->bbcache, B);
  float* dy = k.dyA;
  float* dxn = k.dyB;
  TRYB(launch_copy_f32(dy, dfeat, (size_t)B * c.dims[3], st));
  // Two streams.  `st` carries the chain every block waits on (dy -> da -> dxn -> LayerNorm -> depthwise -> dy of the
  // block before); the filter-gradient GEMMs, their slice reductions and the layer-scale / bias gradients hang off
  // that chain (nothing downstream reads them before the optimiser), so they trail it on `h->side`: in stages 2-3
  // neither the chain's kernels (9216 / 1024 rows) nor these fill the chip on their own.  fork() = the side stream
  // sees everything queued on `st` so far, join() = `st` waits for the side stream (bucket boundaries and the end).
  // G / S / wpart are touched on the side stream only.  BTSBOT_AMD_NO_SIDE_STREAM=1: everything on `st` (A/B).
  hipStream_t sd = st;
  // dwln_bwd_kernel leaves one partial row per workgroup; the column sum that folds them into the arena is queued on
  // the side stream at the next fork (it then sees the kernel's rows) instead of behind the kernel in the chain
  // (every block has partial rows of its own, so several may wait for the same fork: the blocks of a stage whose
  //  filter-gradient GEMMs go out as one batch fork once, behind the stage -- a fork is an event record in the chain's
  //  queue and costs it ~6 us between two kernels)
  struct Pend { const float* part; float* dst; int rows, cols; };   // cols == 0: dw3 compact rows
  Pend pend[8];
  int npend = 0;
  auto fork = [&]() -> int {
    TRYB(side_fork(h, st, &sd));
    for (int q = 0; q < npend; ++q) {
      if (pend[q].cols == 0) TRYB(launch_dw3_rows(pend[q].part, pend[q].dst, pend[q].rows, sd));
      else TRYB(launch_colsum(BTSBOT_F32, pend[q].part, pend[q].dst, pend[q].rows, pend[q].cols, sd));
    }
    npend = 0;
    return BTSBOT_OK;
  };
  auto join = [&]() -> int {
    if (npend > 0) TRYB(fork());
    return side_join(h, st);
  };
  auto add_pend = [&](const float* part, float* dst, int rows, int cols) -> int {
    if (npend == (int)(sizeof(pend) / sizeof(pend[0]))) TRYB(fork());
    pend[npend++] = Pend{part, dst, rows, cols};
    return BTSBOT_OK;
  };
  // 16-bit modes: the depthwise input-gradient kernel that ends a block also writes dy in the operand type (the cast
  // launch in front of the next block is then skipped)
  const bool fold_cast = prec != BTSBOT_F32;
  // G / S (S directly in front of G) are the accumulators of the fc2 / downsample filter-gradient GEMMs: cleared once
  // here, afterwards every consumer (fc2_grads_kernel, unpack_down_grad_kernel) leaves what it read zero
  {
    float* z0 = k.fS0 != nullptr ? k.fS0 : k.S;
    TRYB(launch_fill0(z0, (size_t)((k.G + k.g_floats) - z0), st));
  }
  // (deterministic mode: the per-GEMM launches with their fixed-order column sums)
  const bool batching = k.gb_floats > 0 && det_alloc(0) == nullptr;
  if (batching) TRYB(launch_fill0(k.gb0, k.gb_floats, st));
  // operand-type buffer the consumer after block (i, j) reads its dy from: the block before, else the downsample
  auto next_dyT = [&](int i, int j) -> void* {
    if (j > 0) return k.blk[i][j - 1].dyT;
    return i > 0 ? k.dyT_down[i] : nullptr;
  };
  // the stem filter gradient's operand depends on the triplets only: built on the side stream while it is still idle
  // (at the end of the chain it was 36 us of the step's tail)
  // (no fork of its own: it reads the caller's triplets, which the forward already read on this stream, and the side
  //  stream has the head's backward in front of it, queued behind a fork a few launches ago)
  if (h->use_side && h->side != nullptr) sd = h->side;
  TRYB(launch_stem_im2col(prec, img, k.stem_patches, B, sd));
  bool dyT_ready = false;
  for (int i = 3; i >= 0; --i) {
    const int ch = c.dims[i], hw = STAGE_HW[i], rows = B * hw * hw, H = 4 * ch;
    WgradBatchJob bj[16];
    int nbj = 0;
    for (int j = (int)h->blocks[i].size() - 1; j >= 0; --j) {
      const BlockPk& b = h->blocks[i][j];
      const BlkBuf& s = k.blk[i][j];
      const float* wdw = reinterpret_cast<const float*>(h->extra + b.p_dw);
      const bool batched = batching && s.Gb != nullptr && s.fpart == nullptr && nbj + 2 <= 16;
      if (!dyT_ready)
        TRYB(launch_scale_cast(prec, dy, nullptr, s.dyT, (long)rows * ch, ch, st));
      dyT_ready = false;
      WgradReduceJob red[2];
      const bool adjacent = b.dw_b == b.dw_w + 49 * (int64_t)ch && b.ln_w == b.dw_b + ch && b.ln_b == b.ln_w + ch;
      // (the hidden slices of the 128-channel form hand dxn over as addend planes: dwln_bwd_kernel is their reader)
      const int planes = s.fpart != nullptr ? mlp_bwd_planes(ch) : 1;
      if (planes > 1 && !(s.dwpart != nullptr && adjacent)) {
        btsbot_set_error("backward: the fused MLP backward of a %d-channel block needs dwln_bwd_kernel behind it", ch);
        return BTSBOT_ERR_STATE;
      }
      if (s.fpart != nullptr) {
        // ---- da, dxn = da W1 and both filter gradients of the MLP in one launch (a recomputed from xn; da, g only on chip)
        TRYB(launch_mlp_bwd(prec, ch, s.xn, s.dyT, h->extra + b.p_fc1, h->extra + b.p_fc2t, m + b.fc1_b, dxn, s.fpart,
                            k.G, s.fS, grads + b.fc1_w, grads + b.fc1_b, rows, st, red));
        TRYB(fork());
      } else if (h->s2mlp && b.p_w1tp != 0 && s2mlp_bwd_supported(prec, ch)) {
        // ---- 256 channels: da = (dy (diag(gamma) W2)) * gelu'(a) and dxn = da W1 in one launch (s2mlp_bwd.hip)
        TRYB(launch_s2mlp_bwd(prec, s.dyT, s.a, h->extra + b.p_w2tp, h->extra + b.p_w1tp, s.da, dxn, rows, st));
        if (!batched || h->fork_per_block) TRYB(fork());
      } else {
      // ---- da = (dy (diag(gamma) W2)) * gelu'(a)     (gamma is folded into the packed W2^T)
      TRYB(launch_gemm(prec, EPI_DGELU, s.dyT, h->extra + b.p_fc2t, nullptr, nullptr,
                       reinterpret_cast<const float*>(s.a), s.da, rows, H, ch, st));
      // the side stream may start once da exists; its work is queued below, behind the chain's (a block whose GEMMs wait
      // for the stage's batch has nothing for it yet)
      if (!batched || h->fork_per_block) TRYB(fork());
      // ---- dxn = da W1, then the LayerNorm backward on the depthwise output d = dwconv(x_in) + bias the forward kept
      TRYB(launch_gemm(prec, EPI_PLAIN, s.da, h->extra + b.p_fc1t, nullptr, nullptr, nullptr, dxn,
                       rows, ch, H, st));
      }
      void* nxt = fold_cast ? next_dyT(i, j) : nullptr;
      // (the partial rows follow the arena's layout of conv_dw.weight | conv_dw.bias | norm.weight | norm.bias)
      if (s.dwpart != nullptr && adjacent && dw3_bwd_active(hw, ch)) {
        // ---- 3x3 maps: the same in their own kernel (d recomputed from x_in; compact rows, reduced by launch_dw3_rows)
        TRYB(launch_dw3ln_bwd(m + b.dw_b, dxn, m + b.ln_w, s.xin, wdw, dy, nxt, prec, s.dwpart, B, st, planes,
                              (size_t)rows * ch));
        TRYB(add_pend(s.dwpart, grads + b.dw_w, dw3_rows(B), 0));
      } else if (s.dwpart != nullptr && adjacent) {
        // ---- LayerNorm backward, depthwise filter gradient and dx = dy + conv_flipped(dd) in one launch
        TRYB(launch_dwln_bwd(s.d, dxn, m + b.ln_w, s.xin, wdw, dy, nxt, prec, s.dwpart, B, hw, ch, st, planes, (size_t)rows * ch));
        TRYB(add_pend(s.dwpart, grads + b.dw_w, s.dwrows, 52 * ch));
      } else if (hw == 1 && h->use_dwln && ch <= 640) {
        // ---- 1x1 maps: the same three steps per (alert, channel) in one launch
        TRYB(launch_ln_dw1_bwd(s.d, dxn, m + b.ln_w, s.xin, wdw, dy, nxt, prec, grads + b.ln_w, grads + b.ln_b,
                               grads + b.dw_w, grads + b.dw_b, rows, ch, st));
      } else {
        TRYB(launch_ln_bwd(s.d, dxn, m + b.ln_w, dxn, grads + b.ln_w, grads + b.ln_b, rows, ch, st));
        // ---- depthwise filter gradient.  Stays in the chain: behind a fork of its own (per-block dd buffers) the
        //      step was 0.02-0.06 ms slower, as was a single fork placed here instead of behind da
        TRYB(launch_dw_wgrad(s.xin, dxn, grads + b.dw_w, grads + b.dw_b, k.dwpart, B, hw, ch, st));
        // ---- depthwise input gradient: dx = dy + conv_flipped(dd)
        TRYB(launch_dw_plain(dxn, wdw, 1, nullptr, dy, dy, B, hw, ch, st, nxt, prec));
      }
      // ---- side: fc2 / layer-scale (S = colsum(dy), G = dy^T h), fc1 (dW1 += da^T xn, db1 += colsum(da)).  Queued
      //      after the chain's launches above (the fork itself sits behind da), so the host's five launches here never
      //      stand between the chain and its next kernel (measured: no difference, the host runs ahead either way)
      //      (the two GEMMs' slice reductions share one launch: separate halves of the partial-tile scratch)
      if (batched) {
        // both filter gradients of this block join the stage's batch (launched behind the stage's last block, below)
        bj[nbj++] = WgradBatchJob{s.dyT, s.h, s.Gb, s.Sb, rows, ch, H, H};
        bj[nbj++] = WgradBatchJob{s.da, s.xn, grads + b.fc1_w, grads + b.fc1_b, rows, H, ch, ch};
        dyT_ready = nxt != nullptr;
        continue;
      }
      if (s.fpart == nullptr) {
      TRYB(wgrad_cs(prec, s.dyT, s.h, k.G, k.S, rows, ch, H, H, sd, k.wpart, &red[0]));
      TRYB(wgrad_cs(prec, s.da, s.xn, grads + b.fc1_w, grads + b.fc1_b, rows, H, ch, ch, sd, k.wpart + WPART_FLOATS,
                    &red[1]));
      }
      TRYB(launch_wgrad_reduce(red, 2, sd));
      TRYB(launch_fc2_grads(k.G, s.fpart != nullptr ? s.fS : k.S, m + b.fc2_w, m + b.fc2_b, m + b.gamma, grads + b.fc2_w,
                            grads + b.fc2_b, grads + b.gamma, ch, H, sd));
      dyT_ready = nxt != nullptr;
    }
    auto stage_batch = [&]() -> int {
      // ---- the stage's filter-gradient GEMMs as one launch + one slice reduction, then the fc2 / layer-scale gradients
      //      of every block from its own G / S (side stream; the chain has passed the stage)
      TRYB(launch_wgrad16_batched(prec, bj, nbj, k.wpart, 2 * WPART_FLOATS, hw >= 3 ? 576 : 256, sd));
      for (int j = (int)h->blocks[i].size() - 1; j >= 0; --j) {
        const BlockPk& b = h->blocks[i][j];
        const BlkBuf& s = k.blk[i][j];
        if (s.Gb == nullptr) continue;
        TRYB(launch_fc2_grads(s.Gb, s.Sb, m + b.fc2_w, m + b.fc2_b, m + b.gamma, grads + b.fc2_w, grads + b.fc2_b,
                              grads + b.gamma, ch, H, sd));
      }
      nbj = 0;
      return BTSBOT_OK;
    };
    // (with a downsample in front of the stage the batch shares that one's fork)
    if (nbj > 0 && (i == 0 || h->fork_per_block)) {
      TRYB(fork());
      TRYB(stage_batch());
    }
    if (i > 0) {
      // ---- downsample backward: y = patches(LN(x_prev)) Wd^T + b
      const int cin = c.dims[i - 1], hwp = STAGE_HW[i - 1];
      const long prow = (long)B * hwp * hwp;
      void* dyT = k.dyT_down[i];
      if (!dyT_ready)
        TRYB(launch_scale_cast(prec, dy, nullptr, dyT, (long)rows * ch, ch, st));
      dyT_ready = false;
      TRYB(fork());
      if (nbj > 0) TRYB(stage_batch());
      TRYB(launch_gemm(prec, EPI_PLAIN, dyT, h->extra + h->down[i].p_wt, nullptr, nullptr,
                       nullptr, k.dpat, rows, 4 * cin, ch, st));
      // LN backward per input pixel (x_prev = stage i-1 output), its incoming gradient gathered from the patch matrix
      // (was an unpatch launch); result is the new dy, in the 16-bit modes also as the operand of stage i-1's last
      // block (was a cast launch)
      void* nxt16 = fold_cast && !h->blocks[i - 1].empty() ? k.blk[i - 1].back().dyT : nullptr;
      TRYB(launch_ln_bwd(k.xs[i - 1], k.dpat, m + h->down[i].ln_w, dy, grads + h->down[i].ln_w,
                         grads + h->down[i].ln_b, prow, cin, st, nxt16, prec, hwp));
      dyT_ready = nxt16 != nullptr;
      // (side work queued behind the chain's launches, as in the blocks)
      TRYB(wgrad_cs(prec, dyT, k.patches[i], k.G, grads + h->down[i].b, rows, ch, 4 * cin, 4 * cin, sd, k.wpart));
      TRYB(launch_unpack_down_grad(k.G, grads + h->down[i].w, ch, cin, sd));
    }
    // every gradient of stages.i.* (and, for i = 3, of the heads) is queued: bucket 3 - i is complete once the
    // side stream has drained what it holds AND seen the chain up to here -- so the event is recorded on the side
    // stream behind a fork, and the chain itself does not wait (a join here stalled it ~17 us twice per step)
    if (i >= 2 && h->n_buckets == 3 && h->bucket_fine) {
      TRYB(fork());
      HIP_TRY(hipEventRecord(h->bucket_ev[3 - i], sd));
    }
  }
  // ---- stem: y = LN(patches(img) Ws^T + bs);  dy is d(loss)/d(stem output) [B*225][C0]
  {
    const int c0 = c.dims[0], rows = B * 225;
    TRYB(launch_ln_bwd(k.stem_pre, dy, m + h->stem_lnw, dxn, grads + h->stem_lnw,
                       grads + h->stem_lnb, rows, c0, st, fold_cast ? k.dyT_stem : nullptr, prec));
    if (!fold_cast) TRYB(launch_scale_cast(prec, dxn, nullptr, k.dyT_stem, (long)rows * c0, c0, st));
    TRYB(fork());
    TRYB(wgrad_cs(prec, k.dyT_stem, k.stem_patches, grads + h->stem_w, grads + h->stem_b, rows, c0, 48, 48,
                  sd, k.wpart));
  }
  TRYB(join());
  if (h->n_buckets == 3 && h->bucket_fine) HIP_TRY(hipEventRecord(h->bucket_ev[2], st));
  return BTSBOT_OK;
}
