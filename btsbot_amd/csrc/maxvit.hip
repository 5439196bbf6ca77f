// MaxViT image branch: parameter table, operand images, workspace and the forward schedule.
//
// Reproduces timm's maxvit_tiny_rw_224 as BTSbot uses it (/root/reference/btsbot/architectures.py:25-101:
// MaxViT / mm_MaxViT resize the 63x63 cutouts to 224x224, run the backbone, keep head.global_pool only).
// Parameter names are timm's (stem.conv1 / norm1 / conv2, stages.i.blocks.j.{conv,attn_block,attn_grid}.*,
// norm); the algorithm is restated in oracle/maxvit_oracle.py.  Inference only: BatchNorm2d uses its
// running statistics and is folded into the neighbouring convolution at pack time.
//
// Schedule per MaxxVitBlock (x = fp32 residual map [nb*H*H, C] NHWC, T = staged activation type):
//   MBConv   sc = x | avgpool2(x) | avgpool2(x).Wsc^T            (stride 1 | stride 2 | stride 2 + widen)
//            a  = T(BN_pre(x))                                    mv_bn_cast
//            m1 = silu(a.W1'^T + b1')        [H*H, 4Cin]          GEMM, BN1 folded, EPI_SILU
//            m2 = silu(dw3x3_s(m1)*s2 + b2') [Ho*Ho, 4Cin]        mv_dw3, BN2 folded
//            g  = sigmoid(fc2(silu(fc1(mean m2))))                mv_se
//            x  = sc + (m2 * g).W3^T                              gated GEMM (gate on the A operand)
//   2 x partition attention (windows, then grid):
//            x += proj(attn(qkv(LN1(x))))                         mv_ln, GEMM EPI_BIAS_T, mv_attn, GEMM EPI_RESID
//            x += fc2(gelu(fc1(LN2(x))))                          mv_ln, GEMM EPI_GELU, GEMM EPI_RESID
//            16-bit modes, C = 64 / 128 / 256 (stages 0-2): both lines of a partition attention as ONE launch
//            (maxvit_part.hip: mv_part_kernel / mv_part64_kernel), the next block's BN_pre copy as its post-op
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ctx.h"
#include "maxvit.h"
#include "stage2p.h"

#include "maxvit_tables.h"

namespace {

constexpr int MV_DEPTHS[4] = {2, 2, 5, 2};
constexpr int MV_DIMS[4] = {64, 128, 256, 512};

int64_t mv_add(btsbot_ctx* h, const std::string& name, std::initializer_list<int> shape,
               int is_buffer = 0) {
  ParamRec r;
  r.name = name;
  r.off = h->total_floats;
  r.ndim = (int)shape.size();
  r.numel = 1;
  int i = 0;
  for (int s : shape) {
    r.shape[i++] = s;
    r.numel *= s;
  }
  for (; i < 4; ++i) r.shape[i] = 1;
  r.is_buffer = is_buffer;
  h->total_floats += (r.numel + 3) / 4 * 4;
  h->params.push_back(r);
  return r.off;
}

size_t mv_bump(size_t& cur, size_t bytes) {
  const size_t o = cur;
  cur += (bytes + 255) / 256 * 256;
  return o;
}

BnPk add_bn(btsbot_ctx* h, const std::string& p, int n, size_t& cur) {
  BnPk b;
  b.w = mv_add(h, p + "weight", {n});
  b.b = mv_add(h, p + "bias", {n});
  b.rm = mv_add(h, p + "running_mean", {n}, 1);
  b.rv = mv_add(h, p + "running_var", {n}, 1);
  b.p_scale = mv_bump(cur, (size_t)n * 4);
  b.p_shift = mv_bump(cur, (size_t)n * 4);
  return b;
}

AttnPk add_attn(btsbot_ctx* h, const std::string& p, int c, size_t& cur, int esz) {
  AttnPk a;
  const int heads = c / 32;
  a.n1w = mv_add(h, p + "norm1.weight", {c});
  a.n1b = mv_add(h, p + "norm1.bias", {c});
  a.qkv_w = mv_add(h, p + "attn.qkv.weight", {3 * c, c});
  a.qkv_b = mv_add(h, p + "attn.qkv.bias", {3 * c});
  a.rel = mv_add(h, p + "attn.rel_pos.relative_position_bias_table", {169, heads});
  a.proj_w = mv_add(h, p + "attn.proj.weight", {c, c});
  a.proj_b = mv_add(h, p + "attn.proj.bias", {c});
  a.n2w = mv_add(h, p + "norm2.weight", {c});
  a.n2b = mv_add(h, p + "norm2.bias", {c});
  a.fc1_w = mv_add(h, p + "mlp.fc1.weight", {4 * c, c});
  a.fc1_b = mv_add(h, p + "mlp.fc1.bias", {4 * c});
  a.fc2_w = mv_add(h, p + "mlp.fc2.weight", {c, 4 * c});
  a.fc2_b = mv_add(h, p + "mlp.fc2.bias", {c});
  a.p_qkv = mv_bump(cur, (size_t)3 * c * c * esz);
  a.p_proj = mv_bump(cur, (size_t)c * c * esz);
  a.p_fc1 = mv_bump(cur, (size_t)4 * c * c * esz);
  a.p_fc2 = mv_bump(cur, (size_t)4 * c * c * esz);
  a.p_bias = mv_bump(cur, (size_t)heads * 2401 * 4);
  a.p_bias64 = mv_bump(cur, (size_t)heads * 4096 * 4);
  a.fused = fused_mlp_supported(h->cfg.precision, c);
  a.p_fused = a.fused ? mv_bump(cur, fused_mlp_packed_bytes(c)) : 0;
  a.smlp = !a.fused && stage2p_rows_supported(h->cfg.precision, c);
  a.part = mv_part_supported(h->cfg.precision, c);
  if (a.smlp || a.part) {
    a.p_w1p = mv_bump(cur, (size_t)4 * c * c * esz);
    a.p_w2p = mv_bump(cur, (size_t)4 * c * c * esz);
  }
  if (a.part) {
    a.p_qkvp = mv_bump(cur, (size_t)3 * c * c * esz);
    a.p_projp = mv_bump(cur, (size_t)c * c * esz);
    a.p_biasl = mv_bump(cur, (size_t)heads * 4096 * 4);
  }
  return a;
}

}  // namespace

int maxvit_build_tables(btsbot_ctx* h, size_t* extra_cursor) {
  size_t cur = *extra_cursor;
  const int esz = h->esz();
  MaxVit* mv = new MaxVit();
  h->mv = mv;
  {
    const char* e = getenv("BTSBOT_AMD_MV_ATTN_VALU");
    mv->attn_valu = e != nullptr && e[0] == '1';
    const char* d = getenv("BTSBOT_AMD_MV_DW_PLAIN");
    mv->dw_plain = d != nullptr && d[0] == '1';
    const char* s2 = getenv("BTSBOT_AMD_MV_STEM_IM2COL");
    mv->stem_im2col = s2 != nullptr && s2[0] == '1';
    const char* gg = getenv("BTSBOT_AMD_MV_GATED_GEMM");
    mv->gated_gemm = gg != nullptr && gg[0] == '1';
    const char* nf = getenv("BTSBOT_AMD_MV_NO_FRONT");
    mv->no_front = nf != nullptr && nf[0] == '1';
    const char* lf = getenv("BTSBOT_AMD_MV_NO_LN_FUSE");
    mv->no_ln_fuse = lf != nullptr && lf[0] == '1';
    const char* ab = getenv("BTSBOT_AMD_MV_NO_ATTN_BLOCK");
    mv->no_attn_block = ab != nullptr && ab[0] == '1';
    const char* u = getenv("BTSBOT_AMD_MV_MLP_UNFUSED");
    mv->mlp_unfused = u != nullptr && u[0] == '1';
    const char* na = getenv("BTSBOT_AMD_MV_NO_PART");
    mv->no_part = na != nullptr && na[0] == '1';
    const char* ns = getenv("BTSBOT_AMD_MV_NO_SMLP");
    mv->no_smlp = ns != nullptr && ns[0] == '1';
  }
  char buf[96];
  mv->stem1_w = mv_add(h, "stem.conv1.weight", {32, 3, 3, 3});
  mv->stem_bn = add_bn(h, "stem.norm1.", 32, cur);
  mv->stem2_w = mv_add(h, "stem.conv2.weight", {64, 32, 3, 3});
  mv->p_stem1 = mv_bump(cur, (size_t)32 * 32 * esz);
  mv->p_stem2 = mv_bump(cur, (size_t)64 * 288 * esz);
  mv->p_zero = mv_bump(cur, 2048 * 4);
  mv->p_one = mv_bump(cur, 2048 * 4);
  int cin = 64, hw = 112;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < MV_DEPTHS[i]; ++j) {
      MvBlock b;
      b.cin = cin;
      b.c = MV_DIMS[i];
      b.mid = 4 * cin;
      b.rd = b.mid / 16;
      b.stride = j == 0 ? 2 : 1;
      b.hin = hw;
      b.hout = hw / b.stride;
      snprintf(buf, sizeof buf, "stages.%d.blocks.%d.", i, j);
      const std::string bp(buf), p = bp + "conv.";
      if (b.stride == 2 && b.cin != b.c) {
        b.sc_w = mv_add(h, p + "shortcut.expand.weight", {b.c, b.cin, 1, 1});
        b.p_sc = mv_bump(cur, (size_t)b.c * b.cin * esz);
      } else {
        b.p_sc = 0;
      }
      b.pre = add_bn(h, p + "pre_norm.", b.cin, cur);
      b.c1_w = mv_add(h, p + "conv1_1x1.weight", {b.mid, b.cin, 1, 1});
      b.c1_b = mv_add(h, p + "conv1_1x1.bias", {b.mid});
      b.n1 = add_bn(h, p + "norm1.", b.mid, cur);
      b.c2_w = mv_add(h, p + "conv2_kxk.weight", {b.mid, 1, 3, 3});
      b.c2_b = mv_add(h, p + "conv2_kxk.bias", {b.mid});
      b.n2 = add_bn(h, p + "norm2.", b.mid, cur);
      b.se1_w = mv_add(h, p + "se.fc1.weight", {b.rd, b.mid, 1, 1});
      b.se1_b = mv_add(h, p + "se.fc1.bias", {b.rd});
      b.se2_w = mv_add(h, p + "se.fc2.weight", {b.mid, b.rd, 1, 1});
      b.se2_b = mv_add(h, p + "se.fc2.bias", {b.mid});
      b.c3_w = mv_add(h, p + "conv3_1x1.weight", {b.c, b.mid, 1, 1});
      b.p_c1 = mv_bump(cur, (size_t)b.mid * b.cin * esz);
      b.p_c1b = mv_bump(cur, (size_t)b.mid * 4);
      b.p_dw = mv_bump(cur, (size_t)9 * b.mid * 4);
      b.p_dwb = mv_bump(cur, (size_t)b.mid * 4);
      b.p_c3 = mv_bump(cur, (size_t)b.c * b.mid * esz);
      b.p_se2t = mv_bump(cur, (size_t)b.mid * b.rd * 4);
      b.attn[0] = add_attn(h, bp + "attn_block.", b.c, cur, esz);
      b.attn[1] = add_attn(h, bp + "attn_grid.", b.c, cur, esz);
      mv->blocks.push_back(b);
      cin = b.c;
      hw = b.hout;
    }
  }
  mv->norm_w = mv_add(h, "norm.weight", {512});
  mv->norm_b = mv_add(h, "norm.bias", {512});
  *extra_cursor = cur;
  return BTSBOT_OK;
}

void maxvit_free(btsbot_ctx* h) {
  delete h->mv;
  h->mv = nullptr;
}

#define MTRY(call)                  \
  do {                              \
    int _s = (call);                \
    if (_s != BTSBOT_OK) return _s; \
  } while (0)

static int fold_bn(btsbot_ctx* h, const BnPk& b, int n, hipStream_t st) {
  const float* m = h->mirror;
  return launch_bn_fold(m + b.w, m + b.b, m + b.rm, m + b.rv,
                        reinterpret_cast<float*>(h->extra + b.p_scale),
                        reinterpret_cast<float*>(h->extra + b.p_shift), n, st);
}

int maxvit_pack(btsbot_ctx* h, hipStream_t st) {
  MaxVit* mv = h->mv;
  const int prec = h->cfg.precision;
  const float* m = h->mirror;
  unsigned char* ex = h->extra;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ex + off); };
  MTRY(fold_bn(h, mv->stem_bn, 32, st));
  MTRY(launch_mv_pack_stem1(prec, m + mv->stem1_w, F(mv->stem_bn.p_scale), ex + mv->p_stem1, st));
  MTRY(launch_mv_pack_conv3(prec, m + mv->stem2_w, ex + mv->p_stem2, 64, 32, st));
  MTRY(launch_mv_fill(F(mv->p_zero), 0.f, 2048, st));
  MTRY(launch_mv_fill(F(mv->p_one), 1.f, 2048, st));
  for (const MvBlock& b : mv->blocks) {
    if (b.sc_w >= 0) MTRY(launch_cast(prec, m + b.sc_w, ex + b.p_sc, (int64_t)b.c * b.cin, st));
    MTRY(fold_bn(h, b.pre, b.cin, st));
    MTRY(fold_bn(h, b.n1, b.mid, st));
    MTRY(fold_bn(h, b.n2, b.mid, st));
    // conv1_1x1 followed by BN1: W' = diag(s1) W, b' = b s1 + t1
    MTRY(launch_rowscale_cast(prec, m + b.c1_w, F(b.n1.p_scale), ex + b.p_c1, b.mid, b.cin, st));
    MTRY(launch_mv_fold_bias(m + b.c1_b, F(b.n1.p_scale), F(b.n1.p_shift), F(b.p_c1b), b.mid, st));
    MTRY(launch_mv_pack_dw(m + b.c2_w, F(b.n2.p_scale), F(b.p_dw), b.mid, st));
    MTRY(launch_mv_fold_bias(m + b.c2_b, F(b.n2.p_scale), F(b.n2.p_shift), F(b.p_dwb), b.mid, st));
    MTRY(launch_cast(prec, m + b.c3_w, ex + b.p_c3, (int64_t)b.c * b.mid, st));
    MTRY(launch_transpose_f32(m + b.se2_w, F(b.p_se2t), b.mid, b.rd, st));
    for (const AttnPk& a : b.attn) {
      const int c = b.c;
      MTRY(launch_cast(prec, m + a.qkv_w, ex + a.p_qkv, (int64_t)3 * c * c, st));
      MTRY(launch_cast(prec, m + a.proj_w, ex + a.p_proj, (int64_t)c * c, st));
      MTRY(launch_cast(prec, m + a.fc1_w, ex + a.p_fc1, (int64_t)4 * c * c, st));
      MTRY(launch_cast(prec, m + a.fc2_w, ex + a.p_fc2, (int64_t)4 * c * c, st));
      MTRY(launch_mv_pack_relbias(m + a.rel, F(a.p_bias), c / 32, st));
      MTRY(launch_mv_pack_relbias64(m + a.rel, F(a.p_bias64), c / 32, st));
      if (a.fused) MTRY(launch_pack_fused_mlp(prec, c, m + a.fc1_w, m + a.fc2_w, ex + a.p_fused, st));
      if (a.part) {
        MTRY(launch_pack_s2p(prec, m + a.qkv_w, nullptr, ex + a.p_qkvp, 3 * c, c, 0, 0, nullptr, st));
        MTRY(launch_pack_s2p(prec, m + a.proj_w, nullptr, ex + a.p_projp, c, c, 0, 0, nullptr, st));
        MTRY(launch_mv_pack_relbias_lanes(m + a.rel, F(a.p_biasl), c / 32, st));
      }
      if (a.smlp || a.part) {
        MTRY(launch_pack_s2p(prec, m + a.fc1_w, nullptr, ex + a.p_w1p, 4 * c, c, 0, 0, nullptr, st));
        MTRY(launch_pack_s2p(prec, m + a.fc2_w, nullptr, ex + a.p_w2p, c, 4 * c, 0, 0, nullptr, st));
      }
    }
  }
  return BTSBOT_OK;
}

// per-alert element counts of the workspace buffers (see the schedule at the top)
static void mv_layout(const btsbot_ctx* h, int chunk, MaxVit* out, size_t* total) {
  const size_t esz = (size_t)h->esz(), n = (size_t)chunk;
  size_t cur = 0;
  auto bump = [&](size_t bytes) {
    const size_t o = cur;
    cur += (bytes + 255) / 256 * 256;
    return o;
  };
  MaxVit tmp;
  MaxVit* o = out ? out : &tmp;
  o->o_x = bump(n * 12544 * 64 * 4);       // fp32 residual map (stem output is the largest)
  o->o_x2 = bump(n * 3136 * 64 * 4);       // second fp32 map (outputs of the stride-2 blocks)
  o->o_a = bump(n * 12544 * 288 * esz);    // stem im2col [12544,288]; later m1 [<=12544, 256]
  o->o_b = bump(n * 3136 * 256 * esz);     // conv1 im2col [12544,32]; m2; MLP hidden [3136,256]
  o->o_c = bump(n * 12544 * 64 * esz);     // stem conv1 output [12544,32]; BN / LN outputs
  o->o_d = bump(n * 3136 * 192 * esz);     // qkv
  o->o_e = bump(n * 3136 * 64 * esz);      // attention output; pooled shortcut input
  o->o_gate = bump(n * 2048 * 4);
  o->o_feat = bump(n * 512 * 4);
  o->o_wg = bump(n * 65536 * esz);          // per-alert gated conv3 filters (<= 128 x 512)
  o->o_sescr = bump(n * (2048 + 128) * 4);   // squeeze-excite mean + hidden
  o->o_part = bump(n * 16384 * 4);         // squeeze-excite partial sums [<=56 groups][mid], 14336 floats max
  *total = cur;
}

size_t maxvit_ws_bytes(const btsbot_ctx* h, int chunk) {
  size_t total = 0;
  mv_layout(h, chunk, nullptr, &total);
  return total;
}

// GEMM dispatch of the MaxViT schedule: the LDS-free streaming kernel where it applies (K = 64 / 128)
static int mv_gemm(const MaxVit* mv, int prec, int epi, const void* X, const void* W, const float* bias,
                   const float* gamma, const float* resid, void* out, int M, int N, int K, hipStream_t st) {
  return launch_gemm(prec, epi, X, W, bias, gamma, resid, out, M, N, K, st);
}

template <typename F> static int mv_timed(btsbot_ctx* h, int cat, hipStream_t st, F&& fn) {
  const bool rec = h->prof_on && h->prof_used < PROF_MAX_LAUNCHES;
  if (rec) HIP_TRY(hipEventRecord(h->prof_ev[2 * h->prof_used], st));
  const int s = fn();
  if (s != BTSBOT_OK) return s;
  if (rec) {
    HIP_TRY(hipEventRecord(h->prof_ev[2 * h->prof_used + 1], st));
    h->prof_cat[h->prof_used++] = cat;
  }
  return BTSBOT_OK;
}

int maxvit_chunk(btsbot_ctx* h, const float* img, int nb, hipStream_t st, float** feat_out) {
  MaxVit* mv = h->mv;
  const int prec = h->cfg.precision;
  const float* m = h->mirror;
  unsigned char* ex = h->extra;
  size_t total = 0;
  mv_layout(h, h->max_chunk, mv, &total);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ex + off); };
  float* x = reinterpret_cast<float*>(h->ws + mv->o_x);
  float* x2 = reinterpret_cast<float*>(h->ws + mv->o_x2);
  void* A = h->ws + mv->o_a;
  void* Bb = h->ws + mv->o_b;
  void* Cc = h->ws + mv->o_c;
  void* D = h->ws + mv->o_d;
  void* E = h->ws + mv->o_e;
  float* gate = reinterpret_cast<float*>(h->ws + mv->o_gate);
  float* feat = reinterpret_cast<float*>(h->ws + mv->o_feat);
  float* part = reinterpret_cast<float*>(h->ws + mv->o_part);
  float* sescr = reinterpret_cast<float*>(h->ws + mv->o_sescr);
  void* wg = h->ws + mv->o_wg;
  const float* zero = F(mv->p_zero);
  const float* one = F(mv->p_one);

  bool xn_ready = false, pool_ready = false, next_xn_ready = false;
  // ---- stem: resize + conv3x3 s2 (+BN, SiLU) + conv3x3 s1, both as im2col GEMMs
  const int M0 = nb * 12544;
  if (prec != BTSBOT_F32 && !mv->stem_im2col) {
    MTRY(mv_timed(h, CAT_MV_STEM, st, [&] {
      return launch_mv_stem1(prec, img, ex + mv->p_stem1, F(mv->stem_bn.p_shift), Cc, nb, st);
    }));
  } else {
    MTRY(mv_timed(h, CAT_MV_STEM, st, [&] { return launch_mv_resize_im2col(prec, img, Bb, nb, st); }));
    MTRY(mv_timed(h, CAT_MV_G_STEM, st, [&] {
      return launch_gemm(prec, EPI_SILU, Bb, ex + mv->p_stem1, F(mv->stem_bn.p_shift), nullptr, nullptr,
                         Cc, M0, 32, 32, st);
    }));
  }
  if (prec != BTSBOT_F32 && !mv->stem_im2col) {
    // (writes block 0's pre-norm + cast into the conv1 im2col buffer, which is free by now)
    const MvBlock& b0 = mv->blocks[0];
    // (without debug taps nobody needs the fp32 stem map itself: block 0's shortcut only wants its 2x2
    //  average pool, which the kernel then writes straight into the residual buffer x2)
    pool_ready = !h->debug && b0.stride == 2 && b0.sc_w < 0;
    MTRY(mv_timed(h, CAT_MV_G_STEM, st, [&] {
      return launch_mv_stem2(prec, Cc, ex + mv->p_stem2, pool_ready ? x2 : x, pool_ready ? 1 : 0, Bb,
                             F(b0.pre.p_scale), F(b0.pre.p_shift), nb, st);
    }));
    xn_ready = true;
  } else {
    MTRY(mv_timed(h, CAT_MV_STEM, st, [&] { return launch_mv_im2col3(prec, Cc, A, nb, 112, 32, st); }));
    MTRY(mv_timed(h, CAT_MV_G_STEM, st, [&] {
      return launch_gemm(prec, EPI_BIAS, A, ex + mv->p_stem2, zero, nullptr, nullptr, x, M0, 64, 288,
                         st);
    }));
  }
  if (h->debug && h->taps[0])
    HIP_TRY(hipMemcpyAsync(h->taps[0], x, (size_t)M0 * 64 * 4, hipMemcpyDeviceToDevice, st));

  int stage = 0, jblk = 0;
  // (bool xn_ready: block 0's BN-cast input already sits in Bb)
  for (size_t bi = 0; bi < mv->blocks.size(); ++bi) {
    const MvBlock& b = mv->blocks[bi];
    const int Min = nb * b.hin * b.hin, Mo = nb * b.hout * b.hout;
    // ---- MBConv
    float* resid = x;
    float* dst = x;
    if (b.stride == 2) {
      if (b.sc_w >= 0) {
        MTRY(mv_timed(h, CAT_MV_ELT, st, [&] {
          return launch_mv_avgpool2(prec, x, E, 1, nb, b.hin, b.cin, st);
        }));
        MTRY(mv_timed(h, CAT_MV_G_SC, st, [&] {
          return mv_gemm(mv, prec, EPI_BIAS, E, ex + b.p_sc, zero, nullptr, nullptr, x2, Mo, b.c,
                             b.cin, st);
        }));
      } else if (!(bi == 0 && pool_ready)) {
        MTRY(mv_timed(h, CAT_MV_ELT, st, [&] {
          return launch_mv_avgpool2(prec, x, x2, 0, nb, b.hin, b.cin, st);
        }));
      }
      resid = x2;
      dst = x2;
    }
    const void* c1_in = Cc;
    if (bi == 0 && xn_ready) {
      c1_in = Bb;
    } else if (next_xn_ready) {   // the previous block's last MLP already wrote BN_pre(x) into Cc
      next_xn_ready = false;
    } else {
      MTRY(mv_timed(h, CAT_MV_ELT, st, [&] {
        return launch_mv_bn_cast(prec, x, F(b.pre.p_scale), F(b.pre.p_shift), Cc, (long)Min, b.cin, st);
      }));
    }
    const float inv_hw = 1.0f / (float)(b.hout * b.hout);
    void* m2b = Bb;      // where the gated-conv input (depthwise output) lives
    const bool front = !mv->no_front && mv_mbconv_front_supported(prec, b.hin, b.cin, b.mid, b.stride);
    if (front) {
      // wide stages: conv1 + depthwise + pool partials in one kernel, the expanded map stays on-chip
      m2b = A;
      MTRY(mv_timed(h, CAT_MV_FRONT, st, [&] {
        return launch_mv_mbconv_front(prec, c1_in, ex + b.p_c1, F(b.p_c1b), F(b.p_dw), F(b.p_dwb), m2b, part,
                                      nb, b.hin, b.cin, b.mid, b.stride, st);
      }));
      MTRY(mv_timed(h, CAT_MV_SE, st, [&] {
        return launch_mv_se(BTSBOT_F32, part, m + b.se1_w, m + b.se1_b, F(b.p_se2t), m + b.se2_b, gate,
                            sescr, nb, mv_mbconv_front_tiles(b.hin, b.stride), b.mid, b.rd, inv_hw, st);
      }));
    } else {
    MTRY(mv_timed(h, CAT_MV_G_CONV1, st, [&] {
      return mv_gemm(mv, prec, EPI_SILU, c1_in, ex + b.p_c1, F(b.p_c1b), nullptr, nullptr, A, Min, b.mid,
                         b.cin, st);
    }));
    if (prec != BTSBOT_F32 && !mv->dw_plain) {
      // depthwise conv with the squeeze-excite pool fused (partial sums per workgroup, no second pass)
      MTRY(mv_timed(h, CAT_MV_DW, st, [&] {
        return launch_mv_dw3s(prec, A, F(b.p_dw), F(b.p_dwb), Bb, part, nb, b.hin, b.mid, b.stride, st);
      }));
      MTRY(mv_timed(h, CAT_MV_SE, st, [&] {
        return launch_mv_se(BTSBOT_F32, part, m + b.se1_w, m + b.se1_b, F(b.p_se2t), m + b.se2_b, gate,
                            sescr, nb, mv_dw3s_groups(b.hin, b.mid, b.stride), b.mid, b.rd, inv_hw, st);
      }));
    } else {
      MTRY(mv_timed(h, CAT_MV_DW, st, [&] {
        return launch_mv_dw3(prec, A, F(b.p_dw), F(b.p_dwb), Bb, nb, b.hin, b.mid, b.stride, st);
      }));
      MTRY(mv_timed(h, CAT_MV_SE, st, [&] {
        return launch_mv_se(prec, Bb, m + b.se1_w, m + b.se1_b, F(b.p_se2t), m + b.se2_b, gate, sescr, nb,
                            b.hout * b.hout, b.mid, b.rd, inv_hw, st);
      }));
    }
    }
    const int hw2 = b.hout * b.hout;
    // wide stages (C = 64 / 128): the LayerNorm that follows a residual GEMM is computed in that GEMM's
    // epilogue (the staged output tile holds whole rows)
    // (a block whose partition halves run as mv_part_kernel normalises its rows there: nobody reads a fused LayerNorm copy)
    const bool part_blk = b.attn[0].part && !mv->no_part;
    const bool ln_fuse = prec != BTSBOT_F32 && !mv->no_ln_fuse && (b.c == 64 || b.c == 128) && !part_blk;
    bool ln1_done = false, ln1_grid_done = false;
    if (prec != BTSBOT_F32 && !mv->gated_gemm && (size_t)b.c * b.mid * 4 <= (size_t)hw2 * b.mid) {
      // wide stages: per-alert filters W3 diag(g_b) (a fraction of the map's size) + batched LDS-DMA GEMM
      MTRY(mv_timed(h, CAT_MV_SE, st, [&] {
        return launch_mv_scale_w(prec, m + b.c3_w, gate, wg, nb, b.c, b.mid, st);
      }));
      MTRY(mv_timed(h, CAT_MV_G_CONV3, st, [&] {
        return launch_gemm2_batched_resid(prec, m2b, wg, zero, one, resid, dst, nb, hw2, b.c, b.mid, st,
                                          ln_fuse ? m + b.attn[0].n1w : nullptr,
                                          ln_fuse ? m + b.attn[0].n1b : nullptr, ln_fuse ? Cc : nullptr);
      }));
      ln1_done = ln_fuse;
    } else if (prec != BTSBOT_F32 && !mv->gated_gemm) {
      // narrow stages: the map is small -- gate it in place, then the plain LDS-DMA GEMM
      MTRY(mv_timed(h, CAT_MV_SE, st, [&] { return launch_mv_gate(prec, m2b, gate, nb, hw2, b.mid, st); }));
      MTRY(mv_timed(h, CAT_MV_G_CONV3, st, [&] {
        return launch_gemm(prec, EPI_RESID, m2b, ex + b.p_c3, zero, one, resid, dst, Mo, b.c, b.mid, st);
      }));
    } else {
      MTRY(mv_timed(h, CAT_MV_G_CONV3, st, [&] {
        return launch_gemm_gated(prec, m2b, gate, hw2, ex + b.p_c3, resid, dst, Mo, b.c, b.mid, st);
      }));
    }
    if (b.stride == 2) {   // the block's output lives in x2: swap the roles of the two maps
      float* t = x;
      x = x2;
      x2 = t;
    }
    // ---- window attention, then grid attention
    for (int g = 0; g < 2; ++g) {
      const AttnPk& a = b.attn[g];
      const int c = b.c;
      // C = 64 / 128 / 256 (every block but stage 3's): norm1, qkv, attention, proj, residual, norm2, fc1, GELU, fc2,
      // residual as ONE launch (maxvit_part.hip)
      if (part_blk) {
        MvPartW pw;
        memset(&pw, 0, sizeof(pw));
        pw.ln1w = m + a.n1w;
        pw.ln1b = m + a.n1b;
        pw.wqkvp = ex + a.p_qkvp;
        pw.bqkv = m + a.qkv_b;
        pw.wprojp = ex + a.p_projp;
        pw.bproj = m + a.proj_b;
        pw.biasl = F(a.p_biasl);
        pw.stamps = h->stamps ? h->stamps + 20000 + (c == 256 ? 0 : 32) : nullptr;
        pw.ln2w = m + a.n2w;
        pw.ln2b = m + a.n2b;
        pw.w1p = ex + a.p_w1p;
        pw.b1 = m + a.fc1_b;
        pw.w2p = ex + a.p_w2p;
        pw.b2 = m + a.fc2_b;
        if (g == 1 && bi + 1 < mv->blocks.size()) {   // ... and the next block's pre-norm copy of the finished rows
          pw.post_s = F(mv->blocks[bi + 1].pre.p_scale);
          pw.post_b = F(mv->blocks[bi + 1].pre.p_shift);
          pw.post_out = Cc;
        }
        MTRY(mv_timed(h, CAT_MV_PART, st, [&] { return launch_mv_part(prec, x, pw, nb, b.hout, c, g, st); }));
        if (pw.post_out != nullptr) next_xn_ready = true;
        continue;
      } else {
      if (!(g == 0 && ln1_done) && !(g == 1 && ln1_grid_done)) {
        MTRY(mv_timed(h, CAT_MV_LN, st, [&] {
          return launch_mv_ln(prec, x, m + a.n1w, m + a.n1b, Cc, (long)Mo, c, st);
        }));
      }
      if (!mv->no_attn_block && mv_attn_block_supported(prec, c)) {
        // C = 64: qkv, attention, proj, residual and LN2 in one kernel (qkv never reaches HBM)
        MTRY(mv_timed(h, CAT_MV_ABLK, st, [&] {
          return launch_mv_attn_block(prec, Cc, x, Cc, ex + a.p_qkv, m + a.qkv_b, ex + a.p_proj, m + a.proj_b,
                                      F(a.p_bias64), m + a.n2w, m + a.n2b, nb, b.hout, c, g, st);
        }));
      } else {
      MTRY(mv_timed(h, CAT_MV_G_QKV, st, [&] {
        return mv_gemm(mv, prec, EPI_BIAS_T, Cc, ex + a.p_qkv, m + a.qkv_b, nullptr, nullptr, D, Mo,
                           3 * c, c, st);
      }));
      MTRY(mv_timed(h, CAT_MV_ATTN, st, [&] {
        if (prec != BTSBOT_F32 && !mv->attn_valu)
          return launch_mv_attn_mfma(prec, D, F(a.p_bias64), E, nb, b.hout, c, g, st);
        return launch_mv_attn(prec, D, F(a.p_bias), E, nb, b.hout, c, g, st);
      }));
      if (ln_fuse && gemm2_supported(prec, Mo, c, c)) {   // proj + residual + LN2 in one launch
        MTRY(mv_timed(h, CAT_MV_G_PROJ, st, [&] {
          return launch_gemm2_batched_resid(prec, E, ex + a.p_proj, m + a.proj_b, one, x, x, 1, Mo, c, c, st,
                                            m + a.n2w, m + a.n2b, Cc);
        }));
      } else {
        MTRY(mv_timed(h, CAT_MV_G_PROJ, st, [&] {
          return mv_gemm(mv, prec, EPI_RESID, E, ex + a.p_proj, m + a.proj_b, one, x, x, Mo, c, c, st);
        }));
        if (!(a.smlp && !mv->no_smlp))   // (the streamed MLP below normalises its rows itself)
          MTRY(mv_timed(h, CAT_MV_LN, st, [&] {
            return launch_mv_ln(prec, x, m + a.n2w, m + a.n2b, Cc, (long)Mo, c, st);
          }));
      }
      }
      }   // (part_blk)
      if (a.fused && !mv->mlp_unfused) {   // C = 64 / 128: fc1 -> GELU -> fc2 -> +x with the hidden on-chip
        // ... and the next consumer's normalised copy of x from the same registers: the grid attention's
        // LN1 after the window attention's MLP, the next block's pre-norm BatchNorm after the grid one's.
        // (the kernel reads its input Cc completely before any row is written? no: rows are independent
        //  and a lane reads its row's xn fragments before it writes that row -- in place is safe)
        const float *pw = nullptr, *pb = nullptr;
        int post = 0;
        if (ln_fuse && g == 0) {
          pw = m + b.attn[1].n1w;
          pb = m + b.attn[1].n1b;
          post = 1;
        } else if (ln_fuse && g == 1 && bi + 1 < mv->blocks.size()) {
          pw = F(mv->blocks[bi + 1].pre.p_scale);
          pb = F(mv->blocks[bi + 1].pre.p_shift);
          post = 2;
        }
        MTRY(mv_timed(h, CAT_MV_FUSED, st, [&] {
          return launch_fused_mlp(prec, c, Cc, ex + a.p_fused, m + a.fc1_b, m + a.fc2_b, one, x, Mo, st,
                                  post ? Cc : nullptr, pw, pb, post);
        }));
        if (post == 1) ln1_grid_done = true;
        if (post == 2) next_xn_ready = true;
        continue;
      }
      if (a.smlp && !mv->no_smlp) {
        // C = 256: norm2 -> fc1 -> GELU -> fc2 -> + x as ONE launch of stage2p_kernel's row-tile form: 64 rows resident per
        // workgroup, both filters streamed past them as packed MFMA fragments, the 1024-wide hidden rows never leave the CU
        // (unfused: a LayerNorm launch and two GEMMs that write and re-read 4 KB per row)
        Stage2pBlk sb;
        memset(&sb, 0, sizeof(sb));
        sb.ln_w = m + a.n2w;
        sb.ln_b = m + a.n2b;
        sb.b1 = m + a.fc1_b;
        sb.b2 = m + a.fc2_b;
        sb.gamma = one;
        sb.w1p = ex + a.p_w1p;
        sb.w2p = ex + a.p_w2p;
        MTRY(mv_timed(h, CAT_MV_SMLP, st, [&] { return launch_stage2p_rows(prec, x, (long)Mo, sb, st); }));
        continue;
      }
      MTRY(mv_timed(h, CAT_MV_G_FC1, st, [&] {
        return mv_gemm(mv, prec, EPI_GELU, Cc, ex + a.p_fc1, m + a.fc1_b, nullptr, nullptr, Bb, Mo,
                           4 * c, c, st);
      }));
      MTRY(mv_timed(h, CAT_MV_G_FC2, st, [&] {
        return launch_gemm(prec, EPI_RESID, Bb, ex + a.p_fc2, m + a.fc2_b, one, x, x, Mo, c, 4 * c,
                           st);
      }));
    }
    // debug taps: output of the last block of every stage
    ++jblk;
    if (jblk == MV_DEPTHS[stage]) {
      if (h->debug && h->taps[stage + 1])
        HIP_TRY(hipMemcpyAsync(h->taps[stage + 1], x, (size_t)Mo * b.c * 4, hipMemcpyDeviceToDevice,
                               st));
      ++stage;
      jblk = 0;
    }
  }
  MTRY(mv_timed(h, CAT_MV_LN, st, [&] {
    return launch_mv_final(x, m + mv->norm_w, m + mv->norm_b, feat, nb, 49, 512, st);
  }));
  *feat_out = feat;
  return BTSBOT_OK;
}
