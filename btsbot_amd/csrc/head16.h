// Argument block of the matrix-pipe classifier head (head16.hip).
#pragma once

struct H16Layer {
  const void* w;        // [hi | lo][row tile of 32][k-step][lane 64][8] A fragments (launch_pack_h16)
  const float* bias;    // [N]
  int K, N, act;        // act: ACT_NONE / ACT_GELU / ACT_RELU (common.h)
};

struct H16Step {        // one layer of the chain; buffers: 0 = z (concat row), 1 = t0, 2 = t1
  H16Layer L;
  int in_buf, out_buf, col0, red_buf;   // red_buf: scratch for K-split partial sums (-1: none)
};

struct Head16Args {
  const float* feat;    // [B][feat_dim] fp32 (final 1x1 map), or nullptr
  int feat_dim;
  const float* hn_w;    // head LayerNorm (nullptr -> none)
  const float* hn_b;
  const float* meta;    // [B][n_meta] or nullptr
  int n_meta;
  const float* bn_scale;   // BatchNorm1d folded to scale / shift
  const float* bn_shift;
  H16Layer m1, m2;      // metadata branch (m2.act = ACT_NONE where the wiring has no trailing activation)
  int n_layers;         // fusion MLP: comb[0].K = feat_dim + m2.N (or whichever part exists), comb[n_layers-1].N = 1
  H16Layer comb[3];
  float* logits;
  float* scores;        // may be nullptr
  int B;
  unsigned long long* stamps;     // optional: workgroup 0 / thread 0 stores the shader clock per phase
  int pitch_z, pitch_t, zwidth;   // LDS row pitches / concat width (filled in by launch_head16)
  int n_steps;                    // the layer chain in execution order (filled in by launch_head16)
  H16Step steps[5];
};

size_t head16_packed_bytes(int N, int K);
bool head16_supported(int prec, int feat_dim, int n_meta, int f1, int f2, int n_layers, const int* dims);
// fp32 [N][K] (PyTorch Linear layout) -> the fragment image above
int launch_pack_h16(int prec, const float* w, void* dst, int N, int K, hipStream_t st);
int launch_head16(int prec, const Head16Args& a, hipStream_t st);
