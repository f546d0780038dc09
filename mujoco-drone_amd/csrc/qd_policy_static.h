// qd_policy_static.h -- compile-time specialisations of the policy kernel for the reference's actual networks.
//
// (Round 4: the specialised kernels multiply on v_mfma_f32_16x16x32_f16 with every operand split into two float16 halves -- three
// products per float32 product, 5.3x the rate of the float32 MFMA, as close to the float64 oracle as float32 is; s_dense below.
// The interpreter keeps v_mfma_f32_16x16x4_f32.)
// k_policy (qd_policy.h) interprets any layer program, but with ONE wave per SIMD every interpreted step costs ~500
// instructions at ~5 cycles each: 1.1 us of bookkeeping around 0.1-0.45 us of MFMAs (measured with the stamped
// build, tests/diag_policy_stamps.py).  The networks the reference actually trains are fixed (train_PPO.py:39-45:
// num_states 16, num_params 6, num_actions 4, param_embed_dim 8), so their programs are also compiled into the
// library as constexpr tables and the same algorithm is instantiated with every size, offset and trip count known at
// compile time: addresses fold into immediates, layers unroll, and the weights of the next layer's first k-blocks are
// requested before the current layer's barrier.  qd_policy_create compares the program it is given with these tables
// and uses the specialisation only on an exact match; both kernels read the same packed blob.
#pragma once

#include "qd_policy.h"

namespace qd {

struct SOp {
  int kind, in_buf, in_off, in_dim, out_buf, out_off, out_dim, act, flags;
};
constexpr int SPROG_MAX = 32;
struct SRing {
  int rows, width, period;
};
struct SProg {
  int n_ops;
  SOp op[SPROG_MAX];
  int n_bufs, width[POL_MAX_BUFS];
  int obs_dim, act_dim;
  int logits_buf, logits_off, n_logits, value_buf, value_off;
  int n_rings;
  SRing ring[POL_MAX_RINGS];
  int aux_buf, aux_off, aux_dim;  // auxiliary slice (the RMA networks' embedding z); aux_dim 0: none
};

// ---- layout, the same arithmetic as pol_compile (qd_policy_host.inc) ----
constexpr int sp_w16(const SProg& p, int b) { return (p.width[b] + 15) / 16 * 16; }
constexpr int sp_ld(const SProg& p, int b) { return sp_w16(p, b) + 4; }
constexpr int sp_base(const SProg& p, int b) {
  int off = 0;
  for (int i = 0; i < b; i++) off += POL_TILE * sp_ld(p, i);
  return off;
}
constexpr int sp_k16(const SProg& p, int k) { return (p.op[k].in_dim + 15) / 16; }
constexpr int sp_k32(const SProg& p, int k) { return (p.op[k].in_dim + 31) / 32; }
constexpr bool sp_fused_affine(const SProg& p, int k);
// ---- the float16 mirror of the activations (see s_dense) ----
// Dense layers multiply on v_mfma_f32_16x16x32_f16 with every operand split into two halves, x = hi + lo / 2048.  A layer whose
// whole input slice was written by dense epilogues reads it from a mirror of the activation buffers that already holds the split
// (two planes of halves, written next to the float32 value); any other layer splits the float32 values as it reads them.
constexpr bool sp_overlap(int a0, int a1, int b0, int b1) { return a0 < b1 && b0 < a1; }
constexpr bool sp_mirror_fed(const SProg& p, int k) {
  const SOp& o = p.op[k];
  if (o.kind != POL_DENSE || o.in_off % 8) return false;
  const int lo = o.in_off, hi = o.in_off + o.in_dim, hi_pad = o.in_off + sp_k32(p, k) * 32;
  for (int c = lo; c < hi; c++) {   // every column comes out of a dense epilogue ...
    bool cov = false;
    for (int j = 0; j < p.n_ops; j++)
      if (p.op[j].kind == POL_DENSE && p.op[j].out_buf == o.in_buf && c >= p.op[j].out_off && c < p.op[j].out_off + p.op[j].out_dim) cov = true;
    if (!cov) return false;
  }
  for (int j = 0; j < p.n_ops; j++) {   // ... and nothing else writes into the (padded) slice
    const SOp& w = p.op[j];
    if (w.kind == POL_DENSE || w.kind == POL_RING_PUSH) continue;
    if (w.kind == POL_AFFINE && j > 0 && sp_fused_affine(p, j - 1)) continue;   // applied inside the dense epilogue
    const int n = (w.kind == POL_COPY_OBS || w.kind == POL_COPY_PREV) ? w.in_dim : w.kind == POL_LSTM_CELL ? 2 * w.out_dim : w.out_dim;
    if (w.out_buf == o.in_buf && sp_overlap(w.out_off, w.out_off + n, lo, hi_pad)) return false;
  }
  return true;
}
constexpr bool sp_out_mirrored(const SProg& p, int k) {   // does a mirror-fed layer read what dense op k writes?
  const SOp& o = p.op[k];
  for (int j = 0; j < p.n_ops; j++)
    if (sp_mirror_fed(p, j) && p.op[j].in_buf == o.out_buf &&
        sp_overlap(o.out_off, o.out_off + o.out_dim, p.op[j].in_off, p.op[j].in_off + p.op[j].in_dim))
      return true;
  return false;
}
constexpr int sp_hw(const SProg& p, int b) {   // halves per row of buffer b's mirror (0: no mirror-fed layer reads it)
  int w = 0;
  for (int k = 0; k < p.n_ops; k++)
    if (sp_mirror_fed(p, k) && p.op[k].in_buf == b && p.op[k].in_off + sp_k32(p, k) * 32 > w) w = p.op[k].in_off + sp_k32(p, k) * 32;
  return w;
}
constexpr int sp_hld(const SProg& p, int b) { return sp_hw(p, b) ? sp_hw(p, b) + 8 : 0; }   // +16 bytes: rows start 4 banks apart
constexpr int sp_hbase(const SProg& p, int b) {   // halves, within one plane
  int off = 0;
  for (int i = 0; i < b; i++) off += POL_TILE * sp_hld(p, i);
  return off;
}
constexpr int sp_hplane(const SProg& p) { return sp_hbase(p, p.n_bufs); }
constexpr int sp_mirror_floats(const SProg& p) { return sp_hplane(p); }   // two planes of halves
constexpr int sp_act_floats(const SProg& p) { return sp_base(p, p.n_bufs) + sp_mirror_floats(p) + POL_SCRATCH; }
constexpr int sp_ntiles(const SProg& p, int k) { return (p.op[k].out_dim + 15) / 16; }
constexpr long long sp_ws_at(const SProg& p, int k) {  // floats, within the split-weight region
  long long off = 0;
  for (int i = 0; i < k; i++)
    if (p.op[i].kind == POL_DENSE) off += (long long)sp_ntiles(p, i) * sp_k32(p, i) * 512;
  return off;
}
constexpr long long sp_w_at(const SProg& p, int k) {  // floats, within the weight region
  long long off = 0;
  for (int i = 0; i < k; i++)
    if (p.op[i].kind == POL_DENSE) off += (long long)sp_ntiles(p, i) * sp_k16(p, i) * 256;
  return off;
}
constexpr int sp_ring_fill(const SProg& p, int r) {  // the rings' episode-start values open the small region
  int off = 0;
  for (int i = 0; i < r; i++) off += (p.ring[i].width + 3) / 4 * 4;
  return off;
}
constexpr int sp_ring_off(const SProg& p, int r) {  // floats, within one env's history block
  int off = 0;
  for (int i = 0; i < r; i++) off += p.ring[i].period * p.ring[i].rows * p.ring[i].width;
  return off;
}
constexpr int sp_state_floats(const SProg& p) { return sp_ring_off(p, p.n_rings); }
constexpr int sp_s_at(const SProg& p, int k) {  // floats, within the small region
  int off = sp_ring_fill(p, p.n_rings);
  for (int i = 0; i < k; i++) {
    if (p.op[i].kind == POL_DENSE) off += sp_ntiles(p, i) * 16;
    else if (p.op[i].kind == POL_AFFINE) off += (2 * p.op[i].out_dim + 3) / 4 * 4;
  }
  return off;
}
constexpr int sp_next_dense(const SProg& p, int from) {
  for (int i = from; i < p.n_ops; i++)
    if (p.op[i].kind == POL_DENSE) return i;
  return p.n_ops;
}
constexpr int sp_min(int a, int b) { return a < b ? a : b; }
// is dense op k directly followed by an affine op on exactly its output slice (tanh -> BatchNorm)?  Then the affine map is
// applied in the dense layer's epilogue and the op itself is skipped.
constexpr bool sp_fused_affine(const SProg& p, int k) {
  return k + 1 < p.n_ops && p.op[k].kind == POL_DENSE && p.op[k + 1].kind == POL_AFFINE && p.op[k + 1].out_buf == p.op[k].out_buf &&
         p.op[k + 1].out_off == p.op[k].out_off && p.op[k + 1].out_dim == p.op[k].out_dim && p.op[k + 1].flags == p.op[k].flags;
}
constexpr int sp_small_floats(const SProg& p) {
  int off = sp_ring_fill(p, p.n_rings);
  for (int i = 0; i < p.n_ops; i++) {
    if (p.op[i].kind == POL_DENSE) off += (p.op[i].out_dim + 15) / 16 * 16;
    else if (p.op[i].kind == POL_AFFINE) off += (2 * p.op[i].out_dim + 3) / 4 * 4;
  }
  return (off + 3) / 4 * 4;
}
// leading input gathers (COPY / RING_LOAD ops before anything else): done in the prologue, under one global-load latency
constexpr int sp_leading_copies(const SProg& p) {
  int n = 0;
  while (n < p.n_ops && (p.op[n].kind == POL_COPY_OBS || p.op[n].kind == POL_COPY_PREV || p.op[n].kind == POL_RING_LOAD) &&
         !(p.op[n].flags & POL_FLAG_VALUE_ONLY))
    n++;
  return n;
}

// ---- the reference's networks with the training scripts' sizes (mirrors mujoco_drone_amd/policy.py) ----
constexpr int SX = 0, SP = 1, SA = 2, SB = 3, SV = POL_FLAG_VALUE_ONLY, TANH = POL_ACT_TANH;
struct ArchRmaFull {  // models/PPO/RMA/RMA_model.py:17-110, train_adaptation=False
  static constexpr SProg prog = {13,
      {{POL_COPY_OBS, 0, 0, 16, SX, 0, 16, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 16, 4, 0, 0}, {POL_COPY_OBS, 0, 16, 6, SP, 0, 6, 0, 0},
       {POL_DENSE, SP, 0, 6, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SX, 20, 8, 0, 0},
       {POL_DENSE, SX, 0, 28, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_AFFINE, SB, 0, 128, SB, 0, 128, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, 0}, {POL_DENSE, SA, 0, 128, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      4, {32, 16, 256, 128}, 22, 4, SP, 0, 8, SX, 0, 0, {}, SX, 20, 8};
};
struct ArchRmaModel {  // models/PPO/RMA/RMA_model.py:199-292
  static constexpr SProg prog = {16,
      {{POL_COPY_OBS, 0, 0, 16, SX, 0, 16, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 16, 4, 0, 0}, {POL_COPY_OBS, 0, 16, 6, SP, 0, 6, 0, 0},
       {POL_DENSE, SP, 0, 6, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SX, 20, 8, TANH, 0},
       {POL_DENSE, SX, 0, 28, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, 0},
       {POL_DENSE, SA, 0, 128, SB, 0, 96, TANH, 0}, {POL_AFFINE, SB, 0, 96, SB, 0, 96, 0, 0},
       {POL_DENSE, SB, 0, 96, SA, 0, 64, TANH, 0}, {POL_DENSE, SA, 0, 64, SA, 64, 64, TANH, 0}, {POL_DENSE, SA, 64, 64, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 96, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      4, {32, 16, 256, 128}, 22, 4, SP, 0, 8, SX, 0, 0, {}, SX, 20, 8};
};
struct ArchSimpleMlp {  // models/PPO/SimpleMLP/SimpleMLP.py:18-98
  static constexpr SProg prog = {22,
      {{POL_COPY_OBS, 0, 0, 22, SX, 0, 22, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 22, 4, 0, 0}, {POL_AFFINE, SX, 0, 26, SX, 0, 26, 0, 0},
       {POL_DENSE, SX, 0, 26, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, 0},
       {POL_DENSE, SA, 0, 128, SB, 0, 96, TANH, 0}, {POL_AFFINE, SB, 0, 96, SB, 0, 96, 0, 0},
       {POL_DENSE, SB, 0, 96, SA, 0, 64, TANH, 0}, {POL_DENSE, SA, 0, 64, SA, 128, 64, TANH, 0}, {POL_DENSE, SA, 128, 64, SX, 0, 8, 0, 0},
       {POL_COPY_OBS, 0, 0, 22, SP, 0, 22, 0, SV}, {POL_COPY_PREV, 0, 0, 4, SP, 22, 4, 0, SV}, {POL_AFFINE, SP, 0, 26, SP, 0, 26, 0, SV},
       {POL_DENSE, SP, 0, 26, SA, 0, 256, TANH, SV}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, SV}, {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV},
       {POL_DENSE, SA, 0, 128, SB, 0, 96, TANH, SV}, {POL_AFFINE, SB, 0, 96, SB, 0, 96, 0, SV},
       {POL_DENSE, SB, 0, 96, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SP, 0, 1, 0, SV}},
      4, {32, 32, 256, 128}, 22, 4, SX, 0, 8, SP, 0};
};

constexpr int SH = 4, SC1 = 5, SC2 = 6;
struct ArchRmaFullAdapt {  // RMA_full with train_adaptation=True, adapt_seq_len 32 (train_RMA.py:39-45); see policy.py:_rma_full_adapt
  static constexpr SProg prog = {23,
      {{POL_COPY_OBS, 0, 0, 16, SX, 0, 16, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 16, 4, 0, 0},
       {POL_RING_LOAD, 0, 0, 160, SH, 0, 160, 0, 0}, {POL_RING_LOAD, 1, 0, 128, SC1, 0, 128, 0, 0}, {POL_RING_LOAD, 2, 0, 144, SC2, 0, 144, 0, 0},
       {POL_DENSE, SX, 0, 20, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SB, 0, 32, TANH, 0}, {POL_DENSE, SB, 0, 32, SP, 0, 32, TANH, 0},
       {POL_DENSE, SH, 0, 160, SC1, 128, 32, 0, 0}, {POL_RING_PUSH, SP, 0, 32, 0, 0, 32, 0, 0},
       {POL_DENSE, SC1, 0, 160, SC2, 144, 16, 0, 0}, {POL_RING_PUSH, SC1, 128, 32, 1, 0, 32, 0, 0},
       {POL_DENSE, SC2, 0, 160, SA, 0, 64, TANH, 0}, {POL_RING_PUSH, SC2, 144, 16, 2, 0, 16, 0, 0},
       {POL_DENSE, SA, 0, 64, SX, 20, 8, 0, 0},
       {POL_DENSE, SX, 0, 28, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_AFFINE, SB, 0, 128, SB, 0, 128, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, 0}, {POL_DENSE, SA, 0, 128, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      7, {32, 32, 256, 128, 160, 160, 160}, 22, 4, SP, 0, 8, SX, 0,
      3, {{5, 32, 1}, {4, 32, 2}, {9, 16, 2}}};
};

struct ArchCnnEst {  // CNNestimator, use_estimate=False (StateEstimatorLSTM.py:200-283; train_LSTM.py:51-60): 23-value observation
  static constexpr SProg prog = {9,
      {{POL_COPY_OBS, 0, 0, 19, SX, 0, 19, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 19, 4, 0, 0}, {POL_COPY_OBS, 0, 19, 4, SX, 23, 4, 0, 0},
       {POL_DENSE, SX, 0, 27, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      4, {32, 16, 256, 128}, 23, 4, SP, 0, 8, SX, 0, 0, {}};
};
struct ArchCnnEstHist {  // CNNestimator, use_estimate=True: TimeCNN over the 32-step history, incremental (policy.py:_cnn_estimator_estimate)
  static constexpr SProg prog = {20,
      {{POL_COPY_OBS, 0, 0, 19, SX, 0, 19, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 19, 4, 0, 0},
       {POL_RING_LOAD, 0, 0, 160, SH, 0, 160, 0, 0}, {POL_RING_LOAD, 1, 0, 128, SC1, 0, 128, 0, 0}, {POL_RING_LOAD, 2, 0, 144, SC2, 0, 144, 0, 0},
       {POL_DENSE, SX, 0, 23, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SP, 0, 32, TANH, 0},
       {POL_DENSE, SH, 0, 160, SC1, 128, 32, 0, 0}, {POL_RING_PUSH, SP, 0, 32, 0, 0, 32, 0, 0},
       {POL_DENSE, SC1, 0, 160, SC2, 144, 16, 0, 0}, {POL_RING_PUSH, SC1, 128, 32, 1, 0, 32, 0, 0},
       {POL_DENSE, SC2, 0, 160, SA, 0, 32, TANH, 0}, {POL_RING_PUSH, SC2, 144, 16, 2, 0, 16, 0, 0},
       {POL_DENSE, SA, 0, 32, SX, 23, 4, 0, 0},
       {POL_DENSE, SX, 0, 27, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      7, {32, 32, 256, 128, 160, 160, 160}, 23, 4, SP, 0, 8, SX, 0,
      3, {{5, 32, 1}, {4, 32, 2}, {9, 16, 2}}};
};
struct ArchCustomMlp {  // models/PPO/MLP/CustomMLP.py:17-98
  static constexpr SProg prog = {14,
      {{POL_COPY_OBS, 0, 0, 22, SX, 0, 22, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 22, 4, 0, 0}, {POL_AFFINE, SX, 0, 26, SX, 0, 26, 0, 0},
       {POL_DENSE, SX, 0, 26, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, 0},
       {POL_DENSE, SA, 0, 128, SB, 0, 96, TANH, 0}, {POL_AFFINE, SB, 0, 96, SB, 0, 96, 0, 0},
       {POL_DENSE, SB, 0, 96, SA, 0, 64, TANH, 0}, {POL_DENSE, SA, 0, 64, SA, 64, 64, TANH, 0}, {POL_DENSE, SA, 64, 64, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 96, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      4, {32, 16, 256, 128}, 22, 4, SP, 0, 8, SX, 0, 0, {}};
};
constexpr int SIN = 4, SS = 5, SG = 6;
struct ArchLstmEst {  // LSTMestimator with the nn.LSTM estimate (StateEstimatorLSTM.py:15-197); see policy.py:_lstm_estimator
  static constexpr SProg prog = {22,
      {{POL_COPY_OBS, 0, 0, 15, SX, 0, 15, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SX, 15, 4, 0, 0},
       {POL_RING_LOAD, 0, 0, 15, SIN, 0, 15, 0, 0}, {POL_RING_LOAD, 1, 0, 32, SS, 32, 32, 0, 0}, {POL_RING_LOAD, 2, 0, 32, SS, 64, 32, 0, 0},
       {POL_COPY_OBS, 0, 0, 15, SIN, 15, 15, 0, 0}, {POL_COPY_PREV, 0, 0, 4, SIN, 30, 4, 0, 0},
       {POL_DENSE, SIN, 0, 34, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SS, 0, 32, TANH, 0}, {POL_DENSE, SS, 0, 64, SG, 0, 128, 0, 0},
       {POL_RING_PUSH, SIN, 15, 15, 0, 0, 15, 0, 0}, {POL_LSTM_CELL, SG, 0, 128, SS, 32, 32, 0, 0},
       {POL_RING_PUSH, SS, 32, 32, 1, 0, 32, 0, 0}, {POL_RING_PUSH, SS, 64, 32, 2, 0, 32, 0, 0},
       {POL_DENSE, SS, 0, 64, SA, 0, 32, TANH, 0}, {POL_DENSE, SA, 0, 32, SX, 19, 4, 0, 0},
       {POL_DENSE, SX, 0, 23, SA, 0, 256, TANH, 0}, {POL_DENSE, SA, 0, 256, SB, 0, 128, TANH, 0}, {POL_DENSE, SB, 0, 128, SP, 0, 8, 0, 0},
       {POL_DENSE, SB, 0, 128, SA, 0, 128, TANH, SV}, {POL_DENSE, SA, 0, 128, SA, 128, 128, TANH, SV}, {POL_DENSE, SA, 128, 128, SX, 0, 1, 0, SV}},
      7, {32, 16, 256, 128, 48, 96, 128}, 19, 4, SP, 0, 8, SX, 0,
      3, {{1, 15, 1}, {1, 32, 1}, {1, 32, 1}}};
};
// (the windowed networks and SimpleMLPmodel / CustomMLP / CNNestimator / LSTMestimator declare no auxiliary slice)
// the remaining variants the training scripts import; buffers by index (0 X, 1 P, 2 A, 3 B, 4 S, 5 G), see policy.py
struct ArchRmaSmaller {  // RMA_model_smaller (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {13,
      {{POL_COPY_OBS, 0, 0, 16, 0, 0, 16, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 0, 16, 4, 0, 0}, {POL_COPY_OBS, 0, 16, 6, 1, 0, 6, 0, 0},
       {POL_DENSE, 1, 0, 6, 2, 0, 32, TANH, 0}, {POL_DENSE, 2, 0, 32, 0, 20, 8, TANH, 0}, {POL_DENSE, 0, 0, 28, 2, 0, 256, TANH, 0},
       {POL_DENSE, 2, 0, 256, 3, 0, 128, TANH, 0}, {POL_AFFINE, 3, 0, 128, 3, 0, 128, 0, 0}, {POL_DENSE, 3, 0, 128, 2, 0, 128, TANH, 0},
       {POL_DENSE, 2, 0, 128, 1, 0, 8, 0, 0}, {POL_DENSE, 3, 0, 128, 2, 0, 128, TANH, SV}, {POL_DENSE, 2, 0, 128, 2, 128, 128, TANH, SV},
       {POL_DENSE, 2, 128, 128, 0, 0, 1, 0, SV}},
      4, {32, 16, 256, 128}, 22, 4, 1, 0, 8, 0, 0,
      0, {}, 0, 20, 8};
};
struct ArchRmaSmaller2 {  // RMA_model_smaller2 (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {14,
      {{POL_COPY_OBS, 0, 0, 16, 0, 0, 16, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 0, 16, 4, 0, 0}, {POL_COPY_OBS, 0, 16, 6, 1, 0, 6, 0, 0},
       {POL_DENSE, 1, 0, 6, 2, 0, 32, TANH, 0}, {POL_DENSE, 2, 0, 32, 0, 20, 8, TANH, 0}, {POL_DENSE, 0, 0, 28, 2, 0, 512, TANH, 0},
       {POL_DENSE, 2, 0, 512, 3, 0, 256, TANH, 0}, {POL_AFFINE, 3, 0, 256, 3, 0, 256, 0, 0}, {POL_DENSE, 3, 0, 256, 1, 0, 8, 0, 0},
       {POL_DENSE, 3, 0, 256, 3, 256, 256, TANH, SV}, {POL_DENSE, 3, 0, 512, 2, 0, 128, TANH, SV}, {POL_DENSE, 2, 0, 128, 2, 256, 128, TANH, SV},
       {POL_DENSE, 2, 256, 128, 2, 128, 128, TANH, SV}, {POL_DENSE, 2, 0, 256, 0, 0, 1, 0, SV}},
      4, {32, 16, 512, 512}, 22, 4, 1, 0, 8, 0, 0,
      0, {}, 0, 20, 8};
};
struct ArchCustomLstm {  // CustomLSTM (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {13,
      {{POL_COPY_OBS, 0, 0, 22, 0, 0, 22, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 0, 22, 4, 0, 0}, {POL_RING_LOAD, 0, 0, 64, 4, 64, 64, 0, 0},
       {POL_RING_LOAD, 1, 0, 64, 4, 128, 64, 0, 0}, {POL_DENSE, 0, 0, 26, 4, 0, 64, TANH, 0}, {POL_AFFINE, 4, 0, 64, 4, 0, 64, 0, 0},
       {POL_DENSE, 4, 0, 128, 5, 0, 256, 0, 0}, {POL_LSTM_CELL, 5, 0, 256, 4, 64, 64, 0, 0}, {POL_RING_PUSH, 4, 64, 64, 0, 0, 64, 0, 0},
       {POL_RING_PUSH, 4, 128, 64, 1, 0, 64, 0, 0}, {POL_DENSE, 4, 0, 128, 1, 0, 8, 0, 0}, {POL_DENSE, 4, 0, 64, 2, 0, 128, TANH, SV},
       {POL_DENSE, 2, 0, 128, 0, 0, 1, 0, SV}},
      6, {32, 16, 128, 16, 192, 256}, 22, 4, 1, 0, 8, 0, 0,
      2, {{1, 64, 1}, {1, 64, 1}}, 0, 0, 0};
};
struct ArchLstmBigger {  // CustomLSTMbigger (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {16,
      {{POL_COPY_OBS, 0, 0, 22, 0, 0, 22, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 0, 22, 4, 0, 0}, {POL_RING_LOAD, 0, 0, 64, 4, 64, 64, 0, 0},
       {POL_RING_LOAD, 1, 0, 64, 4, 128, 64, 0, 0}, {POL_DENSE, 0, 0, 26, 2, 0, 64, TANH, 0}, {POL_DENSE, 2, 0, 64, 4, 0, 64, TANH, 0},
       {POL_AFFINE, 4, 0, 64, 4, 0, 64, 0, 0}, {POL_DENSE, 4, 0, 128, 5, 0, 256, 0, 0}, {POL_LSTM_CELL, 5, 0, 256, 4, 64, 64, 0, 0},
       {POL_RING_PUSH, 4, 64, 64, 0, 0, 64, 0, 0}, {POL_RING_PUSH, 4, 128, 64, 1, 0, 64, 0, 0}, {POL_DENSE, 4, 0, 128, 2, 0, 64, TANH, 0},
       {POL_DENSE, 2, 0, 64, 1, 0, 8, 0, 0}, {POL_DENSE, 4, 0, 64, 2, 0, 128, TANH, SV}, {POL_DENSE, 2, 0, 128, 2, 128, 128, TANH, SV},
       {POL_DENSE, 2, 128, 128, 0, 0, 1, 0, SV}},
      6, {32, 16, 256, 16, 192, 256}, 22, 4, 1, 0, 8, 0, 0,
      2, {{1, 64, 1}, {1, 64, 1}}, 0, 0, 0};
};
struct ArchLstmCommonF {  // CustomLSTMbiggerCommonF (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {16,
      {{POL_COPY_OBS, 0, 0, 22, 0, 0, 22, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 0, 22, 4, 0, 0}, {POL_RING_LOAD, 0, 0, 64, 4, 64, 64, 0, 0},
       {POL_RING_LOAD, 1, 0, 64, 4, 128, 64, 0, 0}, {POL_DENSE, 0, 0, 26, 2, 0, 64, TANH, 0}, {POL_DENSE, 2, 0, 64, 4, 0, 64, TANH, 0},
       {POL_AFFINE, 4, 0, 64, 4, 0, 64, 0, 0}, {POL_DENSE, 4, 0, 128, 5, 0, 256, 0, 0}, {POL_LSTM_CELL, 5, 0, 256, 4, 64, 64, 0, 0},
       {POL_RING_PUSH, 4, 64, 64, 0, 0, 64, 0, 0}, {POL_RING_PUSH, 4, 128, 64, 1, 0, 64, 0, 0}, {POL_DENSE, 4, 0, 128, 2, 0, 64, TANH, 0},
       {POL_DENSE, 2, 0, 64, 1, 0, 8, 0, 0}, {POL_DENSE, 4, 0, 128, 2, 0, 128, TANH, SV}, {POL_DENSE, 2, 0, 128, 2, 128, 128, TANH, SV},
       {POL_DENSE, 2, 128, 128, 0, 0, 1, 0, SV}},
      6, {32, 16, 256, 16, 192, 256}, 22, 4, 1, 0, 8, 0, 0,
      2, {{1, 64, 1}, {1, 64, 1}}, 0, 0, 0};
};
struct ArchDsnLstm {  // DSN_LSTM_model (mujoco_drone_amd/policy.py; table printed by tools/emit_policy_arch.py)
  static constexpr SProg prog = {15,
      {{POL_COPY_OBS, 0, 0, 12, 0, 0, 12, 0, 0}, {POL_COPY_PREV, 0, 0, 4, 4, 0, 4, 0, 0}, {POL_RING_LOAD, 0, 0, 160, 4, 84, 160, 0, 0},
       {POL_DENSE, 0, 0, 12, 2, 0, 160, TANH, 0}, {POL_DENSE, 2, 0, 160, 3, 0, 160, TANH, 0}, {POL_DENSE, 3, 0, 160, 4, 4, 80, TANH, 0},
       {POL_AFFINE, 4, 4, 80, 4, 4, 80, 0, 0}, {POL_DENSE, 4, 4, 160, 5, 0, 320, 0, 0}, {POL_LSTM_CELL, 5, 0, 320, 4, 84, 80, 0, 0},
       {POL_RING_PUSH, 4, 84, 160, 0, 0, 160, 0, 0}, {POL_DENSE, 4, 0, 164, 2, 0, 64, TANH, 0}, {POL_DENSE, 2, 0, 64, 1, 0, 8, 0, 0},
       {POL_DENSE, 4, 4, 80, 2, 0, 128, TANH, SV}, {POL_DENSE, 2, 0, 128, 2, 128, 128, TANH, SV}, {POL_DENSE, 2, 128, 128, 0, 0, 1, 0, SV}},
      6, {16, 16, 256, 160, 256, 320}, 22, 4, 1, 0, 8, 0, 0,
      1, {{1, 160, 1}}, 0, 0, 0};
};

// every specialisation and the id qd_policy_kernel() reports for it (0 = the generic interpreter, qd_policy.h)
#define QD_POL_ARCHS(X)                                                                                                  \
  X(1, ArchRmaFull) X(2, ArchRmaModel) X(3, ArchSimpleMlp) X(4, ArchRmaFullAdapt) X(5, ArchCnnEst) X(6, ArchCnnEstHist) \
  X(7, ArchCustomMlp) X(8, ArchLstmEst) X(9, ArchRmaSmaller) X(10, ArchRmaSmaller2) X(11, ArchCustomLstm)              \
  X(12, ArchLstmBigger) X(13, ArchLstmCommonF) X(14, ArchDsnLstm)

// Ops whose result depends on the drone parameters only (the parameter encoder of the RMA networks, writing the auxiliary slice z):
// within one fused rollout the parameters of an env do not change (in-kernel resets keep them; regeneration is a host call
// between fragments), so the fused rollout (qd_rollout_fused.hip) runs them on the first step only and keeps z in LDS.
template <class A> inline constexpr unsigned fused_const_ops = 0u;
template <> inline constexpr unsigned fused_const_ops<ArchRmaFull> = (1u << 3) | (1u << 4);
template <> inline constexpr unsigned fused_const_ops<ArchRmaModel> = (1u << 3) | (1u << 4);
template <> inline constexpr unsigned fused_const_ops<ArchRmaSmaller> = (1u << 3) | (1u << 4);

// ---- the specialised kernel ----
struct SCtx {
  float* lds;
  const float* small;      // LDS mirror of the small region
  _Float16* mir;           // the float16 mirror of the activation buffers: hi plane, lo plane sp_hplane halves further
  // packed split weights, read through a buffer descriptor: the address of a load is descriptor (4 SGPRs) + a scalar byte offset
  // (tile, k-block: wave-uniform) + ONE vector register (16 lane) that every load of the kernel shares.  With plain pointers the
  // compiler kept a 64-bit vector address per 4 KB window of every layer alive across the fused rollout's step loop: 50-70 spilled
  // registers under that kernel's 256-register cap.
  __amdgpu_buffer_rsrc_t wrs;
  int lane16;
  int lane;
  const float* obs;
  const float* prev_actions;
  const uint8_t* prev_truncated;
  int n_envs, env0, tid, wave, li, lg;
  bool want_value;
  const float* small_global;  // the small region in the blob (the prologue reads ring fill values before the LDS mirror exists)
  float* state;            // per-env history rings (windowed networks)
  unsigned counter;        // the caller's step counter: selects ring slots and banks
  unsigned skip_ops;       // bit I set: op I is skipped (fused rollouts: the parameter encoder after the first step); 0 elsewhere
};

typedef _Float16 pol_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pol_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 pol_h2 __attribute__((ext_vector_type(2)));
typedef float pol_f32x2 __attribute__((ext_vector_type(2)));
// x = hi + lo / 2048 with hi, lo in float16: hi = x cut to 11 significant bits (a mask; exact in float16, so the packed conversion
// v_cvt_pk_f16_f32 does not round), lo = what is left, scaled by a power of two so that it stays a normal number wherever hi is
// one, rounded to 11 bits: the pair misses x by less than 2^-21 |x|.  BOUNDED: the values are known to lie inside float16's range
// (tanh outputs); otherwise they saturate at +-6e4 (no network of the reference gets there).  Below 6e-5 hi rounds to a
// subnormal: an absolute error under 3e-8.  Four values at a time: 4 masks, 4 subtractions, 4 products, 4 packed conversions.
template <bool BOUNDED>
__device__ __forceinline__ void pol_split4(const float (&x)[4], pol_h4& h, pol_h4& l) {
  float t[4], r[4];
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const float c = BOUNDED ? x[v] : __builtin_amdgcn_fmed3f(x[v], -6.0e4f, 6.0e4f);
    t[v] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) & 0xFFFFE000u);
    r[v] = (c - t[v]) * 2048.0f;
  }
  const pol_h2 h0 = __builtin_convertvector(pol_f32x2{t[0], t[1]}, pol_h2), h1 = __builtin_convertvector(pol_f32x2{t[2], t[3]}, pol_h2);
  const pol_h2 l0 = __builtin_convertvector(pol_f32x2{r[0], r[1]}, pol_h2), l1 = __builtin_convertvector(pol_f32x2{r[2], r[3]}, pol_h2);
  h = pol_h4{h0[0], h0[1], h1[0], h1[1]};
  l = pol_h4{l0[0], l0[1], l1[0], l1[1]};
}

constexpr int POL_MB = POL_TILE / 16;   // 16-env MFMA blocks per workgroup
constexpr int POL_UMAX = 4 / POL_MB;    // output tiles a wave accumulates at a time: 4 x 1 block or 2 x 2 blocks, 32 accumulator registers
static_assert(POL_MB == 1 || POL_MB == 2, "POL_TILE is 16 or 32");
#ifndef QD_POL_SPF
#define QD_POL_SPF 4
#endif
constexpr int SPF = QD_POL_SPF;  // k-blocks (of 32 inputs) of the next dense layer requested before the current layer's barrier
template <class A, int I> struct SDense {
  static constexpr int K32 = I < A::prog.n_ops ? sp_k32(A::prog, I) : 1;
  static constexpr int NT = I < A::prog.n_ops ? sp_ntiles(A::prog, I) : 1;
  static constexpr int SLOTS = (NT + POL_WAVES - 1) / POL_WAVES;   // tile slots per wave (tile = wave + 4 slot, clamped)
  static constexpr int U0 = sp_min(SLOTS, POL_UMAX);               // tiles of the first group
  static constexpr int PB = sp_min(K32, SPF);                      // prefetched k-blocks (first group only)
};
template <class A, int I> struct SPre {  // prefetched weights of dense op I (empty if I is past the end): [k-block][tile][hi | lo]
  float4 w[SDense<A, I>::PB][SDense<A, I>::U0][2];
};

typedef unsigned pol_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pol_weight_rsrc(const float* base) {
  // raw buffer (stride 0), bounds wide open: the offsets are compile-time functions of the program
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 pol_wload(const SCtx& c, int byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(c.wrs, c.lane16, byte_off, 0));
}
// split weights of one 16-column tile: [k-block of 32][hi | lo][lane][8 halves], W[n = 16 tile + (lane & 15)][k = 32 kb + 8 (lane >> 4) + j];
// returns the tile's byte offset in the split region (wave-uniform)
template <class A, int I>
__device__ __forceinline__ int s_tile_off(const SCtx& c, int slot) {
  constexpr int K32 = SDense<A, I>::K32, NT = SDense<A, I>::NT;
  constexpr int W0 = I < A::prog.n_ops ? (int)(sp_ws_at(A::prog, I) * 4) : 0;
  const int tile = min(c.wave + POL_WAVES * slot, NT - 1);  // waves without a tile in this slot repeat the last one
  return W0 + tile * (K32 * 2048);
}

template <class A, int I>
__device__ __forceinline__ void s_prefetch(const SCtx& c, SPre<A, I>& pre) {
  if constexpr (I < A::prog.n_ops) {
#pragma unroll
    for (int u = 0; u < SDense<A, I>::U0; u++) {
      const int off = s_tile_off<A, I>(c, u);
#pragma unroll
      for (int kb = 0; kb < SDense<A, I>::PB; kb++) {
        pre.w[kb][u][0] = pol_wload(c, off + kb * 2048);
        pre.w[kb][u][1] = pol_wload(c, off + kb * 2048 + 1024);
      }
    }
  }
}

// One dense layer for the tile's envs, 16 at a time (POL_MB blocks).  The product runs transposed, D[n][env] = sum_k W[n][k] x[env][k],
// with the weights as the MFMA's A operand and the activations as its B operand (both: lane = row + 16 (k / 8), eight consecutive k
// per lane), so that a lane ends up with FOUR CONSECUTIVE output features of ONE env (D: lane = env + 16 (n / 4)) -- one 16-byte
// store of the float32 values, one 8-byte store per mirror plane.  Three MFMAs per k-block, tile and env block: hi hi into one
// accumulator, hi lo + lo hi into a second that is scaled by 2^-11 at the end; lo lo (< 2^-22) is dropped.  Against the float32
// MFMA (16x16x4: 32 cycles per 4 k) that is 48 cycles per 32 k.  With two env blocks every weight register is used twice.
template <class A, int I>
__device__ __forceinline__ void s_dense(const SCtx& c, const SPre<A, I>& pre) {
  constexpr SOp op = A::prog.op[I];
  constexpr int K32 = SDense<A, I>::K32, NT = SDense<A, I>::NT, SLOTS = SDense<A, I>::SLOTS, PB = SDense<A, I>::PB;
  constexpr bool MF = sp_mirror_fed(A::prog, I), MO = sp_out_mirrored(A::prog, I);
  constexpr int ld_in = sp_ld(A::prog, op.in_buf), ld_out = sp_ld(A::prog, op.out_buf);
  constexpr int in_base = sp_base(A::prog, op.in_buf) + op.in_off, out_base = sp_base(A::prog, op.out_buf) + op.out_off;
  constexpr int hld_in = sp_hld(A::prog, op.in_buf), hin_base = sp_hbase(A::prog, op.in_buf) + op.in_off;
  constexpr int hld_out = sp_hld(A::prog, op.out_buf), hout_base = sp_hbase(A::prog, op.out_buf) + op.out_off;
  constexpr int HP = sp_hplane(A::prog);
  constexpr int s_at = sp_s_at(A::prog, I);
  constexpr bool VEC = op.out_off % 4 == 0 && op.out_dim % 4 == 0;   // a lane's four features are stored as one vector
  constexpr int MB = POL_MB, UMAX = POL_UMAX;
  const float* x_f32 = c.lds + in_base + c.li * ld_in + c.lg * 8;
  const _Float16* x_mir = c.mir + hin_base + c.li * hld_in + c.lg * 8;
  float* o_ptr = c.lds + out_base + c.li * ld_out + 4 * c.lg;
  _Float16* m_ptr = c.mir + hout_base + c.li * hld_out + 4 * c.lg;
  const float* bias = c.small + s_at + 4 * c.lg;
  constexpr int aff_at = sp_fused_affine(A::prog, I) ? sp_s_at(A::prog, I + 1) : 0;
  const float* aff = c.small + aff_at + 4 * c.lg;  // scale at [col], shift at [out_dim + col] of the affine op that follows
#pragma unroll
  for (int g0 = 0; g0 < SLOTS; g0 += UMAX) {
    const int U = sp_min(SLOTS - g0, UMAX);  // compile-time after unrolling
    pol_f32x4 acc[UMAX][MB], acx[UMAX][MB];
    int wp[UMAX];
#pragma unroll
    for (int u = 0; u < UMAX; u++) {
      if (u < U) {
        const int tile = min(c.wave + POL_WAVES * (g0 + u), NT - 1);
        const float4 b = *reinterpret_cast<const float4*>(bias + tile * 16);
#pragma unroll
        for (int eb = 0; eb < MB; eb++) {
          acc[u][eb] = pol_f32x4{b.x, b.y, b.z, b.w};
          acx[u][eb] = pol_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wp[u] = s_tile_off<A, I>(c, g0 + u);
      }
    }
    // The weights run PB k-blocks ahead of the MFMAs through a ring of registers: the loads of block kb + PB leave when block kb
    // is taken out of the ring.  (Left to itself the compiler requests a block where it is used and keeps three or four loads in
    // flight -- the 256 -> 128 layer then waits for L2 once per k-block: 2.3 us for 0.33 us of MFMAs.)
    float4 wq[PB][UMAX][2];
#pragma unroll
    for (int d = 0; d < PB; d++)
#pragma unroll
      for (int u = 0; u < UMAX; u++)
        if (u < U) {
          if (g0 == 0) {
            wq[d][u][0] = pre.w[d][u < SDense<A, I>::U0 ? u : 0][0];
            wq[d][u][1] = pre.w[d][u < SDense<A, I>::U0 ? u : 0][1];
          } else {
            wq[d][u][0] = pol_wload(c, wp[u] + d * 2048);
            wq[d][u][1] = pol_wload(c, wp[u] + d * 2048 + 1024);
          }
        }
#pragma unroll
    for (int kb = 0; kb < K32; kb++) {
      pol_h8 xh[MB], xl[MB];
#pragma unroll
      for (int eb = 0; eb < MB; eb++) {
        if constexpr (MF) {
          xh[eb] = *reinterpret_cast<const pol_h8*>(x_mir + eb * 16 * hld_in + kb * 32);
          xl[eb] = *reinterpret_cast<const pol_h8*>(x_mir + eb * 16 * hld_in + HP + kb * 32);
        } else {
          const float* xp = x_f32 + eb * 16 * ld_in + kb * 32;
          const float4 v0 = *reinterpret_cast<const float4*>(xp), v1 = *reinterpret_cast<const float4*>(xp + 4);
          float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if ((kb + 1) * 32 > op.in_dim) {   // (compile-time once unrolled) columns past the slice, the k-block's padding, may hold anything: they count as zero
#pragma unroll
            for (int j = 0; j < 8; j++) xv[j] = kb * 32 + c.lg * 8 + j < op.in_dim ? xv[j] : 0.f;
          }
          pol_h4 ha, la, hb, lb;
          const float xa[4] = {xv[0], xv[1], xv[2], xv[3]}, xb[4] = {xv[4], xv[5], xv[6], xv[7]};
          pol_split4<false>(xa, ha, la);
          pol_split4<false>(xb, hb, lb);
          xh[eb] = pol_h8{ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
          xl[eb] = pol_h8{la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
        }
      }
      float4 w[UMAX][2];
#pragma unroll
      for (int u = 0; u < UMAX; u++)
        if (u < U) {
          w[u][0] = wq[kb % PB][u][0];
          w[u][1] = wq[kb % PB][u][1];
        }
#pragma unroll
      for (int u = 0; u < UMAX; u++)
#pragma unroll
        for (int eb = 0; eb < MB; eb++)
          if (u < U) acc[u][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pol_h8, w[u][0]), xh[eb], acc[u][eb], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < UMAX; u++)
#pragma unroll
        for (int eb = 0; eb < MB; eb++)
          if (u < U) acx[u][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pol_h8, w[u][0]), xl[eb], acx[u][eb], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < UMAX; u++)
#pragma unroll
        for (int eb = 0; eb < MB; eb++)
          if (u < U) acx[u][eb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pol_h8, w[u][1]), xh[eb], acx[u][eb], 0, 0, 0);
      if (kb + PB < K32) {   // the slot is free: block kb + PB takes it, requested HERE (the fences keep the scheduler from moving the loads to their uses)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UMAX; u++)
          if (u < U) {
            wq[kb % PB][u][0] = pol_wload(c, wp[u] + (kb + PB) * 2048);
            wq[kb % PB][u][1] = pol_wload(c, wp[u] + (kb + PB) * 2048 + 1024);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int u = 0; u < UMAX; u++) {
      if (u < U) {
        const int tile = c.wave + POL_WAVES * (g0 + u);
        if (tile < NT && tile * 16 + 4 * c.lg < op.out_dim) {
#pragma unroll
          for (int eb = 0; eb < MB; eb++) {
            float y[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
              y[v] = pol_act(fmaf(acx[u][eb][v], 1.0f / 2048.0f, acc[u][eb][v]), op.act);
              if constexpr (sp_fused_affine(A::prog, I)) y[v] = fmaf(y[v], aff[tile * 16 + v], aff[op.out_dim + tile * 16 + v]);  // eval-mode BatchNorm
            }
            float* o = o_ptr + eb * 16 * ld_out + tile * 16;
            if constexpr (VEC) {
              *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (tile * 16 + 4 * c.lg + v < op.out_dim) o[v] = y[v];
            }
            if constexpr (MO) {
              pol_h4 h, l;
              pol_split4<op.act == POL_ACT_TANH && !sp_fused_affine(A::prog, I)>(y, h, l);
              _Float16* m = m_ptr + eb * 16 * hld_out + tile * 16;
              if constexpr (VEC) {
                *reinterpret_cast<pol_h4*>(m) = h;
                *reinterpret_cast<pol_h4*>(m + HP) = l;
              } else {
#pragma unroll
                for (int v = 0; v < 4; v++)
                  if (tile * 16 + 4 * c.lg + v < op.out_dim) { m[v] = h[v]; m[HP + v] = l[v]; }
              }
            }
          }
        }
      }
    }
  }
}

template <class A, int I> struct SCopy {  // one leading gather: COPY_OBS / COPY_PREV row slice or a RING_LOAD window
  static constexpr SOp OP = A::prog.op[I < A::prog.n_ops ? I : 0];
  static constexpr int N = OP.kind == POL_RING_LOAD ? OP.out_dim : OP.in_dim;
  static constexpr int IT = (POL_TILE * N + POL_THREADS - 1) / POL_THREADS;
  float v[IT];
};
template <class A, int I>
__device__ __forceinline__ void s_copy_load(const SCtx& c, SCopy<A, I>& r) {
  constexpr SOp op = A::prog.op[I];
  constexpr int n = SCopy<A, I>::N, obs_dim = A::prog.obs_dim, act_dim = A::prog.act_dim;
#pragma unroll
  for (int it = 0; it < SCopy<A, I>::IT; it++) {
    const int k = c.tid + it * POL_THREADS;
    const int row = k / n, col = k - row * n, e = c.env0 + row;
    float v = 0.f;
    if (k < POL_TILE * n && e < c.n_envs) {
      if constexpr (op.kind == POL_COPY_OBS) {
        v = c.obs[(size_t)e * obs_dim + op.in_off + col];
      } else if constexpr (op.kind == POL_COPY_PREV) {
        if (c.prev_actions && !(c.prev_truncated && c.prev_truncated[e])) v = c.prev_actions[(size_t)e * act_dim + op.in_off + col];
      } else {  // POL_RING_LOAD: the slots written by the last R steps of this step's bank, oldest first
        constexpr int rg = op.in_buf, R = A::prog.ring[rg].rows, W = A::prog.ring[rg].width, per = A::prog.ring[rg].period;
        constexpr int off = sp_ring_off(A::prog, rg), fill = sp_ring_fill(A::prog, rg), S = sp_state_floats(A::prog);
        const unsigned tt = per == 2 ? c.counter >> 1 : c.counter;
        const int bank = per == 2 ? (int)(c.counter & 1u) : 0;
        const int j = col / W, cc = col - j * W;
        const int slot = (int)((tt + (unsigned)j) % (unsigned)R);
        const bool fresh = c.prev_truncated && c.prev_truncated[e];
        // the fill values are read from the blob here (the LDS mirror is not written yet in the prologue)
        v = fresh ? c.small_global[fill + cc] : c.state[(size_t)e * S + off + (bank * R + slot) * W + cc];
      }
    }
    r.v[it] = v;
  }
}
template <class A, int I>
__device__ __forceinline__ void s_copy_store(const SCtx& c, const SCopy<A, I>& r) {
  constexpr SOp op = A::prog.op[I];
  constexpr int n = SCopy<A, I>::N, ld = sp_ld(A::prog, op.out_buf), out_base = sp_base(A::prog, op.out_buf) + op.out_off;
#pragma unroll
  for (int it = 0; it < SCopy<A, I>::IT; it++) {
    const int k = c.tid + it * POL_THREADS;
    const int row = k / n, col = k - row * n;
    if (k < POL_TILE * n) c.lds[out_base + row * ld + col] = r.v[it];
  }
}
template <class A, int I>
__device__ __forceinline__ void s_other(const SCtx& c) {
  constexpr SOp op = A::prog.op[I];
  constexpr int OB = op.kind == POL_RING_PUSH ? 0 : op.out_buf;  // a push's out_buf is a ring index, not a buffer
  constexpr int ld = sp_ld(A::prog, OB), out_base = sp_base(A::prog, OB) + op.out_off;
  constexpr int s_at = sp_s_at(A::prog, I), obs_dim = A::prog.obs_dim, act_dim = A::prog.act_dim;
  float* b = c.lds + out_base;
  if constexpr (op.kind == POL_RING_PUSH) {
    constexpr int rg = op.out_buf, R = A::prog.ring[rg].rows, W = A::prog.ring[rg].width, per = A::prog.ring[rg].period;
    constexpr int off = sp_ring_off(A::prog, rg), fill = sp_ring_fill(A::prog, rg), S = sp_state_floats(A::prog);
    constexpr int ld_in = sp_ld(A::prog, op.in_buf), in_base = sp_base(A::prog, op.in_buf) + op.in_off;
    const unsigned tt = per == 2 ? c.counter >> 1 : c.counter;
    const int bank = per == 2 ? (int)(c.counter & 1u) : 0;
    const int slot_new = (int)(tt % (unsigned)R);
    for (int k = c.tid; k < POL_TILE * W; k += POL_THREADS) {
      const int r = k / W, col = k - r * W, e = c.env0 + r;
      if (e < c.n_envs) {
        float* ring = c.state + (size_t)e * S + off;
        if (c.prev_truncated && c.prev_truncated[e])
          for (int q = 0; q < per * R; q++) ring[q * W + col] = c.small[fill + col];
        ring[(bank * R + slot_new) * W + col] = c.lds[in_base + r * ld_in + col];
      }
    }
  } else if constexpr (op.kind == POL_LSTM_CELL) {
    // torch.nn.LSTM cell, gate order (i, f, g, o); c sits right after h in the output buffer
    constexpr int H = op.out_dim, ldg = sp_ld(A::prog, op.in_buf), g_base = sp_base(A::prog, op.in_buf) + op.in_off;
    for (int k = c.tid; k < POL_TILE * H; k += POL_THREADS) {
      const int r = k / H, j = k - r * H;
      const float* g4 = c.lds + g_base + r * ldg + j;
      float* hc = c.lds + out_base + r * ld + j;
      const float si = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[0])), sf = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[H]));
      const float so = __builtin_amdgcn_rcpf(1.0f + __expf(-g4[3 * H]));
      const float cn = fmaf(sf, hc[H], si * pol_act(g4[2 * H], POL_ACT_TANH));
      hc[H] = cn;
      hc[0] = so * pol_act(cn, POL_ACT_TANH);
    }
  } else if constexpr (op.kind == POL_RING_LOAD) {
    SCopy<A, I> tmp;  // a ring load that is not in the prologue
    s_copy_load<A, I>(c, tmp);
    s_copy_store<A, I>(c, tmp);
  } else if constexpr (op.kind == POL_AFFINE) {
    constexpr int n = op.out_dim;
    const float* sc = c.small + s_at;
    for (int k = c.tid; k < POL_TILE * n; k += POL_THREADS) {
      const int r = k / n, col = k - r * n;
      b[r * ld + col] = fmaf(b[r * ld + col], sc[col], sc[n + col]);
    }
  } else {
    constexpr int n = op.in_dim;
    for (int k = c.tid; k < POL_TILE * n; k += POL_THREADS) {
      const int r = k / n, col = k - r * n;
      const int e = c.env0 + r;
      float v = 0.f;
      if (e < c.n_envs) {
        if constexpr (op.kind == POL_COPY_OBS) v = c.obs[(size_t)e * obs_dim + op.in_off + col];
        else if (c.prev_actions && !(c.prev_truncated && c.prev_truncated[e])) v = c.prev_actions[(size_t)e * act_dim + op.in_off + col];
      }
      b[r * ld + col] = v;
    }
  }
}

// the first LC ops are input gathers: all their loads are issued together (with the small region's) in the prologue
template <class A, int I, int LC> struct SLead {
  SCopy<A, I> cur;
  SLead<A, I + 1, LC> rest;
  __device__ __forceinline__ void load(const SCtx& c) { s_copy_load<A, I>(c, cur); rest.load(c); }
  __device__ __forceinline__ void store(const SCtx& c) const { s_copy_store<A, I>(c, cur); rest.store(c); }
};
template <class A, int LC> struct SLead<A, LC, LC> {
  __device__ __forceinline__ void load(const SCtx&) {}
  __device__ __forceinline__ void store(const SCtx&) const {}
};

// LDS is not cleared as a whole: every slice a dense layer reads is written earlier in the same launch, except the padding
// columns that round its K up to whole k-blocks of 16 (and slices a program deliberately reads before writing them, which
// lie in the same range).  Only those columns are zeroed: they meet zero weights, but must not hold NaNs.
template <class A, int I>
__device__ __forceinline__ void s_zero_pads(const SCtx& c) {
  if constexpr (I < A::prog.n_ops) {
    constexpr SOp op = A::prog.op[I];
    if constexpr (op.kind == POL_DENSE) {
      constexpr int pad0 = op.in_off + op.in_dim, pad1 = op.in_off + sp_k16(A::prog, I) * 16, n = pad1 - pad0;
      if constexpr (n > 0) {
        constexpr int ld = sp_ld(A::prog, op.in_buf), base = sp_base(A::prog, op.in_buf) + pad0;
        for (int k = c.tid; k < POL_TILE * n; k += POL_THREADS) c.lds[base + (k / n) * ld + (k % n)] = 0.f;
      }
    }
    s_zero_pads<A, I + 1>(c);
  }
}

// ops I.. of the program; `pre` holds the prefetched weights of dense op J = the first dense op at or after I
template <class A, int I, int J>
__device__ __forceinline__ void s_run(const SCtx& c, const SPre<A, J>& pre) {
  if constexpr (I < A::prog.n_ops) {
    constexpr SOp op = A::prog.op[I];
    const bool runs = (c.want_value || !(op.flags & POL_FLAG_VALUE_ONLY)) && !((c.skip_ops >> I) & 1u);  // uniform over the workgroup
    if constexpr (op.kind == POL_DENSE) {
      static_assert(I == J, "prefetch bookkeeping");
      constexpr int JN = sp_next_dense(A::prog, I + 1);
      SPre<A, JN> next;
      s_prefetch<A, JN>(c, next);
      if (runs) {
        s_dense<A, I>(c, pre);
        __syncthreads();
        POL_STAMP(2 + I);
      }
      s_run<A, I + 1, JN>(c, next);
    } else {
      constexpr bool fused = I > 0 && sp_fused_affine(A::prog, I > 0 ? I - 1 : 0);  // already applied by the dense op before it
      if constexpr (!fused) {
        if (runs) {
          s_other<A, I>(c);
          __syncthreads();
          POL_STAMP(2 + I);
        }
      }
      s_run<A, I + 1, J>(c, pre);
    }
  }
}

template <class A>
__global__ __launch_bounds__(POL_THREADS) void k_policy_static(PolArgs p, int n_envs, const float* __restrict__ obs,
                                                               const float* __restrict__ prev_actions,
                                                               const uint8_t* __restrict__ prev_truncated, PolSample smp,
                                                               float* __restrict__ state,
                                                               float* __restrict__ actions, float* __restrict__ logp,
                                                               float* __restrict__ logits, float* __restrict__ value,
                                                               float* __restrict__ aux) {
  extern __shared__ float lds[];
  constexpr int ACT = sp_act_floats(A::prog);
  SCtx c;
  c.lds = lds; c.small = lds + ACT;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = c.tid & 63;
  c.li = lane & 15; c.lg = lane >> 4;
  c.mir = reinterpret_cast<_Float16*>(lds + sp_base(A::prog, A::prog.n_bufs));
  c.wrs = pol_weight_rsrc(p.packed + p.wsplit_off); c.lane16 = lane * 16; c.lane = lane;
  c.obs = obs; c.prev_actions = prev_actions; c.prev_truncated = prev_truncated;
  c.n_envs = n_envs; c.env0 = blockIdx.x * POL_TILE; c.want_value = value != nullptr;
  c.small_global = p.packed + p.prog_ints; c.state = state; c.counter = smp.counter; c.skip_ops = 0u;
  // the first dense layer's weights are requested before anything else
  constexpr int J0 = sp_next_dense(A::prog, 0);
  POL_STAMP(0);
  SPre<A, J0> pre;
  s_prefetch<A, J0>(c, pre);
  // prologue: every global read that does not depend on the network (biases / affine parameters, the leading input
  // gathers) is issued at once, LDS is cleared under their latency, then they are written behind one barrier
  constexpr int LC = sp_leading_copies(A::prog), S4 = sp_small_floats(A::prog) / 4, SIT = (S4 + POL_THREADS - 1) / POL_THREADS;
  float4 sm[SIT];
  {
    const float4* src = reinterpret_cast<const float4*>(p.packed + p.prog_ints);
#pragma unroll
    for (int it = 0; it < SIT; it++) {
      const int k = c.tid + it * POL_THREADS;
      sm[it] = k < S4 ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  SLead<A, 0, LC> lead;
  lead.load(c);
  s_zero_pads<A, 0>(c);
  {  // the mirror's padding columns meet zero weights, but must not hold NaNs either
    constexpr int M4 = sp_mirror_floats(A::prog) / 4;
    float4* mz = reinterpret_cast<float4*>(lds + sp_base(A::prog, A::prog.n_bufs));
    for (int k = c.tid; k < M4; k += POL_THREADS) mz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  {
    float4* dst = reinterpret_cast<float4*>(lds + ACT);
#pragma unroll
    for (int it = 0; it < SIT; it++) {
      const int k = c.tid + it * POL_THREADS;
      if (k < S4) dst[k] = sm[it];
    }
  }
  lead.store(c);
  __syncthreads();
  POL_STAMP(1);
  s_run<A, LC, J0>(c, pre);
  // outputs: value, then logits / action / log-prob (MyBetaDist, qd_policy_dist.h)
  constexpr int ldl = sp_ld(A::prog, A::prog.logits_buf), NL = A::prog.n_logits, AD = A::prog.act_dim;
  constexpr int lg_base = sp_base(A::prog, A::prog.logits_buf) + A::prog.logits_off;
  constexpr int VB = A::prog.value_buf < 0 ? 0 : A::prog.value_buf;
  constexpr int v_base = sp_base(A::prog, VB) + A::prog.value_off, v_ld = sp_ld(A::prog, VB);
  constexpr bool has_value = A::prog.value_buf >= 0;
  if (has_value && c.want_value && c.tid < POL_TILE && c.env0 + c.tid < n_envs) value[c.env0 + c.tid] = lds[v_base + c.tid * v_ld];
  if constexpr (A::prog.aux_dim > 0) {
    constexpr int NA = A::prog.aux_dim, a_base = sp_base(A::prog, A::prog.aux_buf) + A::prog.aux_off, a_ld = sp_ld(A::prog, A::prog.aux_buf);
    if (aux)
      for (int k = c.tid; k < POL_TILE * NA; k += POL_THREADS) {
        const int r = k / NA, col = k - r * NA;
        if (c.env0 + r < n_envs) aux[(size_t)(c.env0 + r) * NA + col] = lds[a_base + r * a_ld + col];
      }
  }
  pol_outputs(lds + lg_base, ldl, NL, AD, c.env0, n_envs, c.tid, lds + ACT - POL_SCRATCH, smp, actions, logp, logits, nullptr, p.dist);
  POL_STAMP(2 + A::prog.n_ops);
}

// qd_rollout_fused.hip: the closed policy -> env loop of one fragment in one launch, the env step beside the forward pass (load
// model; SPEC_RMA rows with the feed-forward 22-value networks, arch ids 1, 2, 7, 9, and SPEC_LSTM rows with CNNestimator, arch id
// 5; hipErrorNotSupported for any other -- qd_rollout_policy then runs the two-launch loop).
// `lds_bytes`: the policy's dynamic LDS (qd_policy_create).  `k` as qd_step would pass it.
struct KArgs;
hipError_t launch_rollout_fused_pipe(int arch, const KArgs& k, const PolArgs& pa, size_t lds_bytes, int T, const PolSample& smp,
                                     const float* obs0, const float* prev0, float* obs, float* actions, float* reward, uint8_t* trunc,
                                     float* logp, float* logits, float* value, hipStream_t stream);

// ---- host: does a program handed to qd_policy_create equal one of the tables above? ----
template <class A>
inline bool pol_matches(const qd_policy_desc* d, const qd_policy_op* ops) {
  const SProg& s = A::prog;
  if (d->n_ops != s.n_ops || d->n_bufs != s.n_bufs || d->n_rings != s.n_rings || d->obs_dim != s.obs_dim || d->act_dim != s.act_dim) return false;
  for (int r = 0; r < s.n_rings; r++)
    if (d->ring[r].rows != s.ring[r].rows || d->ring[r].width != s.ring[r].width || d->ring[r].period != s.ring[r].period) return false;
  if (d->logits_buf != s.logits_buf || d->logits_off != s.logits_off || d->n_logits != s.n_logits) return false;
  if (d->value_buf != s.value_buf || d->value_off != s.value_off) return false;
  if (d->aux_dim != s.aux_dim || (s.aux_dim > 0 && (d->aux_buf != s.aux_buf || d->aux_off != s.aux_off))) return false;
  for (int b = 0; b < s.n_bufs; b++)
    if ((d->buf_width[b] + 15) / 16 != (s.width[b] + 15) / 16) return false;
  for (int k = 0; k < s.n_ops; k++) {
    const qd_policy_op& a = ops[k];
    const SOp& b = s.op[k];
    if (a.kind != b.kind || a.in_off != b.in_off || a.in_dim != b.in_dim || a.out_buf != b.out_buf || a.out_off != b.out_off ||
        a.out_dim != b.out_dim || a.act != b.act || a.flags != b.flags)
      return false;
    if ((a.kind == QD_POL_DENSE || a.kind == QD_POL_RING_LOAD || a.kind == QD_POL_RING_PUSH || a.kind == QD_POL_LSTM_CELL) && a.in_buf != b.in_buf) return false;
  }
  return true;
}
inline int pol_arch_of(const qd_policy_desc* d, const qd_policy_op* ops) {
  const char* e = getenv("QD_POLICY_GENERIC");  // testing: force the interpreter
  if (e && atoi(e)) return 0;
#define QD_POL_MATCH(ID, ARCH) \
  if (pol_matches<ARCH>(d, ops)) return ID;
  QD_POL_ARCHS(QD_POL_MATCH)
#undef QD_POL_MATCH
  return 0;
}

}  // namespace qd
