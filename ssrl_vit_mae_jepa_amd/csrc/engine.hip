// The MAE pretrain-step engine: parameter arena, workspace plan and the launch sequences of
// forward_encoder / forward_decoder / backward / optimizer step.
// Reference control flow: src/models/mae.py:54-94 (forward), src/training/mae.py:45-65 (step),
// scripts/training/pretrain_mae.py:124-125 (clip).  Everything here only enqueues kernels on the caller's stream.
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include "kernels.h"

namespace mae {

// ---------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t err, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)err, hipGetErrorString(err), file, line, what);
  return 2;
}

// ---------------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------------
struct ParamInfo {
  std::string name;
  int64_t offset = 0, numel = 0;
  int ndim = 0;
  int64_t shape[4] = {1, 1, 1, 1};
  int flags = 0;
  int64_t t_off = -1;  // element offset of the transposed bf16 copy inside the wcache transposed region
};

struct BlockRefs {  // indices into params
  int ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
};

enum TimerKind { TK_LINEAR = 0, TK_WGRAD, TK_ATTN_FWD, TK_ATTN_BWD, TK_LN_FWD, TK_LN_BWD, TK_DATA, TK_LOSS, TK_OPTIM, TK_COUNT };
static const char* kTimerNames[TK_COUNT] = {"linear_nt", "linear_wgrad", "attention_fwd", "attention_bwd", "layernorm_fwd",
                                            "layernorm_bwd", "token_data_movement", "mse_loss", "clip_adamw"};

struct TimerSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double flops = 0, bytes = 0;
};

}  // namespace mae

using namespace mae;

struct mae_engine {
  mae_config_t cfg;
  int act = MAE_F32;
  int D, depth, H, Dd, dd, Hd, P, L, G, C, img, p, mlp;
  int PO;  // width of the prediction head: P (MAE pixels) or cfg.pred_dim (I-JEPA: the encoder width)
  std::vector<ParamInfo> params;
  int64_t arena_elems = 0, trainable_elems = 0, trans_elems = 0;
  int i_enc_mask, i_cls, i_pos, i_patch_w, i_patch_b, i_norm_w, i_norm_b;
  int i_dec_mask, i_dec_pos, i_de_w, i_de_b, i_dn_w, i_dn_b, i_pred_w, i_pred_b;
  std::vector<BlockRefs> enc, dec;
  bool timers_on = false;
  TimerSlot timers[TK_COUNT];
  // weight-gradient GEMMs are off the backward critical path (their results are only read by the optimizer): they
  // are enqueued on a side stream so they overlap the HBM-bound LayerNorm / attention backward kernels of the main chain
  int side_mode = -1;  // -1 = undecided, 0 = off, 1 = on
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  PartialsTable ln_tab;  // LayerNorm dgamma / dbeta second stages queued by the current backward pass
  hipEvent_t pending[4] = {nullptr, nullptr, nullptr, nullptr};  // last side-stream reader of dres_c / d_hidden / d_qkv / one-off buffers
};

namespace mae {

static int add_param(mae_engine* e, const std::string& name, std::initializer_list<int64_t> shape, int flags) {
  ParamInfo pi;
  pi.name = name;
  pi.ndim = (int)shape.size();
  pi.numel = 1;
  int i = 0;
  for (int64_t s : shape) { pi.shape[i++] = s; pi.numel *= s; }
  pi.flags = flags;
  e->params.push_back(pi);
  return (int)e->params.size() - 1;
}

static BlockRefs add_block(mae_engine* e, const std::string& pfx, int d, int mlp) {
  const int T = MAE_PARAM_TRAINABLE, M = MAE_PARAM_TRAINABLE | MAE_PARAM_MATRIX;
  BlockRefs r;
  r.ln1_w = add_param(e, pfx + ".norm1.weight", {d}, T);
  r.ln1_b = add_param(e, pfx + ".norm1.bias", {d}, T);
  r.qkv_w = add_param(e, pfx + ".attn.qkv.weight", {3 * (int64_t)d, d}, M);
  r.qkv_b = add_param(e, pfx + ".attn.qkv.bias", {3 * (int64_t)d}, T);
  r.proj_w = add_param(e, pfx + ".attn.proj.weight", {d, d}, M);
  r.proj_b = add_param(e, pfx + ".attn.proj.bias", {d}, T);
  r.ln2_w = add_param(e, pfx + ".norm2.weight", {d}, T);
  r.ln2_b = add_param(e, pfx + ".norm2.bias", {d}, T);
  r.fc1_w = add_param(e, pfx + ".mlp.fc1.weight", {(int64_t)mlp * d, d}, M);
  r.fc1_b = add_param(e, pfx + ".mlp.fc1.bias", {(int64_t)mlp * d}, T);
  r.fc2_w = add_param(e, pfx + ".mlp.fc2.weight", {d, (int64_t)mlp * d}, M);
  r.fc2_b = add_param(e, pfx + ".mlp.fc2.bias", {d}, T);
  return r;
}

// ---------------------------------------------------------------------------------------------------
// workspace plan: byte offsets of every saved activation / scratch buffer for (B, k)
// ---------------------------------------------------------------------------------------------------
struct LayerBufs {
  int64_t x_mid, ln1, mean1, rstd1, qkv, lse, att, ln2, mean2, rstd2, fc1_pre, fc1_act;
};

struct Plan {
  int B, k, m;
  int dec_B, dec_T;  // decoder / predictor attention batch and sequence length: (B, L) for MAE, (B * nblk, k + m) for I-JEPA
  int64_t Me, Md, Mp;
  int64_t keep32, mask32, inv, pred_rows;
  int64_t patchA;
  std::vector<int64_t> enc_x, dec_x;
  std::vector<LayerBufs> enc, dec;
  int64_t enc_norm, enc_mean, enc_rstd, xdec, dec_norm, dec_mean, dec_rstd, pred;
  // backward scratch
  int64_t branch_a, branch_b;  // bf16/fp32 outputs of the attention / MLP branches (added by the next LayerNorm kernel)
  int64_t dpred, d_decn, dres, dres_c, d_ln, d_att, d_qkv, d_hidden, d_xdec, dtok;
  int64_t ln_partial, ln_partial_stride, split_partial, wgrad_scratch, loss_scratch;
  int64_t total;
};

// dec_B sequences of dec_T tokens go through the decoder stack, m_seq of them per sequence reach the prediction head
static Plan make_plan_ex(const mae_engine* e, int B, int k, int dec_B, int dec_T, int m_seq) {
  Plan pl;
  pl.B = B; pl.k = k; pl.m = m_seq;
  pl.dec_B = dec_B; pl.dec_T = dec_T;
  pl.Me = (int64_t)B * k; pl.Md = (int64_t)dec_B * dec_T; pl.Mp = (int64_t)dec_B * m_seq;
  const int64_t as = (int64_t)dtype_size(e->act);
  int64_t off = 0;
  auto take = [&](int64_t bytes) { const int64_t o = off; off += round_up(std::max<int64_t>(bytes, 4), 256); return o; };
  pl.keep32 = take(pl.Me * 4);
  pl.mask32 = take(pl.Mp * 4);
  pl.inv = take(pl.Md * 4);
  pl.pred_rows = take(pl.Mp * 4);
  pl.patchA = take(pl.Me * e->P * as);
  auto layers = [&](int n, int64_t M, int d, int heads, int T, std::vector<int64_t>& xs, std::vector<LayerBufs>& ls) {
    const int64_t seqs = M / std::max(T, 1);
    xs.resize(n + 1);
    ls.resize(n);
    xs[0] = take(M * d * 4);
    for (int i = 0; i < n; ++i) {
      LayerBufs& b = ls[i];
      b.ln1 = take(M * d * as); b.mean1 = take(M * 4); b.rstd1 = take(M * 4);
      b.qkv = take(M * 3 * d * as);
      b.lse = take(seqs * heads * T * 4);
      b.att = take(M * d * as);
      b.x_mid = take(M * d * 4);
      b.ln2 = take(M * d * as); b.mean2 = take(M * 4); b.rstd2 = take(M * 4);
      b.fc1_pre = take(M * e->mlp * d * as);
      b.fc1_act = take(M * e->mlp * d * as);
      xs[i + 1] = take(M * d * 4);
    }
  };
  layers(e->depth, pl.Me, e->D, e->H, k, pl.enc_x, pl.enc);
  pl.enc_norm = take(pl.Me * e->D * as); pl.enc_mean = take(pl.Me * 4); pl.enc_rstd = take(pl.Me * 4);
  pl.xdec = take(pl.Me * e->Dd * as);
  layers(e->dd, pl.Md, e->Dd, e->Hd, dec_T, pl.dec_x, pl.dec);
  pl.dec_norm = take(std::max<int64_t>(pl.Mp, 1) * e->Dd * as);
  pl.dec_mean = take(std::max<int64_t>(pl.Mp, 1) * 4); pl.dec_rstd = take(std::max<int64_t>(pl.Mp, 1) * 4);
  pl.pred = take(std::max<int64_t>(pl.Mp, 1) * e->PO * 4);
  // backward scratch, shared by the decoder and encoder sweeps
  const int64_t R = std::max(pl.Me * e->D, pl.Md * e->Dd);
  pl.branch_a = take(R * as);
  pl.branch_b = take(R * as);
  pl.dpred = take(std::max<int64_t>(pl.Mp, 1) * e->PO * as);
  pl.d_decn = take(std::max<int64_t>(pl.Mp, 1) * e->Dd * as);
  pl.dres = take(R * 4);
  pl.dres_c = take(R * as);
  pl.d_ln = take(R * as);
  pl.d_att = take(R * as);
  pl.d_qkv = take(3 * R * as);
  pl.d_hidden = take((int64_t)e->mlp * R * as);
  pl.d_xdec = take(pl.Me * e->Dd * as);
  pl.dtok = take(pl.Me * e->D * as);
  const int maxd = std::max(e->D, e->Dd);
  pl.ln_partial_stride = (int64_t)2 * LN_BWD_MAX_BLOCKS * maxd * 4;  // one slot per LayerNorm of the backward pass: their
  pl.ln_partial = take(pl.ln_partial_stride * std::min(2 * (e->depth + e->dd) + 2, (int)PartialsTable::MAX));  // second stages run in one launch at the end
  pl.split_partial = take((int64_t)512 * maxd * 4);
  int64_t wg = 0;
  auto wgs = [&](int64_t M, int N, int K) { wg = std::max(wg, linear_wgrad_scratch_bytes(M, N, K)); };
  wgs(pl.Me, 3 * e->D, e->D); wgs(pl.Me, e->mlp * e->D, e->D); wgs(pl.Me, e->D, e->mlp * e->D); wgs(pl.Me, e->D, e->D);
  wgs(pl.Md, 3 * e->Dd, e->Dd); wgs(pl.Md, e->mlp * e->Dd, e->Dd); wgs(pl.Md, e->Dd, e->mlp * e->Dd); wgs(pl.Md, e->Dd, e->Dd);
  wgs(pl.Me, e->D, e->P); wgs(pl.Me, e->Dd, e->D); wgs(std::max<int64_t>(pl.Mp, 1), e->PO, e->Dd);
  auto wgp = [&](int64_t M, int d) {  // the paired launches of a block: fc2 + fc1, proj + qkv
    wg = std::max(wg, linear_wgrad_pair_scratch_bytes(M, d, e->mlp * d, e->mlp * d, d));
    wg = std::max(wg, linear_wgrad_pair_scratch_bytes(M, d, d, 3 * d, d));
  };
  wgp(pl.Me, e->D); wgp(pl.Md, e->Dd);
  pl.wgrad_scratch = take(wg);
  pl.loss_scratch = take(4096 * 4);
  pl.total = off;
  return pl;
}
static Plan make_plan(const mae_engine* e, int B, int k) { return make_plan_ex(e, B, k, B, e->L, e->L - k); }

// ---------------------------------------------------------------------------------------------------
// timers
// ---------------------------------------------------------------------------------------------------
struct TimerScope {
  mae_engine* e; int kind; hipStream_t s; hipEvent_t stop = nullptr;
  TimerScope(mae_engine* e_, int kind_, double flops, double bytes, hipStream_t s_) : e(e_), kind(kind_), s(s_) {
    if (!e->timers_on) return;
    TimerSlot& t = e->timers[kind];
    if (t.used == t.ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      t.ev.push_back({a, b});
    }
    (void)hipEventRecord(t.ev[t.used].first, s);
    stop = t.ev[t.used].second;
    t.used++;
    t.flops += flops;
    t.bytes += bytes;
  }
  ~TimerScope() { if (stop) (void)hipEventRecord(stop, s); }
};

#define RUN(kind, flops, bytes, expr)                      \
  do {                                                     \
    TimerScope _ts(e, kind, (double)(flops), (double)(bytes), s); \
    MAE_TRY(expr);                                         \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// side stream for the weight-gradient GEMMs
// ---------------------------------------------------------------------------------------------------
enum DepTag { DEP_DRESC = 0, DEP_HIDDEN = 1, DEP_QKV = 2, DEP_MISC = 3 };

static bool side_enabled(mae_engine* e) {
  if (e->side_mode < 0) {
    const char* v = getenv("MAE_WGRAD_STREAM");  // opt-in: measured neutral on MI355X (the GEMMs already fill every CU)
    e->side_mode = (v && v[0] == '1') ? 1 : 0;
    if (e->side_mode == 1 && hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) e->side_mode = 0;
  }
  return e->side_mode == 1;
}
static hipEvent_t next_event(mae_engine* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}
// the main stream must not overwrite a buffer the side stream may still be reading
static int await_side(mae_engine* e, int tag, hipStream_t s) {
  if (e->pending[tag]) {
    MAE_HIP(hipStreamWaitEvent(s, e->pending[tag], 0));
    e->pending[tag] = nullptr;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// helpers bound to one call
// ---------------------------------------------------------------------------------------------------
struct Ctx {
  mae_engine* e;
  const float* params;
  const char* wcache;  // bf16 copies (null in fp32 mode)
  float* grads;
  char* ws;
  hipStream_t s;
  int act;
  int64_t as;
  void* const* ready = nullptr;  // hipEvent_t per gradient-ready point (entries may be null), or null
  bool fwd_only = false;         // no backward will follow (I-JEPA target encoder): the MLP saves no derivative
  const float* P(int i) const { return params + e->params[i].offset; }
  float* Gp(int i) const { return grads + e->params[i].offset; }
  // GEMM operand view of weight i: (out, in) row-major in the activation dtype
  const void* W(int i) const {
    return act == MAE_BF16 ? (const void*)(wcache + 2 * e->params[i].offset) : (const void*)(params + e->params[i].offset);
  }
  // transposed (in, out) bf16 copy (bf16 mode only)
  const void* WT(int i) const { return wcache + 2 * (e->trainable_elems + e->params[i].t_off); }
  template <class T = void> T* buf(int64_t off) const { return reinterpret_cast<T*>(ws + off); }
};

// algorithmic HBM bytes of one Linear launch: both operands, every output (the GELU epilogues write two) and the (M,N) side input
// of the RESID (fp32 residual) / MUL / DGELU (output-typed) epilogues
static double gemm_bytes(int64_t M, int N, int K, int64_t as, int64_t os, int mode = MAE_EPI_NONE) {
  const int64_t outs = (mode == MAE_EPI_GELU || mode == MAE_EPI_GELU_GRAD) ? 2 : 1;
  const int64_t side = mode == MAE_EPI_RESID ? 4 : (mode == MAE_EPI_MUL || mode == MAE_EPI_DGELU) ? os : 0;
  return (double)(M * K * as + (int64_t)N * K * as + M * N * (outs * os + side));
}

static int linear(const Ctx& c, const void* A, int wi, int bi, int64_t M, int N, int K, int mode, int out_dt, void* out, void* out2,
                  const void* aux) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  Epi ep; ep.mode = mode; ep.bias = bi >= 0 ? c.P(bi) : nullptr; ep.aux = aux; ep.out = out; ep.out2 = out2; ep.out_dt = out_dt;
  RUN(TK_LINEAR, 2.0 * M * N * K, gemm_bytes(M, N, K, c.as, (int64_t)dtype_size(out_dt), mode), launch_linear_fwd(A, c.W(wi), M, N, K, c.act, ep, s));
  return 0;
}

// dX[M,K] = dY[M,N] * W[N,K]
static int dgrad(const Ctx& c, const void* dY, int wi, int64_t M, int N, int K, int mode, void* out, const void* aux) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  Epi ep; ep.mode = mode; ep.aux = aux; ep.out = out; ep.out_dt = c.act;
  if (c.act == MAE_BF16) {
    RUN(TK_LINEAR, 2.0 * M * N * K, gemm_bytes(M, K, N, 2, 2, mode), launch_linear_fwd(dY, c.WT(wi), M, K, N, MAE_BF16, ep, s));
  } else {
    RUN(TK_LINEAR, 2.0 * M * N * K, gemm_bytes(M, K, N, 4, 4, mode), launch_linear_dgrad(dY, c.W(wi), M, N, K, MAE_F32, ep, s));
  }
  return 0;
}

static int wgrad(const Ctx& c, const Plan& pl, const void* dY, const void* A, int64_t M, int N, int K, int wi, int bi, int tag) {
  mae_engine* e = c.e;
  hipStream_t s = c.s;
  const bool side = side_enabled(e);
  if (side) {  // fork: the side stream waits for everything enqueued so far (the producer of dY)
    hipEvent_t ready = next_event(e);
    MAE_REQUIRE(ready, "wgrad: cannot create an event");
    MAE_HIP(hipEventRecord(ready, c.s));
    MAE_HIP(hipStreamWaitEvent(e->side, ready, 0));
    s = e->side;
  }
  RUN(TK_WGRAD, 2.0 * M * N * K, (double)(M * (N + K) * c.as + (int64_t)N * K * 4),
      launch_linear_wgrad(dY, A, M, N, K, c.act, c.Gp(wi), bi >= 0 ? c.Gp(bi) : nullptr, c.buf<>(pl.wgrad_scratch), s));
  if (side) {
    hipEvent_t done = next_event(e);
    MAE_REQUIRE(done, "wgrad: cannot create an event");
    MAE_HIP(hipEventRecord(done, e->side));
    e->pending[tag] = done;
  }
  return 0;
}

// two weight gradients over the same rows in one launch (k_gemm.hip: launch_linear_wgrad_pair); main stream only
static int wgrad_pair(const Ctx& c, const Plan& pl, int64_t M, const void* dY0, const void* A0, int N0, int K0, int w0, int b0,
                      const void* dY1, const void* A1, int N1, int K1, int w1, int b1) {
  mae_engine* e = c.e;
  hipStream_t s = c.s;
  RUN(TK_WGRAD, 2.0 * M * ((double)N0 * K0 + (double)N1 * K1),
      (double)(M * (int64_t)(N0 + K0 + N1 + K1) * c.as + ((int64_t)N0 * K0 + (int64_t)N1 * K1) * 4),
      launch_linear_wgrad_pair(dY0, A0, N0, K0, c.Gp(w0), c.Gp(b0), dY1, A1, N1, K1, c.Gp(w1), c.Gp(b1), M, c.act, c.buf<>(pl.wgrad_scratch), s));
  return 0;
}

// join: everything on the side stream is finished before the main stream continues
static int join_side(const Ctx& c) {
  mae_engine* e = c.e;
  if (e->side_mode == 1) {
    hipEvent_t done = next_event(e);
    MAE_REQUIRE(done, "cannot create an event");
    MAE_HIP(hipEventRecord(done, e->side));
    MAE_HIP(hipStreamWaitEvent(c.s, done, 0));
    for (auto& p : e->pending) p = nullptr;
  }
  return 0;
}

// One transformer block.  The residual adds are done by the LayerNorm kernels: LN1 first forms this block's input
// x_in = x_prev + prev_branch (the previous block's MLP output; none for the first block), LN2 forms x_mid = x_in + proj(att).
// The block's own MLP output is left in pl.branch_b for whoever normalises next.
static int block_forward(const Ctx& c, const Plan& pl, const BlockRefs& r, const LayerBufs& b, int64_t M, int d, int heads, int Bn, int T,
                         int64_t x_prev, bool add_prev, int64_t x_in) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  const int hd = d / heads, hid = e->mlp * d;
  const float eps = 1e-6f;
  if (add_prev) {
    RUN(TK_LN_FWD, 0, M * d * (8 + 2 * c.as), launch_layernorm_fwd(c.buf<float>(x_prev), c.buf<>(pl.branch_b), c.buf<float>(x_in), nullptr, c.P(r.ln1_w), c.P(r.ln1_b), eps, M, d, c.act, c.buf<>(b.ln1), c.buf<float>(b.mean1), c.buf<float>(b.rstd1), s));
  } else {
    RUN(TK_LN_FWD, 0, M * d * (4 + c.as), launch_layernorm_fwd(c.buf<float>(x_in), nullptr, nullptr, nullptr, c.P(r.ln1_w), c.P(r.ln1_b), eps, M, d, c.act, c.buf<>(b.ln1), c.buf<float>(b.mean1), c.buf<float>(b.rstd1), s));
  }
  MAE_TRY(linear(c, c.buf<>(b.ln1), r.qkv_w, r.qkv_b, M, 3 * d, d, MAE_EPI_NONE, c.act, c.buf<>(b.qkv), nullptr, nullptr));
  RUN(TK_ATTN_FWD, 4.0 * Bn * heads * (double)T * T * hd, M * 4 * d * c.as, launch_attention_fwd(c.buf<>(b.qkv), Bn, T, heads, hd, c.act, c.buf<>(b.att), c.buf<float>(b.lse), s));
  MAE_TRY(linear(c, c.buf<>(b.att), r.proj_w, r.proj_b, M, d, d, MAE_EPI_NONE, c.act, c.buf<>(pl.branch_a), nullptr, nullptr));
  RUN(TK_LN_FWD, 0, M * d * (8 + 2 * c.as), launch_layernorm_fwd(c.buf<float>(x_in), c.buf<>(pl.branch_a), c.buf<float>(b.x_mid), nullptr, c.P(r.ln2_w), c.P(r.ln2_b), eps, M, d, c.act, c.buf<>(b.ln2), c.buf<float>(b.mean2), c.buf<float>(b.rstd2), s));
  if (c.fwd_only)
    MAE_TRY(linear(c, c.buf<>(b.ln2), r.fc1_w, r.fc1_b, M, hid, d, MAE_EPI_GELU_ACT, c.act, c.buf<>(b.fc1_act), nullptr, nullptr));
  else
    MAE_TRY(linear(c, c.buf<>(b.ln2), r.fc1_w, r.fc1_b, M, hid, d, MAE_EPI_GELU_GRAD, c.act, c.buf<>(b.fc1_pre), c.buf<>(b.fc1_act), nullptr));  // fc1_pre holds gelu'(pre)
  MAE_TRY(linear(c, c.buf<>(b.fc1_act), r.fc2_w, r.fc2_b, M, d, hid, MAE_EPI_NONE, c.act, c.buf<>(pl.branch_b), nullptr, nullptr));
  return 0;
}

// partial-sum slot of the next LayerNorm backward of this pass (its reduction is queued in e->ln_tab)
static float* ln_slot(const Ctx& c, const Plan& pl) { return c.buf<float>(pl.ln_partial + (int64_t)(c.e->ln_tab.n % PartialsTable::MAX) * pl.ln_partial_stride); }

// in: dres (fp32) / dres_c (act copy) = gradient w.r.t. the block output; out: same buffers = gradient w.r.t. the block input
static int block_backward(const Ctx& c, const Plan& pl, const BlockRefs& r, const LayerBufs& b, int64_t M, int d, int heads, int Bn, int T,
                          int64_t x_in) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  const int hd = d / heads, hid = e->mlp * d;
  float* dres = c.buf<float>(pl.dres);
  void* dres_c = c.buf<>(pl.dres_c);
  if (!side_enabled(e)) {
    // Each branch's two weight gradients go out as ONE launch, placed where both of their dY operands exist and before the
    // LayerNorm backward that rewrites dres_c (fc2 / proj read it).  Same dependencies as the separate launches below.
    MAE_TRY(dgrad(c, dres_c, r.fc2_w, M, d, hid, MAE_EPI_MUL, c.buf<>(pl.d_hidden), c.buf<>(b.fc1_pre)));
    MAE_TRY(wgrad_pair(c, pl, M, dres_c, c.buf<>(b.fc1_act), d, hid, r.fc2_w, r.fc2_b, c.buf<>(pl.d_hidden), c.buf<>(b.ln2), hid, d, r.fc1_w, r.fc1_b));
    MAE_TRY(dgrad(c, c.buf<>(pl.d_hidden), r.fc1_w, M, hid, d, MAE_EPI_NONE, c.buf<>(pl.d_ln), nullptr));
    RUN(TK_LN_BWD, 0, M * d * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_ln), c.act, c.buf<float>(b.x_mid), nullptr, c.P(r.ln2_w), c.buf<float>(b.mean2), c.buf<float>(b.rstd2), M, d, 1, dres, dres_c, c.Gp(r.ln2_w), c.Gp(r.ln2_b), ln_slot(c, pl), s, &e->ln_tab));
    MAE_TRY(dgrad(c, dres_c, r.proj_w, M, d, d, MAE_EPI_NONE, c.buf<>(pl.d_att), nullptr));
    RUN(TK_ATTN_BWD, 10.0 * Bn * heads * (double)T * T * hd, M * 9 * d * c.as, launch_attention_bwd(c.buf<>(b.qkv), c.buf<>(b.att), c.buf<>(pl.d_att), c.buf<float>(b.lse), Bn, T, heads, hd, c.act, c.buf<>(pl.d_qkv), s));
    MAE_TRY(wgrad_pair(c, pl, M, dres_c, c.buf<>(b.att), d, d, r.proj_w, r.proj_b, c.buf<>(pl.d_qkv), c.buf<>(b.ln1), 3 * d, d, r.qkv_w, r.qkv_b));
    MAE_TRY(dgrad(c, c.buf<>(pl.d_qkv), r.qkv_w, M, 3 * d, d, MAE_EPI_NONE, c.buf<>(pl.d_ln), nullptr));
    RUN(TK_LN_BWD, 0, M * d * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_ln), c.act, c.buf<float>(x_in), nullptr, c.P(r.ln1_w), c.buf<float>(b.mean1), c.buf<float>(b.rstd1), M, d, 1, dres, dres_c, c.Gp(r.ln1_w), c.Gp(r.ln1_b), ln_slot(c, pl), s, &e->ln_tab));
    return 0;
  }
  // MLP branch (weight gradients on the side stream, MAE_WGRAD_STREAM=1: separate launches)
  MAE_TRY(wgrad(c, pl, dres_c, c.buf<>(b.fc1_act), M, d, hid, r.fc2_w, r.fc2_b, DEP_DRESC));
  MAE_TRY(await_side(e, DEP_HIDDEN, s));  // the previous block's fc1 wgrad reads d_hidden
  MAE_TRY(dgrad(c, dres_c, r.fc2_w, M, d, hid, MAE_EPI_MUL, c.buf<>(pl.d_hidden), c.buf<>(b.fc1_pre)));
  MAE_TRY(wgrad(c, pl, c.buf<>(pl.d_hidden), c.buf<>(b.ln2), M, hid, d, r.fc1_w, r.fc1_b, DEP_HIDDEN));
  MAE_TRY(dgrad(c, c.buf<>(pl.d_hidden), r.fc1_w, M, hid, d, MAE_EPI_NONE, c.buf<>(pl.d_ln), nullptr));
  MAE_TRY(await_side(e, DEP_DRESC, s));  // fc2 wgrad reads dres_c, which this kernel rewrites
  RUN(TK_LN_BWD, 0, M * d * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_ln), c.act, c.buf<float>(b.x_mid), nullptr, c.P(r.ln2_w), c.buf<float>(b.mean2), c.buf<float>(b.rstd2), M, d, 1, dres, dres_c, c.Gp(r.ln2_w), c.Gp(r.ln2_b), ln_slot(c, pl), s, &e->ln_tab));
  // attention branch
  MAE_TRY(wgrad(c, pl, dres_c, c.buf<>(b.att), M, d, d, r.proj_w, r.proj_b, DEP_DRESC));
  MAE_TRY(dgrad(c, dres_c, r.proj_w, M, d, d, MAE_EPI_NONE, c.buf<>(pl.d_att), nullptr));
  MAE_TRY(await_side(e, DEP_QKV, s));  // the previous block's qkv wgrad reads d_qkv
  RUN(TK_ATTN_BWD, 10.0 * Bn * heads * (double)T * T * hd, M * 9 * d * c.as, launch_attention_bwd(c.buf<>(b.qkv), c.buf<>(b.att), c.buf<>(pl.d_att), c.buf<float>(b.lse), Bn, T, heads, hd, c.act, c.buf<>(pl.d_qkv), s));
  MAE_TRY(wgrad(c, pl, c.buf<>(pl.d_qkv), c.buf<>(b.ln1), M, 3 * d, d, r.qkv_w, r.qkv_b, DEP_QKV));
  MAE_TRY(dgrad(c, c.buf<>(pl.d_qkv), r.qkv_w, M, 3 * d, d, MAE_EPI_NONE, c.buf<>(pl.d_ln), nullptr));
  MAE_TRY(await_side(e, DEP_DRESC, s));  // proj wgrad reads dres_c
  RUN(TK_LN_BWD, 0, M * d * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_ln), c.act, c.buf<float>(x_in), nullptr, c.P(r.ln1_w), c.buf<float>(b.mean1), c.buf<float>(b.rstd1), M, d, 1, dres, dres_c, c.Gp(r.ln1_w), c.Gp(r.ln1_b), ln_slot(c, pl), s, &e->ln_tab));
  return 0;
}

static int check_call(const mae_engine* e, const void* params, const void* wcache, int B, int k, const void* ws, int64_t ws_bytes,
                      Plan* pl, const char* who) {
  MAE_REQUIRE(e, "%s: null engine", who);
  MAE_REQUIRE(params && ws, "%s: null params/workspace", who);
  MAE_REQUIRE(e->act == MAE_F32 || wcache, "%s: bf16 engine needs the weight cache", who);
  MAE_REQUIRE(B > 0 && k >= 1 && k <= e->L, "%s: batch %d / num_keep %d out of range (L = %d)", who, B, k, e->L);
  MAE_REQUIRE((int64_t)B * e->L * std::max(3 * e->D, e->mlp * std::max(e->D, e->Dd)) < (1ll << 40), "%s: batch too large", who);
  *pl = make_plan(e, B, k);
  MAE_REQUIRE(ws_bytes >= pl->total, "%s: workspace too small (%lld < %lld bytes)", who, (long long)ws_bytes, (long long)pl->total);
  MAE_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)params & 15) == 0, "%s: workspace must be 256-byte aligned, params 16-byte", who);
  return 0;
}

static int check_image_dtype(int dt, const char* who) {
  MAE_REQUIRE(dt == MAE_F32 || dt == MAE_U8, "%s: image_dtype must be MAE_F32 or MAE_U8 (got %d)", who, dt);
  return 0;
}

static int forward_encoder_impl(const Ctx& c, const Plan& pl, const void* images, int img_dt, float* x_encoded_out) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  const int32_t* keep32 = c.buf<int32_t>(pl.keep32);
  if (img_dt == MAE_U8)  // whole image rows through LDS: every pixel byte is fetched once
    RUN(TK_DATA, 0, (int64_t)pl.B * e->C * e->img * e->img + pl.Me * e->P * c.as, launch_gather_patches_u8((const uint8_t*)images, keep32, pl.B, pl.k, e->C, e->img, e->p, c.act, c.buf<>(pl.patchA), s));
  else
    RUN(TK_DATA, 0, pl.Me * e->P * (4 + c.as), launch_gather_patches((const float*)images, keep32, pl.B, pl.k, e->C, e->img, e->p, c.act, c.buf<>(pl.patchA), s));
  MAE_TRY(linear(c, c.buf<>(pl.patchA), e->i_patch_w, e->i_patch_b, pl.Me, e->D, e->P, MAE_EPI_NONE, MAE_F32, c.buf<>(pl.enc_x[0]), nullptr, nullptr));
  RUN(TK_DATA, 0, pl.Me * e->D * 12, launch_assemble_visible(c.buf<float>(pl.enc_x[0]), keep32, c.P(e->i_cls), c.P(e->i_pos), pl.Me, e->D, s));
  for (int i = 0; i < e->depth; ++i)
    MAE_TRY(block_forward(c, pl, e->enc[i], pl.enc[i], pl.Me, e->D, e->H, pl.B, pl.k, i ? pl.enc[i - 1].x_mid : 0, i > 0, pl.enc_x[i]));
  RUN(TK_LN_FWD, 0, pl.Me * e->D * (8 + 2 * c.as), launch_layernorm_fwd(c.buf<float>(pl.enc[e->depth - 1].x_mid), c.buf<>(pl.branch_b), c.buf<float>(pl.enc_x[e->depth]), nullptr, c.P(e->i_norm_w), c.P(e->i_norm_b), 1e-6f, pl.Me, e->D, c.act, c.buf<>(pl.enc_norm), c.buf<float>(pl.enc_mean), c.buf<float>(pl.enc_rstd), s));
  if (x_encoded_out) {
    if (c.act == MAE_F32) MAE_HIP(hipMemcpyAsync(x_encoded_out, c.buf<>(pl.enc_norm), (size_t)pl.Me * e->D * 4, hipMemcpyDeviceToDevice, s));
    else RUN(TK_LN_FWD, 0, pl.Me * e->D * 8, launch_layernorm_fwd(c.buf<float>(pl.enc_x[e->depth]), nullptr, nullptr, nullptr, c.P(e->i_norm_w), c.P(e->i_norm_b), 1e-6f, pl.Me, e->D, MAE_F32, x_encoded_out, c.buf<float>(pl.enc_mean), c.buf<float>(pl.enc_rstd), s));
  }
  return 0;
}

// I-JEPA predictor input (no counterpart in the reference; spec in DESIGN.md): per image nblk sequences
// [k context tokens | m mask tokens of target block i], position rows gathered by token id
struct JepaSeq {
  const int32_t* ctx32;  // (B, k) context token ids
  const int32_t* tgt32;  // (B, nblk, m) target token ids
  int nblk;
};

// needs keep32 / mask32 already in the workspace (MAE) or the token lists of `jp` (I-JEPA)
static int forward_decoder_impl(const Ctx& c, const Plan& pl, float* x_pred_out, const JepaSeq* jp = nullptr) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  MAE_REQUIRE(pl.m > 0, "forward_decoder: nothing is masked (num_keep == sequence_length)");
  MAE_TRY(linear(c, c.buf<>(pl.enc_norm), e->i_de_w, e->i_de_b, pl.Me, e->Dd, e->D, MAE_EPI_NONE, c.act, c.buf<>(pl.xdec), nullptr, nullptr));
  if (jp) {
    MAE_TRY(launch_build_tail_row_map(pl.dec_B, pl.dec_T, pl.m, c.buf<int32_t>(pl.pred_rows), s));
    RUN(TK_DATA, 0, pl.Md * e->Dd * 4 + pl.Me * e->Dd * c.as, launch_predictor_assemble(c.buf<>(pl.xdec), c.act, jp->ctx32, jp->tgt32, c.P(e->i_dec_mask), c.P(e->i_dec_pos), pl.B, pl.k, jp->nblk, pl.m, e->L, e->Dd, c.buf<float>(pl.dec_x[0]), s));
  } else {
    MAE_TRY(launch_build_inverse(c.buf<int32_t>(pl.keep32), pl.B, pl.k, e->L, c.buf<int32_t>(pl.inv), s));
    MAE_TRY(launch_build_row_map(c.buf<int32_t>(pl.mask32), pl.B, pl.m, e->L, c.buf<int32_t>(pl.pred_rows), s));
    RUN(TK_DATA, 0, pl.Md * e->Dd * 4 + pl.Me * e->Dd * c.as, launch_decoder_assemble(c.buf<>(pl.xdec), c.act, c.buf<int32_t>(pl.inv), c.P(e->i_dec_mask), c.P(e->i_dec_pos), pl.B, pl.k, e->L, e->Dd, c.buf<float>(pl.dec_x[0]), s));
  }
  for (int i = 0; i < e->dd; ++i)
    MAE_TRY(block_forward(c, pl, e->dec[i], pl.dec[i], pl.Md, e->Dd, e->Hd, pl.dec_B, pl.dec_T, i ? pl.dec[i - 1].x_mid : 0, i > 0, pl.dec_x[i]));
  // decoder_norm on the masked rows only; the residual add of the last MLP branch is done for exactly those rows
  RUN(TK_LN_FWD, 0, pl.Mp * e->Dd * (8 + 2 * c.as), launch_layernorm_fwd(c.buf<float>(pl.dec[e->dd - 1].x_mid), c.buf<>(pl.branch_b), c.buf<float>(pl.dec_x[e->dd]), c.buf<int32_t>(pl.pred_rows), c.P(e->i_dn_w), c.P(e->i_dn_b), 1e-6f, pl.Mp, e->Dd, c.act, c.buf<>(pl.dec_norm), c.buf<float>(pl.dec_mean), c.buf<float>(pl.dec_rstd), s));
  MAE_TRY(linear(c, c.buf<>(pl.dec_norm), e->i_pred_w, e->i_pred_b, pl.Mp, e->PO, e->Dd, MAE_EPI_NONE, MAE_F32, x_pred_out ? (void*)x_pred_out : c.buf<>(pl.pred), nullptr, nullptr));
  return 0;
}

// Gradient-ready points (data-parallel overlap): point 0 = the whole decoder range of the gradient arena is final, point
// i (1 .. depth-1) = encoder block depth-i and everything behind it, point depth = everything.  Reaching a point that the
// caller gave an event for first retires what was deferred (LayerNorm dgamma / dbeta second stages, side-stream wgrads).
static int reach_point(const Ctx& c, int j) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  if (!c.ready || !c.ready[j]) return 0;
  if (e->ln_tab.n > 0) {
    RUN(TK_LN_BWD, 0, 0, launch_sum_partials_many(e->ln_tab, s));
    e->ln_tab.n = 0;  // slots are reused in stream order behind the reduction just enqueued
  }
  MAE_TRY(join_side(c));
  MAE_HIP(hipEventRecord((hipEvent_t)c.ready[j], s));
  return 0;
}

static void backward_begin(const Ctx& c) {
  mae_engine* e = c.e;
  e->ev_used = 0;
  for (auto& p : e->pending) p = nullptr;
  e->ln_tab.n = 0;
}
static int backward_end(const Ctx& c) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  if (e->ln_tab.n > 0) RUN(TK_LN_BWD, 0, 0, launch_sum_partials_many(e->ln_tab, s));  // dgamma / dbeta of every LayerNorm not yet retired: one launch
  e->ln_tab.n = 0;
  return join_side(c);
}

// Decoder half: dpred (act dtype) must already sit in the workspace; leaves d(x_encoded) (act dtype) in pl.d_ln and
// every decoder gradient in its arena range [offset(decoder.mask_token), trainable_elems).
static int backward_decoder_impl(const Ctx& c, const Plan& pl, const JepaSeq* jp = nullptr) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  float* dres = c.buf<float>(pl.dres);
  void* dres_c = c.buf<>(pl.dres_c);
  // prediction head
  MAE_TRY(wgrad(c, pl, c.buf<>(pl.dpred), c.buf<>(pl.dec_norm), pl.Mp, e->PO, e->Dd, e->i_pred_w, e->i_pred_b, DEP_MISC));
  MAE_TRY(dgrad(c, c.buf<>(pl.dpred), e->i_pred_w, pl.Mp, e->PO, e->Dd, MAE_EPI_NONE, c.buf<>(pl.d_decn), nullptr));
  // decoder_norm over the masked rows only: every other row of the residual gradient is zero
  MAE_TRY(await_side(e, DEP_DRESC, s));
  RUN(TK_DATA, 0, (pl.Md - pl.Mp) * e->Dd * (4 + c.as), launch_zero_unpredicted_rows(jp ? nullptr : c.buf<int32_t>(pl.inv), pl.Md, pl.dec_T, pl.m, e->Dd, c.act, dres, dres_c, s));
  RUN(TK_LN_BWD, 0, pl.Mp * e->Dd * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_decn), c.act, c.buf<float>(pl.dec_x[e->dd]), c.buf<int32_t>(pl.pred_rows), c.P(e->i_dn_w), c.buf<float>(pl.dec_mean), c.buf<float>(pl.dec_rstd), pl.Mp, e->Dd, 0, dres, dres_c, c.Gp(e->i_dn_w), c.Gp(e->i_dn_b), ln_slot(c, pl), s, &e->ln_tab));
  for (int i = e->dd - 1; i >= 0; --i)
    MAE_TRY(block_backward(c, pl, e->dec[i], pl.dec[i], pl.Md, e->Dd, e->Hd, pl.dec_B, pl.dec_T, pl.dec_x[i]));
  if (jp)
    RUN(TK_DATA, 0, pl.Md * e->Dd * 4 + pl.Me * e->Dd * c.as, launch_predictor_assemble_bwd(dres, pl.B, pl.k, jp->nblk, pl.m, e->Dd, c.act, c.buf<>(pl.d_xdec), c.Gp(e->i_dec_mask), c.buf<float>(pl.split_partial), s));
  else
    RUN(TK_DATA, 0, pl.Md * e->Dd * 4 + pl.Me * e->Dd * c.as, launch_decoder_assemble_bwd(dres, c.buf<int32_t>(pl.inv), c.buf<int32_t>(pl.keep32), pl.B, pl.k, e->L, e->Dd, c.act, c.buf<>(pl.d_xdec), c.Gp(e->i_dec_mask), c.buf<float>(pl.split_partial), s));
  // decoder_embed
  MAE_TRY(wgrad(c, pl, c.buf<>(pl.d_xdec), c.buf<>(pl.enc_norm), pl.Me, e->Dd, e->D, e->i_de_w, e->i_de_b, DEP_MISC));
  MAE_TRY(dgrad(c, c.buf<>(pl.d_xdec), e->i_de_w, pl.Me, e->Dd, e->D, MAE_EPI_NONE, c.buf<>(pl.d_ln), nullptr));
  return reach_point(c, 0);
}

// Encoder half: d(x_encoded) (act dtype) sits in pl.d_ln; writes the arena range [0, offset(decoder.mask_token)).
static int backward_encoder_impl(const Ctx& c, const Plan& pl) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  float* dres = c.buf<float>(pl.dres);
  void* dres_c = c.buf<>(pl.dres_c);
  MAE_TRY(await_side(e, DEP_DRESC, s));
  RUN(TK_LN_BWD, 0, pl.Me * e->D * (12 + 2 * c.as), launch_layernorm_bwd(c.buf<>(pl.d_ln), c.act, c.buf<float>(pl.enc_x[e->depth]), nullptr, c.P(e->i_norm_w), c.buf<float>(pl.enc_mean), c.buf<float>(pl.enc_rstd), pl.Me, e->D, 0, dres, dres_c, c.Gp(e->i_norm_w), c.Gp(e->i_norm_b), ln_slot(c, pl), s, &e->ln_tab));
  for (int i = e->depth - 1; i >= 0; --i) {
    MAE_TRY(block_backward(c, pl, e->enc[i], pl.enc[i], pl.Me, e->D, e->H, pl.B, pl.k, pl.enc_x[i]));
    if (i > 0) MAE_TRY(reach_point(c, e->depth - i));
  }
  // token assembly and patch projection
  RUN(TK_DATA, 0, pl.Me * e->D * (4 + c.as), launch_visible_grad_split(dres, c.buf<int32_t>(pl.keep32), pl.Me, e->D, c.act, c.buf<>(pl.dtok), c.Gp(e->i_cls), c.buf<float>(pl.split_partial), s));
  MAE_TRY(wgrad(c, pl, c.buf<>(pl.dtok), c.buf<>(pl.patchA), pl.Me, e->D, e->P, e->i_patch_w, e->i_patch_b, DEP_MISC));
  return 0;
}

static int backward_impl(const Ctx& c, const Plan& pl, const float* d_x_encoded_extra = nullptr, const JepaSeq* jp = nullptr) {
  mae_engine* e = c.e; hipStream_t s = c.s;
  backward_begin(c);
  MAE_TRY(backward_decoder_impl(c, pl, jp));
  if (d_x_encoded_extra) MAE_TRY(launch_add_into(d_x_encoded_extra, c.buf<>(pl.d_ln), c.act, pl.Me * e->D, s));
  MAE_TRY(backward_encoder_impl(c, pl));
  MAE_TRY(backward_end(c));
  if (c.ready && c.ready[e->depth]) MAE_HIP(hipEventRecord((hipEvent_t)c.ready[e->depth], s));
  return 0;
}

}  // namespace mae

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" const char* mae_last_error(void) { return mae::g_err; }
extern "C" int mae_abi_version(void) { return MAE_ABI_VERSION; }

extern "C" int mae_engine_create(const mae_config_t* cfg, mae_engine_t** out) {
  MAE_REQUIRE(cfg && out, "mae_engine_create: null argument");
  MAE_REQUIRE(cfg->image_size > 0 && cfg->patch_size > 0 && cfg->image_size % cfg->patch_size == 0,
              "image_size %d must be a positive multiple of patch_size %d", cfg->image_size, cfg->patch_size);
  MAE_REQUIRE(cfg->in_chans > 0 && cfg->embed_dim > 0 && cfg->depth > 0 && cfg->decoder_embed_dim > 0 && cfg->decoder_depth > 0,
              "mae_engine_create: non-positive model dimension");
  MAE_REQUIRE(cfg->num_heads > 0 && cfg->embed_dim % cfg->num_heads == 0, "dim should be divisible by num_heads (%d %% %d)", cfg->embed_dim, cfg->num_heads);
  MAE_REQUIRE(cfg->decoder_num_heads > 0 && cfg->decoder_embed_dim % cfg->decoder_num_heads == 0,
              "decoder dim should be divisible by decoder_num_heads (%d %% %d)", cfg->decoder_embed_dim, cfg->decoder_num_heads);
  MAE_REQUIRE(cfg->embed_dim % 4 == 0 && cfg->decoder_embed_dim % 4 == 0 && cfg->embed_dim <= 1024 && cfg->decoder_embed_dim <= 1024,
              "embed dims must be multiples of 4 and <= 1024");
  MAE_REQUIRE(cfg->act_dtype == MAE_F32 || cfg->act_dtype == MAE_BF16, "act_dtype must be MAE_F32 or MAE_BF16");
  for (int hd : {cfg->embed_dim / cfg->num_heads, cfg->decoder_embed_dim / cfg->decoder_num_heads})
    MAE_REQUIRE(hd == 16 || hd == 24 || hd == 32 || hd == 48 || hd == 64,
                "head dim %d is not supported by the attention kernels (16, 24, 32, 48 or 64): choose num_heads accordingly", hd);
  const int P = cfg->patch_size * cfg->patch_size * cfg->in_chans;
  MAE_REQUIRE(P % 4 == 0, "patch_size^2 * in_chans must be a multiple of 4");
  MAE_REQUIRE(cfg->pred_dim >= 0 && cfg->pred_dim % 4 == 0, "pred_dim must be 0 (pixels) or a multiple of 4");
  mae_engine* e = new mae_engine();
  e->cfg = *cfg;
  e->act = cfg->act_dtype;
  e->D = cfg->embed_dim; e->depth = cfg->depth; e->H = cfg->num_heads;
  e->Dd = cfg->decoder_embed_dim; e->dd = cfg->decoder_depth; e->Hd = cfg->decoder_num_heads;
  e->C = cfg->in_chans; e->img = cfg->image_size; e->p = cfg->patch_size;
  e->G = e->img / e->p; e->L = e->G * e->G + 1; e->P = P;
  e->mlp = cfg->mlp_ratio > 0 ? cfg->mlp_ratio : 4;
  e->PO = cfg->pred_dim > 0 ? cfg->pred_dim : P;
  const int T = MAE_PARAM_TRAINABLE, M = MAE_PARAM_TRAINABLE | MAE_PARAM_MATRIX;
  // state_dict order
  e->i_enc_mask = add_param(e, "encoder.mask_token", {1, 1, e->D}, MAE_PARAM_UNUSED);
  e->i_cls = add_param(e, "encoder.vit.cls_token", {1, 1, e->D}, T);
  e->i_pos = add_param(e, "encoder.vit.pos_embed", {1, e->L, e->D}, MAE_PARAM_FROZEN);
  e->i_patch_w = add_param(e, "encoder.vit.patch_embed.proj.weight", {e->D, e->C, e->p, e->p}, M);
  e->i_patch_b = add_param(e, "encoder.vit.patch_embed.proj.bias", {e->D}, T);
  for (int i = 0; i < e->depth; ++i) e->enc.push_back(add_block(e, "encoder.vit.blocks." + std::to_string(i), e->D, e->mlp));
  e->i_norm_w = add_param(e, "encoder.vit.norm.weight", {e->D}, T);
  e->i_norm_b = add_param(e, "encoder.vit.norm.bias", {e->D}, T);
  e->i_dec_mask = add_param(e, "decoder.mask_token", {1, 1, e->Dd}, T);
  e->i_dec_pos = add_param(e, "decoder.decoder_pos_embed", {1, e->L, e->Dd}, MAE_PARAM_FROZEN);
  e->i_de_w = add_param(e, "decoder.decoder_embed.weight", {e->Dd, e->D}, M);
  e->i_de_b = add_param(e, "decoder.decoder_embed.bias", {e->Dd}, T);
  for (int i = 0; i < e->dd; ++i) e->dec.push_back(add_block(e, "decoder.decoder_blocks." + std::to_string(i), e->Dd, e->mlp));
  e->i_dn_w = add_param(e, "decoder.decoder_norm.weight", {e->Dd}, T);
  e->i_dn_b = add_param(e, "decoder.decoder_norm.bias", {e->Dd}, T);
  e->i_pred_w = add_param(e, "decoder.decoder_pred.weight", {e->PO, e->Dd}, M);
  e->i_pred_b = add_param(e, "decoder.decoder_pred.bias", {e->PO}, T);
  // arena: trainable-on-path first, then frozen / unused; 64-element alignment
  int64_t off = 0, toff = 0;
  for (auto& pi : e->params)
    if (pi.flags & MAE_PARAM_TRAINABLE) {
      pi.offset = off; off += round_up(pi.numel, 64);
      if (pi.flags & MAE_PARAM_MATRIX) { pi.t_off = toff; toff += round_up(pi.numel, 64); }
    }
  e->trainable_elems = off;
  for (auto& pi : e->params)
    if (!(pi.flags & MAE_PARAM_TRAINABLE)) { pi.offset = off; off += round_up(pi.numel, 64); }
  e->arena_elems = off;
  e->trans_elems = toff;
  *out = e;
  return 0;
}

extern "C" void mae_engine_destroy(mae_engine_t* e) {
  if (!e) return;
  for (auto& t : e->timers)
    for (auto& pr : t.ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->side) (void)hipStreamDestroy(e->side);
  delete e;
}

extern "C" int64_t mae_engine_num_params(const mae_engine_t* e) { return e ? (int64_t)e->params.size() : 0; }
extern "C" int64_t mae_engine_arena_elems(const mae_engine_t* e) { return e ? e->arena_elems : 0; }
extern "C" int64_t mae_engine_trainable_elems(const mae_engine_t* e) { return e ? e->trainable_elems : 0; }

extern "C" int mae_engine_param_info(const mae_engine_t* e, int64_t index, const char** name, int64_t* offset, int64_t* numel,
                                     int32_t* ndim, int64_t shape[4], int32_t* flags) {
  MAE_REQUIRE(e && index >= 0 && index < (int64_t)e->params.size(), "mae_engine_param_info: index out of range");
  const ParamInfo& pi = e->params[index];
  if (name) *name = pi.name.c_str();
  if (offset) *offset = pi.offset;
  if (numel) *numel = pi.numel;
  if (ndim) *ndim = pi.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = pi.shape[i];
  if (flags) *flags = pi.flags;
  return 0;
}

extern "C" int64_t mae_engine_workspace_bytes(const mae_engine_t* e, int32_t batch, int32_t num_keep) {
  if (!e || batch <= 0 || num_keep < 1 || num_keep > e->L) return -1;
  return make_plan(e, batch, num_keep).total;
}

extern "C" int64_t mae_engine_wcache_bytes(const mae_engine_t* e) {
  if (!e || e->act != MAE_BF16) return 256;
  return round_up(2 * (e->trainable_elems + e->trans_elems), 256);
}

static int refresh_transposed(mae_engine* e, const float* params, void* wcache, hipStream_t s) {
  bf16* tbase = reinterpret_cast<bf16*>(wcache) + e->trainable_elems;
  TransposeTable tab;
  auto flush = [&]() -> int {
    if (tab.n == 0) return 0;
    tab.total_tiles = tab.tile_begin[tab.n];
    const int r = launch_transpose_many(params, tbase, tab, s);
    tab.n = 0;
    return r;
  };
  tab.tile_begin[0] = 0;
  for (const auto& pi : e->params) {
    if (!(pi.flags & MAE_PARAM_MATRIX) || pi.t_off < 0) continue;
    const int rows = (int)pi.shape[0], cols = (int)(pi.numel / pi.shape[0]);
    const int i = tab.n++;
    tab.src_off[i] = pi.offset; tab.dst_off[i] = pi.t_off; tab.rows[i] = rows; tab.cols[i] = cols;
    tab.tile_begin[i + 1] = tab.tile_begin[i] + (int)(cdiv(rows, TransposeTable::TILE) * cdiv(cols, TransposeTable::TILE));
    if (tab.n == TransposeTable::MAX) { MAE_TRY(flush()); tab.tile_begin[0] = 0; }
  }
  return flush();
}

extern "C" int mae_engine_refresh_weights(mae_engine_t* e, const float* params, void* wcache, void* stream) {
  MAE_REQUIRE(e && params, "mae_engine_refresh_weights: null argument");
  if (e->act != MAE_BF16) return 0;
  MAE_REQUIRE(wcache, "mae_engine_refresh_weights: null weight cache");
  hipStream_t s = (hipStream_t)stream;
  MAE_TRY(launch_f32_to_bf16(params, reinterpret_cast<bf16*>(wcache), e->trainable_elems, s));
  return refresh_transposed(e, params, wcache, s);
}

extern "C" int mae_engine_forward_encoder(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                                          int32_t image_dtype, const int64_t* idx_keep, int32_t batch, int32_t num_keep,
                                          void* workspace, int64_t workspace_bytes, float* x_encoded, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, "mae_engine_forward_encoder"));
  MAE_REQUIRE(images && idx_keep, "mae_engine_forward_encoder: null images/idx_keep");
  MAE_TRY(check_image_dtype(image_dtype, "mae_engine_forward_encoder"));
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, nullptr, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_idx_to_i32(idx_keep, c.buf<int32_t>(pl.keep32), pl.Me, s));
  return forward_encoder_impl(c, pl, images, image_dtype, x_encoded);
}

extern "C" int mae_engine_forward_decoder(mae_engine_t* e, const float* params, const void* wcache, const float* x_encoded,
                                          const int64_t* idx_keep, const int64_t* idx_mask, int32_t batch, int32_t num_keep,
                                          int32_t num_mask, void* workspace, int64_t workspace_bytes, float* x_pred, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, "mae_engine_forward_decoder"));
  MAE_REQUIRE(idx_keep && idx_mask, "mae_engine_forward_decoder: null indices");
  MAE_REQUIRE(num_mask == pl.m, "mae_engine_forward_decoder: num_keep + num_mask must equal the sequence length (%d + %d != %d)", num_keep, num_mask, e->L);
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, nullptr, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_idx_to_i32(idx_keep, c.buf<int32_t>(pl.keep32), pl.Me, s));
  MAE_TRY(launch_idx_to_i32(idx_mask, c.buf<int32_t>(pl.mask32), pl.Mp, s));
  if (x_encoded) MAE_TRY(launch_cast(x_encoded, MAE_F32, c.buf<>(pl.enc_norm), e->act, pl.Me * e->D, s));
  return forward_decoder_impl(c, pl, x_pred);
}

extern "C" int mae_engine_backward(mae_engine_t* e, const float* params, const void* wcache, const float* d_pred,
                                   const float* d_x_encoded_extra, int32_t batch, int32_t num_keep, int32_t num_mask,
                                   void* workspace, int64_t workspace_bytes, float* grads, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, "mae_engine_backward"));
  MAE_REQUIRE(d_pred && grads, "mae_engine_backward: null d_pred/grads");
  MAE_REQUIRE(num_mask == pl.m && pl.m > 0, "mae_engine_backward: num_mask mismatch");
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, grads, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_cast(d_pred, MAE_F32, c.buf<>(pl.dpred), e->act, pl.Mp * e->PO, s));
  return backward_impl(c, pl, d_x_encoded_extra);
}

extern "C" int mae_engine_backward_decoder(mae_engine_t* e, const float* params, const void* wcache, const float* d_pred,
                                           int32_t batch, int32_t num_keep, int32_t num_mask, void* workspace,
                                           int64_t workspace_bytes, float* grads, float* d_x_encoded, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, "mae_engine_backward_decoder"));
  MAE_REQUIRE(d_pred && grads, "mae_engine_backward_decoder: null d_pred/grads");
  MAE_REQUIRE(num_mask == pl.m && pl.m > 0, "mae_engine_backward_decoder: num_mask mismatch");
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, grads, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_cast(d_pred, MAE_F32, c.buf<>(pl.dpred), e->act, pl.Mp * e->PO, s));
  backward_begin(c);
  MAE_TRY(backward_decoder_impl(c, pl));
  MAE_TRY(backward_end(c));
  if (d_x_encoded) MAE_TRY(launch_cast(c.buf<>(pl.d_ln), e->act, d_x_encoded, MAE_F32, pl.Me * e->D, s));
  return 0;
}

extern "C" int mae_engine_backward_encoder(mae_engine_t* e, const float* params, const void* wcache, const float* d_x_encoded,
                                           int32_t batch, int32_t num_keep, void* workspace, int64_t workspace_bytes,
                                           float* grads, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, "mae_engine_backward_encoder"));
  MAE_REQUIRE(d_x_encoded && grads, "mae_engine_backward_encoder: null d_x_encoded/grads");
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, grads, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_cast(d_x_encoded, MAE_F32, c.buf<>(pl.d_ln), e->act, pl.Me * e->D, s));
  backward_begin(c);
  MAE_TRY(backward_encoder_impl(c, pl));
  return backward_end(c);
}

extern "C" int64_t mae_engine_encoder_grad_elems(const mae_engine_t* e) { return e ? e->params[e->i_dec_mask].offset : 0; }

// decoder.decode(x) of lightly MAEDecoderTIMM (src/models/mae.py:71): x + decoder_pos_embed -> blocks -> decoder_norm, every row
extern "C" int mae_engine_decoder_decode(mae_engine_t* e, const float* params, const void* wcache, const float* x, int32_t batch,
                                         void* workspace, int64_t workspace_bytes, float* out, void* stream) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, 1, workspace, workspace_bytes, &pl, "mae_engine_decoder_decode"));
  MAE_REQUIRE(x && out, "mae_engine_decoder_decode: null x/out");
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, nullptr, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  MAE_TRY(launch_add_rows_pos(x, c.P(e->i_dec_pos), pl.Md, e->L, e->Dd, c.buf<float>(pl.dec_x[0]), s));
  for (int i = 0; i < e->dd; ++i)
    MAE_TRY(block_forward(c, pl, e->dec[i], pl.dec[i], pl.Md, e->Dd, e->Hd, pl.B, e->L, i ? pl.dec[i - 1].x_mid : 0, i > 0, pl.dec_x[i]));
  // the fused add + LayerNorm kernel reads the branch in its OUTPUT dtype: first pass in the activation dtype (forms the last
  // residual sum in dec_x[dd]; its normalised output goes to scratch), second pass writes fp32 rows for the caller
  float* stat = c.buf<float>(pl.dres);  // statistics of all B*L rows (dec_mean / dec_rstd hold the masked rows only)
  MAE_TRY(launch_layernorm_fwd(c.buf<float>(pl.dec[e->dd - 1].x_mid), c.buf<>(pl.branch_b), c.buf<float>(pl.dec_x[e->dd]), nullptr,
                               c.P(e->i_dn_w), c.P(e->i_dn_b), 1e-6f, pl.Md, e->Dd, c.act, c.buf<>(pl.d_ln), stat, stat + pl.Md, s));
  return launch_layernorm_fwd(c.buf<float>(pl.dec_x[e->dd]), nullptr, nullptr, nullptr, c.P(e->i_dn_w), c.P(e->i_dn_b), 1e-6f, pl.Md,
                              e->Dd, MAE_F32, out, stat, stat + pl.Md, s);
}

static int loss_and_grads_impl(mae_engine_t* e, const float* params, const void* wcache, const void* images, int32_t image_dtype,
                               const float* noise, int32_t batch, int32_t num_keep, float grad_scale, void* workspace, int64_t workspace_bytes, float* grads,
                               float* loss_out, int64_t* idx_keep_out, int64_t* idx_mask_out, void* const* ready, void* stream,
                               const char* who) {
  Plan pl;
  MAE_TRY(check_call(e, params, wcache, batch, num_keep, workspace, workspace_bytes, &pl, who));
  MAE_REQUIRE(images && noise && grads && loss_out, "%s: null argument", who);
  MAE_TRY(check_image_dtype(image_dtype, who));
  MAE_REQUIRE(pl.m > 0, "%s: nothing is masked", who);
  hipStream_t s = (hipStream_t)stream;
  Ctx c{e, params, (const char*)wcache, grads, (char*)workspace, s, e->act, (int64_t)dtype_size(e->act)};
  c.ready = ready;
  {
    TimerScope ts(e, TK_DATA, 0, (double)pl.Md * 12, s);
    MAE_TRY(launch_mask_from_noise(noise, batch, e->L, num_keep, idx_keep_out, idx_mask_out, c.buf<int32_t>(pl.keep32), c.buf<int32_t>(pl.mask32), s));
  }
  MAE_TRY(forward_encoder_impl(c, pl, images, image_dtype, nullptr));
  MAE_TRY(forward_decoder_impl(c, pl, nullptr));
  if (image_dtype == MAE_U8)
    RUN(TK_LOSS, 0, pl.Mp * e->P * (4 + c.as) + (int64_t)batch * e->C * e->img * e->img, launch_mse_from_images_u8(c.buf<float>(pl.pred), (const uint8_t*)images, c.buf<int32_t>(pl.mask32), batch, pl.m, e->C, e->img, e->p, grad_scale, loss_out, c.buf<>(pl.dpred), e->act, c.buf<float>(pl.loss_scratch), s));
  else
    RUN(TK_LOSS, 0, pl.Mp * e->P * (8 + c.as), launch_mse_from_images(c.buf<float>(pl.pred), (const float*)images, c.buf<int32_t>(pl.mask32), batch, pl.m, e->C, e->img, e->p, grad_scale, loss_out, c.buf<>(pl.dpred), e->act, c.buf<float>(pl.loss_scratch), s));
  return backward_impl(c, pl);
}

extern "C" int mae_engine_loss_and_grads(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                                         int32_t image_dtype, const float* noise, int32_t batch, int32_t num_keep, float grad_scale, void* workspace,
                                         int64_t workspace_bytes, float* grads, float* loss_out, int64_t* idx_keep_out,
                                         int64_t* idx_mask_out, void* stream) {
  return loss_and_grads_impl(e, params, wcache, images, image_dtype, noise, batch, num_keep, grad_scale, workspace, workspace_bytes, grads,
                             loss_out, idx_keep_out, idx_mask_out, nullptr, stream, "mae_engine_loss_and_grads");
}

extern "C" int32_t mae_engine_grad_ready_points(const mae_engine_t* e, int64_t* offsets, int32_t max_points) {
  if (!e) return 0;
  const int n = e->depth + 1;
  if (offsets) {
    for (int j = 0; j < n && j < max_points; ++j) {
      if (j == 0) offsets[j] = e->params[e->i_dec_mask].offset;
      else if (j < e->depth) offsets[j] = e->params[e->enc[e->depth - j].ln1_w].offset;
      else offsets[j] = 0;
    }
  }
  return n;
}

extern "C" int mae_engine_loss_and_grads_phased(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                                                int32_t image_dtype, const float* noise, int32_t batch, int32_t num_keep, float grad_scale, void* workspace,
                                                int64_t workspace_bytes, float* grads, float* loss_out, int64_t* idx_keep_out,
                                                int64_t* idx_mask_out, void* const* ready_events, int32_t num_ready, void* stream) {
  MAE_REQUIRE(e, "mae_engine_loss_and_grads_phased: null engine");
  MAE_REQUIRE(ready_events && num_ready == e->depth + 1,
              "mae_engine_loss_and_grads_phased: need one event slot per gradient-ready point (%d), got %d", e->depth + 1, num_ready);
  return loss_and_grads_impl(e, params, wcache, images, image_dtype, noise, batch, num_keep, grad_scale, workspace, workspace_bytes, grads,
                             loss_out, idx_keep_out, idx_mask_out, ready_events, stream, "mae_engine_loss_and_grads_phased");
}

// =====================================================================================================
// I-JEPA step (BASELINE.json configs[2], [4]; NO reference code -- specification: DESIGN.md section "I-JEPA")
//   target encoder (EMA weights, every patch token, no gradient) -> parameter-free LayerNorm -> target rows h
//   context encoder on the context tokens -> predictor on [context | mask tokens of block i] for each target block
//   -> latent regression loss -> backward through predictor and context encoder
// The engine's "decoder" tensors are the predictor (decoder_embed = predictor_embed, decoder_pred: Dd -> D, pred_dim = D).
// =====================================================================================================
namespace mae {

struct JepaPlan {
  int64_t ctx32, tgt32, tgt_rows, ones, zeros, stat, h, phase, total;
  Plan tgt, ctx;   // plans of the two phases, both placed at `phase`
  int64_t xenc;    // fp32 output of the target encoder, behind the target-phase plan
};

static JepaPlan make_jepa_plan(const mae_engine* e, int B, int k, int nblk, int m) {
  JepaPlan jp;
  const int N = e->L - 1;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { const int64_t o = off; off += round_up(std::max<int64_t>(bytes, 4), 256); return o; };
  const int64_t Mp = (int64_t)B * nblk * m;
  jp.ctx32 = take((int64_t)B * k * 4);
  jp.tgt32 = take(Mp * 4);
  jp.tgt_rows = take(Mp * 4);
  jp.ones = take(e->D * 4);
  jp.zeros = take(e->D * 4);
  jp.stat = take(2 * Mp * 4);
  jp.h = take(Mp * e->D * 4);
  jp.phase = off;
  jp.tgt = make_plan_ex(e, B, N, 1, 1, 1);                       // encoder over every patch token; a token-sized decoder stub
  jp.xenc = round_up(jp.tgt.total, 256);
  const int64_t tgt_total = jp.xenc + round_up((int64_t)B * N * e->D * 4, 256);
  jp.ctx = make_plan_ex(e, B, k, B * nblk, k + m, m);
  jp.total = jp.phase + std::max(tgt_total, jp.ctx.total);
  return jp;
}

}  // namespace mae

extern "C" int64_t mae_engine_jepa_workspace_bytes(const mae_engine_t* e, int32_t batch, int32_t num_context, int32_t num_blocks,
                                                   int32_t block_tokens) {
  if (!e || batch <= 0 || num_context < 1 || num_context >= e->L || num_blocks < 1 || block_tokens < 1 || block_tokens >= e->L) return -1;
  return make_jepa_plan(e, batch, num_context, num_blocks, block_tokens).total;
}

extern "C" int mae_engine_jepa_loss_and_grads(mae_engine_t* e, const float* params, const void* wcache, const float* target_params,
                                              const void* target_wcache, const void* images, int32_t image_dtype,
                                              const int64_t* idx_context, const int64_t* idx_target, int32_t batch, int32_t num_context,
                                              int32_t num_blocks, int32_t block_tokens, int32_t loss_kind, float grad_scale,
                                              void* workspace, int64_t workspace_bytes, float* grads, float* loss_out, float* h_out,
                                              float* pred_out, void* const* ready_events, int32_t num_ready, void* stream) {
  const char* who = "mae_engine_jepa_loss_and_grads";
  MAE_REQUIRE(e, "%s: null engine", who);
  MAE_REQUIRE(e->PO == e->D, "%s: the engine was not created with pred_dim = embed_dim (I-JEPA predicts latents of the encoder width)", who);
  MAE_REQUIRE(params && target_params && images && idx_context && idx_target && workspace && loss_out, "%s: null argument", who);
  MAE_REQUIRE(e->act == MAE_F32 || (wcache && target_wcache), "%s: bf16 engine needs both weight caches", who);
  MAE_REQUIRE(batch > 0 && num_context >= 1 && num_blocks >= 1 && block_tokens >= 1 && num_context + block_tokens <= e->L - 1 + block_tokens &&
              num_context < e->L && block_tokens < e->L, "%s: bad token counts (context %d, blocks %d x %d, %d patches)", who, num_context, num_blocks, block_tokens, e->L - 1);
  MAE_REQUIRE(loss_kind == MAE_LOSS_MSE || loss_kind == MAE_LOSS_SMOOTH_L1, "%s: loss_kind must be MAE_LOSS_MSE or MAE_LOSS_SMOOTH_L1", who);
  MAE_REQUIRE(!ready_events || num_ready == e->depth + 1, "%s: need one event slot per gradient-ready point (%d)", who, e->depth + 1);
  MAE_TRY(check_image_dtype(image_dtype, who));
  MAE_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)params & 15) == 0 && ((uintptr_t)target_params & 15) == 0, "%s: workspace must be 256-byte aligned, params 16-byte", who);
  const JepaPlan jp = make_jepa_plan(e, batch, num_context, num_blocks, block_tokens);
  MAE_REQUIRE(workspace_bytes >= jp.total, "%s: workspace too small (%lld < %lld bytes)", who, (long long)workspace_bytes, (long long)jp.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int N = e->L - 1;
  const int64_t Mp = (int64_t)batch * num_blocks * block_tokens;
  int32_t* ctx32 = reinterpret_cast<int32_t*>(ws + jp.ctx32);
  int32_t* tgt32 = reinterpret_cast<int32_t*>(ws + jp.tgt32);
  int32_t* tgt_rows = reinterpret_cast<int32_t*>(ws + jp.tgt_rows);
  float* h = h_out ? h_out : reinterpret_cast<float*>(ws + jp.h);
  MAE_TRY(launch_idx_to_i32(idx_context, ctx32, (int64_t)batch * num_context, s));
  MAE_TRY(launch_idx_to_i32(idx_target, tgt32, Mp, s));
  // ---- phase A: target encoder over all patch tokens (EMA weights), fp32 output, parameter-free LayerNorm of the target rows
  {
    Ctx c{e, target_params, (const char*)target_wcache, nullptr, ws + jp.phase, s, e->act, (int64_t)dtype_size(e->act)};
    c.fwd_only = true;
    float* xenc = reinterpret_cast<float*>(ws + jp.phase + jp.xenc);
    MAE_TRY(launch_iota_tokens(c.buf<int32_t>(jp.tgt.keep32), batch, N, s));
    MAE_TRY(forward_encoder_impl(c, jp.tgt, images, image_dtype, xenc));
    MAE_TRY(launch_rows_from_tokens(tgt32, batch, num_blocks * block_tokens, N, tgt_rows, s));
    MAE_TRY(launch_fill(reinterpret_cast<float*>(ws + jp.ones), 1.0f, e->D, s));
    MAE_HIP(hipMemsetAsync(ws + jp.zeros, 0, (size_t)e->D * 4, s));
    float* stat = reinterpret_cast<float*>(ws + jp.stat);
    RUN(TK_LN_FWD, 0, Mp * e->D * 8, launch_layernorm_fwd(xenc, nullptr, nullptr, tgt_rows, reinterpret_cast<float*>(ws + jp.ones), reinterpret_cast<float*>(ws + jp.zeros),
                                                          1e-5f, Mp, e->D, MAE_F32, h, stat, stat + Mp, s));  // F.layer_norm default eps
  }
  if (!grads) return 0;  // targets only
  // ---- phase B: context encoder, predictor, loss, backward
  Ctx c{e, params, (const char*)wcache, grads, ws + jp.phase, s, e->act, (int64_t)dtype_size(e->act)};
  c.ready = ready_events;
  const Plan& pl = jp.ctx;
  MAE_HIP(hipMemcpyAsync(c.buf<int32_t>(pl.keep32), ctx32, (size_t)batch * num_context * 4, hipMemcpyDeviceToDevice, s));
  MAE_TRY(forward_encoder_impl(c, pl, images, image_dtype, nullptr));
  const JepaSeq seq{ctx32, tgt32, num_blocks};
  MAE_TRY(forward_decoder_impl(c, pl, pred_out, &seq));
  const float* pred = pred_out ? pred_out : c.buf<float>(pl.pred);
  if (loss_kind == MAE_LOSS_SMOOTH_L1)
    RUN(TK_LOSS, 0, Mp * e->D * (8 + c.as), launch_smooth_l1(pred, h, Mp * e->D, grad_scale, loss_out, c.buf<>(pl.dpred), e->act, c.buf<float>(pl.loss_scratch), s));
  else
    RUN(TK_LOSS, 0, Mp * e->D * (8 + c.as), launch_mse(pred, h, Mp * e->D, grad_scale, loss_out, c.buf<>(pl.dpred), e->act, c.buf<float>(pl.loss_scratch), s));
  return backward_impl(c, pl, nullptr, &seq);
}

static int optimizer_step_impl(mae_engine_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wcache,
                               float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int64_t step,
                               float* stats_out, float* scratch, float* ema_target, void* ema_wcache, float ema_momentum, void* stream) {
  MAE_REQUIRE(e && params && grads && exp_avg && exp_avg_sq && stats_out && scratch, "mae_engine_optimizer_step: null argument");
  MAE_REQUIRE(step >= 1, "mae_engine_optimizer_step: step is 1-based");
  MAE_REQUIRE(e->act == MAE_F32 || wcache, "mae_engine_optimizer_step: bf16 engine needs the weight cache");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = e->trainable_elems;
  const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - std::pow((double)beta2, (double)step));
  RUN(TK_OPTIM, 0, n * 4, launch_grad_norm(grads, n, max_norm, stats_out, scratch, s));
  RUN(TK_OPTIM, 0, n * (28 + (e->act == MAE_BF16 ? 2 : 0)),
      launch_adamw(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, stats_out,
                   e->act == MAE_BF16 ? reinterpret_cast<bf16*>(wcache) : nullptr, s, ema_target,
                   e->act == MAE_BF16 ? reinterpret_cast<bf16*>(ema_wcache) : nullptr, ema_target ? e->params[e->i_dec_mask].offset : 0, ema_momentum));
  if (e->act == MAE_BF16) RUN(TK_OPTIM, 0, e->trans_elems * 6, refresh_transposed(e, params, wcache, s));
  return 0;
}

extern "C" int mae_engine_optimizer_step(mae_engine_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wcache,
                                         float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int64_t step,
                                         float* stats_out, float* scratch, void* stream) {
  return optimizer_step_impl(e, params, grads, exp_avg, exp_avg_sq, wcache, lr, beta1, beta2, eps, weight_decay, max_norm, step, stats_out,
                             scratch, nullptr, nullptr, 0.f, stream);
}

// ---- optimizer on a shard of the arena (data-parallel ranks that own 1 / world of the parameters) ----------------
extern "C" int mae_engine_grad_sumsq_range(mae_engine_t* e, const float* grads, int64_t lo, int64_t count, float* sumsq_out, float* scratch,
                                           void* stream) {
  MAE_REQUIRE(e && grads && sumsq_out && scratch, "mae_engine_grad_sumsq_range: null argument");
  MAE_REQUIRE(lo >= 0 && count >= 0 && lo % 4 == 0 && count % 4 == 0 && lo + count <= e->trainable_elems,
              "mae_engine_grad_sumsq_range: range [%lld, +%lld) outside the %lld trainable elements or not a multiple of 4", (long long)lo,
              (long long)count, (long long)e->trainable_elems);
  return launch_grad_sumsq(grads + lo, count, sumsq_out, scratch, (hipStream_t)stream);
}

extern "C" int mae_engine_clip_from_sumsq(mae_engine_t* e, const float* sumsq, float max_norm, float* stats_out, void* stream) {
  MAE_REQUIRE(e && sumsq && stats_out, "mae_engine_clip_from_sumsq: null argument");
  return launch_clip_from_sumsq(sumsq, max_norm, stats_out, (hipStream_t)stream);
}

extern "C" int mae_engine_adamw_range(mae_engine_t* e, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* wcache,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* stats,
                                      int64_t lo, int64_t count, void* stream) {
  MAE_REQUIRE(e && params && grads && exp_avg && exp_avg_sq && stats, "mae_engine_adamw_range: null argument");
  MAE_REQUIRE(step >= 1, "mae_engine_adamw_range: step is 1-based");
  MAE_REQUIRE(lo >= 0 && count >= 0 && lo % 4 == 0 && count % 4 == 0 && lo + count <= e->trainable_elems,
              "mae_engine_adamw_range: range [%lld, +%lld) outside the %lld trainable elements or not a multiple of 4", (long long)lo,
              (long long)count, (long long)e->trainable_elems);
  MAE_REQUIRE(e->act == MAE_F32 || wcache, "mae_engine_adamw_range: bf16 engine needs the weight cache");
  if (count == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float bc1 = (float)(1.0 - std::pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - std::pow((double)beta2, (double)step));
  RUN(TK_OPTIM, 0, count * (28 + (e->act == MAE_BF16 ? 2 : 0)),
      launch_adamw(params + lo, grads + lo, exp_avg + lo, exp_avg_sq + lo, count, lr, beta1, beta2, eps, weight_decay, bc1, bc2, stats,
                   e->act == MAE_BF16 ? reinterpret_cast<bf16*>(wcache) + lo : nullptr, s));
  return 0;   // the bf16 operand copies of the OTHER ranks' shards and the transposes: mae_engine_refresh_weights after the all-gather
}

extern "C" int mae_engine_optimizer_step_ema(mae_engine_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wcache,
                                             float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int64_t step,
                                             float* stats_out, float* scratch, float* target_params, void* target_wcache,
                                             float ema_momentum, void* stream) {
  MAE_REQUIRE(target_params && (!e || e->act == MAE_F32 || target_wcache), "mae_engine_optimizer_step_ema: null target arena / weight cache");
  MAE_REQUIRE(ema_momentum >= 0.f && ema_momentum <= 1.f, "mae_engine_optimizer_step_ema: momentum %g outside [0, 1]", (double)ema_momentum);
  return optimizer_step_impl(e, params, grads, exp_avg, exp_avg_sq, wcache, lr, beta1, beta2, eps, weight_decay, max_norm, step, stats_out,
                             scratch, target_params, target_wcache, ema_momentum, stream);
}

extern "C" int mae_engine_timers_enable(mae_engine_t* e, int32_t on) {
  MAE_REQUIRE(e, "null engine");
  e->timers_on = on != 0;
  return 0;
}
extern "C" int32_t mae_engine_timer_count(const mae_engine_t*) { return TK_COUNT; }
extern "C" const char* mae_engine_timer_name(const mae_engine_t*, int32_t kind) { return (kind >= 0 && kind < TK_COUNT) ? kTimerNames[kind] : ""; }
extern "C" int mae_engine_timers_reset(mae_engine_t* e) {
  MAE_REQUIRE(e, "null engine");
  for (auto& t : e->timers) { t.used = 0; t.flops = 0; t.bytes = 0; }
  return 0;
}
extern "C" int mae_engine_timer_read(mae_engine_t* e, int32_t kind, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  MAE_REQUIRE(e && kind >= 0 && kind < TK_COUNT, "mae_engine_timer_read: bad kind");
  TimerSlot& t = e->timers[kind];
  double ms = 0;
  for (size_t i = 0; i < t.used; ++i) {
    MAE_HIP(hipEventSynchronize(t.ev[i].second));
    float f = 0;
    MAE_HIP(hipEventElapsedTime(&f, t.ev[i].first, t.ev[i].second));
    ms += f;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)t.used;
  if (flops) *flops = t.flops;
  if (bytes) *bytes = t.bytes;
  return 0;
}
