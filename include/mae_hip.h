/*
 * mae_hip.h -- C ABI of libmae_hip.so, the MI355X (gfx950) engine behind the reference's
 * MaskedAutoencoder boundary (reference: src/models/mae.py:12-94; step semantics
 * src/training/mae.py:40-83, scripts/training/pretrain_mae.py:116-126).
 *
 * The reference is pure Python and has no FFI of its own; the symbols below are what a
 * ctypes binding in the reference's src/models/mae.py would load.
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm caching allocator);
 *     the library borrows it for the duration of the call, allocates nothing on the device
 *     and keeps no pointer after the call returns;
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising; it is safe to capture into a hipGraph;
 *   - return value 0 = ok; non-zero = error, text via mae_last_error() (thread-local);
 *     no exception ever crosses this boundary;
 *   - "act dtype" selects the arithmetic of the branch tensors (LayerNorm outputs, q/k/v,
 *     MLP hidden, their gradients): MAE_F32 = exact fp32 path (parity), MAE_BF16 = bf16
 *     operands with fp32 accumulation on the MFMA units (throughput).  The residual stream,
 *     LayerNorm statistics, losses, gradients of parameters and optimizer state are fp32
 *     in both modes;
 *   - token indices are int64 on the API (as torch.argsort returns them) and int32 inside;
 *   - images are (batch, C, H, W) NCHW, either MAE_F32 (already normalised, what the reference's
 *     DataLoader yields after ToTensor + Normalize(.5,.5), src/data.py:15-24) or MAE_U8 (raw pixels
 *     as the STL-10 file stores them): with MAE_U8 the kernels that read pixels apply
 *     (u8/255 - 0.5)/0.5 themselves, bit-identical to the torch expression, and fetch every
 *     image byte once per kernel.
 */
#ifndef MAE_HIP_H
#define MAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAE_ABI_VERSION 4

enum { MAE_F32 = 0, MAE_BF16 = 1, MAE_U8 = 2 /* images only */ };

/* parameter flags (mae_engine_param_info) */
enum {
  MAE_PARAM_TRAINABLE = 1,   /* receives a gradient on the MAE path; clip + AdamW touch it     */
  MAE_PARAM_FROZEN    = 2,   /* requires_grad=False in the reference (sin-cos position tables)  */
  MAE_PARAM_UNUSED    = 4,   /* trainable in the reference but unreachable on this path:
                                encoder.mask_token (encode() is called with idx_mask=None,
                                src/models/mae.py:55) -> grad None -> skipped by AdamW/clip     */
  MAE_PARAM_MATRIX    = 8    /* 2-D GEMM weight (out, in): has a bf16 copy and a transposed copy */
};

/* Replaces the three ctor dicts of MaskedAutoencoder.__init__ (src/models/mae.py:15-52). */
typedef struct mae_config {
  int32_t image_size, patch_size, in_chans;
  int32_t embed_dim, depth, num_heads;
  int32_t decoder_embed_dim, decoder_depth, decoder_num_heads;
  int32_t mlp_ratio;      /* timm default 4 */
  int32_t act_dtype;      /* MAE_F32 | MAE_BF16 */
  int32_t pred_dim;       /* width of the prediction head: 0 = patch_size^2 * in_chans (MAE pixel targets);
                             embed_dim for the I-JEPA engine (latent targets), where the "decoder" is the predictor */
  int32_t reserved[4];
} mae_config_t;

typedef struct mae_engine mae_engine_t;

const char* mae_last_error(void);
int         mae_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Engine: the whole path of MaskedAutoencoder.forward + MSE + backward + clip + AdamW.
 * ---------------------------------------------------------------------------------------- */
int  mae_engine_create(const mae_config_t* cfg, mae_engine_t** out);   /* src/models/mae.py:15-52 */
void mae_engine_destroy(mae_engine_t* e);

/* Parameter arena: one flat fp32 buffer.  Trainable-on-path tensors come first (so that clip and
 * AdamW run over one contiguous range [0, mae_engine_trainable_elems)), then frozen/unused ones.
 * Each tensor starts at a multiple of 64 elements; padding is zero and stays zero. */
int64_t mae_engine_num_params(const mae_engine_t* e);
int64_t mae_engine_arena_elems(const mae_engine_t* e);
int64_t mae_engine_trainable_elems(const mae_engine_t* e);
/* name: state_dict key, shape[4], flags: MAE_PARAM_*.  Order = state_dict order. */
int mae_engine_param_info(const mae_engine_t* e, int64_t index, const char** name, int64_t* offset,
                          int64_t* numel, int32_t* ndim, int64_t shape[4], int32_t* flags);

/* Workspace (saved activations + scratch) for `batch` images with `num_keep` visible tokens. */
int64_t mae_engine_workspace_bytes(const mae_engine_t* e, int32_t batch, int32_t num_keep);

/* Refresh the bf16 / transposed-bf16 operand copies of the GEMM weights from the fp32 arena.
 * Needed after the caller changed parameters behind the engine's back (load_state_dict, an
 * external optimizer).  mae_engine_optimizer_step does it itself.  No-op in MAE_F32 mode.
 * wcache: mae_engine_wcache_bytes() bytes. */
int64_t mae_engine_wcache_bytes(const mae_engine_t* e);
int mae_engine_refresh_weights(mae_engine_t* e, const float* params, void* wcache, void* stream);

/* lightly utils.random_token_mask with the noise draw made explicit (src/models/mae.py:79-83):
 * noise (batch, L) fp32 -> idx_keep (batch, num_keep), idx_mask (batch, L-num_keep), int64,
 * ascending noise, column 0 forced to -1 (class token always kept, always first), ties broken
 * by lower index (stable). */
int mae_mask_from_noise(const float* noise, int32_t batch, int32_t seq_len, int32_t num_keep,
                        int64_t* idx_keep, int64_t* idx_mask, void* stream);

/* MaskedAutoencoder.forward_encoder(images, idx_keep) (src/models/mae.py:54-55).
 * images (batch, C, H, W) NCHW in image_dtype (MAE_F32 | MAE_U8); idx_keep (batch, num_keep) int64 token ids in [0, L);
 * x_encoded (batch, num_keep, D) fp32.  Saves what backward needs in `workspace`. */
int mae_engine_forward_encoder(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                               int32_t image_dtype, const int64_t* idx_keep, int32_t batch, int32_t num_keep,
                               void* workspace, int64_t workspace_bytes, float* x_encoded, void* stream);

/* MaskedAutoencoder.forward_decoder(x_encoded, idx_keep, idx_mask) (src/models/mae.py:57-75).
 * x_encoded may be NULL = "use the encoder output already in workspace" (the fused forward).
 * x_pred (batch, L-num_keep... = num_mask, p*p*C) fp32. */
int mae_engine_forward_decoder(mae_engine_t* e, const float* params, const void* wcache, const float* x_encoded,
                               const int64_t* idx_keep, const int64_t* idx_mask, int32_t batch, int32_t num_keep,
                               int32_t num_mask, void* workspace, int64_t workspace_bytes, float* x_pred,
                               void* stream);

/* lightly utils.patchify + get_at_index(idx_mask-1) (src/models/mae.py:90-92):
 * target (batch, num_mask, p*p*C) fp32, per-patch order (py, px, c). */
int mae_patchify_gather(const void* images, int32_t image_dtype, const int64_t* idx_mask, int32_t batch,
                        int32_t in_chans, int32_t image_size, int32_t patch_size, int32_t num_mask, float* target,
                        void* stream);

/* The augmentation step in front of the path: transforms.RandomResizedCrop(96, scale=(0.8, 1.0)) +
 * RandomHorizontalFlip() on the uint8 image, before ToTensor (src/data.py:15-20).  params (batch, 5)
 * int32 = (top, left, height, width, flip) per image, drawn by the caller (torchvision's get_params
 * rule, ssrl_vit_mae_jepa_amd/data.py); the box is resampled bilinearly to image_size x image_size
 * (pixel centres aligned, border clamp), rounded to uint8, columns mirrored when flip != 0.
 * images, out (batch, C, S, S) uint8, out != images. */
int mae_augment_crop_flip_u8(const uint8_t* images, const int32_t* params, int32_t batch, int32_t in_chans,
                             int32_t image_size, uint8_t* out, void* stream);

/* torch.nn.MSELoss() (src/training/mae.py:40,48) over n elements, and its gradient w.r.t. pred
 * scaled by grad_scale: loss[0] = mean((pred-target)^2); d_pred = grad_scale*2*(pred-target)/n.
 * scratch: >= 4096 floats. */
int mae_mse_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss,
                 float* d_pred /* may be NULL */, float* scratch, void* stream);

/* Backward of forward_decoder(forward_encoder(.)) given d_pred (batch, num_mask, P) fp32.
 * Writes (not accumulates) every trainable gradient into grads[0 .. trainable_elems).
 * Requires the workspace of the matching forward calls.  d_x_encoded_extra: optional extra
 * gradient (batch, num_keep, D) fp32 added at the encoder output (NULL on the MAE path): a second consumer of
 * x_encoded, e.g. a probe head trained next to the reconstruction loss. */
int mae_engine_backward(mae_engine_t* e, const float* params, const void* wcache, const float* d_pred,
                        const float* d_x_encoded_extra, int32_t batch, int32_t num_keep, int32_t num_mask,
                        void* workspace, int64_t workspace_bytes, float* grads, void* stream);

/* The two halves of mae_engine_backward, for callers that hold forward_encoder and forward_decoder as separate autograd
 * nodes (the reference's fine-tuning hands encoder.vit to a classifier: scripts/training/train_mae.py:143,
 * src/models/classifier.py:47-57).
 * _decoder: d_pred -> every decoder gradient (arena range [mae_engine_encoder_grad_elems, trainable_elems)) and, when
 *           d_x_encoded is not NULL, the gradient w.r.t. forward_decoder's x_encoded input (batch, num_keep, D) fp32.
 * _encoder: d_x_encoded (batch, num_keep, D) fp32 -> every encoder gradient (arena range [0, encoder_grad_elems)).
 * Each writes only its own range of `grads` and needs the workspace of the matching forward call. */
int64_t mae_engine_encoder_grad_elems(const mae_engine_t* e);
int mae_engine_backward_decoder(mae_engine_t* e, const float* params, const void* wcache, const float* d_pred,
                                int32_t batch, int32_t num_keep, int32_t num_mask, void* workspace,
                                int64_t workspace_bytes, float* grads, float* d_x_encoded, void* stream);
int mae_engine_backward_encoder(mae_engine_t* e, const float* params, const void* wcache, const float* d_x_encoded,
                                int32_t batch, int32_t num_keep, void* workspace, int64_t workspace_bytes,
                                float* grads, void* stream);

/* decoder.decode(x) (lightly MAEDecoderTIMM, called at src/models/mae.py:71): x (batch, L, Dd) fp32 ->
 * decoder_norm(blocks(x + decoder_pos_embed)) for EVERY row, (batch, L, Dd) fp32.  Inference only (saves nothing).
 * decoder.embed / decoder.predict (src/models/mae.py:59,73) are plain Linears: mae_linear_fwd. */
int mae_engine_decoder_decode(mae_engine_t* e, const float* params, const void* wcache, const float* x, int32_t batch,
                              void* workspace, int64_t workspace_bytes, float* out, void* stream);

/* One fused pass: zero_grad + mask + forward + MSE + backward (training_step + loss.backward(),
 * src/training/mae.py:45-50).  noise (batch, L) fp32.  loss_out[0] = batch-mean MSE.
 * grad_scale multiplies the loss gradient (1/world_size for data-parallel sum-all-reduce).
 * idx_keep_out/idx_mask_out: optional int64 outputs (may be NULL). */
int mae_engine_loss_and_grads(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                              int32_t image_dtype, const float* noise, int32_t batch, int32_t num_keep, float grad_scale,
                              void* workspace, int64_t workspace_bytes, float* grads, float* loss_out,
                              int64_t* idx_keep_out, int64_t* idx_mask_out, void* stream);

/* Data-parallel overlap (no counterpart in the reference, which runs devices=1, scripts/training/pretrain_mae.py:118).
 * The backward pass finishes the gradient arena from its END: the decoder's tensors first, then encoder block depth-1,
 * ..., block 0, patch projection and class token last.  mae_engine_grad_ready_points reports those points in the order
 * they are reached: reaching point j means grads[offsets[j] .. trainable_elems) is final.  Returns the number of
 * points (encoder depth + 1; the last one has offset 0); fills at most max_points offsets (offsets may be NULL). */
int32_t mae_engine_grad_ready_points(const mae_engine_t* e, int64_t* offsets, int32_t max_points);
/* mae_engine_loss_and_grads that also records ready_events[j] (a hipEvent_t passed as void*; NULL = skip the point) on
 * `stream` when point j is reached, so that the caller can start the all-reduce of that arena range on another stream
 * while the rest of the backward pass still runs.  num_ready must equal mae_engine_grad_ready_points(). Results are
 * bit-identical to mae_engine_loss_and_grads. */
int mae_engine_loss_and_grads_phased(mae_engine_t* e, const float* params, const void* wcache, const void* images,
                                     int32_t image_dtype, const float* noise, int32_t batch, int32_t num_keep, float grad_scale,
                                     void* workspace, int64_t workspace_bytes, float* grads, float* loss_out,
                                     int64_t* idx_keep_out, int64_t* idx_mask_out, void* const* ready_events,
                                     int32_t num_ready, void* stream);

/* clip_grad_norm_(max_norm, L2) (scripts/training/pretrain_mae.py:124-125) followed by
 * torch.optim.AdamW single-group step (src/training/mae.py:59-65) over the trainable range,
 * then the operand-copy refresh.  step is 1-based.  stats_out[0] = total grad norm (pre-clip),
 * stats_out[1] = clip coefficient.  exp_avg / exp_avg_sq: trainable_elems floats each.
 * scratch: >= 4096 floats. */
int mae_engine_optimizer_step(mae_engine_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                              void* wcache, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float max_norm, int64_t step, float* stats_out, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * I-JEPA step (BASELINE.json configs[2] and [4]).  The reference has NO I-JEPA code ("JEPA" only in README.md:1,9 and
 * pyproject.toml:2), so nothing here replaces a reference function: the specification is DESIGN.md section "I-JEPA"
 * (I-JEPA paper), built from the same ViT pieces.  The engine must be created with pred_dim = embed_dim; its
 * "decoder" tensors are the predictor.  Token ids are patch tokens 1 .. N (0, the class token, is not used).
 *   idx_context (batch, num_context) int64: context tokens of every image (same count for all: truncated to the minimum);
 *   idx_target  (batch, num_blocks, block_tokens) int64: the target blocks;
 *   target_params / target_wcache: the EMA target encoder, an arena of the same layout (only the encoder part is read);
 *   loss over all batch * num_blocks * block_tokens * D elements between predictor output and
 *   layer_norm(target_encoder(images))[targets] (no affine, eps 1e-5); loss_kind MAE_LOSS_MSE | MAE_LOSS_SMOOTH_L1.
 * grads == NULL: compute the targets only.  h_out / pred_out: optional fp32 copies (batch*num_blocks*block_tokens, D).
 * ready_events: as mae_engine_loss_and_grads_phased (may be NULL, then num_ready is ignored). */
enum { MAE_LOSS_MSE = 0, MAE_LOSS_SMOOTH_L1 = 1 };
int64_t mae_engine_jepa_workspace_bytes(const mae_engine_t* e, int32_t batch, int32_t num_context, int32_t num_blocks,
                                        int32_t block_tokens);
int mae_engine_jepa_loss_and_grads(mae_engine_t* e, const float* params, const void* wcache, const float* target_params,
                                   const void* target_wcache, const void* images, int32_t image_dtype,
                                   const int64_t* idx_context, const int64_t* idx_target, int32_t batch,
                                   int32_t num_context, int32_t num_blocks, int32_t block_tokens, int32_t loss_kind,
                                   float grad_scale, void* workspace, int64_t workspace_bytes, float* grads,
                                   float* loss_out, float* h_out, float* pred_out, void* const* ready_events,
                                   int32_t num_ready, void* stream);
/* The optimizer on a SHARD of the parameter arena (ABI v4; data-parallel ranks that each own 1 / world of the trainable range, the
 * reduce-scatter -> shard sum of squares -> scalar all-reduce -> AdamW on the shard -> all-gather scheme).  The reference runs one
 * replicated clip_grad_norm_ + AdamW (scripts/training/pretrain_mae.py:124-125, src/training/mae.py:59-65); these three calls are that
 * step cut at the two points where ranks have to talk.  lo, count: float offsets into the arenas, multiples of 4, inside
 * [0, mae_engine_trainable_elems).
 *   mae_engine_grad_sumsq_range: sumsq_out[0] = sum of grads[lo .. lo+count)^2 (count may be 0)
 *   mae_engine_clip_from_sumsq : stats_out = {sqrt(sumsq[0]), min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))} -- after the ranks' sums were added
 *   mae_engine_adamw_range     : AdamW on [lo, lo+count) with gradients scaled by stats[1]; writes the bf16 operand copy of that range only.
 *                                Call mae_engine_refresh_weights once the shards have been gathered. */
int mae_engine_grad_sumsq_range(mae_engine_t* e, const float* grads, int64_t lo, int64_t count, float* sumsq_out, float* scratch,
                                void* stream);
int mae_engine_clip_from_sumsq(mae_engine_t* e, const float* sumsq, float max_norm, float* stats_out, void* stream);
int mae_engine_adamw_range(mae_engine_t* e, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* wcache,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* stats,
                           int64_t lo, int64_t count, void* stream);

/* mae_engine_optimizer_step with the EMA update of the target encoder fused into the AdamW sweep:
 * target[0 .. encoder_grad_elems) = m * target + (1 - m) * params_new (+ the bf16 operand copy in target_wcache).
 * max_norm = +inf disables clipping (I-JEPA trains unclipped). */
int mae_engine_optimizer_step_ema(mae_engine_t* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                  void* wcache, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  float max_norm, int64_t step, float* stats_out, float* scratch, float* target_params,
                                  void* target_wcache, float ema_momentum, void* stream);

/* Per-kernel-class device timing with HIP events on `stream` (bench.py roofline): enable, run
 * steps, then read.  kind: index into mae_engine_timer_name().  Reading synchronises the events. */
int         mae_engine_timers_enable(mae_engine_t* e, int32_t on);
int32_t     mae_engine_timer_count(const mae_engine_t* e);
const char* mae_engine_timer_name(const mae_engine_t* e, int32_t kind);
int         mae_engine_timer_read(mae_engine_t* e, int32_t kind, double* total_ms, int64_t* launches,
                                  double* flops, double* bytes);
int         mae_engine_timers_reset(mae_engine_t* e);

/* ------------------------------------------------------------------------------------------
 * Single kernels (the pieces the engine is made of; exported for parity tests and reuse).
 * dtype arguments are MAE_F32 / MAE_BF16 and describe the `void*` activation tensors.
 * ---------------------------------------------------------------------------------------- */

/* LayerNorm(eps) forward over rows of x (fp32, rows x dim): y[r] = LN(x[row_map ? row_map[r] : r]).
 * mean/rstd: (rows) fp32 saved for backward. */
int mae_layernorm_fwd(const float* x, const int32_t* row_map, const float* gamma, const float* beta, float eps,
                      int64_t rows, int32_t dim, int32_t y_dtype, void* y, float* mean, float* rstd, void* stream);
/* Same with the residual add fused in: v = x[row] + branch[row] (branch in y_dtype); x_out[row] = v (fp32); y = LN(v).
 * This is `x = x + attn(...)` / `x = x + mlp(...)` of timm Block followed by the next norm. */
int mae_add_layernorm_fwd(const float* x, const void* branch, float* x_out, const int32_t* row_map, const float* gamma,
                          const float* beta, float eps, int64_t rows, int32_t dim, int32_t y_dtype, void* y, float* mean,
                          float* rstd, void* stream);
/* Backward: dx_io[row] = (accumulate ? dx_io[row] : 0) + dLN/dx ; dx_copy (dtype, may be NULL) gets the
 * same value in the activation dtype; dgamma/dbeta (dim) written.  partial: >= 2*1024*dim floats. */
int mae_layernorm_bwd(const void* dy, int32_t dy_dtype, const float* x, const int32_t* row_map, const float* gamma,
                      const float* mean, const float* rstd, int64_t rows, int32_t dim, int32_t accumulate,
                      float* dx_io, void* dx_copy, float* dgamma, float* dbeta, float* partial, void* stream);

/* Epilogues of the GEMM family. */
enum {
  MAE_EPI_NONE  = 0,  /* out = acc (+bias)                                        */
  MAE_EPI_GELU  = 1,  /* out = acc+bias (pre-activation), out2 = gelu_erf(out)    */
  MAE_EPI_RESID = 2,  /* out(fp32) = resid(fp32) + acc + bias                     */
  MAE_EPI_DGELU = 3,  /* out = acc * gelu_erf'(aux[m][n])  (aux = saved pre-act)  */
  MAE_EPI_GELU_GRAD = 4, /* v = acc+bias rounded to the output type; out = gelu_erf'(v), out2 = gelu_erf(v): the forward
                            saves the derivative instead of the pre-activation, so backward only multiplies  */
  MAE_EPI_MUL   = 5,  /* out = acc * aux[m][n]                                  */
  MAE_EPI_GELU_ACT = 6 /* out = gelu_erf(acc+bias rounded to the output type): one output, for forward-only passes
                          (the I-JEPA target encoder never runs backward, so nothing but the activation is needed) */
};
/* out[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) -- torch.nn.functional.linear.  A, W in `dtype`;
 * out/out2/aux in out_dtype (MAE_EPI_RESID: out and resid fp32). */
int mae_linear_fwd(const void* A, const void* W, const float* bias, int64_t M, int32_t N, int32_t K, int32_t dtype,
                   int32_t epilogue, int32_t out_dtype, void* out, void* out2, const void* aux_or_resid,
                   void* stream);
/* dW[N,K] = dY[M,N]^T * A[M,K] (fp32 out, written), db[N] = column sums of dY (may be NULL).
 * scratch: mae_linear_wgrad_scratch_bytes(M, N, K) bytes. */
int64_t mae_linear_wgrad_scratch_bytes(int64_t M, int32_t N, int32_t K);
int mae_linear_wgrad(const void* dY, const void* A, int64_t M, int32_t N, int32_t K, int32_t dtype, float* dW,
                     float* db, void* scratch, void* stream);

/* Two weight gradients over the same M rows in one launch (the engine pairs a block's fc2 + fc1 and proj + qkv:
 * torch.autograd's dW = dY^T X of two nn.Linear, timm Block / Attention).  Same results contract as two
 * mae_linear_wgrad calls (fp32 dW / db written); falls back to exactly those when a shape is outside the
 * ring kernel.  scratch >= mae_linear_wgrad_pair_scratch_bytes. */
int64_t mae_linear_wgrad_pair_scratch_bytes(int64_t M, int32_t N0, int32_t K0, int32_t N1, int32_t K1);
int mae_linear_wgrad_pair(const void* dY0, const void* A0, int32_t N0, int32_t K0, float* dW0, float* db0,
                          const void* dY1, const void* A1, int32_t N1, int32_t K1, float* dW1, float* db1,
                          int64_t M, int32_t dtype, void* scratch, void* stream);

/* Multi-head self-attention core of timm Attention (F.scaled_dot_product_attention, no mask, no dropout).
 * qkv: (batch, T, 3, H, hd) in `dtype` as the qkv Linear emits it; out: (batch, T, H*hd) in `dtype`;
 * lse: (batch, H, T) fp32 log-sum-exp of the scaled scores, saved for backward. */
int mae_attention_fwd(const void* qkv, int32_t batch, int32_t T, int32_t H, int32_t hd, int32_t dtype, void* out,
                      float* lse, void* stream);
int mae_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int32_t batch,
                      int32_t T, int32_t H, int32_t hd, int32_t dtype, void* d_qkv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAE_HIP_H */
