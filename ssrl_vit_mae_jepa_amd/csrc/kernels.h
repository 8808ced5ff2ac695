// Internal launch API of libmae_hip.so: one function per kernel family.  Every function validates its
// operand shapes on the host before launching and returns 0 / non-zero (message via mae_last_error()).
#pragma once
#include "common.cuh"

namespace mae {

// ---- k_index.hip: masks and token-index maps ---------------------------------------------------
// noise (B, L) -> ascending-noise permutation with column 0 forced first; any output may be null.
int launch_mask_from_noise(const float* noise, int B, int L, int k, int64_t* keep64, int64_t* mask64,
                           int32_t* keep32, int32_t* mask32, hipStream_t s);
int launch_idx_to_i32(const int64_t* src, int32_t* dst, int64_t n, hipStream_t s);
int launch_idx_to_i64(const int32_t* src, int64_t* dst, int64_t n, hipStream_t s);
// inv[b][t] = j with keep[b][j] == t, else -1
int launch_build_inverse(const int32_t* keep32, int B, int k, int L, int32_t* inv, hipStream_t s);
// rows[b*m + j] = b*L + mask[b][j]  (row map of the masked tokens inside the (B*L) decoder matrix)
int launch_build_row_map(const int32_t* idx32, int B, int n_per, int L, int32_t* rows, hipStream_t s);

// ---- k_patch.hip: pixels <-> tokens ---------------------------------------------------------------
// A[(b,j)][c*p*p + py*p + px] = images[b][c][ph*p+py][pw*p+px] for token t = tok[b][j] >= 1 (patch t-1), zeros for t == 0
int launch_gather_patches(const float* images, const int32_t* tok, int B, int k, int C, int img, int p, int dt,
                          void* out, hipStream_t s);
// in place on x (B*k, D) fp32: x[r] = (t == 0 ? cls : x[r]) + pos[t]
int launch_assemble_visible(float* x, const int32_t* tok, const float* cls, const float* pos, int64_t rows, int D,
                            hipStream_t s);
// dtok (dt) = dx with class-token rows zeroed; dcls[D] = sum of class-token rows.  partial: >= 512*D floats
int launch_visible_grad_split(const float* dx, const int32_t* tok, int64_t rows, int D, int dt, void* dtok,
                              float* dcls, float* partial, hipStream_t s);
// out[b][t] = (inv[b][t] >= 0 ? xdec[b*k + inv] : mask_token) + pos[t]   (fp32 out)
int launch_decoder_assemble(const void* xdec, int dt, const int32_t* inv, const float* mask_token, const float* pos,
                            int B, int k, int L, int Dd, float* out, hipStream_t s);
// d_xdec[b*k+j] (dt) = dx[b][keep[b][j]];  d_mask_token[Dd] = sum over rows with inv < 0.  partial: >= 512*Dd floats
int launch_zero_unpredicted_rows(const int32_t* inv, int64_t rows, int Tseq, int m, int D, int act, float* dres, void* dres_c, hipStream_t s);
int launch_decoder_assemble_bwd(const float* dx, const int32_t* inv, const int32_t* keep, int B, int k, int L, int Dd,
                                int dt, void* d_xdec, float* d_mask_token, float* partial, hipStream_t s);
// target (B*m, P) fp32 in (py, px, c) order for patch max(mask[b][j]-1, 0)
int launch_patchify_gather(const float* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                           float* target, hipStream_t s);
// ---- k_jepa.hip: I-JEPA predictor input / target rows / latent loss (no reference code: spec in DESIGN.md) -----------------
int launch_iota_tokens(int32_t* keep32, int B, int N, hipStream_t s);                 // keep[b][j] = j + 1
int launch_fill(float* p, float v, int64_t n, hipStream_t s);
int launch_build_tail_row_map(int seqs, int T, int m, int32_t* rows, hipStream_t s);  // last m rows of every sequence of T
int launch_rows_from_tokens(const int32_t* tok, int B, int per_image, int N, int32_t* rows, hipStream_t s);  // b*N + tok-1
int launch_predictor_assemble(const void* xdec, int dt, const int32_t* ctx32, const int32_t* tgt32, const float* mask_token, const float* pos,
                              int B, int k, int nblk, int m, int L, int Dd, float* out, hipStream_t s);
int launch_predictor_assemble_bwd(const float* dx, int B, int k, int nblk, int m, int Dd, int dt, void* d_xdec, float* d_mask_token,
                                  float* partial, hipStream_t s);
int launch_smooth_l1(const float* pred, const float* target, int64_t n, float grad_scale, float* loss, void* d_pred, int dpred_dt,
                     float* scratch, hipStream_t s);

// ---- k_pixels_u8.hip: the same three pixel readers on uint8 images, ToTensor + Normalize(.5,.5) fused into the read -----
int launch_gather_patches_u8(const uint8_t* images, const int32_t* tok, int B, int k, int C, int img, int p, int dt, void* out, hipStream_t s);
int launch_mse_from_images_u8(const float* pred, const uint8_t* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                              float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch, hipStream_t s);
int launch_patchify_gather_u8(const uint8_t* images, const int32_t* mask32, int B, int m, int C, int img, int p, float* target, hipStream_t s);
int launch_patchify_gather_u8_i64(const uint8_t* images, const int64_t* mask64, int B, int m, int C, int img, int p, float* target, hipStream_t s);
int launch_augment_crop_flip_u8(const uint8_t* in, const int32_t* params, int B, int C, int S, uint8_t* out, hipStream_t s);
int launch_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, hipStream_t s);
// out[r] = x[r] + pos[r mod L] (fp32 rows of D)
int launch_add_rows_pos(const float* x, const float* pos, int64_t rows, int L, int D, float* out, hipStream_t s);
// dst (dt) += src (fp32)
int launch_add_into(const float* src, void* dst, int dst_dt, int64_t n, hipStream_t s);
// out[c] = sum_{i<G} partial[i][c] in row order (deterministic); columns [0,split) go to out0, [split,C) to out1
int launch_sum_partials(const float* partial, int G, int C, float* out0, float* out1, int split, hipStream_t s);

// ---- k_loss_optim.hip -----------------------------------------------------------------------------
// loss[0] = mean((pred-target)^2); d_pred (dt, may be null) = grad_scale*2*(pred-target)/n.  scratch >= 1024+ floats
int launch_mse(const float* pred, const float* target, int64_t n, float grad_scale, float* loss, void* d_pred,
               int dpred_dt, float* scratch, hipStream_t s);
// same with the target read straight from the image (never materialised)
// band walk for float images (k_pixels_u8.hip); -1 = geometry not covered
int launch_mse_from_images_band_f32(const float* pred, const float* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                                    float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch, hipStream_t s);
int launch_mse_from_images(const float* pred, const float* images, const int32_t* mask32, int B, int m, int C, int img,
                           int p, float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch,
                           hipStream_t s);
// stats[0] = ||g||_2, stats[1] = min(1, max_norm/(norm+1e-6)).  scratch >= 1024+ floats
int launch_grad_norm(const float* g, int64_t n, float max_norm, float* stats, float* scratch, hipStream_t s);
int launch_grad_sumsq(const float* g, int64_t n, float* out, float* scratch, hipStream_t s);
int launch_clip_from_sumsq(const float* sumsq, float max_norm, float* stats, hipStream_t s);
// AdamW with g scaled by stats[1]; optionally refresh bf16 copy of the params (wbf may be null)
// ema_target != null: target[0 .. ema_n) = ema_momentum * target + (1 - ema_momentum) * p_new in the same sweep (+ its bf16 copy)
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, float bc1, float bc2, const float* stats, bf16* wbf, hipStream_t s, float* ema_target = nullptr,
                 bf16* ema_wbf = nullptr, int64_t ema_n = 0, float ema_momentum = 0.f);
int launch_f32_to_bf16(const float* src, bf16* dst, int64_t n, hipStream_t s);
// dst[c][r] = (bf16) src[r][c]
int launch_transpose_to_bf16(const float* src, bf16* dst, int rows, int cols, hipStream_t s);
// the same for a whole list of matrices living in one arena (one launch); passed by value as a kernel argument
struct TransposeTable {
  static constexpr int MAX = 64, TILE = 64;   // matrices per launch, tile edge
  int n = 0, total_tiles = 0;
  int64_t src_off[MAX], dst_off[MAX];
  int rows[MAX], cols[MAX], tile_begin[MAX + 1];
};
int launch_transpose_many(const float* params, bf16* tbase, const TransposeTable& tab, hipStream_t s);

// ---- k_layernorm.hip --------------------------------------------------------------------------------
// y = LN(x [+ branch]); with a branch (dtype y_dt) the sum is also written to x_out (fp32): the fused residual add
int launch_layernorm_fwd(const float* x, const void* branch, float* x_out, const int32_t* row_map, const float* gamma,
                         const float* beta, float eps, int64_t rows, int dim, int y_dt, void* y, float* mean, float* rstd,
                         hipStream_t s);
// Second stages of several two-stage column reductions, run by ONE launch (passed by value as a kernel argument): the
// backward pass queues the dgamma / dbeta reduction of each of its LayerNorms here instead of launching 27 small,
// latency-bound kernels between the big ones.
struct PartialsTable {
  static constexpr int MAX = 64;
  int n = 0;
  const float* partial[MAX];
  float *out0[MAX], *out1[MAX];
  int G[MAX], C[MAX], split[MAX], block_begin[MAX + 1];
};
int launch_sum_partials_many(const PartialsTable& tab, hipStream_t s);
// `defer` != null: the dgamma / dbeta second stage is appended to *defer (the caller keeps `partial` alive and launches
// launch_sum_partials_many later) instead of being launched here
int launch_layernorm_bwd(const void* dy, int dy_dt, const float* x, const int32_t* row_map, const float* gamma,
                         const float* mean, const float* rstd, int64_t rows, int dim, int accumulate, float* dx_io,
                         void* dx_copy, float* dgamma, float* dbeta, float* partial, hipStream_t s, PartialsTable* defer = nullptr);
constexpr int LN_BWD_MAX_BLOCKS = 1024;

// Compute units of the current device (256 on MI355X), queried once: the persistent GEMMs launch one workgroup per CU.
int num_cus();

// ---- GEMM family ------------------------------------------------------------------------------------
struct Epi {
  int mode = MAE_EPI_NONE;
  const float* bias = nullptr;
  const void* aux = nullptr;  // RESID: fp32 residual (M,N); DGELU: saved pre-activation (M,N) in out_dt
  void* out = nullptr;
  void* out2 = nullptr;       // GELU: activated output
  int out_dt = MAE_F32;
};
// out[M,N] = A[M,K] * W[N,K]^T
int launch_linear_fwd(const void* A, const void* W, int64_t M, int N, int K, int dt, const Epi& epi, hipStream_t s);
// dX[M,K] = dY[M,N] * W[N,K]   (epi.mode NONE or DGELU; bias unused)
int launch_linear_dgrad(const void* dY, const void* W, int64_t M, int N, int K, int dt, const Epi& epi, hipStream_t s);
// dW[N,K] (fp32, written) = dY[M,N]^T * A[M,K]; db[N] = colsum(dY) (may be null)
int64_t linear_wgrad_scratch_bytes(int64_t M, int N, int K);
int launch_linear_wgrad(const void* dY, const void* A, int64_t M, int N, int K, int dt, float* dW, float* db,
                        void* scratch, hipStream_t s);
int64_t linear_wgrad_pair_scratch_bytes(int64_t M, int N0, int K0, int N1, int K1);
int launch_linear_wgrad_pair(const void* dY0, const void* A0, int N0, int K0, float* dW0, float* db0, const void* dY1, const void* A1,
                             int N1, int K1, float* dW1, float* db1, int64_t M, int dt, void* scratch, hipStream_t s);

// ---- attention ------------------------------------------------------------------------------------------
int launch_attention_fwd(const void* qkv, int B, int T, int H, int hd, int dt, void* out, float* lse, hipStream_t s);
int launch_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int B, int T, int H,
                         int hd, int dt, void* d_qkv, hipStream_t s);

}  // namespace mae
