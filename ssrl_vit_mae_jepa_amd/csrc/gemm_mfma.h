// bf16 MFMA GEMM kernels (k_gemm_mfma.hip): entry points used by the dispatch in k_gemm.hip.
// Each returns MFMA_UNSUPPORTED when the shape is outside what the tiles assume; the caller then falls
// back to the any-shape kernel.
#pragma once
#include "kernels.h"

namespace mae {

constexpr int MFMA_UNSUPPORTED = -7777;

// out[M,N] = A[M,K] * W[N,K]^T with the Epi epilogues; needs K % 32 == 0, N % 16 == 0
int mfma_linear_fwd(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& epi, hipStream_t s);

// persistent LDS-DMA ring variant (k_gemm_mfma2.hip): NONE / GELU epilogues, N % 128 == 0, K % 64 == 0, K >= 192
int mfma_linear_fwd_v2(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& epi, hipStream_t s);

// round-3 K-loop on the same tiles (k_gemm_nt3.hip): NONE / GELU_GRAD / GELU_ACT / MUL epilogues, same shapes as v2
int mfma_linear_fwd_v3(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& epi, hipStream_t s);

// dW[N,K] = dY[M,N]^T * A[M,K] (fp32, written)
int64_t mfma_wgrad_scratch_bytes(int64_t M, int N, int K);
// also db[N] = column sums of dY when db != null
int mfma_linear_wgrad(const bf16* dY, const bf16* A, int64_t M, int N, int K, float* dW, float* db, void* slab, hipStream_t s);

// two weight gradients that share M in one launch (one tile list, M-splits chosen for the sum); MFMA_UNSUPPORTED when either
// shape is not one of the 192 x 192 ring kernel's, M < 8192, a bias gradient is missing, or an A/B switch is set
int64_t mfma_wgrad_pair_scratch_bytes(int64_t M, int N0, int K0, int N1, int K1);
int mfma_linear_wgrad_pair(const bf16* dY0, const bf16* A0, int N0, int K0, float* dW0, float* db0, const bf16* dY1, const bf16* A1,
                           int N1, int K1, float* dW1, float* db1, int64_t M, void* slab, hipStream_t s);

// attention (k_attention_mfma.hip)
int mfma_attention_fwd(const bf16* qkv, int B, int T, int H, int hd, bf16* out, float* lse, hipStream_t s);
int mfma_attention_bwd(const bf16* qkv, const bf16* out, const bf16* d_out, const float* lse, int B, int T, int H, int hd,
                       bf16* d_qkv, hipStream_t s);

}  // namespace mae
