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

// dW[N,K] = dY[M,N]^T * A[M,K] (fp32, written)
int64_t mfma_wgrad_scratch_bytes(int64_t M, int N, int K);
// also db[N] = column sums of dY when db != null
int mfma_linear_wgrad(const bf16* dY, const bf16* A, int64_t M, int N, int K, float* dW, float* db, void* slab, hipStream_t s);

// attention (k_attention_mfma.hip)
int mfma_attention_fwd(const bf16* qkv, int B, int T, int H, int hd, bf16* out, float* lse, hipStream_t s);
int mfma_attention_bwd(const bf16* qkv, const bf16* out, const bf16* d_out, const float* lse, int B, int T, int H, int hd,
                       bf16* d_qkv, hipStream_t s);

}  // namespace mae
