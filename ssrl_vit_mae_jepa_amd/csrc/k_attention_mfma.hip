// bf16 MFMA attention kernels for gfx950 (placeholder: falls back to the any-length kernel).
#include "gemm_mfma.h"

namespace mae {

int mfma_attention_fwd(const bf16*, int, int, int, int, bf16*, float*, hipStream_t) { return MFMA_UNSUPPORTED; }
int mfma_attention_bwd(const bf16*, const bf16*, const bf16*, const float*, int, int, int, int, bf16*, hipStream_t) {
  return MFMA_UNSUPPORTED;
}

}  // namespace mae
