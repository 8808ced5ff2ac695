// bf16 MFMA GEMM kernels for gfx950 (placeholder: shapes not yet covered fall back to the any-shape kernel).
#include "gemm_mfma.h"

namespace mae {

int mfma_linear_fwd(const bf16*, const bf16*, int64_t, int, int, const Epi&, hipStream_t) { return MFMA_UNSUPPORTED; }
int64_t mfma_wgrad_scratch_bytes(int64_t, int, int) { return 0; }
int mfma_linear_wgrad(const bf16*, const bf16*, int64_t, int, int, float*, void*, hipStream_t) { return MFMA_UNSUPPORTED; }

}  // namespace mae
