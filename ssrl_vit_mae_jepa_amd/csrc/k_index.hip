// Mask generation and token-index maps.
// Reference behaviour: lightly utils.random_token_mask as called at src/models/mae.py:79-83
// (noise[:,0] = -1; argsort ascending; keep = first k, mask = rest).  Integer outputs are bit-exact:
// rank(t) = #{j : n[j] < n[t] or (n[j] == n[t] and j < t)}  == position of t in a stable ascending sort.
#include "kernels.h"

namespace mae {

// One 256-thread workgroup per row.  L <= 4096 keys live in LDS; L^2 compares per row (145^2 here).
__global__ void __launch_bounds__(256) mask_rank_kernel(const float* __restrict__ noise, int L, int k,
                                                        int64_t* __restrict__ keep64, int64_t* __restrict__ mask64,
                                                        int32_t* __restrict__ keep32, int32_t* __restrict__ mask32) {
  extern __shared__ __attribute__((aligned(16))) float keys[];
  const int b = blockIdx.x;
  const float* row = noise + (int64_t)b * L;
  for (int t = threadIdx.x; t < L; t += 256) keys[t] = (t == 0) ? -1.0f : row[t];
  __syncthreads();
  const int m = L - k;
  for (int t = threadIdx.x; t < L; t += 256) {
    const float v = keys[t];
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float u = keys[j];  // LDS broadcast
      rank += (u < v) || (u == v && j < t);
    }
    if (rank < k) {
      if (keep64) keep64[(int64_t)b * k + rank] = t;
      if (keep32) keep32[(int64_t)b * k + rank] = t;
    } else {
      if (mask64) mask64[(int64_t)b * m + (rank - k)] = t;
      if (mask32) mask32[(int64_t)b * m + (rank - k)] = t;
    }
  }
}

int launch_mask_from_noise(const float* noise, int B, int L, int k, int64_t* keep64, int64_t* mask64,
                           int32_t* keep32, int32_t* mask32, hipStream_t s) {
  MAE_REQUIRE(noise && B > 0 && L > 0 && L <= 4096, "mask_from_noise: need 0 < seq_len <= 4096 (got %d), batch %d", L, B);
  MAE_REQUIRE(k >= 1 && k <= L, "mask_from_noise: num_keep %d out of [1, %d]", k, L);
  hipLaunchKernelGGL(mask_rank_kernel, dim3(B), dim3(256), L * sizeof(float), s, noise, L, k, keep64, mask64, keep32,
                     mask32);
  MAE_LAUNCH_CHECK();
  return 0;
}

__global__ void idx_to_i32_kernel(const int64_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (int32_t)src[i];
}
__global__ void idx_to_i64_kernel(const int32_t* __restrict__ src, int64_t* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

int launch_idx_to_i32(const int64_t* src, int32_t* dst, int64_t n, hipStream_t s) {
  MAE_REQUIRE(src && dst && n > 0, "idx_to_i32: bad arguments");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  hipLaunchKernelGGL(idx_to_i32_kernel, dim3(grid), dim3(256), 0, s, src, dst, n);
  MAE_LAUNCH_CHECK();
  return 0;
}
int launch_idx_to_i64(const int32_t* src, int64_t* dst, int64_t n, hipStream_t s) {
  MAE_REQUIRE(src && dst && n > 0, "idx_to_i64: bad arguments");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  hipLaunchKernelGGL(idx_to_i64_kernel, dim3(grid), dim3(256), 0, s, src, dst, n);
  MAE_LAUNCH_CHECK();
  return 0;
}

// inv[b][t] = -1, then inv[b][keep[b][j]] = j.  One workgroup per image.
__global__ void __launch_bounds__(256) build_inverse_kernel(const int32_t* __restrict__ keep, int k, int L,
                                                            int32_t* __restrict__ inv) {
  const int b = blockIdx.x;
  int32_t* row = inv + (int64_t)b * L;
  for (int t = threadIdx.x; t < L; t += 256) row[t] = -1;
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 256) {
    const int t = keep[(int64_t)b * k + j];
    if (t >= 0 && t < L) row[t] = j;  // out-of-range ids are rejected on the host; never fault here
  }
}

int launch_build_inverse(const int32_t* keep32, int B, int k, int L, int32_t* inv, hipStream_t s) {
  MAE_REQUIRE(keep32 && inv && B > 0 && k > 0 && k <= L, "build_inverse: bad arguments");
  hipLaunchKernelGGL(build_inverse_kernel, dim3(B), dim3(256), 0, s, keep32, k, L, inv);
  MAE_LAUNCH_CHECK();
  return 0;
}

__global__ void build_row_map_kernel(const int32_t* __restrict__ idx, int64_t n, int n_per, int L,
                                     int32_t* __restrict__ rows) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / n_per;
    int t = idx[i];
    t = t < 0 ? 0 : (t >= L ? L - 1 : t);
    rows[i] = (int32_t)(b * L + t);
  }
}

int launch_build_row_map(const int32_t* idx32, int B, int n_per, int L, int32_t* rows, hipStream_t s) {
  MAE_REQUIRE(idx32 && rows && B > 0 && n_per > 0, "build_row_map: bad arguments");
  MAE_REQUIRE((int64_t)B * L < (1ll << 31), "build_row_map: batch*seq_len overflows int32");
  const int64_t n = (int64_t)B * n_per;
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 2048);
  hipLaunchKernelGGL(build_row_map_kernel, dim3(grid), dim3(256), 0, s, idx32, n, n_per, L, rows);
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae

extern "C" int mae_mask_from_noise(const float* noise, int32_t batch, int32_t seq_len, int32_t num_keep,
                                   int64_t* idx_keep, int64_t* idx_mask, void* stream) {
  MAE_REQUIRE(idx_keep && (idx_mask || num_keep == seq_len), "mae_mask_from_noise: null output");
  return mae::launch_mask_from_noise(noise, batch, seq_len, num_keep, idx_keep, idx_mask, nullptr, nullptr,
                                     (hipStream_t)stream);
}
