// I-JEPA pieces that the MAE path does not have (BASELINE.json configs[2] and [4]).  The reference holds NO I-JEPA code
// ("JEPA" appears in README.md:1,9 and pyproject.toml:2 only): these kernels follow the specification written down in
// DESIGN.md from the I-JEPA paper (Assran et al. 2023: context encoder, EMA target encoder, narrow predictor fed with the
// context tokens plus one mask token per target position, latent regression loss), on the reference's own ViT pieces.
//   predictor_assemble(+bwd)  rows of the predictor input: nblk sequences per image = [k context tokens | m mask tokens],
//                             each with the position row of its token id
//   build_tail_row_map        the last m rows of every sequence (what the predictor's norm + projection run on)
//   rows_from_tokens          token ids -> rows of the target encoder's (B * N) output matrix
//   smooth_l1                 latent loss alternative to MSE (torch.nn.functional.smooth_l1_loss, beta = 1)
// All HBM-bound, a few hundred MB per launch at most.
#include "kernels.h"

namespace mae {

// keep[b][j] = j + 1: every patch token, no class token (the target encoder sees the whole image)
__global__ void iota_tokens_kernel(int32_t* __restrict__ keep, int64_t n, int N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) keep[i] = (int32_t)(i % N) + 1;
}
int launch_iota_tokens(int32_t* keep32, int B, int N, hipStream_t s) {
  MAE_REQUIRE(keep32 && B > 0 && N > 0, "iota_tokens: bad arguments");
  const int64_t n = (int64_t)B * N;
  hipLaunchKernelGGL(iota_tokens_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s, keep32, n, N);
  MAE_LAUNCH_CHECK();
  return 0;
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill(float* p, float v, int64_t n, hipStream_t s) {
  MAE_REQUIRE(p && n > 0, "fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s, p, v, n);
  MAE_LAUNCH_CHECK();
  return 0;
}

// rows[seq * m + j] = seq * T + (T - m) + j
__global__ void tail_row_map_kernel(int32_t* __restrict__ rows, int64_t n, int T, int m) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t seq = i / m;
    rows[i] = (int32_t)(seq * T + (T - m) + (i - seq * m));
  }
}
int launch_build_tail_row_map(int seqs, int T, int m, int32_t* rows, hipStream_t s) {
  MAE_REQUIRE(rows && seqs > 0 && m > 0 && m <= T, "tail_row_map: bad arguments");
  MAE_REQUIRE((int64_t)seqs * T < (1ll << 31), "tail_row_map: sequences * length overflows int32");
  const int64_t n = (int64_t)seqs * m;
  hipLaunchKernelGGL(tail_row_map_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s, rows, n, T, m);
  MAE_LAUNCH_CHECK();
  return 0;
}

// rows[i] = b * N + clamp(tok[i] - 1, 0, N - 1), b = i / per_image
__global__ void rows_from_tokens_kernel(const int32_t* __restrict__ tok, int64_t n, int per_image, int N, int32_t* __restrict__ rows) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int t = tok[i] - 1;
    t = t < 0 ? 0 : (t >= N ? N - 1 : t);
    rows[i] = (int32_t)((i / per_image) * N + t);
  }
}
int launch_rows_from_tokens(const int32_t* tok, int B, int per_image, int N, int32_t* rows, hipStream_t s) {
  MAE_REQUIRE(tok && rows && B > 0 && per_image > 0 && (int64_t)B * N < (1ll << 31), "rows_from_tokens: bad arguments");
  const int64_t n = (int64_t)B * per_image;
  hipLaunchKernelGGL(rows_from_tokens_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s, tok, n, per_image, N, rows);
  MAE_LAUNCH_CHECK();
  return 0;
}

// out[(b, blk, t)] = (t < k ? xdec[b*k + t] + pos[ctx[b][t]] : mask_token + pos[tgt[b][blk][t - k]])      fp32 rows of Dd
template <class T>
__global__ void __launch_bounds__(256) predictor_assemble_kernel(const T* __restrict__ xdec, const int32_t* __restrict__ ctx,
                                                                 const int32_t* __restrict__ tgt, const float* __restrict__ mask_token,
                                                                 const float* __restrict__ pos, int64_t rows, int k, int nblk, int m, int L,
                                                                 int D4, float* __restrict__ out) {
  const int rpb = 256 / D4, ro = threadIdx.x / D4, d = (threadIdx.x - ro * D4) * 4;
  if (ro >= rpb) return;
  const int Tq = k + m;
  for (int64_t r = (int64_t)blockIdx.x * rpb + ro; r < rows; r += (int64_t)gridDim.x * rpb) {
    const int seq = (int)((uint32_t)r / (uint32_t)Tq);  // rows < 2^31 is checked by the launcher
    const int t = (int)(r - (int64_t)seq * Tq);
    const int b = seq / nblk;
    int tokid;
    f32x4 v;
    if (t < k) { tokid = ctx[(int64_t)b * k + t]; v = load4(xdec + ((int64_t)b * k + t) * (D4 * 4) + d); }
    else { tokid = tgt[(int64_t)seq * m + (t - k)]; v = load4(mask_token + d); }
    tokid = tokid < 0 ? 0 : (tokid >= L ? L - 1 : tokid);  // ids are range-checked on the host; never fault here
    v += load4(pos + (int64_t)tokid * (D4 * 4) + d);
    store4_nt(out + r * (D4 * 4) + d, v);
  }
}
int launch_predictor_assemble(const void* xdec, int dt, const int32_t* ctx32, const int32_t* tgt32, const float* mask_token, const float* pos,
                              int B, int k, int nblk, int m, int L, int Dd, float* out, hipStream_t s) {
  MAE_REQUIRE(xdec && ctx32 && tgt32 && mask_token && pos && out && Dd % 4 == 0 && Dd / 4 <= 256, "predictor_assemble: bad arguments");
  const int64_t rows = (int64_t)B * nblk * (k + m);
  MAE_REQUIRE(rows < (1ll << 31), "predictor_assemble: B * nblk * (k + m) < 2^31");
  const int grid = (int)std::min<int64_t>(cdiv(rows, 256 / (Dd / 4)), 256 * 16);
  if (dt == MAE_BF16)
    hipLaunchKernelGGL((predictor_assemble_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)xdec, ctx32, tgt32, mask_token, pos, rows, k, nblk, m, L, Dd / 4, out);
  else
    hipLaunchKernelGGL((predictor_assemble_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)xdec, ctx32, tgt32, mask_token, pos, rows, k, nblk, m, L, Dd / 4, out);
  MAE_LAUNCH_CHECK();
  return 0;
}

// adjoint: d_xdec[b*k + t] = sum over the nblk copies of context row t; partial[block] = column sums of the mask-token rows
template <class T>
__global__ void __launch_bounds__(256) predictor_assemble_bwd_kernel(const float* __restrict__ dx, int64_t ctx_rows, int k, int nblk, int m,
                                                                     int D, T* __restrict__ d_xdec, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [RPI][D]
  const int D4 = D / 4, RPI = 256 / D4, Tq = k + m;
  const int ro = threadIdx.x / D4, d0 = (threadIdx.x - ro * D4) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ro < RPI) {
    // context rows: r = b*k + t
    for (int64_t r = (int64_t)blockIdx.x * RPI + ro; r < ctx_rows; r += (int64_t)gridDim.x * RPI) {
      const int64_t b = r / k;
      const int t = (int)(r - b * k);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int blk = 0; blk < nblk; ++blk) v += load4(dx + ((b * nblk + blk) * Tq + t) * (int64_t)D + d0);
      store4(d_xdec + r * D + d0, v);
    }
    // mask-token rows: q = seq*m + j
    const int64_t mask_rows = (ctx_rows / k) * nblk * m;
    for (int64_t q = (int64_t)blockIdx.x * RPI + ro; q < mask_rows; q += (int64_t)gridDim.x * RPI) {
      const int64_t seq = q / m;
      acc += load4(dx + (seq * Tq + k + (q - seq * m)) * (int64_t)D + d0);
    }
    store4(red + ro * D + d0, acc);
  }
  __syncthreads();
  if (ro == 0) {
    for (int i = 1; i < RPI; ++i) acc += load4(red + i * D + d0);
    store4(partial + (int64_t)blockIdx.x * D + d0, acc);
  }
}
int launch_predictor_assemble_bwd(const float* dx, int B, int k, int nblk, int m, int Dd, int dt, void* d_xdec, float* d_mask_token,
                                  float* partial, hipStream_t s) {
  MAE_REQUIRE(dx && d_xdec && d_mask_token && partial && Dd % 4 == 0 && Dd <= 1024, "predictor_assemble_bwd: bad arguments");
  const int64_t ctx_rows = (int64_t)B * k;
  const int RPI = 256 / (Dd / 4);
  const int G = (int)std::min<int64_t>(cdiv(std::max<int64_t>(ctx_rows, (int64_t)B * nblk * m), RPI), 512);  // partial: >= 512 * Dd floats
  const size_t lds = (size_t)RPI * Dd * sizeof(float);
  if (dt == MAE_BF16)
    hipLaunchKernelGGL((predictor_assemble_bwd_kernel<bf16>), dim3(G), dim3(256), lds, s, dx, ctx_rows, k, nblk, m, Dd, (bf16*)d_xdec, partial);
  else
    hipLaunchKernelGGL((predictor_assemble_bwd_kernel<float>), dim3(G), dim3(256), lds, s, dx, ctx_rows, k, nblk, m, Dd, (float*)d_xdec, partial);
  MAE_LAUNCH_CHECK();
  return launch_sum_partials(partial, G, Dd, d_mask_token, nullptr, Dd, s);
}

// smooth L1 (beta = 1): mean over n of (|d| < 1 ? 0.5 d^2 : |d| - 0.5); d_pred = grad_scale * clamp(d, -1, 1) / n
template <class T, bool HAS_GRAD>
__global__ void __launch_bounds__(256) smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n4, float gscale,
                                                        float* __restrict__ partial, T* __restrict__ dpred) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 d = load4(pred + i * 4) - load4(target + i * 4);
    f32x4 g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = fabsf(d[j]);
      acc += a < 1.0f ? 0.5f * d[j] * d[j] : a - 0.5f;
      g[j] = fminf(fmaxf(d[j], -1.0f), 1.0f) * gscale;
    }
    if (HAS_GRAD) store4(dpred + i * 4, g);
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) mean_finalize_l1_kernel(const float* __restrict__ partial, int nb, float inv_n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = acc * inv_n;
}
int launch_smooth_l1(const float* pred, const float* target, int64_t n, float grad_scale, float* loss, void* d_pred, int dpred_dt,
                     float* scratch, hipStream_t s) {
  MAE_REQUIRE(pred && target && loss && scratch && n > 0 && n % 4 == 0, "smooth_l1: need n %% 4 == 0 and non-null buffers");
  const int64_t n4 = n / 4;
  const int grid = (int)std::min<int64_t>(cdiv(n4, 256), 1024);
  const float gs = grad_scale / (float)n;
  if (!d_pred) hipLaunchKernelGGL((smooth_l1_kernel<float, false>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (float*)nullptr);
  else if (dpred_dt == MAE_BF16) hipLaunchKernelGGL((smooth_l1_kernel<bf16, true>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (bf16*)d_pred);
  else hipLaunchKernelGGL((smooth_l1_kernel<float, true>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (float*)d_pred);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_finalize_l1_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 1.0f / (float)n, loss);
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae
