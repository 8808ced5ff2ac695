// Masked-pixel MSE loss (+ its gradient), global grad-norm clip and AdamW.  All HBM-bound streaming kernels.
// Reference behaviour:
//   mse            == torch.nn.MSELoss() mean reduction                     (src/training/mae.py:40,48)
//   grad_norm      == torch.nn.utils.clip_grad_norm_(params, 1.0, L2)       (scripts/training/pretrain_mae.py:124-125)
//   adamw          == torch.optim.AdamW(lr, betas=(.9,.999), eps=1e-8, weight_decay) single group
//                                                                          (src/training/mae.py:59-65)
#include "kernels.h"
#include <cstdlib>

namespace mae {

constexpr int RED_BLOCKS = 1024;  // stage-1 partial count upper bound (scratch holds RED_BLOCKS floats + 8)

// ---------------------------------------------------------------------------------------------------
// MSE
// ---------------------------------------------------------------------------------------------------
template <class T, bool HAS_GRAD>
__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  int64_t n4, float gscale, float* __restrict__ partial,
                                                  T* __restrict__ dpred) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 d = load4(pred + i * 4) - load4(target + i * 4);
    acc += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    if (HAS_GRAD) store4(dpred + i * 4, d * gscale);
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// target read straight from the image: element e of row r=(b,j) is (py, px, c) of patch mask[b][j]-1.
// One thread per (row, py, px/VP): C-channel pixels for VP consecutive px.
template <class T, bool HAS_GRAD>
__global__ void __launch_bounds__(256) mse_images_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ images,
                                                         const int32_t* __restrict__ mask32, int64_t rows, int m, int C,
                                                         int img, int p, float gscale, float* __restrict__ partial,
                                                         T* __restrict__ dpred) {
  __shared__ float red[4];
  const int g = img / p, pp = p * p;
  const int64_t total = rows * pp;
  float acc = 0.f;
  for (int64_t u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t r = u / pp;
    const int q = (int)(u - r * pp);
    const int py = q / p, px = q - py * p;
    const int64_t b = r / m;
    int n = mask32[r] - 1;
    n = n < 0 ? 0 : n;
    const int ph = n / g, pw = n - ph * g;
    const float* src = images + (b * C * (int64_t)img + (ph * p + py)) * img + pw * p + px;
    const int64_t o = r * (int64_t)(pp * C) + q * C;
    for (int c = 0; c < C; ++c) {
      const float d = pred[o + c] - src[(int64_t)c * img * img];
      acc += d * d;
      if (HAS_GRAD) dpred[o + c] = from_f<T>(d * gscale);
    }
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) mean_finalize_kernel(const float* __restrict__ partial, int nb, float inv_n,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = acc * inv_n;
}

int launch_mse(const float* pred, const float* target, int64_t n, float grad_scale, float* loss, void* d_pred,
               int dpred_dt, float* scratch, hipStream_t s) {
  MAE_REQUIRE(pred && target && loss && scratch && n > 0 && n % 4 == 0, "mse: need n %% 4 == 0 and non-null buffers");
  const int64_t n4 = n / 4;
  const int grid = (int)std::min<int64_t>(cdiv(n4, 256), RED_BLOCKS);
  const float gs = grad_scale * 2.0f / (float)n;
  if (!d_pred)
    hipLaunchKernelGGL((mse_kernel<float, false>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (float*)nullptr);
  else if (dpred_dt == MAE_BF16)
    hipLaunchKernelGGL((mse_kernel<bf16, true>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (bf16*)d_pred);
  else
    hipLaunchKernelGGL((mse_kernel<float, true>), dim3(grid), dim3(256), 0, s, pred, target, n4, gs, scratch, (float*)d_pred);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 1.0f / (float)n, loss);
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_mse_from_images(const float* pred, const float* images, const int32_t* mask32, int B, int m, int C, int img,
                           int p, float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch,
                           hipStream_t s) {
  MAE_REQUIRE(pred && images && mask32 && loss && scratch && B > 0 && m > 0 && img % p == 0, "mse_from_images: bad arguments");
  {
    static const bool band = [] { const char* v = getenv("MAE_MSE_BAND"); return !v || v[0] != '0'; }();   // MAE_MSE_BAND=0: the per-pixel gather (A/B)
    const int r = band ? launch_mse_from_images_band_f32(pred, images, mask32, B, m, C, img, p, grad_scale, loss, d_pred, dpred_dt, scratch, s) : -1;
    if (r >= 0) return r;
  }
  const int64_t rows = (int64_t)B * m;
  const int64_t n = rows * p * p * C;
  const int grid = (int)std::min<int64_t>(cdiv(rows * p * p, 256), RED_BLOCKS);
  const float gs = grad_scale * 2.0f / (float)n;
  if (!d_pred)
    hipLaunchKernelGGL((mse_images_kernel<float, false>), dim3(grid), dim3(256), 0, s, pred, images, mask32, rows, m, C, img, p, gs, scratch, (float*)nullptr);
  else if (dpred_dt == MAE_BF16)
    hipLaunchKernelGGL((mse_images_kernel<bf16, true>), dim3(grid), dim3(256), 0, s, pred, images, mask32, rows, m, C, img, p, gs, scratch, (bf16*)d_pred);
  else
    hipLaunchKernelGGL((mse_images_kernel<float, true>), dim3(grid), dim3(256), 0, s, pred, images, mask32, rows, m, C, img, p, gs, scratch, (float*)d_pred);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 1.0f / (float)n, loss);
  MAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// global L2 norm of the flat gradient arena + clip coefficient
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = load4(g + i * 4);
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) norm_finalize_kernel(const float* __restrict__ partial, int nb, float max_norm,
                                                            float* __restrict__ stats) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    const float total = sqrtf(acc);
    stats[0] = total;
    stats[1] = fminf(max_norm / (total + 1e-6f), 1.0f);  // clip_grad_norm_: clamp(max_norm/(total+1e-6), max=1)
  }
}

// the two halves of launch_grad_norm for a gradient that is sharded over ranks: sum of squares of a range, then (after the ranks'
// sums were added) norm + clip coefficient from the total
__global__ void __launch_bounds__(256) sumsq_finalize_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = acc;
}
__global__ void clip_from_sumsq_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ stats) {
  const float total = sqrtf(sumsq[0]);
  stats[0] = total;
  stats[1] = fminf(max_norm / (total + 1e-6f), 1.0f);
}
int launch_grad_sumsq(const float* g, int64_t n, float* out, float* scratch, hipStream_t s) {
  MAE_REQUIRE(g && out && scratch && n >= 0 && n % 4 == 0, "grad_sumsq: need n %% 4 == 0 and non-null buffers");
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n / 4, 256), RED_BLOCKS));
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, g, n / 4, scratch);   // n == 0: every partial is zero
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_finalize_kernel, dim3(1), dim3(256), 0, s, scratch, grid, out);
  MAE_LAUNCH_CHECK();
  return 0;
}
int launch_clip_from_sumsq(const float* sumsq, float max_norm, float* stats, hipStream_t s) {
  MAE_REQUIRE(sumsq && stats, "clip_from_sumsq: null argument");
  hipLaunchKernelGGL(clip_from_sumsq_kernel, dim3(1), dim3(1), 0, s, sumsq, max_norm, stats);
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_grad_norm(const float* g, int64_t n, float max_norm, float* stats, float* scratch, hipStream_t s) {
  MAE_REQUIRE(g && stats && scratch && n > 0 && n % 4 == 0, "grad_norm: need n %% 4 == 0 and non-null buffers");
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), RED_BLOCKS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, g, n / 4, scratch);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, s, scratch, grid, max_norm, stats);
  MAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// AdamW, one pass over p/g/m/v (+ optional bf16 operand copy of the new parameters)
// ---------------------------------------------------------------------------------------------------
// EMA (I-JEPA target encoder, no counterpart in the reference): the first ema_n4 float4 of a second parameter arena follow the
// updated parameters, tgt = mom * tgt + (1 - mom) * p_new, in the same sweep (+ the bf16 operand copy of the target weights)
template <bool WBF, bool EMA>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n4, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, const float* __restrict__ stats,
                                                    bf16* __restrict__ wbf, float* __restrict__ tgt, bf16* __restrict__ tgt_wbf,
                                                    int64_t ema_n4, float mom) {
  const float coef = stats ? stats[1] : 1.0f;
  const float decay = 1.0f - lr * wd;
  const float step = lr * inv_bc1;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 pp = load4(p + i * 4);
    const f32x4 gg = load4(g + i * 4) * coef;
    f32x4 mm = load4(m + i * 4), vv = load4(v + i * 4);
    pp *= decay;
    mm = mm * b1 + gg * (1.0f - b1);
    vv = vv * b2 + gg * gg * (1.0f - b2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= step * (mm[j] / denom);
    }
    store4(p + i * 4, pp);
    store4(m + i * 4, mm);
    store4(v + i * 4, vv);
    if (WBF) store4(wbf + i * 4, pp);
    if (EMA && i < ema_n4) {
      const f32x4 tt = load4(tgt + i * 4) * mom + pp * (1.0f - mom);
      store4(tgt + i * 4, tt);
      if (WBF) store4(tgt_wbf + i * 4, tt);
    }
  }
}

int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, float bc1, float bc2, const float* stats, bf16* wbf, hipStream_t s, float* ema_target,
                 bf16* ema_wbf, int64_t ema_n, float ema_momentum) {
  MAE_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0, "adamw: need n %% 4 == 0 and non-null buffers");
  MAE_REQUIRE(bc1 > 0.f && bc2 > 0.f, "adamw: bias corrections must be positive (step >= 1)");
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), 256 * 16);
  const float inv_bc1 = 1.0f / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  MAE_REQUIRE(!ema_target || (ema_n > 0 && ema_n % 4 == 0 && ema_n <= n && (!wbf || ema_wbf)), "adamw: bad EMA range");
#define ADAMW(W, E) hipLaunchKernelGGL((adamw_kernel<W, E>), dim3(grid), dim3(256), 0, s, p, g, m, v, n / 4, lr, b1, b2, eps, wd, inv_bc1, inv_sqrt_bc2, stats, wbf, ema_target, ema_wbf, ema_n / 4, ema_momentum)
  if (wbf) { if (ema_target) ADAMW(true, true); else ADAMW(true, false); }
  else     { if (ema_target) ADAMW(false, true); else ADAMW(false, false); }
#undef ADAMW
  MAE_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n4) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) store4(dst + i * 4, load4(src + i * 4));
}

int launch_f32_to_bf16(const float* src, bf16* dst, int64_t n, hipStream_t s) {
  MAE_REQUIRE(src && dst && n > 0 && n % 4 == 0, "f32_to_bf16: need n %% 4 == 0");
  const int grid = (int)std::min<int64_t>(cdiv(n / 4, 256), 256 * 16);
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, s, src, dst, n / 4);
  MAE_LAUNCH_CHECK();
  return 0;
}

// 32x32 LDS-tiled transpose + cast: dst[c][r] = src[r][c]
__global__ void __launch_bounds__(256) transpose_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows,
                                                             int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * rows + r] = (bf16)tile[tx][i];
  }
}

// every GEMM weight in ONE launch: the matrix table travels as a kernel argument (<= 64 entries); block b finds its matrix
// by a scan over the cumulative tile counts.  64 x 64 tiles (TransposeTable::TILE): rows are read as float4 (256 B per row of a tile)
// and the transposed rows written as 4 bf16 per thread (128 B = one line per row of a tile); 32 x 32 tiles with 4-byte reads and
// 2-byte writes ran at 1.8 TB/s.  Matrices whose dims are not multiples of 4 take the element-wise path inside the same tile.
__global__ void __launch_bounds__(256) transpose_many_kernel(const float* __restrict__ params, bf16* __restrict__ tbase, TransposeTable tab) {
  constexpr int TL = TransposeTable::TILE;
  __shared__ float tile[TL][TL + 1];
  int i = 0;
  while (i + 1 < tab.n && (int)blockIdx.x >= tab.tile_begin[i + 1]) ++i;
  const int rows = tab.rows[i], cols = tab.cols[i];
  const int local = blockIdx.x - tab.tile_begin[i];
  const int tiles_c = (cols + TL - 1) / TL;
  const int r0 = (local / tiles_c) * TL, c0 = (local % tiles_c) * TL;
  const float* src = params + tab.src_off[i];
  bf16* dst = tbase + tab.dst_off[i];
  const int t16 = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 threads x 4 elements across a tile row, 16 rows per pass
  const bool vec = (rows & 3) == 0 && (cols & 3) == 0 && ((tab.src_off[i] | tab.dst_off[i]) & 3) == 0;
  if (vec) {
    for (int k = ty; k < TL; k += 16) {
      const int r = r0 + k, c = c0 + 4 * t16;
      const f32x4 v = (r < rows && c < cols) ? load4(src + (int64_t)r * cols + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      tile[k][4 * t16] = v[0]; tile[k][4 * t16 + 1] = v[1]; tile[k][4 * t16 + 2] = v[2]; tile[k][4 * t16 + 3] = v[3];
    }
    __syncthreads();
    for (int k = ty; k < TL; k += 16) {
      const int c = c0 + k, r = r0 + 4 * t16;
      if (c < cols && r < rows) store4(dst + (int64_t)c * rows + r, f32x4{tile[4 * t16][k], tile[4 * t16 + 1][k], tile[4 * t16 + 2][k], tile[4 * t16 + 3][k]});
    }
  } else {
    const int tx = threadIdx.x & 63, tz = threadIdx.x >> 6;
    for (int k = tz; k < TL; k += 4) {
      const int r = r0 + k, c = c0 + tx;
      tile[k][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = tz; k < TL; k += 4) {
      const int c = c0 + k, r = r0 + tx;
      if (c < cols && r < rows) dst[(int64_t)c * rows + r] = (bf16)tile[tx][k];
    }
  }
}

int launch_transpose_many(const float* params, bf16* tbase, const TransposeTable& tab, hipStream_t s) {
  MAE_REQUIRE(params && tbase && tab.n > 0 && tab.n <= TransposeTable::MAX, "transpose_many: bad table");
  hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)tab.total_tiles), dim3(256), 0, s, params, tbase, tab);
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_transpose_to_bf16(const float* src, bf16* dst, int rows, int cols, hipStream_t s) {
  MAE_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose_to_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((int)cdiv(cols, 32), (int)cdiv(rows, 32)), dim3(256), 0, s, src, dst, rows, cols);
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae

extern "C" int mae_mse_loss(const float* pred, const float* target, int64_t n, float grad_scale, float* loss,
                            float* d_pred, float* scratch, void* stream) {
  return mae::launch_mse(pred, target, n, grad_scale, loss, d_pred, MAE_F32, scratch, (hipStream_t)stream);
}
