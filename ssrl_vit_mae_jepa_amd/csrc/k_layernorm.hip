// LayerNorm(eps) forward / backward over fp32 residual-stream rows; LPR lanes per row (64, 32 or 16: a wave walks 1, 2 or 4
// rows at a time so that 384- and 192-wide rows = 96 / 48 float4 fill every lane; at 64 lanes per row a quarter of the lanes
// idled), the row held in registers, mean/variance by shuffles inside the lane group.  HBM-bound: fwd reads 4*D B/row,
// writes sizeof(T)*D.
// Reference behaviour: torch.nn.LayerNorm(dim, eps=1e-6) as timm VisionTransformer / lightly MAEDecoderTIMM
// instantiate it (norm1/norm2/norm/decoder_norm), computed in fp32 (autocast keeps layer_norm in fp32).
#include "kernels.h"

#ifndef MAE_LN_NT
#define MAE_LN_NT 5  // bit 0: fp32 residual-stream / dx stores stream (measured -0.4 ms per step); bit 1: the bf16 operand copies too (no gain); bit 2: the read-once inputs (x, branch, dy) are loaded non-temporally (-0.2 ms, mostly in the GEMMs that follow)
#endif
#if MAE_LN_NT & 1
#define LN_ST_A store4_nt
#else
#define LN_ST_A store4
#endif
#if MAE_LN_NT & 2
#define LN_ST_B store4_nt
#else
#define LN_ST_B store4
#endif
#if MAE_LN_NT & 4
#define LN_LD load4_nt
#else
#define LN_LD load4
#endif

namespace mae {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {  // sum over the LPR consecutive lanes that share a row
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// lanes per row for a row of d4 float4: the widest group that wastes no more lanes than a narrower one would
static int ln_lanes_per_row(int d4) {
  int best = 64;
  double best_fill = (double)d4 / (64.0 * (double)cdiv(d4, 64));
  for (int lpr : {32, 16}) {
    if (cdiv(d4, lpr) > 4) continue;
    const double fill = (double)d4 / ((double)lpr * (double)cdiv(d4, lpr));
    if (fill > best_fill + 1e-9) { best_fill = fill; best = lpr; }
  }
  return best;
}

// NV = float4 vectors per lane: covers dim <= 4*LPR*NV
// ADD: the LayerNorm input is x + branch (the residual add of the preceding attention / MLP branch, whose GEMM then
// keeps the plain bf16 epilogue); the sum is written to x_out as the new fp32 residual stream.
template <class T, int NV, bool ADD, int LPR>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, const T* __restrict__ branch,
                                                            float* __restrict__ x_out, const int32_t* __restrict__ row_map,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, int64_t rows, int dim, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  constexpr int RPW = 64 / LPR;  // rows per wave and iteration
  const int lane = threadIdx.x & 63, li = lane & (LPR - 1), sub = lane / LPR;
  const int D4 = dim >> 2;
  const float inv_d = 1.0f / (float)dim;
  for (int64_t r0 = (blockIdx.x * 4ll + (threadIdx.x >> 6)) * RPW; r0 < rows; r0 += (int64_t)gridDim.x * 4 * RPW) {
    const bool live = r0 + sub < rows;                 // lane groups past the last row compute on it and store nothing
    const int64_t r = live ? r0 + sub : rows - 1;
    const int64_t src = row_map ? (int64_t)row_map[r] : r;
    const float* px = x + src * dim;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = li + LPR * i;
      if (c < D4) {
        v[i] = LN_LD(px + c * 4);
        if (ADD) {
          v[i] += LN_LD(branch + src * dim + c * 4);
          if (live) LN_ST_A(x_out + src * dim + c * 4, v[i]);
        }
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = group_sum<LPR>(sum) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = li + LPR * i;
      if (c < D4) {
        const f32x4 d = v[i] - mean;
        sq += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      }
    }
    const float var = group_sum<LPR>(sq) * inv_d;
    const float rstd = rsqrtf(var + eps);
    T* py = y + r * dim;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = li + LPR * i;
      if (c < D4 && live) {
        const f32x4 o = (v[i] - mean) * rstd * load4(gamma + c * 4) + load4(beta + c * 4);
        LN_ST_B(py + c * 4, o);
      }
    }
    if (li == 0 && live) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
    }
  }
}

int launch_layernorm_fwd(const float* x, const void* branch, float* x_out, const int32_t* row_map, const float* gamma,
                         const float* beta, float eps, int64_t rows, int dim, int y_dt, void* y, float* mean, float* rstd,
                         hipStream_t s) {
  MAE_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0, "layernorm_fwd: null buffer");
  MAE_REQUIRE(!branch || x_out, "layernorm_fwd: the fused residual add needs x_out");
  MAE_REQUIRE(dim % 4 == 0 && dim >= 4 && dim <= 1024, "layernorm: dim %d must be a multiple of 4 in [4, 1024]", dim);
  const int lpr = ln_lanes_per_row(dim / 4), nv = (int)cdiv(dim / 4, lpr);
  const int grid = (int)std::min<int64_t>(cdiv(rows, 4 * (64 / lpr)), 256 * 32);
#define LN(T, NV, ADD, LPR) hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV, ADD, LPR>), dim3(grid), dim3(256), 0, s, x, (const T*)branch, x_out, row_map, gamma, beta, eps, rows, dim, (T*)y, mean, rstd)
#define LN_NV(T, ADD, LPR) switch (nv) { case 1: LN(T, 1, ADD, LPR); break; case 2: LN(T, 2, ADD, LPR); break; case 3: LN(T, 3, ADD, LPR); break; default: LN(T, 4, ADD, LPR); }
#define LN_LPR(T, ADD) switch (lpr) { case 16: LN_NV(T, ADD, 16) break; case 32: LN_NV(T, ADD, 32) break; default: LN_NV(T, ADD, 64) }
  if (y_dt == MAE_BF16) { if (branch) { LN_LPR(bf16, true) } else { LN_LPR(bf16, false) } }
  else { if (branch) { LN_LPR(float, true) } else { LN_LPR(float, false) } }
#undef LN_LPR
#undef LN_NV
#undef LN
  MAE_LAUNCH_CHECK();
  return 0;
}

// Backward.  xhat = (x-mean)*rstd, g = dy*gamma:
//   dx = rstd * (g - mean(g) - xhat*mean(g*xhat));  dgamma = sum_rows dy*xhat;  dbeta = sum_rows dy
// Each wave keeps per-lane column partials of dgamma/dbeta across the rows it walks; the block's 4 waves are
// combined through LDS into partial[block][2][dim]; a second kernel adds the blocks in order (deterministic).
template <class T, int NV, int LPR>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                            const int32_t* __restrict__ row_map,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int64_t rows, int dim,
                                                            int accumulate, float* __restrict__ dx_io, T* __restrict__ dx_copy,
                                                            float* __restrict__ partial) {
  constexpr int RPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves x RPW lane groups][2][dim]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & (LPR - 1), sub = lane / LPR;
  const int D4 = dim >> 2;
  const float inv_d = 1.0f / (float)dim;
  f32x4 gam[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = li + LPR * i;
    gam[i] = (c < D4) ? load4(gamma + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t r0 = (blockIdx.x * 4ll + wave) * RPW; r0 < rows; r0 += (int64_t)gridDim.x * 4 * RPW) {
    const bool live = r0 + sub < rows;
    const int64_t r = live ? r0 + sub : rows - 1;
    const int64_t src = row_map ? (int64_t)row_map[r] : r;
    const float mu = mean[r], rs = rstd[r];
    f32x4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = li + LPR * i;
      if (c < D4) {
        const f32x4 d = LN_LD(dy + r * dim + c * 4);
        xh[i] = (LN_LD(x + src * dim + c * 4) - mu) * rs;
        g[i] = d * gam[i];
        if (live) {
          dg[i] += d * xh[i];
          db[i] += d;
        }
        s1 += g[i][0] + g[i][1] + g[i][2] + g[i][3];
        s2 += g[i][0] * xh[i][0] + g[i][1] * xh[i][1] + g[i][2] * xh[i][2] + g[i][3] * xh[i][3];
      }
    }
    s1 = group_sum<LPR>(s1) * inv_d;
    s2 = group_sum<LPR>(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = li + LPR * i;
      if (c < D4 && live) {
        f32x4 d = (g[i] - s1 - xh[i] * s2) * rs;
        float* pd = dx_io + src * dim + c * 4;
        if (accumulate) d += load4(pd);
        LN_ST_A(pd, d);
        if (dx_copy) LN_ST_B(dx_copy + src * dim + c * 4, d);
      }
    }
  }
  // block reduce of the column partials: 4 * RPW lane groups hold partials of the same columns
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = li + LPR * i;
    if (c < D4) {
      store4(red + ((wave * RPW + sub) * 2 + 0) * dim + c * 4, dg[i]);
      store4(red + ((wave * RPW + sub) * 2 + 1) * dim + c * 4, db[i]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * dim; c += 256) {
    float v = 0.f;
#pragma unroll
    for (int gidx = 0; gidx < 4 * RPW; ++gidx) v += red[gidx * 2 * dim + c];
    partial[(int64_t)blockIdx.x * 2 * dim + c] = v;
  }
}

int launch_layernorm_bwd(const void* dy, int dy_dt, const float* x, const int32_t* row_map, const float* gamma,
                         const float* mean, const float* rstd, int64_t rows, int dim, int accumulate, float* dx_io,
                         void* dx_copy, float* dgamma, float* dbeta, float* partial, hipStream_t s, PartialsTable* defer) {
  MAE_REQUIRE(dy && x && gamma && mean && rstd && dx_io && dgamma && dbeta && partial && rows > 0, "layernorm_bwd: null buffer");
  MAE_REQUIRE(dim % 4 == 0 && dim >= 4 && dim <= 1024, "layernorm: dim %d must be a multiple of 4 in [4, 1024]", dim);
  if (defer && defer->n == PartialsTable::MAX) {  // deep models: run what is queued before `partial` (slot 0 again) is rewritten
    MAE_TRY(launch_sum_partials_many(*defer, s));
    defer->n = 0;
  }
  const int lpr = ln_lanes_per_row(dim / 4), nv = (int)cdiv(dim / 4, lpr);
  const int grid = (int)std::min<int64_t>(cdiv(rows, 4 * (64 / lpr)), LN_BWD_MAX_BLOCKS);
  const size_t lds = (size_t)8 * (64 / lpr) * dim * sizeof(float);
#define LNB(T, NV, LPR) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV, LPR>), dim3(grid), dim3(256), lds, s, (const T*)dy, x, row_map, gamma, mean, rstd, rows, dim, accumulate, dx_io, (T*)dx_copy, partial)
#define LNB_NV(T, LPR) switch (nv) { case 1: LNB(T, 1, LPR); break; case 2: LNB(T, 2, LPR); break; case 3: LNB(T, 3, LPR); break; default: LNB(T, 4, LPR); }
#define LNB_LPR(T) switch (lpr) { case 16: LNB_NV(T, 16) break; case 32: LNB_NV(T, 32) break; default: LNB_NV(T, 64) }
  if (dy_dt == MAE_BF16) { LNB_LPR(bf16) } else { LNB_LPR(float) }
#undef LNB_LPR
#undef LNB_NV
#undef LNB
  MAE_LAUNCH_CHECK();
  if (defer) {
    const int i = defer->n++;
    if (i == 0) defer->block_begin[0] = 0;
    defer->partial[i] = partial; defer->out0[i] = dgamma; defer->out1[i] = dbeta;
    defer->G[i] = grid; defer->C[i] = 2 * dim; defer->split[i] = dim;
    defer->block_begin[i + 1] = defer->block_begin[i] + (int)cdiv(2 * dim, 32);
    return 0;
  }
  return launch_sum_partials(partial, grid, 2 * dim, dgamma, dbeta, dim, s);
}

}  // namespace mae

extern "C" int mae_layernorm_fwd(const float* x, const int32_t* row_map, const float* gamma, const float* beta, float eps,
                                 int64_t rows, int32_t dim, int32_t y_dtype, void* y, float* mean, float* rstd,
                                 void* stream) {
  return mae::launch_layernorm_fwd(x, nullptr, nullptr, row_map, gamma, beta, eps, rows, dim, y_dtype, y, mean, rstd, (hipStream_t)stream);
}

extern "C" int mae_add_layernorm_fwd(const float* x, const void* branch, float* x_out, const int32_t* row_map, const float* gamma,
                                     const float* beta, float eps, int64_t rows, int32_t dim, int32_t y_dtype, void* y, float* mean,
                                     float* rstd, void* stream) {
  MAE_REQUIRE(branch && x_out, "mae_add_layernorm_fwd: null branch/x_out");
  return mae::launch_layernorm_fwd(x, branch, x_out, row_map, gamma, beta, eps, rows, dim, y_dtype, y, mean, rstd, (hipStream_t)stream);
}

extern "C" int mae_layernorm_bwd(const void* dy, int32_t dy_dtype, const float* x, const int32_t* row_map,
                                 const float* gamma, const float* mean, const float* rstd, int64_t rows, int32_t dim,
                                 int32_t accumulate, float* dx_io, void* dx_copy, float* dgamma, float* dbeta,
                                 float* partial, void* stream) {
  return mae::launch_layernorm_bwd(dy, dy_dtype, x, row_map, gamma, mean, rstd, rows, dim, accumulate, dx_io, dx_copy,
                                   dgamma, dbeta, partial, (hipStream_t)stream);
}
