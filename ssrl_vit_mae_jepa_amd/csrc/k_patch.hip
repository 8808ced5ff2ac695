// Pixel <-> token data movement: visible-patch gather (im2col of the kept patches only), token assembly
// (cls / position table), decoder scatter with mask tokens, target patchify-gather.  All HBM-bound.
// Reference behaviour:
//   gather_patches + patch GEMM + assemble_visible == timm PatchEmbed conv (k = s = patch) -> cat(cls) -> +pos_embed
//     -> gather(idx_keep)           (lightly MaskedVisionTransformerTIMM.preprocess, via src/models/mae.py:55)
//   decoder_assemble == repeat_token(mask_token) -> set_at_index(idx_keep, x_decode) -> + decoder_pos_embed
//                                   (src/models/mae.py:61-71)
//   patchify_gather == utils.patchify + get_at_index(clamp(idx_mask-1, 0))   (src/models/mae.py:90-92)
#include "kernels.h"

namespace mae {

// ---------------------------------------------------------------------------------------------------
// visible-patch gather: row (b,j) <- patch tok[b][j]-1 in conv-weight order (c, py, px)
// ---------------------------------------------------------------------------------------------------
template <class T, int VEC>
__global__ void __launch_bounds__(256) gather_patches_kernel(const float* __restrict__ images,
                                                             const int32_t* __restrict__ tok, int64_t rows, int k,
                                                             int C, int img, int p, T* __restrict__ out) {
  const int g = img / p;
  const int pv = p / VEC;                 // vector units per patch row
  const int units_per_row = C * p * pv;   // per token row
  const int64_t total = rows * units_per_row;
  for (int64_t u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t r = u / units_per_row;
    int rem = (int)(u - r * units_per_row);
    const int c = rem / (p * pv);
    rem -= c * p * pv;
    const int py = rem / pv;
    const int px = (rem - py * pv) * VEC;
    const int t = tok[r];
    const int64_t b = r / k;
    T* dst = out + r * (int64_t)(C * p * p) + (c * p + py) * p + px;
    if (t <= 0) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = from_f<T>(0.f);
    } else {
      const int n = t - 1, ph = n / g, pw = n - ph * g;
      const float* src = images + ((b * C + c) * (int64_t)img + (ph * p + py)) * img + pw * p + px;
      if (VEC == 4) {
        store4(dst, load4(src));
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = from_f<T>(src[i]);
      }
    }
  }
}

int launch_gather_patches(const float* images, const int32_t* tok, int B, int k, int C, int img, int p, int dt,
                          void* out, hipStream_t s) {
  MAE_REQUIRE(images && tok && out && B > 0 && k > 0, "gather_patches: bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0, "gather_patches: image_size %d not divisible by patch_size %d", img, p);
  const int64_t rows = (int64_t)B * k;
  const bool vec = (p % 4 == 0) && (img % 4 == 0);
  const int64_t units = rows * C * p * (vec ? p / 4 : p);
  const int grid = (int)std::min<int64_t>(cdiv(units, 256), 256 * 16);
#define GP(T, V) hipLaunchKernelGGL((gather_patches_kernel<T, V>), dim3(grid), dim3(256), 0, s, images, tok, rows, k, C, img, p, (T*)out)
  if (dt == MAE_BF16) { if (vec) GP(bf16, 4); else GP(bf16, 1); }
  else                { if (vec) GP(float, 4); else GP(float, 1); }
#undef GP
  MAE_LAUNCH_CHECK();
  return 0;
}

// x[r] = (t == 0 ? cls : x[r]) + pos[t], fp32 in place, D % 4 == 0
__global__ void __launch_bounds__(256) assemble_visible_kernel(float* __restrict__ x, const int32_t* __restrict__ tok,
                                                               const float* __restrict__ cls,
                                                               const float* __restrict__ pos, int64_t rows, int D4) {
  const int64_t total = rows * D4;
  for (int64_t u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t r = u / D4;
    const int d = (int)(u - r * D4) * 4;
    const int t = tok[r];
    float* px = x + r * (int64_t)D4 * 4 + d;
    f32x4 v = (t == 0) ? load4(cls + d) : load4(px);
    v += load4(pos + (int64_t)t * D4 * 4 + d);
    store4(px, v);
  }
}

int launch_assemble_visible(float* x, const int32_t* tok, const float* cls, const float* pos, int64_t rows, int D,
                            hipStream_t s) {
  MAE_REQUIRE(x && tok && cls && pos && rows > 0 && D % 4 == 0, "assemble_visible: bad arguments (D %% 4 must be 0)");
  const int grid = (int)std::min<int64_t>(cdiv(rows * (D / 4), 256), 256 * 16);
  hipLaunchKernelGGL(assemble_visible_kernel, dim3(grid), dim3(256), 0, s, x, tok, cls, pos, rows, D / 4);
  MAE_LAUNCH_CHECK();
  return 0;
}

// Second stage of every two-stage column reduction: 32 columns per block (8 lanes x float4), 32 row lanes, fixed order.
__global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ partial, int G, int C,
                                                           float* __restrict__ out0, float* __restrict__ out1, int split) {
  __shared__ __attribute__((aligned(16))) float red[32][36];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 32 + cl * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int i = rl; i < G; i += 32) acc += load4(partial + (int64_t)i * C + c);
  store4(&red[rl][cl * 4], acc);
  __syncthreads();
  if (threadIdx.x < 32) {
    const int cc = blockIdx.x * 32 + threadIdx.x;
    if (cc < C) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) v += red[j][threadIdx.x];
      if (cc < split) out0[cc] = v; else out1[cc - split] = v;
    }
  }
}

// the same for a table of reductions: block -> (entry, 32-column group)
__global__ void __launch_bounds__(256) sum_partials_many_kernel(const PartialsTable tab) {
  __shared__ __attribute__((aligned(16))) float red[32][36];
  int e = 0;
  while (e + 1 < tab.n && (int)blockIdx.x >= tab.block_begin[e + 1]) ++e;
  const int blk = blockIdx.x - tab.block_begin[e];
  const float* __restrict__ partial = tab.partial[e];
  const int G = tab.G[e], C = tab.C[e], split = tab.split[e];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blk * 32 + cl * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int i = rl; i < G; i += 32) acc += load4(partial + (int64_t)i * C + c);
  store4(&red[rl][cl * 4], acc);
  __syncthreads();
  if (threadIdx.x < 32) {
    const int cc = blk * 32 + threadIdx.x;
    if (cc < C) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) v += red[j][threadIdx.x];
      if (cc < split) tab.out0[e][cc] = v; else tab.out1[e][cc - split] = v;
    }
  }
}

int launch_sum_partials_many(const PartialsTable& tab, hipStream_t s) {
  if (tab.n == 0) return 0;
  MAE_REQUIRE(tab.n <= PartialsTable::MAX, "sum_partials_many: %d entries", tab.n);
  hipLaunchKernelGGL(sum_partials_many_kernel, dim3((unsigned)tab.block_begin[tab.n]), dim3(256), 0, s, tab);
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_sum_partials(const float* partial, int G, int C, float* out0, float* out1, int split, hipStream_t s) {
  MAE_REQUIRE(partial && out0 && G > 0 && C > 0 && C % 4 == 0 && (split >= C || out1), "sum_partials: bad arguments (C %% 4 == 0)");
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(C, 32)), dim3(256), 0, s, partial, G, C, out0, out1, split);
  MAE_LAUNCH_CHECK();
  return 0;
}

// Row-walk helper shared by the two "split" kernels: thread (ro, c4) owns 4 columns of row offset ro; a block
// covers RPI = 256 / (D/4) rows per iteration and G blocks stride the matrix.  Column sums of the selected rows
// are reduced over ro in LDS and written to partial[block][D].
constexpr int SPLIT_BLOCKS = 512;

template <class T>
__global__ void __launch_bounds__(256) visible_grad_split_kernel(const float* __restrict__ dx,
                                                                 const int32_t* __restrict__ tok, int64_t rows, int D,
                                                                 T* __restrict__ dtok, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [RPI][D]
  const int D4 = D / 4, RPI = 256 / D4;
  const int ro = threadIdx.x / D4, d0 = (threadIdx.x - ro * D4) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ro < RPI) {
    constexpr int U = 4;   // rows in flight per thread (see decoder_assemble_bwd_kernel); same summation order
    const int64_t stride = (int64_t)gridDim.x * RPI;
    for (int64_t r0 = (int64_t)blockIdx.x * RPI + ro; r0 < rows; r0 += U * stride) {
      f32x4 v[U];
      bool is_cls[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        const bool live = r < rows;
        v[u] = live ? load4(dx + r * D + d0) : f32x4{0.f, 0.f, 0.f, 0.f};
        is_cls[u] = live && tok[r] == 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        if (is_cls[u]) acc += v[u];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) store4(dtok + r * D + d0, is_cls[u] ? z : v[u]);
      }
    }
    store4(red + ro * D + d0, acc);
  }
  __syncthreads();
  if (ro == 0) {
    for (int i = 1; i < RPI; ++i) acc += load4(red + i * D + d0);
    store4(partial + (int64_t)blockIdx.x * D + d0, acc);
  }
}

int launch_visible_grad_split(const float* dx, const int32_t* tok, int64_t rows, int D, int dt, void* dtok, float* dcls,
                              float* partial, hipStream_t s) {
  MAE_REQUIRE(dx && tok && dtok && dcls && partial && rows > 0 && D % 4 == 0 && D <= 1024,
              "visible_grad_split: bad arguments (need D %% 4 == 0, D <= 1024)");
  const int RPI = 256 / (D / 4);
  const int G = (int)std::min<int64_t>(cdiv(rows, RPI), SPLIT_BLOCKS);
  const size_t lds = (size_t)RPI * D * sizeof(float);
  if (dt == MAE_BF16)
    hipLaunchKernelGGL((visible_grad_split_kernel<bf16>), dim3(G), dim3(256), lds, s, dx, tok, rows, D, (bf16*)dtok, partial);
  else
    hipLaunchKernelGGL((visible_grad_split_kernel<float>), dim3(G), dim3(256), lds, s, dx, tok, rows, D, (float*)dtok, partial);
  MAE_LAUNCH_CHECK();
  return launch_sum_partials(partial, G, D, dcls, nullptr, D, s);
}

// ---------------------------------------------------------------------------------------------------
// decoder input assembly and its adjoint
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) decoder_assemble_kernel(const T* __restrict__ xdec,
                                                               const int32_t* __restrict__ inv,
                                                               const float* __restrict__ mask_token,
                                                               const float* __restrict__ pos, int64_t rows, int k, int L,
                                                               int D4, float* __restrict__ out) {
  // D4 threads per row, 256 / D4 rows per block pass: 32-bit index math only (a 64-bit divide per element made this
  // kernel run at half the HBM rate)
  const int rpb = 256 / D4, ro = threadIdx.x / D4, d = (threadIdx.x - ro * D4) * 4;
  if (ro >= rpb) return;
  // four rows per thread and pass: the row's index, then its source row, are two dependent round trips -- with one row in flight per
  // thread the kernel ran at 3.2 TB/s
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * rpb;
  for (int64_t r0 = (int64_t)blockIdx.x * rpb + ro; r0 < rows; r0 += U * stride) {
    int j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t r = r0 + u * stride; j[u] = r < rows ? inv[r] : -1; }
    f32x4 v[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride < rows ? r0 + u * stride : rows - 1;
      const int b = (int)((uint32_t)r / (uint32_t)L);  // rows < 2^31 is checked by the launcher
      const int t = (int)(r - (int64_t)b * L);
      v[u] = (j[u] >= 0) ? load4(xdec + ((int64_t)b * k + j[u]) * (D4 * 4) + d) : load4(mask_token + d);
      pz[u] = load4(pos + t * (D4 * 4) + d);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      if (r < rows) store4(out + r * (D4 * 4) + d, v[u] + pz[u]);   // (ordinary stores: 74.6 us against 77.1 non-temporal)
    }
  }
}

int launch_decoder_assemble(const void* xdec, int dt, const int32_t* inv, const float* mask_token, const float* pos,
                            int B, int k, int L, int Dd, float* out, hipStream_t s) {
  MAE_REQUIRE(xdec && inv && mask_token && pos && out && Dd % 4 == 0, "decoder_assemble: bad arguments");
  const int64_t rows = (int64_t)B * L;
  MAE_REQUIRE(Dd / 4 <= 256 && rows < (1ll << 31), "decoder_assemble: Dd <= 1024 and B * L < 2^31");
  const int grid = (int)std::min<int64_t>(cdiv(rows, 256 / (Dd / 4)), 256 * 16);
  if (dt == MAE_BF16)
    hipLaunchKernelGGL((decoder_assemble_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)xdec, inv, mask_token, pos, rows, k, L, Dd / 4, out);
  else
    hipLaunchKernelGGL((decoder_assemble_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)xdec, inv, mask_token, pos, rows, k, L, Dd / 4, out);
  MAE_LAUNCH_CHECK();
  return 0;
}

template <class T>
__global__ void __launch_bounds__(256) decoder_assemble_bwd_kernel(const float* __restrict__ dx,
                                                                   const int32_t* __restrict__ inv, int64_t rows, int k,
                                                                   int L, int D, T* __restrict__ d_xdec,
                                                                   float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [RPI][D]
  const int D4 = D / 4, RPI = 256 / D4;
  const int ro = threadIdx.x / D4, d0 = (threadIdx.x - ro * D4) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ro < RPI) {
    // four rows per thread and pass in flight (512 blocks x one 16-byte load per thread ran at 2.6 TB/s); the masked rows are added in
    // the same order as before (u ascending inside a pass = r ascending)
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * RPI;
    for (int64_t r0 = (int64_t)blockIdx.x * RPI + ro; r0 < rows; r0 += U * stride) {
      f32x4 v[U];
      int j[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        const bool live = r < rows;
        v[u] = live ? load4(dx + r * D + d0) : f32x4{0.f, 0.f, 0.f, 0.f};
        j[u] = live ? inv[r] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * stride;
        if (j[u] < 0) {
          acc += v[u];   // rows past the end contribute zeros
        } else {
          const int64_t b = (uint32_t)r / (uint32_t)L;  // rows < 2^31 (checked by the launcher)
          store4(d_xdec + (b * k + j[u]) * (int64_t)D + d0, v[u]);
        }
      }
    }
    store4(red + ro * D + d0, acc);
  }
  __syncthreads();
  if (ro == 0) {
    for (int i = 1; i < RPI; ++i) acc += load4(red + i * D + d0);
    store4(partial + (int64_t)blockIdx.x * D + d0, acc);
  }
}

int launch_decoder_assemble_bwd(const float* dx, const int32_t* inv, const int32_t* keep, int B, int k, int L, int Dd,
                                int dt, void* d_xdec, float* d_mask_token, float* partial, hipStream_t s) {
  (void)keep;
  MAE_REQUIRE(dx && inv && d_xdec && d_mask_token && partial && Dd % 4 == 0 && Dd <= 1024,
              "decoder_assemble_bwd: bad arguments (need Dd %% 4 == 0, Dd <= 1024)");
  const int64_t rows = (int64_t)B * L;
  MAE_REQUIRE(rows < (1ll << 31), "decoder_assemble_bwd: B * L < 2^31");
  const int RPI = 256 / (Dd / 4);
  const int G = (int)std::min<int64_t>(cdiv(rows, RPI), SPLIT_BLOCKS);
  const size_t lds = (size_t)RPI * Dd * sizeof(float);
  if (dt == MAE_BF16)
    hipLaunchKernelGGL((decoder_assemble_bwd_kernel<bf16>), dim3(G), dim3(256), lds, s, dx, inv, rows, k, L, Dd, (bf16*)d_xdec, partial);
  else
    hipLaunchKernelGGL((decoder_assemble_bwd_kernel<float>), dim3(G), dim3(256), lds, s, dx, inv, rows, k, L, Dd, (float*)d_xdec, partial);
  MAE_LAUNCH_CHECK();
  return launch_sum_partials(partial, G, Dd, d_mask_token, nullptr, Dd, s);
}

// ---------------------------------------------------------------------------------------------------
// pixel targets: target[(b,j)][(py*p + px)*C + c] = images[b][c][ph*p+py][pw*p+px], patch = max(mask-1, 0)
// one thread per (row, py, px): reads C strided pixels, writes C contiguous floats
// ---------------------------------------------------------------------------------------------------
template <class I>
__global__ void __launch_bounds__(256) patchify_gather_kernel(const float* __restrict__ images,
                                                              const I* __restrict__ mask32, int64_t rows, int m,
                                                              int C, int img, int p, float* __restrict__ target) {
  const int g = img / p, pp = p * p;
  const int64_t total = rows * pp;
  for (int64_t u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int64_t r = u / pp;
    const int q = (int)(u - r * pp);
    const int py = q / p, px = q - py * p;
    const int64_t b = r / m;
    int n = (int)mask32[r] - 1;
    n = n < 0 ? 0 : (n >= g * g ? g * g - 1 : n);
    const int ph = n / g, pw = n - ph * g;
    const float* src = images + (b * C * (int64_t)img + (ph * p + py)) * img + pw * p + px;
    float* dst = target + r * (int64_t)(pp * C) + q * C;
    for (int c = 0; c < C; ++c) dst[c] = src[(int64_t)c * img * img];
  }
}

int launch_patchify_gather(const float* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                           float* target, hipStream_t s) {
  MAE_REQUIRE(images && mask32 && target && B > 0 && m > 0, "patchify_gather: bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0, "patchify: image_size %d not divisible by patch_size %d", img, p);
  const int64_t rows = (int64_t)B * m;
  const int grid = (int)std::min<int64_t>(cdiv(rows * p * p, 256), 256 * 16);
  hipLaunchKernelGGL((patchify_gather_kernel<int32_t>), dim3(grid), dim3(256), 0, s, images, mask32, rows, m, C, img, p, target);
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_patchify_gather_i64(const float* images, const int64_t* mask64, int B, int m, int C, int img, int p,
                               float* target, hipStream_t s) {
  MAE_REQUIRE(images && mask64 && target && B > 0 && m > 0, "patchify_gather: bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0, "patchify: image_size %d not divisible by patch_size %d", img, p);
  const int64_t rows = (int64_t)B * m;
  const int grid = (int)std::min<int64_t>(cdiv(rows * p * p, 256), 256 * 16);
  hipLaunchKernelGGL((patchify_gather_kernel<int64_t>), dim3(grid), dim3(256), 0, s, images, mask64, rows, m, C, img, p, target);
  MAE_LAUNCH_CHECK();
  return 0;
}

template <class S, class D_>
__global__ void __launch_bounds__(256) cast_kernel(const S* __restrict__ src, D_* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = from_f<D_>(to_f(src[i]));
}

int launch_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, hipStream_t s) {
  MAE_REQUIRE(src && dst && n > 0, "cast: bad arguments");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 256 * 16);
  if (src_dt == MAE_F32 && dst_dt == MAE_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16*)dst, n);
  else if (src_dt == MAE_BF16 && dst_dt == MAE_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (float*)dst, n);
  else if (src_dt == MAE_F32 && dst_dt == MAE_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n);
  MAE_LAUNCH_CHECK();
  return 0;
}

// out[r] = x[r] + pos[r mod L]  (fp32, D/4 float4 per row): the `x + decoder_pos_embed` of decoder.decode()
__global__ void add_rows_pos_kernel(const float* __restrict__ x, const float* __restrict__ pos, int64_t rows, int L, int D4,
                                    float* __restrict__ out) {
  const int64_t n = rows * D4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D4;
    const int c = (int)(i - r * D4);
    store4(out + i * 4, load4(x + i * 4) + load4(pos + ((r % L) * D4 + c) * 4));
  }
}
int launch_add_rows_pos(const float* x, const float* pos, int64_t rows, int L, int D, float* out, hipStream_t s) {
  MAE_REQUIRE(x && pos && out && rows > 0 && L > 0 && D % 4 == 0, "add_rows_pos: bad arguments");
  const int grid = (int)std::min<int64_t>(cdiv(rows * (D / 4), 256), 256 * 16);
  hipLaunchKernelGGL(add_rows_pos_kernel, dim3(grid), dim3(256), 0, s, x, pos, rows, L, D / 4, out);
  MAE_LAUNCH_CHECK();
  return 0;
}

// dst[i] (dt) += src[i] (fp32): an extra upstream gradient joins the engine's own one at a tensor boundary
template <class T>
__global__ void add_into_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f<T>(to_f(dst[i]) + src[i]);
}
int launch_add_into(const float* src, void* dst, int dst_dt, int64_t n, hipStream_t s) {
  MAE_REQUIRE(src && dst && n > 0, "add_into: bad arguments");
  const int grid = (int)std::min<int64_t>(cdiv(n, 256), 256 * 16);
  if (dst_dt == MAE_BF16) hipLaunchKernelGGL((add_into_kernel<bf16>), dim3(grid), dim3(256), 0, s, src, (bf16*)dst, n);
  else hipLaunchKernelGGL((add_into_kernel<float>), dim3(grid), dim3(256), 0, s, src, (float*)dst, n);
  MAE_LAUNCH_CHECK();
  return 0;
}

// The backward pass of the decoder starts from a residual gradient that is zero on every row the prediction head never saw: the
// visible rows (inv[row] >= 0, MAE) or the first T - m rows of every sequence (inv == nullptr, I-JEPA's predictor).  Only those
// rows are cleared here (a quarter of the rows at mask ratio 0.75); the decoder_norm backward then writes the others.
template <class T>
__global__ void __launch_bounds__(256) zero_unpredicted_rows_kernel(const int32_t* __restrict__ inv, int64_t rows, int Tseq, int m, int D4,
                                                                    float* __restrict__ dres, T* __restrict__ dres_c) {
  const int lanes = D4 <= 64 ? 64 : 128;   // threads per row: one float4 each, two passes at most for D <= 1024
  const int rpb = 256 / lanes, sub = threadIdx.x / lanes, l = threadIdx.x % lanes;
  for (int64_t row = (int64_t)blockIdx.x * rpb + sub; row < rows; row += (int64_t)gridDim.x * rpb) {
    const bool clear = inv ? inv[row] >= 0 : (int)(row % Tseq) < Tseq - m;
    if (!clear) continue;
    for (int c = l; c < D4; c += lanes) {
      store4(dres + row * (int64_t)D4 * 4 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f});
      store4(dres_c + row * (int64_t)D4 * 4 + c * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
}

int launch_zero_unpredicted_rows(const int32_t* inv, int64_t rows, int Tseq, int m, int D, int act, float* dres, void* dres_c, hipStream_t s) {
  MAE_REQUIRE(dres && dres_c && rows > 0 && D % 4 == 0 && Tseq > 0 && m >= 0 && m <= Tseq, "zero_unpredicted_rows: bad arguments");
  const int rpb = D / 4 <= 64 ? 4 : 2;
  const int grid = (int)std::min<int64_t>(cdiv(rows, rpb), 8192);
  if (act == MAE_BF16)
    hipLaunchKernelGGL(zero_unpredicted_rows_kernel<bf16>, dim3(grid), dim3(256), 0, s, inv, rows, Tseq, m, D / 4, dres, reinterpret_cast<bf16*>(dres_c));
  else
    hipLaunchKernelGGL(zero_unpredicted_rows_kernel<float>, dim3(grid), dim3(256), 0, s, inv, rows, Tseq, m, D / 4, dres, reinterpret_cast<float*>(dres_c));
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae

extern "C" int mae_patchify_gather(const void* images, int32_t image_dtype, const int64_t* idx_mask, int32_t batch, int32_t in_chans,
                                   int32_t image_size, int32_t patch_size, int32_t num_mask, float* target,
                                   void* stream) {
  MAE_REQUIRE(image_dtype == MAE_F32 || image_dtype == MAE_U8, "mae_patchify_gather: image_dtype must be MAE_F32 or MAE_U8 (got %d)", image_dtype);
  if (image_dtype == MAE_U8)
    return mae::launch_patchify_gather_u8_i64((const uint8_t*)images, idx_mask, batch, num_mask, in_chans, image_size, patch_size, target,
                                              (hipStream_t)stream);
  return mae::launch_patchify_gather_i64((const float*)images, idx_mask, batch, num_mask, in_chans, image_size, patch_size, target,
                                         (hipStream_t)stream);
}
