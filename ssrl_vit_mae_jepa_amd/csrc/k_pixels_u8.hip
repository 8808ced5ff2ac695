// uint8 pixels read inside the engine: the three kernels that touch the image (visible-patch gather, masked-pixel MSE,
// target patchify) with ToTensor + Normalize(.5, .5) fused into the read (reference: src/data.py:15-24 builds
// transforms.ToTensor() -> x / 255, then Normalize(mean .5, std .5) -> (x - .5) / .5; here v = (u8 / 255 - 0.5) / 0.5 with
// an IEEE-rounded division, bit-identical to the torch fp32 expression).
//
// Data movement: a patch row of a uint8 image is p bytes (8 for ViT-S/8), far below a 128-byte HBM line, so a gather by
// patch would pull 16x the bytes it uses.  Instead one workgroup owns one BAND = one row of patches of one image
// (C x p x W bytes: 2.3 KB at 96 px / p 8, 10.5 KB at 224 px / p 16), streams it into LDS with coalesced 4-byte loads,
// then serves the band's visible (or masked) tokens from LDS.  Every image byte is fetched exactly once per kernel:
// 27.6 KB per image instead of the 110.6 KB fp32 image read as scattered 32-byte patch rows (3.9x / 2.3x over-fetch
// measured in round 1).  All HBM-bound, 55 MB per launch at batch 2000.
#include "kernels.h"

namespace mae {

namespace {

// The tokens of one band are listed in LDS behind the band image: n_tok slots, so that even an index list that names one
// band over and over (legal for a caller-made idx_keep) is served completely.

__device__ __forceinline__ float norm_u8(unsigned u) { return (__fdiv_rn((float)u, 255.0f) - 0.5f) / 0.5f; }

// band (b, ph) -> LDS image [C][p][W] bytes (W % 4 == 0)
__device__ __forceinline__ void load_band(const uint8_t* __restrict__ images, int64_t b, int ph, int C, int img, int p, unsigned* lds32) {
  const int wpr = img >> 2, words = C * p * wpr;
  for (int w = threadIdx.x; w < words; w += 256) {
    const int row = w / wpr, col = w - row * wpr;
    const int c = row / p, py = row - c * p;
    const unsigned* src = reinterpret_cast<const unsigned*>(images + ((b * C + c) * (int64_t)img + (ph * p + py)) * img);
    lds32[w] = src[col];
  }
}

// tokens of image b that lie in band ph -> list (j index, patch column); class token (t == 0) belongs to band 0 when with_cls
template <class I>
__device__ __forceinline__ void band_tokens(const I* __restrict__ tok, int64_t b, int n_tok, int g, int ph, bool with_cls, int* list, int* cnt) {
  for (int j = threadIdx.x; j < n_tok; j += 256) {
    const int t = (int)tok[b * n_tok + j];
    bool mine;
    int pw;
    if (t <= 0 && with_cls) { mine = ph == 0; pw = -1; }  // class token: a zero row
    else {  // idx - 1 clamped into the patch grid (the reference's clamp(idx_mask - 1, min=0), src/models/mae.py:91)
      const int n = t <= 0 ? 0 : (t - 1 < g * g ? t - 1 : g * g - 1); mine = n / g == ph; pw = n - (n / g) * g;
    }
    if (mine) {
      const int slot = atomicAdd(cnt, 1);
      list[slot] = (j << 8) | (pw & 0xff);
    }
  }
}

// the list in ascending order (entries are distinct): a kernel that SUMS over the band's tokens then adds them in an order that does
// not depend on which thread won which slot -- the loss comes out bit-identical from run to run
__device__ __forceinline__ void sort_band_list(const int* list, int n, int* sorted) {
  for (int i = threadIdx.x; i < n; i += 256) {
    const int v = list[i];
    int rank = 0;
    for (int o = 0; o < n; ++o) rank += list[o] < v;
    sorted[rank] = v;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// visible-patch gather: out[(b, j)][c*p*p + py*p + px] = normalised pixel, zero row for the class token
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) gather_patches_u8_kernel(const uint8_t* __restrict__ images, const int32_t* __restrict__ tok, int B,
                                                                int k, int C, int img, int p, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds32[];
  int* list = reinterpret_cast<int*>(lds32 + C * p * (img >> 2));
  __shared__ int cnt;
  const int g = img / p, P = C * p * p, p4 = p >> 2, units = C * p * p4;
  const uint8_t* band = reinterpret_cast<const uint8_t*>(lds32);
  for (int64_t bb = blockIdx.x; bb < (int64_t)B * g; bb += gridDim.x) {
    const int64_t b = bb / g;
    const int ph = (int)(bb - b * g);
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    load_band(images, b, ph, C, img, p, lds32);
    band_tokens(tok, b, k, g, ph, true, list, &cnt);
    __syncthreads();
    const int n = cnt;
    for (int it = threadIdx.x; it < n * units; it += 256) {
      const int li = it / units, u = it - li * units;
      const int j = list[li] >> 8, pw = list[li] & 0xff;
      const int c = u / (p * p4), rem = u - c * (p * p4);
      const int py = rem / p4, px = (rem - py * p4) << 2;
      T* dst = out + (b * k + j) * (int64_t)P + (c * p + py) * p + px;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pw != 0xff) {
        const unsigned w = *reinterpret_cast<const unsigned*>(band + (c * p + py) * img + pw * p + px);
        v = f32x4{norm_u8(w & 0xff), norm_u8((w >> 8) & 0xff), norm_u8((w >> 16) & 0xff), norm_u8(w >> 24)};
      }
      store4(dst, v);
    }
    __syncthreads();
  }
}

int launch_gather_patches_u8(const uint8_t* images, const int32_t* tok, int B, int k, int C, int img, int p, int dt, void* out,
                             hipStream_t s) {
  MAE_REQUIRE(images && tok && out && B > 0 && k > 0, "gather_patches(u8): bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0 && p % 4 == 0 && img / p <= 64, "uint8 images need patch_size %% 4 == 0 and at most 64 patches per side (got image %d, patch %d)", img, p);
  MAE_REQUIRE(C * p * img <= 96 * 1024 && k <= 8192, "uint8 images: one row of patches (%d bytes) does not fit the LDS band", C * p * img);
  const int grid = (int)std::min<int64_t>((int64_t)B * (img / p), 256 * 16);
  const size_t lds = (size_t)C * p * img + (size_t)k * 4;
  // band + token list may exceed the 64 KiB a kernel gets by default (up to 96 + 32 KiB are admitted above)
  if (lds > 64 * 1024) {
    MAE_HIP(hipFuncSetAttribute((const void*)gather_patches_u8_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MAE_HIP(hipFuncSetAttribute((const void*)gather_patches_u8_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (dt == MAE_BF16) hipLaunchKernelGGL((gather_patches_u8_kernel<bf16>), dim3(grid), dim3(256), lds, s, images, tok, B, k, C, img, p, (bf16*)out);
  else hipLaunchKernelGGL((gather_patches_u8_kernel<float>), dim3(grid), dim3(256), lds, s, images, tok, B, k, C, img, p, (float*)out);
  MAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// masked-pixel MSE against the image (target never materialised) + dpred; element order of a row: (py, px, c)
// ---------------------------------------------------------------------------------------------------
template <class T, bool HAS_GRAD>
__global__ void __launch_bounds__(256) mse_images_u8_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ images,
                                                            const int32_t* __restrict__ mask32, int B, int m, int C, int img, int p,
                                                            float gscale, float* __restrict__ partial, T* __restrict__ dpred) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds32[];
  int* list = reinterpret_cast<int*>(lds32 + C * p * (img >> 2));
  int* sorted = list + m;   // the band's token list in ascending order
  __shared__ int cnt;
  __shared__ float red[4];
  const int g = img / p, P = C * p * p, p4 = p >> 2, units = p * p4;
  const uint8_t* band = reinterpret_cast<const uint8_t*>(lds32);
  float acc = 0.f;
  for (int64_t bb = blockIdx.x; bb < (int64_t)B * g; bb += gridDim.x) {
    const int64_t b = bb / g;
    const int ph = (int)(bb - b * g);
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    load_band(images, b, ph, C, img, p, lds32);
    band_tokens(mask32, b, m, g, ph, false, list, &cnt);
    __syncthreads();
    const int n = cnt;
    sort_band_list(list, n, sorted);
    __syncthreads();
    for (int it = threadIdx.x; it < n * units; it += 256) {
      const int li = it / units, u = it - li * units;
      const int j = sorted[li] >> 8, pw = sorted[li] & 0xff;
      const int py = u / p4, px = (u - py * p4) << 2;
      const int64_t o = (b * m + j) * (int64_t)P + (py * p + px) * C;  // 4 pixels x C channels = 4C contiguous elements
      const uint8_t* src = band + py * img + pw * p + px;
      for (int v = 0; v < C; ++v) {
        const f32x4 pr = load4(pred + o + 4 * v);
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * v + i, pxi = e / C, c = e - pxi * C;
          d[i] = pr[i] - norm_u8(src[c * p * img + pxi]);
          acc += d[i] * d[i];
        }
        if (HAS_GRAD) store4(dpred + o + 4 * v, d * gscale);
      }
    }
    __syncthreads();
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) mean_finalize_u8_kernel(const float* __restrict__ partial, int nb, float inv_n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = acc * inv_n;
}

int launch_mse_from_images_u8(const float* pred, const uint8_t* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                              float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch, hipStream_t s) {
  MAE_REQUIRE(pred && images && mask32 && loss && scratch && B > 0 && m > 0, "mse_from_images(u8): bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0 && p % 4 == 0 && img / p <= 64 && C * p * img <= 96 * 1024 && m <= 8192, "uint8 images: unsupported geometry (image %d, patch %d)", img, p);
  const int64_t n = (int64_t)B * m * p * p * C;
  const int grid = (int)std::min<int64_t>((int64_t)B * (img / p), 1024);  // stage-1 partials: scratch holds 1024 floats + 8
  const float gs = grad_scale * 2.0f / (float)n;
  const size_t lds = (size_t)C * p * img + (size_t)m * 8;   // band + token list + its sorted copy
  if (lds > 64 * 1024) {
    MAE_HIP(hipFuncSetAttribute((const void*)mse_images_u8_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MAE_HIP(hipFuncSetAttribute((const void*)mse_images_u8_kernel<bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MAE_HIP(hipFuncSetAttribute((const void*)mse_images_u8_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (!d_pred)
    hipLaunchKernelGGL((mse_images_u8_kernel<float, false>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (float*)nullptr);
  else if (dpred_dt == MAE_BF16)
    hipLaunchKernelGGL((mse_images_u8_kernel<bf16, true>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (bf16*)d_pred);
  else
    hipLaunchKernelGGL((mse_images_u8_kernel<float, true>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (float*)d_pred);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_finalize_u8_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 1.0f / (float)n, loss);
  MAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// The same band walk for NORMALISED FLOAT images (the parity format): one workgroup streams one row of patches ([C][p][W] floats)
// into LDS with coalesced loads and serves the band's masked tokens from there.  The per-(row, pixel) gather it replaces pulled
// 759 MB for 221 MB of pixels at batch 2000 (32-byte patch rows out of 128-byte lines, PMC round 3).
// ---------------------------------------------------------------------------------------------------
template <class T, bool HAS_GRAD>
__global__ void __launch_bounds__(256) mse_images_band_f32_kernel(const float* __restrict__ pred, const float* __restrict__ images,
                                                                  const int32_t* __restrict__ mask32, int B, int m, int C, int img, int p,
                                                                  float gscale, float* __restrict__ partial, T* __restrict__ dpred) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds32[];
  float* band = reinterpret_cast<float*>(lds32);
  int* list = reinterpret_cast<int*>(lds32 + C * p * img);
  int* sorted = list + m;
  __shared__ int cnt;
  __shared__ float red[4];
  const int g = img / p, P = C * p * p, p4 = p >> 2, units = p * p4, w4 = img >> 2;
  float acc = 0.f;
  for (int64_t bb = blockIdx.x; bb < (int64_t)B * g; bb += gridDim.x) {
    const int64_t b = bb / g;
    const int ph = (int)(bb - b * g);
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int w = threadIdx.x; w < C * p * w4; w += 256) {   // 16 bytes per thread: a band row is W floats, contiguous in the image
      const int row = w / w4, col = w - row * w4;
      const int c = row / p, py = row - c * p;
      store4(band + row * img + col * 4, load4(images + ((b * C + c) * (int64_t)img + (ph * p + py)) * img + col * 4));
    }
    band_tokens(mask32, b, m, g, ph, false, list, &cnt);
    __syncthreads();
    const int n = cnt;
    sort_band_list(list, n, sorted);
    __syncthreads();
    for (int it = threadIdx.x; it < n * units; it += 256) {
      const int li = it / units, u = it - li * units;
      const int j = sorted[li] >> 8, pw = sorted[li] & 0xff;
      const int py = u / p4, px = (u - py * p4) << 2;
      const int64_t o = (b * m + j) * (int64_t)P + (py * p + px) * C;  // 4 pixels x C channels = 4C contiguous elements
      const float* src = band + py * img + pw * p + px;
      for (int v = 0; v < C; ++v) {
        const f32x4 pr = load4(pred + o + 4 * v);
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * v + i, pxi = e / C, c = e - pxi * C;
          d[i] = pr[i] - src[c * p * img + pxi];
          acc += d[i] * d[i];
        }
        if (HAS_GRAD) store4(dpred + o + 4 * v, d * gscale);
      }
    }
    __syncthreads();
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// returns -1 when the geometry does not fit the band walk (the caller then takes the per-pixel gather)
int launch_mse_from_images_band_f32(const float* pred, const float* images, const int32_t* mask32, int B, int m, int C, int img, int p,
                                    float grad_scale, float* loss, void* d_pred, int dpred_dt, float* scratch, hipStream_t s) {
  if (!(p > 0 && img % p == 0 && p % 4 == 0 && img / p <= 64 && (int64_t)C * p * img * 4 <= 96 * 1024 && m <= 8192)) return -1;
  if ((((uintptr_t)pred | (uintptr_t)images | (uintptr_t)d_pred) & 15) != 0 || (C * p * p) % 4 != 0) return -1;
  const int64_t n = (int64_t)B * m * p * p * C;
  const int grid = (int)std::min<int64_t>((int64_t)B * (img / p), 1024);  // stage-1 partials: scratch holds 1024 floats + 8
  const float gs = grad_scale * 2.0f / (float)n;
  const size_t lds = (size_t)C * p * img * 4 + (size_t)m * 8;
  MAE_HIP(hipFuncSetAttribute((const void*)mse_images_band_f32_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MAE_HIP(hipFuncSetAttribute((const void*)mse_images_band_f32_kernel<bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MAE_HIP(hipFuncSetAttribute((const void*)mse_images_band_f32_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (!d_pred)
    hipLaunchKernelGGL((mse_images_band_f32_kernel<float, false>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (float*)nullptr);
  else if (dpred_dt == MAE_BF16)
    hipLaunchKernelGGL((mse_images_band_f32_kernel<bf16, true>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (bf16*)d_pred);
  else
    hipLaunchKernelGGL((mse_images_band_f32_kernel<float, true>), dim3(grid), dim3(256), lds, s, pred, images, mask32, B, m, C, img, p, gs, scratch, (float*)d_pred);
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_finalize_u8_kernel, dim3(1), dim3(256), 0, s, scratch, grid, 1.0f / (float)n, loss);
  MAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// target (B*m, P) fp32 in (py, px, c) order for patch max(mask - 1, 0): utils.patchify + get_at_index on uint8 pixels
// ---------------------------------------------------------------------------------------------------
template <class I>
__global__ void __launch_bounds__(256) patchify_gather_u8_kernel(const uint8_t* __restrict__ images, const I* __restrict__ mask, int B, int m,
                                                                 int C, int img, int p, float* __restrict__ target) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds32[];
  int* list = reinterpret_cast<int*>(lds32 + C * p * (img >> 2));
  __shared__ int cnt;
  const int g = img / p, P = C * p * p, p4 = p >> 2, units = p * p4;
  const uint8_t* band = reinterpret_cast<const uint8_t*>(lds32);
  for (int64_t bb = blockIdx.x; bb < (int64_t)B * g; bb += gridDim.x) {
    const int64_t b = bb / g;
    const int ph = (int)(bb - b * g);
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    load_band(images, b, ph, C, img, p, lds32);
    // clamp(idx - 1, 0): a class-token id (0) reads patch 0, like the reference's clamp (it never occurs in idx_mask)
    for (int j = threadIdx.x; j < m; j += 256) {
      int n = (int)mask[b * m + j] - 1;
      n = n < 0 ? 0 : (n >= g * g ? g * g - 1 : n);
      if (n / g == ph) {
        const int slot = atomicAdd(&cnt, 1);
        list[slot] = (j << 8) | (n - (n / g) * g);
      }
    }
    __syncthreads();
    const int n = cnt;
    for (int it = threadIdx.x; it < n * units; it += 256) {
      const int li = it / units, u = it - li * units;
      const int j = list[li] >> 8, pw = list[li] & 0xff;
      const int py = u / p4, px = (u - py * p4) << 2;
      const int64_t o = (b * m + j) * (int64_t)P + (py * p + px) * C;
      const uint8_t* src = band + py * img + pw * p + px;
      for (int v = 0; v < C; ++v) {
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * v + i, pxi = e / C, c = e - pxi * C;
          d[i] = norm_u8(src[c * p * img + pxi]);
        }
        store4(target + o + 4 * v, d);
      }
    }
    __syncthreads();
  }
}

template <class I>
static int launch_patchify_u8(const uint8_t* images, const I* mask, int B, int m, int C, int img, int p, float* target, hipStream_t s) {
  MAE_REQUIRE(images && mask && target && B > 0 && m > 0, "patchify_gather(u8): bad arguments");
  MAE_REQUIRE(p > 0 && img % p == 0 && p % 4 == 0 && img / p <= 64 && C * p * img <= 96 * 1024, "uint8 images: unsupported geometry (image %d, patch %d)", img, p);
  MAE_REQUIRE(m <= 8192, "patchify_gather(u8): too many tokens per image");
  const int grid = (int)std::min<int64_t>((int64_t)B * (img / p), 256 * 16);
  const size_t lds = (size_t)C * p * img + (size_t)m * 4;
  if (lds > 64 * 1024) MAE_HIP(hipFuncSetAttribute((const void*)patchify_gather_u8_kernel<I>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((patchify_gather_u8_kernel<I>), dim3(grid), dim3(256), lds, s, images, mask, B, m, C, img, p, target);
  MAE_LAUNCH_CHECK();
  return 0;
}
int launch_patchify_gather_u8(const uint8_t* images, const int32_t* mask32, int B, int m, int C, int img, int p, float* target, hipStream_t s) {
  return launch_patchify_u8(images, mask32, B, m, C, img, p, target, s);
}
int launch_patchify_gather_u8_i64(const uint8_t* images, const int64_t* mask64, int B, int m, int C, int img, int p, float* target, hipStream_t s) {
  return launch_patchify_u8(images, mask64, B, m, C, img, p, target, s);
}


// ---------------------------------------------------------------------------------------------------
// The augmentation step in front of the path (src/data.py:15-20: RandomResizedCrop(size, scale=(0.8, 1.0)) +
// RandomHorizontalFlip on the PIL uint8 image, before ToTensor): per image a crop box (top, left, h, w) and a flip flag are
// drawn on the host side (ssrl_vit_mae_jepa_amd/data.py: random_resized_crop_params) and this kernel resamples the box to
// the full size x size frame, bilinear, pixel centres aligned as F.grid_sample(align_corners=False, padding_mode="border")
// does, rounds to uint8 like the PIL resize and mirrors columns when flip is set.  uint8 in, uint8 out: the batch stays
// 1 B per pixel all the way into the engine's pixel readers.  One thread per output pixel, all channels.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) augment_crop_flip_u8_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ params,
                                                                   int B, int C, int S, uint8_t* __restrict__ out) {
  const int64_t total = (int64_t)B * S * S;
  const float inv2s = 0.5f / (float)S;
  for (int64_t u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
    const int b = (int)(u / (S * S));
    const int r = (int)(u - (int64_t)b * S * S);
    const int oy = r / S, ox = r - oy * S;
    const int top = params[5 * b + 0], left = params[5 * b + 1], h = params[5 * b + 2], w = params[5 * b + 3], flip = params[5 * b + 4];
    // source coordinate of the output pixel centre: left + w * (2 ox' + 1) / (2 S) - 0.5, ox' mirrored under a flip
    const int oxm = flip ? S - 1 - ox : ox;
    float fx = (float)left + (float)w * (float)(2 * oxm + 1) * inv2s - 0.5f;
    float fy = (float)top + (float)h * (float)(2 * oy + 1) * inv2s - 0.5f;
    fx = fminf(fmaxf(fx, 0.f), (float)(S - 1));   // border padding: coordinates are clipped before interpolation
    fy = fminf(fmaxf(fy, 0.f), (float)(S - 1));
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    const int x1 = x0 + 1 < S ? x0 + 1 : S - 1, y1 = y0 + 1 < S ? y0 + 1 : S - 1;
    const float ax = fx - (float)x0, ay = fy - (float)y0;
    const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
    const uint8_t* src = in + (int64_t)b * C * S * S;
    uint8_t* dst = out + (int64_t)b * C * S * S + oy * S + ox;
    for (int c = 0; c < C; ++c) {
      const uint8_t* pc = src + (int64_t)c * S * S;
      const float v = w00 * (float)pc[y0 * S + x0] + w01 * (float)pc[y0 * S + x1] + w10 * (float)pc[y1 * S + x0] + w11 * (float)pc[y1 * S + x1];
      dst[(int64_t)c * S * S] = (uint8_t)(int)fminf(fmaxf(rintf(v), 0.f), 255.f);
    }
  }
}

int launch_augment_crop_flip_u8(const uint8_t* in, const int32_t* params, int B, int C, int S, uint8_t* out, hipStream_t s) {
  MAE_REQUIRE(in && params && out && B > 0 && C > 0 && S > 0, "augment_crop_flip_u8: bad arguments");
  MAE_REQUIRE(in != out, "augment_crop_flip_u8: in-place resampling is not supported");
  const int grid = (int)std::min<int64_t>(cdiv((int64_t)B * S * S, 256), 256 * 32);
  hipLaunchKernelGGL(augment_crop_flip_u8_kernel, dim3(grid), dim3(256), 0, s, in, params, B, C, S, out);
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae

extern "C" int mae_augment_crop_flip_u8(const uint8_t* images, const int32_t* params, int32_t batch, int32_t in_chans, int32_t image_size,
                                        uint8_t* out, void* stream) {
  return mae::launch_augment_crop_flip_u8(images, params, batch, in_chans, image_size, out, (hipStream_t)stream);
}
