// Shared device helpers for libmae_hip.so (gfx950 only: wave = 64 lanes, no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>
#include "../../include/mae_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

namespace mae {

// ---- error plumbing (host) ---------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t err, const char* what, const char* file, int line);

#define MAE_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) return ::mae::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)
#define MAE_LAUNCH_CHECK() MAE_HIP(hipGetLastError())
#define MAE_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ::mae::set_error(__VA_ARGS__);           \
      return 1;                                \
    }                                          \
  } while (0)
#define MAE_TRY(expr)            \
  do {                           \
    int _r = (expr);             \
    if (_r) return _r;           \
  } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }
inline size_t dtype_size(int dt) { return dt == MAE_BF16 ? 2 : 4; }

// ---- device conversions --------------------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <class T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// 4-wide vector load/store in either activation type
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *reinterpret_cast<bf16x4*>(p) = r;
}

// streaming (non-temporal) forms: for tensors written once and consumed by a later kernel
__device__ __forceinline__ void store4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
__device__ __forceinline__ void store4_nt(bf16* p, f32x4 v) {
  bf16x4 r = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  __builtin_nontemporal_store(r, reinterpret_cast<bf16x4*>(p));
}
__device__ __forceinline__ f32x4 load4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ f32x4 load4_nt(const bf16* p) {
  const bf16x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a 256-thread block; `red` = 4 floats of LDS; result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// exact (erf) GELU and its derivative: torch.nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16-path GELU: erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7), one v_exp + one v_rcp shared by value and slope.
// Written once over a vector type so that the f32x4 form compiles to packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32:
// two lanes' worth per instruction); every operation is spelled out (explicit fma) so the scalar and the packed forms
// round identically.
template <class V>
__device__ __forceinline__ V vfma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
template <>
__device__ __forceinline__ float vfma<float>(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ float v_abs(float x) { return fabsf(x); }
__device__ __forceinline__ f32x4 v_abs(f32x4 x) { return f32x4{fabsf(x[0]), fabsf(x[1]), fabsf(x[2]), fabsf(x[3])}; }
__device__ __forceinline__ float v_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f32x4 v_rcp(f32x4 x) { return f32x4{__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1]), __builtin_amdgcn_rcpf(x[2]), __builtin_amdgcn_rcpf(x[3])}; }
__device__ __forceinline__ float v_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x4 v_exp2(f32x4 x) { return f32x4{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1]), __builtin_amdgcn_exp2f(x[2]), __builtin_amdgcn_exp2f(x[3])}; }
__device__ __forceinline__ float v_copysign(float m, float s) { return copysignf(m, s); }
__device__ __forceinline__ f32x4 v_copysign(f32x4 m, f32x4 s) { return f32x4{copysignf(m[0], s[0]), copysignf(m[1], s[1]), copysignf(m[2], s[2]), copysignf(m[3], s[3])}; }
template <class V>
__device__ __forceinline__ V vsplat(float c);
template <>
__device__ __forceinline__ float vsplat<float>(float c) { return c; }
template <>
__device__ __forceinline__ f32x4 vsplat<f32x4>(float c) { return f32x4{c, c, c, c}; }

template <class V>
__device__ __forceinline__ void gelu_fast_pair(V x, V& act, V& slope) {
  const V t = v_rcp(vfma(v_abs(x), vsplat<V>(0.70710678118654752440f * 0.3275911f), vsplat<V>(1.0f)));
  const V e = v_exp2((x * x) * vsplat<V>(-0.5f * 1.44269504088896340736f));  // exp(-x^2/2)
  V p = vfma(t, vsplat<V>(1.061405429f), vsplat<V>(-1.453152027f));
  p = vfma(p, t, vsplat<V>(1.421413741f));
  p = vfma(p, t, vsplat<V>(-0.284496736f));
  p = vfma(p, t, vsplat<V>(0.254829592f));
  p = p * t;
  const V h = vfma(p * e, vsplat<V>(-0.5f), vsplat<V>(0.5f));  // erf(|x|/sqrt 2) / 2, in [0, 0.5]
  const V cdf = v_copysign(h, x) + vsplat<V>(0.5f);
  act = x * cdf;
  slope = vfma(x * vsplat<V>(0.39894228040143267794f), e, cdf);
}
__device__ __forceinline__ float gelu_fast(float x) { float a, g; gelu_fast_pair(x, a, g); return a; }
__device__ __forceinline__ float gelu_grad_fast(float x) { float a, g; gelu_fast_pair(x, a, g); return g; }

}  // namespace mae
