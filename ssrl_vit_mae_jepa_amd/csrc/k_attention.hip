// Multi-head self-attention core, exact-fp32 / any-length kernels (MAE_F32 parity path and shape fallback),
// plus the dispatch to the bf16 MFMA kernels (k_attention_mfma.hip).
// Reference behaviour: timm Attention.forward: qkv.reshape(B,T,3,H,hd).permute(2,0,3,1,4);
// F.scaled_dot_product_attention(q,k,v) (scale hd^-0.5, softmax over keys, no mask, no dropout);
// transpose(1,2).reshape(B,T,C).  Sequences here are 36 (encoder, visible tokens) and 145 (decoder).
#include "kernels.h"
#include "gemm_mfma.h"

namespace mae {

// One workgroup per (image, head).  K and V rows of the head pass through LDS as fp32 in blocks of KB rows (one block = the whole
// sequence whenever it fits: every BASELINE shape); each thread owns query rows and runs an online softmax over the keys in index order
// (LDS broadcast reads), so the block size changes no result.  Sequences longer than the workgroup take several passes of query rows.
template <class T, int HD>
__global__ void __launch_bounds__(256) attn_fwd_generic_kernel(const T* __restrict__ qkv, int Tn, int H, float scale,
                                                               T* __restrict__ out, float* __restrict__ lse, int KB) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + (size_t)KB * HD;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int64_t rs = 3ll * H * HD;
  const T* base = qkv + (int64_t)b * Tn * rs + h * HD;
  for (int t0 = 0; t0 < Tn; t0 += blockDim.x) {
    const int t = t0 + threadIdx.x;
    const bool live = t < Tn;
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = live ? to_f(base[t * rs + d]) * scale : 0.f; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < Tn; j0 += KB) {
      const int nb = Tn - j0 < KB ? Tn - j0 : KB;
      __syncthreads();   // the previous block (or pass) has been read by everyone
      for (int i = threadIdx.x; i < nb * HD; i += blockDim.x) {
        const int tt = j0 + i / HD, d = i % HD;
        Ks[i] = to_f(base[tt * rs + (int64_t)H * HD + d]);
        Vs[i] = to_f(base[tt * rs + 2ll * H * HD + d]);
      }
      __syncthreads();
      if (live)
        for (int j = 0; j < nb; ++j) {
          float sc = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) sc = fmaf(q[d], Ks[j * HD + d], sc);
          const float mn = fmaxf(m, sc);
          const float a = __expf(m - mn), p = __expf(sc - mn);
          l = l * a + p;
#pragma unroll
          for (int d = 0; d < HD; ++d) o[d] = fmaf(p, Vs[j * HD + d], o[d] * a);
          m = mn;
        }
    }
    if (live) {
      const float inv = 1.0f / l;
      T* po = out + ((int64_t)b * Tn + t) * H * HD + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) po[d] = from_f<T>(o[d] * inv);
      lse[((int64_t)b * H + h) * Tn + t] = m + __logf(l);
    }
  }
}

// Backward, two phases inside one workgroup per (image, head):
//  A (thread per query t): dq_t = scale * sum_j ds_tj k_j,  ds_tj = p_tj (do_t.v_j - D_t),  D_t = do_t.o_t
//  B (thread per key j):   dv_j = sum_t p_tj do_t,  dk_j = scale * sum_t ds_tj q_t
// p_tj = exp(scale q_t.k_j - lse_t) is recomputed from the saved log-sum-exp.  The other side's rows (K, V in phase A; Q, dO in
// phase B) pass through LDS in blocks of KB rows, summed in index order: the block size changes no result.
template <class T, int HD>
__global__ void __launch_bounds__(256) attn_bwd_generic_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ d_out, const float* __restrict__ lse,
                                                               int Tn, int H, float scale, T* __restrict__ d_qkv, int KB) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* S0 = sm;                          // phase A: K, phase B: Q
  float* S1 = sm + (size_t)KB * HD;        // phase A: V, phase B: dO
  float* Lse = sm + 2 * (size_t)KB * HD;   // [Tn]
  float* Dt = Lse + Tn;                    // [Tn]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int64_t rs = 3ll * H * HD, os = (int64_t)H * HD;
  const T* base = qkv + (int64_t)b * Tn * rs + h * HD;
  const T* obase = out + (int64_t)b * Tn * os + h * HD;
  const T* dobase = d_out + (int64_t)b * Tn * os + h * HD;
  T* dbase = d_qkv + (int64_t)b * Tn * rs + h * HD;
  for (int t = threadIdx.x; t < Tn; t += blockDim.x) Lse[t] = lse[((int64_t)b * H + h) * Tn + t];
  // ---- phase A
  for (int t0 = 0; t0 < Tn; t0 += blockDim.x) {
    const int t = t0 + threadIdx.x;
    const bool live = t < Tn;
    float q[HD], dov[HD], dq[HD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = live ? to_f(base[t * rs + d]) : 0.f;
      dov[d] = live ? to_f(dobase[t * os + d]) : 0.f;
      D = fmaf(dov[d], live ? to_f(obase[t * os + d]) : 0.f, D);
      dq[d] = 0.f;
    }
    for (int j0 = 0; j0 < Tn; j0 += KB) {
      const int nb = Tn - j0 < KB ? Tn - j0 : KB;
      __syncthreads();
      for (int i = threadIdx.x; i < nb * HD; i += blockDim.x) {
        const int tt = j0 + i / HD, d = i % HD;
        S0[i] = to_f(base[tt * rs + os + d]);
        S1[i] = to_f(base[tt * rs + 2 * os + d]);
      }
      __syncthreads();   // also publishes Lse on the first block
      if (live) {
        const float ls = Lse[t];
        for (int j = 0; j < nb; ++j) {
          float sc = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) { sc = fmaf(q[d], S0[j * HD + d], sc); dp = fmaf(dov[d], S1[j * HD + d], dp); }
          const float p = __expf(sc * scale - ls);
          const float ds = p * (dp - D);
#pragma unroll
          for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, S0[j * HD + d], dq[d]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int d = 0; d < HD; ++d) dbase[t * rs + d] = from_f<T>(dq[d] * scale);
      Dt[t] = D;
    }
  }
  // ---- phase B: Q and dO blocks through the same LDS
  for (int j0k = 0; j0k < Tn; j0k += blockDim.x) {
    const int j = j0k + threadIdx.x;
    const bool live = j < Tn;
    float kj[HD], vj[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      kj[d] = live ? to_f(base[j * rs + os + d]) : 0.f;
      vj[d] = live ? to_f(base[j * rs + 2 * os + d]) : 0.f;
      dk[d] = 0.f;
      dv[d] = 0.f;
    }
    for (int t0 = 0; t0 < Tn; t0 += KB) {
      const int nb = Tn - t0 < KB ? Tn - t0 : KB;
      __syncthreads();   // phase A's (or the previous block's) readers are done; Dt is complete
      for (int i = threadIdx.x; i < nb * HD; i += blockDim.x) {
        const int tt = t0 + i / HD, d = i % HD;
        S0[i] = to_f(base[tt * rs + d]);
        S1[i] = to_f(dobase[tt * os + d]);
      }
      __syncthreads();
      if (live)
        for (int t = 0; t < nb; ++t) {
          float sc = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) { sc = fmaf(S0[t * HD + d], kj[d], sc); dp = fmaf(S1[t * HD + d], vj[d], dp); }
          const float p = __expf(sc * scale - Lse[t0 + t]);
          const float ds = p * (dp - Dt[t0 + t]);
#pragma unroll
          for (int d = 0; d < HD; ++d) { dv[d] = fmaf(p, S1[t * HD + d], dv[d]); dk[d] = fmaf(ds, S0[t * HD + d], dk[d]); }
        }
    }
    if (live) {
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dbase[j * rs + os + d] = from_f<T>(dk[d] * scale);
        dbase[j * rs + 2 * os + d] = from_f<T>(dv[d]);
      }
    }
  }
}

static int attn_block(int T) { return (int)std::min<int64_t>(round_up(T, 64), 256); }

template <class T>
static int run_attn_fwd(const void* qkv, int B, int Tn, int H, int hd, void* out, float* lse, hipStream_t s) {
  const int KB = (int)std::min<int64_t>(Tn, (160 * 1024) / (2 * hd * (int)sizeof(float)));   // rows of K and V per LDS block
  const size_t lds = (size_t)2 * KB * hd * sizeof(float);
  const float scale = 1.0f / sqrtf((float)hd);
  const dim3 grid((unsigned)B * H), block(attn_block(Tn));
#define AF(HD) { MAE_HIP(hipFuncSetAttribute((const void*)attn_fwd_generic_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((attn_fwd_generic_kernel<T, HD>), grid, block, lds, s, (const T*)qkv, Tn, H, scale, (T*)out, lse, KB); }
  switch (hd) {
    case 16: AF(16) break; case 24: AF(24) break; case 32: AF(32) break; case 48: AF(48) break; case 64: AF(64) break;
    default: set_error("attention: head_dim %d unsupported (16, 24, 32, 48, 64)", hd); return 1;
  }
#undef AF
  MAE_LAUNCH_CHECK();
  return 0;
}

template <class T>
static int run_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int B, int Tn, int H, int hd,
                        void* d_qkv, hipStream_t s) {
  MAE_REQUIRE((int64_t)Tn * 8 <= 96 * 1024, "attention_bwd: T=%d is beyond the fallback kernel (log-sum-exp and D rows stay in LDS)", Tn);
  const int KB = (int)std::min<int64_t>(Tn, (160 * 1024 - (int64_t)Tn * 8) / (2 * hd * (int)sizeof(float)));   // rows per LDS block
  const size_t lds = ((size_t)2 * KB * hd + 2 * Tn) * sizeof(float);
  const float scale = 1.0f / sqrtf((float)hd);
  const dim3 grid((unsigned)B * H), block(attn_block(Tn));
#define AB(HD) { MAE_HIP(hipFuncSetAttribute((const void*)attn_bwd_generic_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((attn_bwd_generic_kernel<T, HD>), grid, block, lds, s, (const T*)qkv, (const T*)out, (const T*)d_out, lse, Tn, H, scale, (T*)d_qkv, KB); }
  switch (hd) {
    case 16: AB(16) break; case 24: AB(24) break; case 32: AB(32) break; case 48: AB(48) break; case 64: AB(64) break;
    default: set_error("attention: head_dim %d unsupported (16, 24, 32, 48, 64)", hd); return 1;
  }
#undef AB
  MAE_LAUNCH_CHECK();
  return 0;
}

int launch_attention_fwd(const void* qkv, int B, int T, int H, int hd, int dt, void* out, float* lse, hipStream_t s) {
  MAE_REQUIRE(qkv && out && lse && B > 0 && T > 0 && H > 0 && hd > 0, "attention_fwd: bad arguments");
  MAE_REQUIRE((int64_t)B * H < (1ll << 31), "attention_fwd: batch*heads overflows the grid");
  if (dt == MAE_BF16) {
    const int r = mfma_attention_fwd((const bf16*)qkv, B, T, H, hd, (bf16*)out, lse, s);
    if (r != MFMA_UNSUPPORTED) return r;
    return run_attn_fwd<bf16>(qkv, B, T, H, hd, out, lse, s);
  }
  MAE_REQUIRE(dt == MAE_F32, "attention_fwd: bad dtype %d", dt);
  return run_attn_fwd<float>(qkv, B, T, H, hd, out, lse, s);
}

int launch_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int B, int T, int H, int hd,
                         int dt, void* d_qkv, hipStream_t s) {
  MAE_REQUIRE(qkv && out && d_out && lse && d_qkv && B > 0 && T > 0 && H > 0 && hd > 0, "attention_bwd: bad arguments");
  MAE_REQUIRE((int64_t)B * H < (1ll << 31), "attention_bwd: batch*heads overflows the grid");
  if (dt == MAE_BF16) {
    const int r = mfma_attention_bwd((const bf16*)qkv, (const bf16*)out, (const bf16*)d_out, lse, B, T, H, hd, (bf16*)d_qkv, s);
    if (r != MFMA_UNSUPPORTED) return r;
    return run_attn_bwd<bf16>(qkv, out, d_out, lse, B, T, H, hd, d_qkv, s);
  }
  MAE_REQUIRE(dt == MAE_F32, "attention_bwd: bad dtype %d", dt);
  return run_attn_bwd<float>(qkv, out, d_out, lse, B, T, H, hd, d_qkv, s);
}

}  // namespace mae

extern "C" int mae_attention_fwd(const void* qkv, int32_t batch, int32_t T, int32_t H, int32_t hd, int32_t dtype, void* out,
                                 float* lse, void* stream) {
  return mae::launch_attention_fwd(qkv, batch, T, H, hd, dtype, out, lse, (hipStream_t)stream);
}

extern "C" int mae_attention_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, int32_t batch,
                                 int32_t T, int32_t H, int32_t hd, int32_t dtype, void* d_qkv, void* stream) {
  return mae::launch_attention_bwd(qkv, out, d_out, lse, batch, T, H, hd, dtype, d_qkv, (hipStream_t)stream);
}
