// bf16 MFMA attention for gfx950: the short sequences of MAE (36 visible tokens in the encoder, 145 in the decoder).
//
// One workgroup per (image, head).  Q, K, V (and dO in backward) of the head are staged once into LDS as row-major
// [token][hd] images with a 32-byte row pad; one image serves both access kinds with no bank conflicts:
//   * row fragments   (operand indexed by token, contraction over hd)     -> ds_read_b128
//   * transposed frags (operand indexed by hd,   contraction over tokens)  -> 2 x ds_read_b64_tr_b16
// Products are formed transposed (S^T = K Q^T, "key/query on the lane") so that the accumulator of the first
// product IS the B operand of the second (P^T for O^T = V^T P^T, dS^T for dQ^T = K^T dS^T, ...): the 32-token
// contraction chunk uses the permuted index kk(h,g,q) = 16h + 4g + q on both operands, so no lane movement and
// no LDS round trip for P.  Softmax is online over 32-key chunks (wave shuffles across the 4 lane groups).
//
// Reference behaviour: F.scaled_dot_product_attention inside timm Attention.forward (no mask, no dropout).
#include "gemm_mfma.h"

#ifndef MAE_ATT_DMA_POLICY
#define MAE_ATT_DMA_POLICY 0  // cache-policy bits of the staging DMAs (2 = nt)
#endif
#ifdef MAE_ATT_NT
#define AT_ST store4_nt
#else
#define AT_ST store4
#endif

namespace mae {

namespace {

constexpr float kLog2e = 1.4426950408889634f;

template <int HD>
struct AT {
  static constexpr int RS = HD * 2 + 32;  // LDS row stride in bytes
  static constexpr int NKS = HD / 32;     // 32-deep MFMA steps across the head dimension
  static constexpr int NDT = HD / 16;     // 16-wide output tiles across the head dimension
  static constexpr int CPR = HD / 8;      // 16-byte chunks per row
};

// stage rows [0,T) of a (T, HD) matrix with global row stride gs (elements) into an LDS image, zero rows [T,Tp)
template <int HD>
__device__ __forceinline__ void stage_image(char* s, const bf16* g, int64_t gs, int T, int Tp) {
  for (int c = threadIdx.x; c < Tp * AT<HD>::CPR; c += blockDim.x) {
    const int row = c / AT<HD>::CPR, cc = c - row * AT<HD>::CPR;
    uint4 v = uint4{0, 0, 0, 0};
    if (row < T) v = *reinterpret_cast<const uint4*>(g + row * gs + cc * 8);
    *reinterpret_cast<uint4*>(s + row * AT<HD>::RS + cc * 16) = v;
  }
}

// Same image filled by LDS-DMA (global_load_lds_dwordx4): every 16-byte slot of the padded image is one lane of one
// DMA, all of a wave's DMAs are in flight together (the register-staged loop above serialises ~3 global round trips
// per image).  Rows past T and the pad chunks read clamped (valid, finite) data: every use of them is multiplied by
// an exactly-zero probability, so they only have to be finite.  img_bytes is a multiple of 1 KiB.
template <int HD>
__device__ __forceinline__ void stage_image_dma(char* s, int img_bytes, const bf16* g, int64_t gs, int T, int wave, int nwaves, int lane, int cprv) {
  constexpr int SPR = AT<HD>::RS / 16;  // 16-byte slots per padded row
  for (int blk = wave; blk * 1024 < img_bytes; blk += nwaves) {
    const int slot = blk * 64 + lane;
    int row = slot / SPR, c = slot - row * SPR;
    row = row < T ? row : T - 1;
    c = c < cprv ? c : cprv - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + row * gs + c * 8),
                                     (__attribute__((address_space(3))) void*)(s + blk * 1024), 16, 0, MAE_ATT_DMA_POLICY);
  }
}

// Head dims below the template's HD (the tiny config's 24 runs the HD = 32 kernels): the 16-byte slots [cprv, CPR) of every
// row take part in the 32-deep contractions, so they are zeroed after the DMAs have landed (a DMA cannot write zeros).
template <int HD>
__device__ __forceinline__ void zero_pad_chunks(char* s, int rows, int cprv) {
  const int npad = AT<HD>::CPR - cprv;
  for (int i = threadIdx.x; i < rows * npad; i += blockDim.x) {
    const int row = i / npad, c = cprv + (i - row * npad);
    *reinterpret_cast<uint4*>(s + row * AT<HD>::RS + c * 16) = uint4{0, 0, 0, 0};
  }
}

// operand indexed by token row (row0 + lane&15), elements hd = ks*32 + 8*(lane>>4) .. +7
// CL: the image holds round_up(T, 16) rows instead of the chunk loops' round_up(T, 32); rows past it read the last row (their probability is zero).
// Only the short unrolled variants trim (a fourth encoder-backward workgroup fits a CU); the clamp costs the long, VALU-bound ones 4-6 %.
template <int HD, bool CL>
__device__ __forceinline__ bf16x8 rowfrag(const char* s, int row0, int ks, int lane, int rmax) {
  const int row = CL ? min(row0 + (lane & 15), rmax) : row0 + (lane & 15);
  return *reinterpret_cast<const bf16x8*>(s + row * AT<HD>::RS + (ks * 32 + 8 * (lane >> 4)) * 2);
}

// operand indexed by hd column (c0 + lane&15), elements = tokens j0 + 16h + 4*(lane>>4) + q, (h,q) = element>>2, &3
template <int HD, bool CL>
__device__ __forceinline__ bf16x8 trfrag(const char* s, int j0, int c0, int lane, int rmax) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r = j0 + 4 * g + q;
  const char* a = s + (CL ? min(r, rmax) : r) * AT<HD>::RS + (c0 + 4 * p) * 2;
  const char* a2 = CL ? s + min(r + 16, rmax) * AT<HD>::RS + (c0 + 4 * p) * 2 : a + 16 * AT<HD>::RS;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a2));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}

// reductions across the 4 lane groups that share lane&15 (lanes l, l^16, l^32, l^48), by VALU lane swaps:
// permlane16_swap(x, x) leaves {own, partner-of-l^16} in the two results for every lane; permlane32_swap likewise for l^32
__device__ __forceinline__ float group_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

}  // namespace

// NCH > 0: the number of 32-token chunks is a compile-time constant, the chunk loops are fully unrolled and the
// compiler overlaps the MFMA / exp chains of different chunks (a runtime loop serialises them); NCH == 0: any length.
// Every workgroup runs load (HBM) -> compute -> store, and the workgroups resident on the chip fall into step: all load,
// then all compute (measured: the phases add up, e.g. decoder backward 155 us of staging + 330 us of compute = 483 us).
// The first generation of workgroups is therefore started out of phase: the workgroup in residency slot r of its CU
// (blockIdx / 256, one slot per CU filled first) waits r * stagger ticks of the 100 MHz wall clock before its loads.
__device__ __forceinline__ void first_generation_stagger(int stagger, int gen1) {
  if (stagger > 0 && (int)blockIdx.x >= 256 && (int)blockIdx.x < gen1) {
    const uint64_t t0 = wall_clock64(), wait = (uint64_t)(blockIdx.x >> 8) * stagger;
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(4);
  }
}

// Workgroup -> (image, head).  The hardware deals consecutive blockIdx round-robin over the 8 XCDs (one L2 each).  With 32-wide heads a
// 128-byte line of a qkv / out / d_out row belongs to TWO heads: dealt in blockIdx order they land on different XCDs and the line is
// fetched from HBM twice (PMC, round 2: decoder attention moved 1.75x its algorithmic bytes).  The remap gives XCD x the contiguous
// range [x G/8, (x+1) G/8) of (image, head) pairs, so the two heads of a line run on the same L2 within a few dispatches of each other.
__device__ __forceinline__ int att_block(int remap) {
  if (!remap) return (int)blockIdx.x;
  const int nb = (int)gridDim.x, bid = (int)blockIdx.x;
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Threads per block: one wave per 16-token tile while the variant's registers allow 4 waves per SIMD (<= 128 VGPRs);
// the wide unrolled variants keep 8 waves and walk their tiles in passes.
template <int HD> struct FwdCap { static constexpr int v = 512; };  // (10 waves for the 145-token decoder forward measured 12 % slower than 5 waves x 2 passes)
template <int NCH> struct BwdCap { static constexpr int v = NCH == 0 ? 1024 : 512; };

// row fragment of the caller's own tile straight from global memory (rows past T clamp to T - 1: masked or multiplied by zero downstream)
template <int HD>
__device__ __forceinline__ bf16x8 rowfrag_global(const bf16* g, int64_t gs, int row0, int ks, int lane, int T) {
  int row = row0 + (lane & 15);
  row = row < T ? row : T - 1;
  return *reinterpret_cast<const bf16x8*>(g + row * gs + ks * 32 + 8 * (lane >> 4));
}

// FQ: the query tile's fragments come straight from global memory and only K and V are staged (sequences whose three images exceed the LDS)
template <int HD, int NCH, bool FQ = false>
__global__ void __launch_bounds__(FwdCap<HD>::v) attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, int T, int Tp, int H, float scale,
                                                            bf16* __restrict__ out, float* __restrict__ lse, int stagger, int gen1, int hdv, int remap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  first_generation_stagger(stagger, gen1);
  constexpr bool CL = NCH >= 1 && NCH <= 3;
  const int Ti = CL ? (T + 15) & ~15 : Tp, rmax = Ti - 1;   // rows an image holds (Tp = the 32-key chunk padding of the loops)
  const int img = (Ti * AT<HD>::RS + 1023) & ~1023;
  char* sQ = smem;
  char* sK = FQ ? smem : sQ + img;
  char* sV = sK + img;
  const int vb = att_block(remap);
  const int b = vb / H, h = vb - b * H;
  const int64_t gs = 3ll * H * hdv;  // hdv = head dim in memory (<= HD)
  const int cprv = hdv >> 3;
  const bf16* base = qkv + (int64_t)b * T * gs + h * hdv;
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (!FQ) stage_image_dma<HD>(sQ, img, base, gs, T, wave, nwaves, lane, cprv);
  stage_image_dma<HD>(sK, img, base + (int64_t)H * hdv, gs, T, wave, nwaves, lane, cprv);
  stage_image_dma<HD>(sV, img, base + 2ll * H * hdv, gs, T, wave, nwaves, lane, cprv);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (hdv < HD) {
    if (!FQ) zero_pad_chunks<HD>(sQ, Ti, cprv);
    zero_pad_chunks<HD>(sK, Ti, cprv);
    zero_pad_chunks<HD>(sV, Ti, cprv);
  }
  __syncthreads();
  const int g = lane >> 4, i = lane & 15;
  const float sl2 = scale * kLog2e;
  const int nq = (T + 15) >> 4, nchunks = NCH > 0 ? NCH : (Tp >> 5);
  for (int qt = wave; qt < nq; qt += nwaves) {
    bf16x8 qf[AT<HD>::NKS];
#pragma unroll
    for (int ks = 0; ks < AT<HD>::NKS; ++ks) qf[ks] = FQ ? rowfrag_global<HD>(base, gs, qt * 16, ks, lane, T) : rowfrag<HD, CL>(sQ, qt * 16, ks, lane, rmax);
    f32x4 oacc[AT<HD>::NDT];
#pragma unroll
    for (int dt = 0; dt < AT<HD>::NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    if constexpr (NCH > 0) {
      // unrolled lengths: exact two-pass softmax.  All the scores of the row tile stay in registers (8 per chunk), the
      // maximum is taken once, and the per-chunk rescale of the online form (lane-group max, alpha = exp2(m - m'),
      // oacc *= alpha, lsum update: ~12 of ~60 VALU instructions per chunk in a VALU-bound kernel) disappears.
      constexpr int NC = NCH > 0 ? NCH : 1;
      f32x4 s0[NC], s1[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        s0[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[c] = s0[c];
#pragma unroll
        for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
          s0[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32, ks, lane, rmax), qf[ks], s0[c], 0, 0, 0);
          s1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32 + 16, ks, lane, rmax), qf[ks], s1[c], 0, 0, 0);
        }
      }
      {  // only the last chunk holds keys past T
        const int j0 = (NC - 1) * 32 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (j0 + r >= T) s0[NC - 1][r] = -INFINITY;
          if (j0 + 16 + r >= T) s1[NC - 1][r] = -INFINITY;
        }
      }
      float mc = -INFINITY;
#pragma unroll
      for (int c = 0; c < NC; ++c)
        mc = fmaxf(mc, fmaxf(fmaxf(fmaxf(s0[c][0], s0[c][1]), fmaxf(s0[c][2], s0[c][3])), fmaxf(fmaxf(s1[c][0], s1[c][1]), fmaxf(s1[c][2], s1[c][3]))));
      m = group_max(mc);  // finite: key 0 is always valid
      const float mn2 = m * sl2;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0[c][r] = __builtin_amdgcn_exp2f(fmaf(s0[c][r], sl2, -mn2));  // raw v_exp_f32: arguments <= 0, exp2(-inf) = 0
          s1[c][r] = __builtin_amdgcn_exp2f(fmaf(s1[c][r], sl2, -mn2));
          lsum += s0[c][r] + s1[c][r];
        }
        const bf16x8 pf = pack8(s0[c], s1[c]);
#pragma unroll
        for (int dt = 0; dt < AT<HD>::NDT; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sV, c * 32, dt * 16, lane, rmax), pf, oacc[dt], 0, 0, 0);
      }
    } else {
    for (int c = 0; c < nchunks; ++c) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32, ks, lane, rmax), qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32 + 16, ks, lane, rmax), qf[ks], s1, 0, 0, 0);
      }
      // lane: query i, keys j = c*32 + 4g + r (s0) and c*32 + 16 + 4g + r (s1)
      const int j0 = c * 32 + 4 * g;
      if (c * 32 + 32 > T) {  // only the last chunk holds keys past T (wave-uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (j0 + r >= T) s0[r] = -INFINITY;
          if (j0 + 16 + r >= T) s1[r] = -INFINITY;
        }
      }
      float mc = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      mc = group_max(mc);
      const float mn = fmaxf(m, mc);  // finite from chunk 0 on (key 0 is always valid)
      // raw v_exp_f32 (arguments are <= 0; exp2(-inf) = 0): the library exp2f adds a denormal-range rescale
      // (v_cmp + 2 v_cndmask + v_ldexp per element) that an attention probability does not need
      const float mn2 = mn * sl2;
      const float alpha = __builtin_amdgcn_exp2f(m * sl2 - mn2);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], sl2, -mn2));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], sl2, -mn2));
        ps += s0[r] + s1[r];
      }
      lsum = lsum * alpha + ps;
      const bf16x8 pf = pack8(s0, s1);
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt) {
        oacc[dt] *= alpha;
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sV, c * 32, dt * 16, lane, rmax), pf, oacc[dt], 0, 0, 0);
      }
      m = mn;
    }
    }
    lsum = group_sum(lsum);
    const float inv = 1.0f / lsum;
    const int tq = qt * 16 + i;
    if (tq < T) {
      bf16* po = out + ((int64_t)b * T + tq) * H * hdv + h * hdv + 4 * g;
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt)
        if (dt * 16 + 4 * g < hdv) AT_ST(po + dt * 16, oacc[dt] * inv);
      if (g == 0) lse[((int64_t)b * H + h) * T + tq] = m * scale + __logf(lsum);
    }
  }
}

// PH = 0: both phases from four staged images (Q, K, V, dO).  Sequences whose four images do not fit the LDS run as TWO launches that stage
// two images each and take their own 16-token tile's fragments straight from global memory: PH = 1 (dQ: K, V staged) and PH = 2 (dK / dV:
// Q, dO staged).  Same arithmetic in the same order as PH = 0.
template <int HD, int NCH, int PH = 0>
__global__ void __launch_bounds__(BwdCap<NCH>::v) attn_bwd_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                            const bf16* __restrict__ d_out, const float* __restrict__ lse, int T,
                                                            int Tp, int H, float scale, bf16* __restrict__ d_qkv, int stagger, int gen1, int hdv, int remap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  first_generation_stagger(stagger, gen1);
  constexpr bool CL = NCH >= 1 && NCH <= 3;
  const int Ti = CL ? (T + 15) & ~15 : Tp, rmax = Ti - 1;   // rows an image holds (Tp = the 32-key chunk padding of the loops)
  const int img = (Ti * AT<HD>::RS + 1023) & ~1023;
  char* sQ = smem;
  char* sK = PH == 1 ? smem : sQ + img;
  char* sV = sK + img;
  char* sdO = PH == 2 ? smem + img : sV + img;
  float* sLse = reinterpret_cast<float*>(smem + (PH == 0 ? 4 : 2) * img);  // pre-multiplied by log2(e); 1e30 on padded rows
  float* sD = sLse + Tp;
  const int vb = att_block(remap);
  const int b = vb / H, h = vb - b * H;
  const int64_t gs = 3ll * H * hdv, os = (int64_t)H * hdv;  // hdv = head dim in memory (<= HD)
  const int cprv = hdv >> 3;
  const bf16* base = qkv + (int64_t)b * T * gs + h * hdv;
  const bf16* obase = out + (int64_t)b * T * os + h * hdv;
  const bf16* dobase = d_out + (int64_t)b * T * os + h * hdv;
  bf16* dbase = d_qkv + (int64_t)b * T * gs + h * hdv;
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (PH != 1) stage_image_dma<HD>(sQ, img, base, gs, T, wave, nwaves, lane, cprv);
  if (PH != 2) stage_image_dma<HD>(sK, img, base + os, gs, T, wave, nwaves, lane, cprv);
  if (PH != 2) stage_image_dma<HD>(sV, img, base + 2 * os, gs, T, wave, nwaves, lane, cprv);
  if (PH != 1) stage_image_dma<HD>(sdO, img, dobase, os, T, wave, nwaves, lane, cprv);
  // D_t = dO_t . O_t with CPR lanes per row (dO chunk from the staged image, O chunk from global), shuffle-reduced.
  // The O chunks and the log-sum-exps are fetched into registers BEFORE the wait on the staging DMAs, so the two global
  // round trips of a workgroup's prologue overlap instead of following each other (up to PF slots per thread; longer
  // sequences take the plain loop).
  // DPM (the unrolled lengths): D_t is not formed from dO_t . O_t at all but as sum_j P_tj dP_tj inside phase A, where a wave holds every
  // P and dP of its 16 queries in registers (the two sums are the same number: O = P V, dP = dO V^T) -- the O rows (a ninth of the
  // kernel's bytes) are never read, and the prologue shrinks to the log-sum-exps.
  constexpr bool DPM = NCH > 0 && PH == 0;
  constexpr int CPR = AT<HD>::CPR, PF = 4;
  const int total = Tp * CPR;
  const bool pf = !DPM && PH == 0 && total <= PF * (int)blockDim.x;
  if (DPM) {
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) {
      sLse[t] = t < T ? lse[((int64_t)b * H + h) * T + t] * kLog2e : 1e30f;
      sD[t] = 0.f;
    }
  }
  bf16x8 o_pf[PF];
  float l_pf[PF];
  if (pf) {
#pragma unroll
    for (int it = 0; it < PF; ++it) {
      const int idx = it * blockDim.x + threadIdx.x;
      const int t = idx / CPR, cc = idx - t * CPR;
      const bool live = idx < total && t < T && cc < cprv;
      const int tc = live ? t : 0;
      o_pf[it] = *reinterpret_cast<const bf16x8*>(obase + tc * os + (live ? cc : 0) * 8);
      l_pf[it] = lse[((int64_t)b * H + h) * T + tc];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (hdv < HD) {
    if (PH != 1) zero_pad_chunks<HD>(sQ, Ti, cprv);
    if (PH != 2) zero_pad_chunks<HD>(sK, Ti, cprv);
    if (PH != 2) zero_pad_chunks<HD>(sV, Ti, cprv);
    if (PH != 1) zero_pad_chunks<HD>(sdO, Ti, cprv);
  }
  __syncthreads();
  if (pf) {
#pragma unroll
    for (int it = 0; it < PF; ++it) {
      const int idx = it * blockDim.x + threadIdx.x;
      const int t = idx / CPR, cc = idx - t * CPR;
      float D = 0.f;
      if (idx < total && t < T && cc < cprv) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sdO + t * AT<HD>::RS + cc * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) D = fmaf((float)a[e], (float)o_pf[it][e], D);
      }
#pragma unroll
      for (int off = CPR / 2; off > 0; off >>= 1) D += __shfl_xor(D, off, 64);
      if (idx < total && cc == 0) {
        sD[t] = D;
        sLse[t] = t < T ? l_pf[it] * kLog2e : 1e30f;
      }
    }
  } else if (!DPM) {
    for (int base = 0; base < total; base += blockDim.x) {
      const int idx = base + threadIdx.x;
      const int t = idx / CPR, cc = idx - t * CPR;
      float D = 0.f;
      if (idx < total && t < T && cc < cprv) {
        const bf16x8 a = PH == 1 ? *reinterpret_cast<const bf16x8*>(dobase + t * os + cc * 8) : *reinterpret_cast<const bf16x8*>(sdO + t * AT<HD>::RS + cc * 16);
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(obase + t * os + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) D = fmaf((float)a[e], (float)o[e], D);
      }
#pragma unroll
      for (int off = CPR / 2; off > 0; off >>= 1) D += __shfl_xor(D, off, 64);
      if (idx < total && cc == 0) {
        sD[t] = D;
        sLse[t] = t < T ? lse[((int64_t)b * H + h) * T + t] * kLog2e : 1e30f;
      }
    }
  }
  if (!DPM) __syncthreads();
  const int g = lane >> 4, i = lane & 15;
  const float sl2 = scale * kLog2e;
  const int nt16 = (T + 15) >> 4, nchunks = NCH > 0 ? NCH : (Tp >> 5);

  // ---- phase A: wave owns 16 queries; dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]
#ifdef MAE_DBG_ATT_NO_A
  if (T < 0)
#endif
  if (PH != 2)
  for (int qt = wave; qt < nt16; qt += nwaves) {
    bf16x8 qf[AT<HD>::NKS], dof[AT<HD>::NKS];
#pragma unroll
    for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
      qf[ks] = PH == 1 ? rowfrag_global<HD>(base, gs, qt * 16, ks, lane, T) : rowfrag<HD, CL>(sQ, qt * 16, ks, lane, rmax);
      dof[ks] = PH == 1 ? rowfrag_global<HD>(dobase, os, qt * 16, ks, lane, T) : rowfrag<HD, CL>(sdO, qt * 16, ks, lane, rmax);
    }
    const float li = sLse[qt * 16 + i];
    f32x4 dq[AT<HD>::NDT];
#pragma unroll
    for (int dt = 0; dt < AT<HD>::NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (DPM) {
      constexpr int NC = NCH > 0 ? NCH : 1;
      f32x4 P0[NC], P1[NC], G0[NC], G1[NC];   // probabilities and dP = dO V^T of the tile's 16 queries against every key
      float dpart = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        G0[c] = s0; G1[c] = s0;
#pragma unroll
        for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32, ks, lane, rmax), qf[ks], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32 + 16, ks, lane, rmax), qf[ks], s1, 0, 0, 0);
          G0[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sV, c * 32, ks, lane, rmax), dof[ks], G0[c], 0, 0, 0);
          G1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sV, c * 32 + 16, ks, lane, rmax), dof[ks], G1[c], 0, 0, 0);
        }
        if (c == NC - 1) {  // only the last chunk holds keys past T
          const int j0 = c * 32 + 4 * g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (j0 + r >= T) s0[r] = -INFINITY;
            if (j0 + 16 + r >= T) s1[r] = -INFINITY;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P0[c][r] = __builtin_amdgcn_exp2f(fmaf(s0[r], sl2, -li));
          P1[c][r] = __builtin_amdgcn_exp2f(fmaf(s1[r], sl2, -li));
          dpart = fmaf(P0[c][r], G0[c][r], dpart);
          dpart = fmaf(P1[c][r], G1[c][r], dpart);
        }
      }
      const float Di = group_sum(dpart);   // the keys of a query are spread over the four lane groups
      if (g == 0) sD[qt * 16 + i] = Di;    // phase B reads it per query
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 d0, d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d0[r] = P0[c][r] * (G0[c][r] - Di);
          d1[r] = P1[c][r] * (G1[c][r] - Di);
        }
        const bf16x8 dsf = pack8(d0, d1);
#pragma unroll
        for (int dt = 0; dt < AT<HD>::NDT; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sK, c * 32, dt * 16, lane, rmax), dsf, dq[dt], 0, 0, 0);
      }
    } else {
    const float Di = sD[qt * 16 + i];
#pragma unroll
    for (int c = 0; c < nchunks; ++c) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
      for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32, ks, lane, rmax), qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sK, c * 32 + 16, ks, lane, rmax), qf[ks], s1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sV, c * 32, ks, lane, rmax), dof[ks], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sV, c * 32 + 16, ks, lane, rmax), dof[ks], p1, 0, 0, 0);
      }
      const int j0 = c * 32 + 4 * g;
      if (NCH > 0 ? c == NCH - 1 : c * 32 + 32 > T) {  // only the last chunk holds keys past T (static when unrolled, else wave-uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (j0 + r >= T) s0[r] = -INFINITY;
          if (j0 + 16 + r >= T) s1[r] = -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pa = __builtin_amdgcn_exp2f(fmaf(s0[r], sl2, -li));  // raw v_exp_f32: exp2(-inf) = 0, no denormal rescale
        const float pb = __builtin_amdgcn_exp2f(fmaf(s1[r], sl2, -li));
        s0[r] = pa * (p0[r] - Di);
        s1[r] = pb * (p1[r] - Di);
      }
      const bf16x8 dsf = pack8(s0, s1);
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sK, c * 32, dt * 16, lane, rmax), dsf, dq[dt], 0, 0, 0);
    }
    }
    const int tq = qt * 16 + i;
    if (tq < T) {
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt)
        if (dt * 16 + 4 * g < hdv) AT_ST(dbase + tq * gs + dt * 16 + 4 * g, dq[dt] * scale);
    }
  }
  if (DPM) __syncthreads();   // every query's D is in sD before the key-owner phase reads it

  // ---- phase B: wave owns 16 keys; dV^T[d][j] = sum_i dO^T[d][i] P[i][j], dK^T[d][j] = sum_i Q^T[d][i] dS[i][j]
#ifdef MAE_DBG_ATT_NO_B
  if (T < 0)
#endif
  if (PH != 1)
  for (int jt = wave; jt < nt16; jt += nwaves) {
    bf16x8 kf[AT<HD>::NKS], vf[AT<HD>::NKS];
#pragma unroll
    for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
      kf[ks] = PH == 2 ? rowfrag_global<HD>(base + os, gs, jt * 16, ks, lane, T) : rowfrag<HD, CL>(sK, jt * 16, ks, lane, rmax);
      vf[ks] = PH == 2 ? rowfrag_global<HD>(base + 2 * os, gs, jt * 16, ks, lane, T) : rowfrag<HD, CL>(sV, jt * 16, ks, lane, rmax);
    }
    f32x4 dk[AT<HD>::NDT], dv[AT<HD>::NDT];
#pragma unroll
    for (int dt = 0; dt < AT<HD>::NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
#pragma unroll
    for (int c = 0; c < nchunks; ++c) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
      for (int ks = 0; ks < AT<HD>::NKS; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sQ, c * 32, ks, lane, rmax), kf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sQ, c * 32 + 16, ks, lane, rmax), kf[ks], s1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sdO, c * 32, ks, lane, rmax), vf[ks], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag<HD, CL>(sdO, c * 32 + 16, ks, lane, rmax), vf[ks], p1, 0, 0, 0);
      }
      // lane: key j = jt*16 + i; queries c*32 + 4g + r (tile 0), c*32 + 16 + 4g + r (tile 1); padded queries: lse = 1e30
      const f32x4 l0 = load4(sLse + c * 32 + 4 * g), l1 = load4(sLse + c * 32 + 16 + 4 * g);
      const f32x4 D0 = load4(sD + c * 32 + 4 * g), D1 = load4(sD + c * 32 + 16 + 4 * g);
      f32x4 ds0, ds1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], sl2, -l0[r]));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], sl2, -l1[r]));
        ds0[r] = s0[r] * (p0[r] - D0[r]);
        ds1[r] = s1[r] * (p1[r] - D1[r]);
      }
      const bf16x8 pf = pack8(s0, s1), dsf = pack8(ds0, ds1);
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sdO, c * 32, dt * 16, lane, rmax), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag<HD, CL>(sQ, c * 32, dt * 16, lane, rmax), dsf, dk[dt], 0, 0, 0);
      }
    }
    const int tj = jt * 16 + i;
    if (tj < T) {
#pragma unroll
      for (int dt = 0; dt < AT<HD>::NDT; ++dt) {
        if (dt * 16 + 4 * g < hdv) {
          AT_ST(dbase + tj * gs + os + dt * 16 + 4 * g, dk[dt] * scale);
          AT_ST(dbase + tj * gs + 2 * os + dt * 16 + 4 * g, dv[dt]);
        }
      }
    }
  }
}

static int attn_waves(int T, int max_threads) {
  const int nt = (T + 15) / 16, maxw = max_threads / 64;
  const int passes = (nt + maxw - 1) / maxw;
  return (nt + passes - 1) / passes;  // <= maxw waves, balanced over the 16-token tiles
}

static int attn_remap() {   // MAE_ATT_XCD=0: blockIdx order (A/B)
  static const int v = [] { const char* e = getenv("MAE_ATT_XCD"); return e ? atoi(e) : 1; }();
  return v;
}
static int attn_stagger() {
  static const int v = [] { const char* e = getenv("MAE_ATT_STAGGER"); return e ? atoi(e) : 0; }();
  return v;
}
// workgroups resident per CU (LDS and wave-slot limits; registers are covered by the launch bounds)
static int attn_resident(size_t lds, int threads) {
  const int by_lds = (int)((160 * 1024) / lds), by_waves = 32 / (threads / 64);
  return std::max(1, std::min(std::min(by_lds, by_waves), 8));
}

static bool attn_supported(int T, int H, int hd) {
  return (hd == 24 || hd == 32 || hd == 64) && T >= 1 && T <= 1024 && ((int64_t)H * hd) % 8 == 0;  // 24 runs zero-padded in the 32 kernels
}

template <int HD, int NCH>
static int launch_attn_fwd(const bf16* qkv, int B, int T, int Tp, int H, int hdv, size_t lds, float scale, bf16* out, float* lse, hipStream_t s) {
  auto kern = attn_fwd_mfma_kernel<HD, NCH>;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)B * H), dim3(64 * attn_waves(T, FwdCap<HD>::v)), lds, s, qkv, T, Tp, H, scale, out, lse, attn_stagger(), 256 * attn_resident(lds, 64 * attn_waves(T, FwdCap<HD>::v)), hdv, attn_remap());
  MAE_LAUNCH_CHECK();
  return 0;
}
template <int HD, int NCH>
static int launch_attn_bwd(const bf16* qkv, const bf16* out, const bf16* d_out, const float* lse, int B, int T, int Tp, int H, int hdv, size_t lds,
                           float scale, bf16* d_qkv, hipStream_t s) {
  auto kern = attn_bwd_mfma_kernel<HD, NCH>;
  static const int lds_pad = [] { const char* e = getenv("MAE_ATT_LDS_PAD"); return e ? atoi(e) : 0; }();   // residency experiment: extra LDS bytes per workgroup
  lds += (size_t)lds_pad;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)B * H), dim3(64 * attn_waves(T, BwdCap<NCH>::v)), lds, s, qkv, out, d_out, lse, T, Tp, H, scale, d_qkv, attn_stagger(), 256 * attn_resident(lds, 64 * attn_waves(T, BwdCap<NCH>::v)), hdv, attn_remap());
  MAE_LAUNCH_CHECK();
  return 0;
}
// long sequences: the backward pass as two launches staging two images each (PH = 1: dQ, PH = 2: dK / dV)
template <int HD>
static int launch_attn_bwd_split(const bf16* qkv, const bf16* out, const bf16* d_out, const float* lse, int B, int T, int Tp, int H, int hdv, size_t lds,
                                 float scale, bf16* d_qkv, hipStream_t s) {
  const int threads = 64 * attn_waves(T, BwdCap<0>::v);
  auto k1 = attn_bwd_mfma_kernel<HD, 0, 1>;
  auto k2 = attn_bwd_mfma_kernel<HD, 0, 2>;
  MAE_HIP(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MAE_HIP(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k1, dim3((unsigned)B * H), dim3(threads), lds, s, qkv, out, d_out, lse, T, Tp, H, scale, d_qkv, 0, 0, hdv, attn_remap());
  MAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k2, dim3((unsigned)B * H), dim3(threads), lds, s, qkv, out, d_out, lse, T, Tp, H, scale, d_qkv, 0, 0, hdv, attn_remap());
  MAE_LAUNCH_CHECK();
  return 0;
}
#define ATTN_DISPATCH(FN, BIG, ...)                                         \
  switch (hdt * 100 + (Tp >> 5)) {                                          \
    case 6401: return FN<64, 1>(__VA_ARGS__);                               \
    case 6402: return FN<64, 2>(__VA_ARGS__);                               \
    case 6403: return FN<64, 3>(__VA_ARGS__);                               \
    case 6405: return FN<64, BIG>(__VA_ARGS__);  /* backward: 5 unrolled chunks cost occupancy (217-256 VGPRs): runtime loop */ \
    case 3201: return FN<32, 1>(__VA_ARGS__);                               \
    case 3202: return FN<32, 2>(__VA_ARGS__);                               \
    case 3203: return FN<32, 3>(__VA_ARGS__);                               \
    case 3205: return FN<32, BIG>(__VA_ARGS__);                             \
    default: return hdt == 64 ? FN<64, 0>(__VA_ARGS__) : FN<32, 0>(__VA_ARGS__); \
  }

template <int HD>
static int launch_attn_fwd_fq(const bf16* qkv, int B, int T, int Tp, int H, int hdv, size_t lds, float scale, bf16* out, float* lse, hipStream_t s) {
  auto kern = attn_fwd_mfma_kernel<HD, 0, true>;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)B * H), dim3(64 * attn_waves(T, FwdCap<HD>::v)), lds, s, qkv, T, Tp, H, scale, out, lse, 0, 0, hdv, attn_remap());
  MAE_LAUNCH_CHECK();
  return 0;
}

int mfma_attention_fwd(const bf16* qkv, int B, int T, int H, int hd, bf16* out, float* lse, hipStream_t s) {
  if (!attn_supported(T, H, hd) || (((uintptr_t)qkv | (uintptr_t)out) & 15)) return MFMA_UNSUPPORTED;
  const int Tp = (int)round_up(T, 32), hdt = hd == 24 ? 32 : hd;
  const size_t img = (size_t)round_up(((Tp >> 5) <= 3 ? round_up((int64_t)T, 16) : (int64_t)Tp) * (hdt * 2 + 32), 1024);   // the short unrolled variants hold round_up(T, 16) rows
  const size_t lds = 3 * img;
  const float scale = 1.0f / sqrtf((float)hd);
  if (lds > 160 * 1024) {   // three images do not fit: K and V only, query fragments from global memory
    if (2 * img > 160 * 1024 || hd != hdt) return MFMA_UNSUPPORTED;
    return hdt == 64 ? launch_attn_fwd_fq<64>(qkv, B, T, Tp, H, hd, 2 * img, scale, out, lse, s) : launch_attn_fwd_fq<32>(qkv, B, T, Tp, H, hd, 2 * img, scale, out, lse, s);
  }
  ATTN_DISPATCH(launch_attn_fwd, 5, qkv, B, T, Tp, H, hd, lds, scale, out, lse, s)
}

int mfma_attention_bwd(const bf16* qkv, const bf16* out, const bf16* d_out, const float* lse, int B, int T, int H, int hd,
                       bf16* d_qkv, hipStream_t s) {
  if (!attn_supported(T, H, hd) || (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)d_qkv) & 15)) return MFMA_UNSUPPORTED;
  const int Tp = (int)round_up(T, 32), hdt = hd == 24 ? 32 : hd;
  const size_t img = (size_t)round_up(((Tp >> 5) <= 3 ? round_up((int64_t)T, 16) : (int64_t)Tp) * (hdt * 2 + 32), 1024);   // the short unrolled variants hold round_up(T, 16) rows
  const size_t lds = 4 * img + (size_t)2 * Tp * 4;
  const float scale = 1.0f / sqrtf((float)hd);
  if (lds > 160 * 1024) {
    const size_t lds2 = 2 * img + (size_t)2 * Tp * 4;
    if (lds2 > 160 * 1024 || hd != hdt) return MFMA_UNSUPPORTED;
    return hdt == 64 ? launch_attn_bwd_split<64>(qkv, out, d_out, lse, B, T, Tp, H, hd, lds2, scale, d_qkv, s)
                     : launch_attn_bwd_split<32>(qkv, out, d_out, lse, B, T, Tp, H, hd, lds2, scale, d_qkv, s);
  }
  ATTN_DISPATCH(launch_attn_bwd, 0, qkv, out, d_out, lse, B, T, Tp, H, hd, lds, scale, d_qkv, s)
}

}  // namespace mae
