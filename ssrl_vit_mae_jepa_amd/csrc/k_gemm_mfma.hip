// bf16 MFMA GEMMs for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate), fused epilogues.
//
//  gemm_nt_kernel   out[M,N] = A[M,K] * W[N,K]^T      Linear forward, and dgrad via the transposed weight copy.
//                   Block tile 128 x (64|128) x 64, 4 waves (2x2), register-staged global->LDS with an XOR
//                   swizzle (conflict-free ds_read_b128 fragments), double-buffered LDS, one barrier per K step.
//                   Operands are passed to the MFMA swapped (W as A-operand) so every lane ends up with 4
//                   consecutive output columns of one row: 8-byte (bf16) / 16-byte (fp32) epilogue stores.
//  gemm_tn_kernel   dW[N,K] = dY[M,N]^T * X[M,K]      weight gradient: reduction over the long M dimension.
//                   Both operand tiles are [m][*] row-major in LDS and read transposed with
//                   ds_read_b64_tr_b16; split over M with fp32 slabs + an ordered (deterministic) reduce.
//
// M here is batch*tokens (72 000 .. 290 000), N/K are 192..1536: every GEMM is short-K and output-bound
// (arithmetic intensity ~ the bf16 ridge), so the epilogue stores and the A-panel L2 reuse matter as much as the
// MFMA schedule.  Blocks are remapped so that one XCD walks the N tiles of the same A panel (L2 reuse).
#include "gemm_mfma.h"

namespace mae {

__device__ __forceinline__ void store8(float* p, const f32x4& a, const f32x4& b) { store4(p, a); store4(p + 4, b); }
__device__ __forceinline__ void store8(bf16* p, const f32x4& a, const f32x4& b) {
  *reinterpret_cast<bf16x8*>(p) = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
__device__ __forceinline__ void load8(const float* p, f32x4& a, f32x4& b) { a = load4(p); b = load4(p + 4); }
__device__ __forceinline__ void load8(const bf16* p, f32x4& a, f32x4& b) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}

struct EpiArgs {
  const float* bias;
  const void* aux;
  void* out;
  void* out2;
};

// bijective XCD remap: blocks b and b+8 share an XCD; give every XCD a contiguous run of tiles
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nb) {
  const int64_t q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// RG ("ragged"): N and K only have to be multiples of 8 (16-byte rows): the last tile column and the last K-step are
// zero-filled on load (predicated, nothing is read out of bounds) and the stores are guarded by column.  This is the
// MFMA path of the reference's shipped tiny config (D = 144: K = 144 / 576, N = 432 / 144 / 576).
template <int MODE, class TO, int NI, bool RG>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int64_t M,
                                                         int N, int K, EpiArgs ep, int tiles_n) {
  constexpr int BM = 128, BN = 32 * NI, BK = 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int NB = BN / 32;  // 16-byte chunks of the W tile per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                    // [2][A_BYTES]
  char* sB = smem + 2 * A_BYTES;      // [2][B_BYTES]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t t = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (t / tiles_n) * BM;
  const int n0 = (int)(t % tiles_n) * BN;

  // global -> register staging: chunk c = tid + 256 i -> (row c>>3, 16-byte chunk c&7).  Named scalars, not
  // arrays: hipcc kept uint4 staging arrays in scratch memory here (a vmcnt(0) + scratch round trip per load).
  const int srow = tid >> 3, skc = tid & 7;  // chunk i covers row srow + 32 i
  const bf16 *pa0, *pa1, *pa2, *pa3, *pb0, *pb1, *pb2 = nullptr, *pb3 = nullptr;
  {
    auto arow = [&](int i) { int64_t gm = m0 + srow + 32 * i; return A + (gm < M ? gm : M - 1) * K + skc * 8; };  // clamp: rows past M are loaded, never stored
    pa0 = arow(0); pa1 = arow(1); pa2 = arow(2); pa3 = arow(3);
    pb0 = W + (int64_t)(n0 + srow) * K + skc * 8;
    pb1 = pb0 + 32ll * K;
    if (NB > 2) { pb2 = pb0 + 64ll * K; pb3 = pb0 + 96ll * K; }
  }
  // ragged form: which of this thread's W rows exist (rows past N are zero-filled)
  const bool wok0 = !RG || n0 + srow < N, wok1 = !RG || n0 + srow + 32 < N;
  const bool wok2 = !RG || n0 + srow + 64 < N, wok3 = !RG || n0 + srow + 96 < N;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2 = uint4{0, 0, 0, 0}, rb3 = uint4{0, 0, 0, 0};
  const int soff = srow * 128 + ((skc ^ (srow & 7)) << 4);  // (srow + 32 i) & 7 == srow & 7
#define LD16(p, k0) (*reinterpret_cast<const uint4*>((p) + (k0)))
#define LD16P(p, k0, ok) ((ok) ? LD16(p, k0) : uint4{0, 0, 0, 0})
#define NT_G_LOAD(k0)                                                                 \
  {                                                                                   \
    if (RG) {                                                                         \
      const bool kok = (k0) + skc * 8 < K;                                            \
      ra0 = LD16P(pa0, k0, kok); ra1 = LD16P(pa1, k0, kok); ra2 = LD16P(pa2, k0, kok); ra3 = LD16P(pa3, k0, kok); \
      rb0 = LD16P(pb0, k0, kok && wok0); rb1 = LD16P(pb1, k0, kok && wok1);           \
      if (NB > 2) { rb2 = LD16P(pb2, k0, kok && wok2); rb3 = LD16P(pb3, k0, kok && wok3); } \
    } else {                                                                          \
      ra0 = LD16(pa0, k0); ra1 = LD16(pa1, k0); ra2 = LD16(pa2, k0); ra3 = LD16(pa3, k0); \
      rb0 = LD16(pb0, k0); rb1 = LD16(pb1, k0);                                       \
      if (NB > 2) { rb2 = LD16(pb2, k0); rb3 = LD16(pb3, k0); }                       \
    }                                                                                 \
  }
#define ST16(base, i, v) (*reinterpret_cast<uint4*>((base) + soff + (i) * 32 * 128) = (v))
#define NT_S_STORE(buf)                                                               \
  {                                                                                   \
    char* _a = sA + (buf) * A_BYTES;                                                  \
    char* _b = sB + (buf) * B_BYTES;                                                  \
    ST16(_a, 0, ra0); ST16(_a, 1, ra1); ST16(_a, 2, ra2); ST16(_a, 3, ra3);           \
    ST16(_b, 0, rb0); ST16(_b, 1, rb1);                                               \
    if (NB > 2) { ST16(_b, 2, rb2); ST16(_b, 3, rb3); }                               \
  }

  f32x4 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = RG ? (K + BK - 1) / BK : K / BK;
  const int fr = lane & 15, fq = lane >> 4;
  NT_G_LOAD(0)
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    NT_S_STORE(buf)
    __syncthreads();
    if (kt + 1 < nk) NT_G_LOAD((kt + 1) * BK)
    const char* a_base = sA + buf * A_BYTES + (wm * 64 + fr) * 128;
    const char* b_base = sB + buf * B_BYTES + (wn * (NI * 16) + fr) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int sw = ((ks * 4 + fq) ^ (fr & 7)) << 4;
      bf16x8 af[4], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 16 * 128 + sw);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 16 * 128 + sw);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // epilogue.  After the MFMAs lane (fq, fr) holds, per 16x16 tile ni, 4 consecutive columns (4 fq ..) of row fr.
  // One v_permlane16_swap per register pair (tiles 2j, 2j+1; lanes l <-> l+16) regroups that into 8 consecutive
  // columns per lane: 16-byte bf16 stores, and the 4 lanes of a row cover 64 contiguous bytes per store.
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int j = 0; j < NI / 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mi][2 * j][r]), __float_as_uint(acc[mi][2 * j + 1][r]), false, false);
        acc[mi][2 * j][r] = __uint_as_float(sw[0]);
        acc[mi][2 * j + 1][r] = __uint_as_float(sw[1]);
      }
  const int gb = (fq & 1) ? 3 + fq : fq;  // 4-column group of this lane inside a 32-column half: {0,4,2,6}[fq]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = m0 + wm * 64 + mi * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      const int n = n0 + wn * (NI * 16) + 32 * j + 4 * gb;
      if (RG && n >= N) continue;  // N % 8 == 0: a lane's 8 columns are all inside or all outside
      const int64_t o = m * N + n;
      f32x4 v0 = acc[mi][2 * j], v1 = acc[mi][2 * j + 1];
      if (ep.bias) { v0 += load4(ep.bias + n); v1 += load4(ep.bias + n + 4); }
      if (MODE == MAE_EPI_NONE) {
        store8(reinterpret_cast<TO*>(ep.out) + o, v0, v1);
      } else if (MODE == MAE_EPI_GELU) {
        f32x4 a0, a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = to_f(from_f<TO>(v0[r]));
          v1[r] = to_f(from_f<TO>(v1[r]));
          a0[r] = gelu_fast(v0[r]);
          a1[r] = gelu_fast(v1[r]);
        }
        store8(reinterpret_cast<TO*>(ep.out) + o, v0, v1);
        store8(reinterpret_cast<TO*>(ep.out2) + o, a0, a1);
      } else if (MODE == MAE_EPI_RESID) {
        v0 += load4(reinterpret_cast<const float*>(ep.aux) + o);
        v1 += load4(reinterpret_cast<const float*>(ep.aux) + o + 4);
        store8(reinterpret_cast<TO*>(ep.out) + o, v0, v1);
      } else if (MODE == MAE_EPI_DGELU) {
        f32x4 p0, p1;
        load8(reinterpret_cast<const TO*>(ep.aux) + o, p0, p1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] *= gelu_grad_fast(p0[r]);
          v1[r] *= gelu_grad_fast(p1[r]);
        }
        store8(reinterpret_cast<TO*>(ep.out) + o, v0, v1);
      } else if (MODE == MAE_EPI_GELU_GRAD) {
        f32x4 a0, a1, g0, g1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          { float _a, _g; gelu_fast_pair(to_f(from_f<TO>(v0[r])), _a, _g); a0[r] = _a; g0[r] = _g; }
          { float _a, _g; gelu_fast_pair(to_f(from_f<TO>(v1[r])), _a, _g); a1[r] = _a; g1[r] = _g; }
        }
        store8(reinterpret_cast<TO*>(ep.out) + o, g0, g1);
        store8(reinterpret_cast<TO*>(ep.out2) + o, a0, a1);
      } else if (MODE == MAE_EPI_GELU_ACT) {
        f32x4 a0, a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = gelu_fast(to_f(from_f<TO>(v0[r])));
          a1[r] = gelu_fast(to_f(from_f<TO>(v1[r])));
        }
        store8(reinterpret_cast<TO*>(ep.out) + o, a0, a1);
      } else {  // MAE_EPI_MUL
        f32x4 p0, p1;
        load8(reinterpret_cast<const TO*>(ep.aux) + o, p0, p1);
        store8(reinterpret_cast<TO*>(ep.out) + o, v0 * p0, v1 * p1);
      }
    }
  }
}

template <int MODE, class TO, int NI, bool RG = false>
static int launch_nt(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  constexpr int BN = 32 * NI;
  const int tiles_n = RG ? (int)cdiv(N, BN) : N / BN;
  const int64_t tiles = cdiv(M, 128) * tiles_n;
  MAE_REQUIRE(tiles < (1ll << 31), "gemm: grid too large");
  const size_t lds = 2 * (128 * 64 * 2) + 2 * (BN * 64 * 2);
  auto kern = gemm_nt_kernel<MODE, TO, NI, RG>;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  EpiArgs ea{e.bias, e.aux, e.out, e.out2};
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, s, A, W, M, N, K, ea, tiles_n);
  MAE_LAUNCH_CHECK();
  return 0;
}

template <int MODE, class TO>
static int launch_nt_ni(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (K % 64 != 0 || N % 64 != 0) return launch_nt<MODE, TO, 2, true>(A, W, M, N, K, e, s);  // ragged: 128 x 64 tiles
  if (N % 128 == 0) return launch_nt<MODE, TO, 4>(A, W, M, N, K, e, s);
  return launch_nt<MODE, TO, 2>(A, W, M, N, K, e, s);
}

int mfma_linear_fwd(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (K % 8 != 0 || N % 8 != 0 || K < 32 || N < 16 || M < 1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)e.out | (uintptr_t)e.out2 | (uintptr_t)e.aux | (uintptr_t)e.bias) & 15) != 0)
    return MFMA_UNSUPPORTED;
  const bool f32out = e.out_dt == MAE_F32;
  switch (e.mode) {
    case MAE_EPI_NONE: return f32out ? launch_nt_ni<MAE_EPI_NONE, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_NONE, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU: return f32out ? launch_nt_ni<MAE_EPI_GELU, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_GELU, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_RESID: return f32out ? launch_nt_ni<MAE_EPI_RESID, float>(A, W, M, N, K, e, s) : MFMA_UNSUPPORTED;
    case MAE_EPI_DGELU: return f32out ? launch_nt_ni<MAE_EPI_DGELU, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_DGELU, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_GRAD: return f32out ? launch_nt_ni<MAE_EPI_GELU_GRAD, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_GELU_GRAD, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_MUL: return f32out ? launch_nt_ni<MAE_EPI_MUL, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_MUL, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_ACT: return f32out ? launch_nt_ni<MAE_EPI_GELU_ACT, float>(A, W, M, N, K, e, s) : launch_nt_ni<MAE_EPI_GELU_ACT, bf16>(A, W, M, N, K, e, s);
    default: return MFMA_UNSUPPORTED;
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: dW[n][k] = sum_m dY[m][n] X[m][k]
// ---------------------------------------------------------------------------------------------------
constexpr int TN_RS = 288;  // LDS row stride in bytes for a 128-column bf16 tile row (256 B + 32 B pad): consecutive rows
                            // shift by 8 banks, so the 8 rows one 32-lane half reads transposed are conflict-free

__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p));
}

// NI / KI: 16-wide tiles per wave along n / k (block tile = 32*NI x 32*KI), reduction step 64 rows of m
// RG ("ragged"): N and K multiples of 8 only; tile columns past N / K are zero-filled on load, stores are guarded.
template <int NI, int KI, bool RG>
__global__ void __launch_bounds__(256, 2) gemm_tn_kernel(const bf16* __restrict__ dY, const bf16* __restrict__ X, int64_t M, int N,
                                                         int K, float* __restrict__ out, float* __restrict__ db,
                                                         int64_t split_stride, int tiles_n, int tiles_k, int64_t m_chunk) {
  constexpr int TNB = 32 * NI, TKB = 32 * KI, BR = 64;
  constexpr int Y_BYTES = BR * TN_RS, X_BYTES = BR * TN_RS;
  constexpr int YC = TNB / 8 * BR / 256;  // 16-byte chunks per thread for the dY tile (TNB/8 chunks per row)
  constexpr int XC = TKB / 8 * BR / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sY = smem;                 // [2][Y_BYTES]
  char* sX = smem + 2 * Y_BYTES;   // [2][X_BYTES]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  // XCD-aware order: every tile of one M-chunk ("split") runs on the same XCD at the same time, so the chunk's dY and X
  // rows are fetched from HBM once and shared through that XCD's L2 (round-robin placement re-fetched them per XCD:
  // 908 MB of HBM reads per launch against 221 MB of operands, rocprofv3 FETCH_SIZE)
  const int vb = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tile = vb % (tiles_n * tiles_k);
  const int split = vb / (tiles_n * tiles_k);
  const int n0 = (tile / tiles_k) * TNB, k0 = (tile % tiles_k) * TKB;
  const int64_t mbeg = (int64_t)split * m_chunk;
  const int64_t mend = mbeg + m_chunk < M ? mbeg + m_chunk : M;

  uint4 ry[YC], rx[XC];
#define TN_G_LOAD(mb)                                                                         \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < YC; ++i) {                                          \
      const int c = tid + 256 * i, row = c / (TNB / 8), cc = c % (TNB / 8);                   \
      const int64_t m = (mb) + row;                                                           \
      ry[i] = (m < mend && (!RG || n0 + cc * 8 < N)) ? *reinterpret_cast<const uint4*>(dY + m * N + n0 + cc * 8) : uint4{0, 0, 0, 0}; \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < XC; ++i) {                                          \
      const int c = tid + 256 * i, row = c / (TKB / 8), cc = c % (TKB / 8);                   \
      const int64_t m = (mb) + row;                                                           \
      rx[i] = (m < mend && (!RG || k0 + cc * 8 < K)) ? *reinterpret_cast<const uint4*>(X + m * K + k0 + cc * 8) : uint4{0, 0, 0, 0}; \
    }                                                                                         \
  }
#define TN_S_STORE(buf)                                                                       \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < YC; ++i) {                                          \
      const int c = tid + 256 * i, row = c / (TNB / 8), cc = c % (TNB / 8);                   \
      *reinterpret_cast<uint4*>(sY + (buf) * Y_BYTES + row * TN_RS + cc * 16) = ry[i];        \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < XC; ++i) {                                          \
      const int c = tid + 256 * i, row = c / (TKB / 8), cc = c % (TKB / 8);                   \
      *reinterpret_cast<uint4*>(sX + (buf) * X_BYTES + row * TN_RS + cc * 16) = rx[i];        \
    }                                                                                         \
  }

  f32x4 acc[KI][NI], accb[NI];
#pragma unroll
  for (int i = 0; i < KI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NI; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY: one extra MFMA per n-tile against an all-ones operand, done by the waves
  // that own the first k tile (wave-uniform condition)
  const bool do_bias = db != nullptr && k0 == 0 && wk == 0;
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

  // transposed-read address of this lane inside a 32-row k-substep: rows 16h + 4g + q, columns cb + 4p .. 4p+3
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_off = (4 * g + q) * TN_RS + p * 8;

  const int64_t nsteps = (mend - mbeg + BR - 1) / BR;
#ifndef MAE_DBG_TN_NO_LOAD
  TN_G_LOAD(mbeg)
#else
  for (int i = 0; i < YC; ++i) ry[i] = uint4{0, 0, 0, 0};
  for (int i = 0; i < XC; ++i) rx[i] = uint4{0, 0, 0, 0};
#endif
  for (int64_t st = 0; st < nsteps; ++st) {
    const int buf = (int)(st & 1);
    TN_S_STORE(buf)
    __syncthreads();
#ifndef MAE_DBG_TN_NO_LOAD
    if (st + 1 < nsteps) TN_G_LOAD(mbeg + (st + 1) * BR)
#endif
#ifdef MAE_DBG_TN_NO_MFMA
    continue;
#endif
    const char* yb = sY + buf * Y_BYTES + lane_off + (wn * NI * 16) * 2;
    const char* xb = sX + buf * X_BYTES + lane_off + (wk * KI * 16) * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 yf[NI], xf[KI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const bf16x4 lo = lds_read_tr(yb + (ks * 32) * TN_RS + ni * 32);
        const bf16x4 hi = lds_read_tr(yb + (ks * 32 + 16) * TN_RS + ni * 32);
        yf[ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int ki = 0; ki < KI; ++ki) {
        const bf16x4 lo = lds_read_tr(xb + (ks * 32) * TN_RS + ki * 32);
        const bf16x4 hi = lds_read_tr(xb + (ks * 32 + 16) * TN_RS + ki * 32);
        xf[ki] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int ki = 0; ki < KI; ++ki)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ki], yf[ni], acc[ki][ni], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[ni], accb[ni], 0, 0, 0);
      }
    }
  }
  // D[i = k][j = n]: lane holds n = tile col (lane&15), k = 4*(lane>>4) + r -> 16-byte store along k
  float* o = out + (int64_t)split * split_stride;
  if (do_bias && lane < 16) {  // every row of the ones-product is the same column sum: take row 0
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * NI * 16 + ni * 16 + lane;
      if (!RG || n < N) db[(int64_t)split * split_stride + n] = accb[ni][0];
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * NI * 16 + ni * 16 + (lane & 15);
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      const int k = k0 + wk * KI * 16 + ki * 16 + (lane >> 4) * 4;
      if (!RG || (n < N && k < K)) store4(o + (int64_t)n * K + k, acc[ki][ni]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad v2: the same contraction on 192 x 192 tiles with an LDS-DMA ring (no register staging, no ds_write).
//
//   block = 512 threads = 8 waves as 4 (n) x 2 (k), wave tile 48 (n) x 96 (k): NI = 3, KI = 6 MFMA tiles, 72 accumulators
//   stage = 64 reduction rows of X[:, k0:k0+192] then of dY[:, n0:n0+192], 384 B per row, 48 KiB; 3 stages, two K-steps of
//           DMA in flight behind the compute (waves 0-3 fetch X, waves 4-7 fetch dY, 6 x 1 KiB global_load_lds each)
//   LDS image: rows are NOT padded (the DMA writes 1 KiB runs); instead the 32-byte granule g of row r sits at
//           g ^ ((r >> 1) & 3).  A 32-lane half of ds_read_b64_tr_b16 takes 8 consecutive rows x 32 B: rows of equal
//           parity differ in (r >> 1) & 3, so they land in 4 different granules of one aligned 128-B group, and odd rows
//           are 32 banks away from even rows (384-B row stride) -> all 64 banks, conflict-free.  The swizzle is applied
//           on the DMA SOURCE address (the LDS destination of a DMA is lane-linear).
//   Every Linear of the ViT-S/8 and YAML-decoder shapes has N and K multiples of 192; other shapes use gemm_tn_kernel.
//   Staged bytes per flop are 0.65x those of the 128 x 128 register-staged kernel, and the ds_write_b128 traffic
//   (79 B/clk, the v1 limiter together with the transposed reads) is gone.
// ---------------------------------------------------------------------------------------------------
constexpr int T2 = 192, T2_BR = 64, T2_RS = T2 * 2, T2_HALF = T2_BR * T2_RS, T2_STAGE = 2 * T2_HALF, T2_NSTAGE = 3;
constexpr int T2_GPW = 6, T2_NI = 3, T2_KI = 6, T2_CPR = T2_RS / 16;  // 24 16-byte chunks per row

__device__ __forceinline__ void tn_glds16(const bf16* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
// The same DMA as one opaque instruction pair.  hipcc's waitcnt pass knows that the builtin writes LDS and, because the
// transposed-read builtin carries no alias information, puts `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16
// after every DMA issue -- which drains the two-steps-ahead ring at every step (the DMA phase and the MFMA phase of the
// wgrad kernel added up for exactly this reason).  Issued from inline asm the DMA is invisible to that pass; the counted
// tn_wait_vm<> + barrier below are what orders it against the reads, checked against the all-drained build
// (MAE_DBG_VMCNT0) like the NT kernel's.  m0 is written behind the compiler's back: nothing else in these kernels uses it.
__device__ __forceinline__ void tn_glds16_raw(const bf16* src, uint32_t lds_addr) {
  // (s_nop 0: an SALU write of m0 needs one wait state before an LDS-DMA reads it; hipcc's hazard recognizer, which inserts
  //  it for the builtin, does not look inside inline asm)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}
template <bool RAW>
__device__ __forceinline__ void tn_dma(const bf16* src, char* dst) {
  if (RAW) tn_glds16_raw(src, (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)dst));
  else tn_glds16(src, dst);
}
template <int N_>
__device__ __forceinline__ void tn_wait_vm() {
#ifdef MAE_DBG_VMCNT0  // see wait_vm() in k_gemm_mfma2.hip: the all-drained build the counted waits are checked against
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
#endif
}

template <bool RAW>
__global__ void __launch_bounds__(512, 2) gemm_tn2_kernel(const bf16* __restrict__ dY, const bf16* __restrict__ X, int64_t M, int N, int K,
                                                          float* __restrict__ out, float* __restrict__ db, int64_t split_stride,
                                                          int tiles_n, int tiles_k, int64_t m_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int vb = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int tile = vb % (tiles_n * tiles_k);
  const int split = vb / (tiles_n * tiles_k);
  const int n0 = (tile / tiles_k) * T2, k0 = (tile % tiles_k) * T2;
  const int64_t mbeg = (int64_t)split * m_chunk;
  const int64_t mend = mbeg + m_chunk < M ? mbeg + m_chunk : M;
  const int nsteps = mend > mbeg ? (int)((mend - mbeg + T2_BR - 1) / T2_BR) : 0;
  const int last_valid = nsteps ? (int)(mend - mbeg - (int64_t)(nsteps - 1) * T2_BR) : 0;  // rows of the last step (1..64)

  // ---- producer: this wave's 6 DMA instructions of a stage; lane -> (row, 16-byte slot) of the operand's 64 x 24 chunks
  const bool isY = wave >= 4;
  const bf16* gbase = isY ? dY + n0 : X + k0;
  const int64_t ld = isY ? N : K;
  int row_[T2_GPW], sc_[T2_GPW];
#pragma unroll
  for (int q = 0; q < T2_GPW; ++q) {
    const int c = ((wave & 3) * T2_GPW + q) * 64 + lane;
    const int row = c / T2_CPR, slot = c % T2_CPR;
    row_[q] = row;
    sc_[q] = ((((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1)) * 8;  // source element offset of the chunk stored in `slot`
  }
  const bf16 *p0 = gbase + (mbeg + row_[0]) * ld + sc_[0], *p1 = gbase + (mbeg + row_[1]) * ld + sc_[1],
             *p2 = gbase + (mbeg + row_[2]) * ld + sc_[2], *p3 = gbase + (mbeg + row_[3]) * ld + sc_[3],
             *p4 = gbase + (mbeg + row_[4]) * ld + sc_[4], *p5 = gbase + (mbeg + row_[5]) * ld + sc_[5];
  const int64_t inc = (int64_t)T2_BR * ld;
  int is_step = 0, is_stage = 0;
  auto issue = [&]() {
#ifdef MAE_DBG_TN_NO_LOAD
    return;
#endif
    char* dst = smem + is_stage * T2_STAGE + (isY ? T2_HALF : 0) + (wave & 3) * (T2_GPW * 1024);
    if (is_step == nsteps - 1 && last_valid < T2_BR) {
      // ragged last step: rows past the chunk are fetched from its last valid row (finite data, in bounds); the dY rows
      // among them are zeroed in LDS before use, which also removes their X rows from the product
      const int64_t mb = mbeg + (int64_t)is_step * T2_BR;
#pragma unroll
      for (int q = 0; q < T2_GPW; ++q) {
        const int64_t m = mb + (row_[q] < last_valid ? row_[q] : last_valid - 1);
        tn_dma<RAW>(gbase + m * ld + sc_[q], dst + q * 1024);
      }
    } else {
      tn_dma<RAW>(p0, dst); tn_dma<RAW>(p1, dst + 1024); tn_dma<RAW>(p2, dst + 2048);
      tn_dma<RAW>(p3, dst + 3072); tn_dma<RAW>(p4, dst + 4096); tn_dma<RAW>(p5, dst + 5120);
      p0 += inc; p1 += inc; p2 += inc; p3 += inc; p4 += inc; p5 += inc;
    }
    is_stage = is_stage == T2_NSTAGE - 1 ? 0 : is_stage + 1;
    ++is_step;
  };

  f32x4 acc[T2_KI][T2_NI], accb[T2_NI];
#pragma unroll
  for (int i = 0; i < T2_KI; ++i)
#pragma unroll
    for (int j = 0; j < T2_NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < T2_NI; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = db != nullptr && k0 == 0 && wk == 0;
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

  // ---- consumer: transposed-read addresses.  rows 4g + q (+16, +32), 8 bytes at column 4p of a 16-column tile
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3;
  const int sw = ((g & 1) << 1) | (q4 >> 1);  // (row >> 1) & 3 of every row this lane reads
  const int lane_off = (4 * g + q4) * T2_RS + p * 8;
  int yo[T2_NI], xo[T2_KI];
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) yo[ni] = T2_HALF + lane_off + (((wn * T2_NI + ni) ^ sw) * 32);
#pragma unroll
  for (int ki = 0; ki < T2_KI; ++ki) xo[ki] = lane_off + (((wk * T2_KI + ki) ^ sw) * 32);

  bf16x8 yf[2][T2_NI], xf[2][T2_KI];
#define TN2_READ_FRAGS(sb)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                     \
      _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni) {                                            \
        const bf16x4 lo = lds_read_tr((sb) + yo[ni] + (32 * h) * T2_RS);                                \
        const bf16x4 hi = lds_read_tr((sb) + yo[ni] + (32 * h + 16) * T2_RS);                           \
        yf[h][ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                     \
      }                                                                                                 \
      _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki) {                                            \
        const bf16x4 lo = lds_read_tr((sb) + xo[ki] + (32 * h) * T2_RS);                                \
        const bf16x4 hi = lds_read_tr((sb) + xo[ki] + (32 * h + 16) * T2_RS);                           \
        xf[h][ki] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                     \
      }                                                                                                 \
    }                                                                                                   \
  }
#if defined(MAE_DBG_TN_SLEEP)
#define TN2_MFMA() { __builtin_amdgcn_s_sleep(10); }
#elif defined(MAE_DBG_TN_NO_MFMA)
#define TN2_MFMA() {}
#else
#define TN2_MFMA()                                                                                      \
  {                                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                  \
      _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki)                                              \
        _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                            \
          acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks][ki], yf[ks][ni], acc[ki][ni], 0, 0, 0); \
      if (do_bias) {                                                                                    \
        _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                            \
          accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[ks][ni], accb[ni], 0, 0, 0);      \
      }                                                                                                 \
    }                                                                                                   \
  }
#endif
  // Ping-pong: waves 0-3 (group A) and 4-7 (group B) sit pairwise on the same SIMDs.  Between the two barriers of a step
  // A reads its fragments of step st while B runs the MFMAs of step st-1; after the second barrier A runs the MFMAs of
  // step st while B reads.  The LDS latency burst that follows a barrier is thus always covered by the other group's
  // MFMAs (one barrier per step with both groups in phase left the MFMA pipe idle ~55 % of the time).
  const bool grpB = wave >= 4;
  if (nsteps > 0) issue();
  if (nsteps > 1) issue();
  int cs = 0;
#define TN2_STEP_HEAD()                                                                                 \
    if (st + 1 < nsteps) tn_wait_vm<T2_GPW>(); else tn_wait_vm<0>();                                    \
    __builtin_amdgcn_s_barrier();                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    asm volatile("" ::: "memory");                                                                      \
    const char* sb = smem + cs * T2_STAGE;                                                              \
    if (st == nsteps - 1 && last_valid < T2_BR) {                                                       \
      for (int i = tid; i < (T2_BR - last_valid) * T2_CPR; i += 512)                                    \
        *reinterpret_cast<uint4*>(smem + cs * T2_STAGE + T2_HALF + (last_valid + i / T2_CPR) * T2_RS + (i % T2_CPR) * 16) = uint4{0, 0, 0, 0}; \
      __syncthreads();                                                                                  \
    }                                                                                                   \
    if (st + 2 < nsteps) issue(); /* refills the stage group B finished reading before this step's first barrier */
#define TN2_STEP_MID()                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __builtin_amdgcn_s_barrier();                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    asm volatile("" ::: "memory");
  // the two groups run separate copies of the loop (no control-flow merges between them: merged, the fragment registers
  // of the two schedules were kept apart by copies and the kernel spilled)
  if (!grpB) {
    for (int st = 0; st < nsteps; ++st) {
      TN2_STEP_HEAD()
      TN2_READ_FRAGS(sb)
      TN2_STEP_MID()
      TN2_MFMA()
      __builtin_amdgcn_sched_barrier(0);
      cs = cs == T2_NSTAGE - 1 ? 0 : cs + 1;
    }
  } else {
    for (int st = 0; st < nsteps; ++st) {
      TN2_STEP_HEAD()
      if (st > 0) TN2_MFMA()
      TN2_STEP_MID()
      TN2_READ_FRAGS(sb)
      __builtin_amdgcn_sched_barrier(0);
      cs = cs == T2_NSTAGE - 1 ? 0 : cs + 1;
    }
    if (nsteps > 0) TN2_MFMA()
  }
#undef TN2_STEP_HEAD
#undef TN2_STEP_MID
#undef TN2_READ_FRAGS
#undef TN2_MFMA
  // D[i = k][j = n]: lane holds n = tile col (lane & 15), k = 4 * (lane >> 4) + r -> 16-byte store along k
  float* o = out + (int64_t)split * split_stride;
  if (do_bias && lane < 16) {
#pragma unroll
    for (int ni = 0; ni < T2_NI; ++ni) db[(int64_t)split * split_stride + n0 + wn * (T2_NI * 16) + ni * 16 + lane] = accb[ni][0];
  }
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) {
    const int n = n0 + wn * (T2_NI * 16) + ni * 16 + (lane & 15);
#pragma unroll
    for (int ki = 0; ki < T2_KI; ++ki) {
      const int k = k0 + wk * (T2_KI * 16) + ki * 16 + (lane >> 4) * 4;
      store4(o + (int64_t)n * K + k, acc[ki][ni]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad v3 (the default; MAE_WGRAD=v2 selects the kernel above): the v2 tile, ring and LDS image with ONE barrier per step and all eight waves in
// phase; the fragment reads are software-pipelined ACROSS the barrier instead of ping-ponged between wave groups:
//     step st:  wait DMA(st) | lgkmcnt(0) | barrier | DMA(st+2) -> the stage read during step st-1
//               reads (st, rows 0-31)  interleaved 1:1 with the MFMAs of (st-1, rows 32-63)
//               reads (st, rows 32-63) interleaved 1:1 with the MFMAs of (st,   rows 0-31)
// Every accumulator sees the same products in the same order as in v2, so the two kernels agree bit for bit.
// ---------------------------------------------------------------------------------------------------
// One launch can serve TWO weight gradients that share their row count M (the engine pairs fc2 + fc1 and proj + qkv of a
// block): the tiles of both problems form one list, so the M-splits are chosen for the sum.  Alone, the 384 x 384 proj gradient
// needs 64 splits of 18 steps to fill the chip (38 MB of fp32 partials for a 0.6 MB result, 0.47 PF/s); next to qkv it takes
// 16 splits of 70 steps.  `out` / `db` point at the problem's slot inside split 0 of the slab (or at dW / db when there is one
// split); the kernel adds split * split_stride.
struct TnProb {
  const bf16* dY; const bf16* X; float* out; float* db;
  int N, K, tiles_k, tile_begin;
};
struct TnGroup {
  TnProb p[2];
  int nprob, total_tiles;
};

template <bool RAW>
__global__ void __launch_bounds__(512, 2) gemm_tn3_kernel(TnGroup grp, int64_t M, int64_t split_stride, int64_t m_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int vb = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int gtile = vb % grp.total_tiles;
  const int split = vb / grp.total_tiles;
  const bool second = grp.nprob > 1 && gtile >= grp.p[1].tile_begin;   // workgroup-uniform
  const bf16* __restrict__ dY = second ? grp.p[1].dY : grp.p[0].dY;
  const bf16* __restrict__ X = second ? grp.p[1].X : grp.p[0].X;
  float* __restrict__ out = second ? grp.p[1].out : grp.p[0].out;
  float* __restrict__ db = second ? grp.p[1].db : grp.p[0].db;
  const int N = second ? grp.p[1].N : grp.p[0].N, K = second ? grp.p[1].K : grp.p[0].K;
  const int tiles_k = second ? grp.p[1].tiles_k : grp.p[0].tiles_k;
  const int tile = gtile - (second ? grp.p[1].tile_begin : 0);
  const int n0 = (tile / tiles_k) * T2, k0 = (tile % tiles_k) * T2;
  const int64_t mbeg = (int64_t)split * m_chunk;
  const int64_t mend = mbeg + m_chunk < M ? mbeg + m_chunk : M;
  const int nsteps = mend > mbeg ? (int)((mend - mbeg + T2_BR - 1) / T2_BR) : 0;
  const int last_valid = nsteps ? (int)(mend - mbeg - (int64_t)(nsteps - 1) * T2_BR) : 0;

  const bool isY = wave >= 4;
  const bf16* gbase = isY ? dY + n0 : X + k0;
  const int64_t ld = isY ? N : K;
  int row_[T2_GPW], sc_[T2_GPW];
#pragma unroll
  for (int q = 0; q < T2_GPW; ++q) {
    const int c = ((wave & 3) * T2_GPW + q) * 64 + lane;
    const int row = c / T2_CPR, slot = c % T2_CPR;
    row_[q] = row;
    sc_[q] = ((((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1)) * 8;
    // a last tile column that sticks out of the matrix (widths that are not multiples of 192): its chunks are fetched from the
    // tile's first column instead (valid memory); they only ever reach accumulators whose stores are guarded out below
    if (sc_[q] >= (isY ? N - n0 : K - k0)) sc_[q] = 0;
  }
  const bf16 *p0 = gbase + (mbeg + row_[0]) * ld + sc_[0], *p1 = gbase + (mbeg + row_[1]) * ld + sc_[1],
             *p2 = gbase + (mbeg + row_[2]) * ld + sc_[2], *p3 = gbase + (mbeg + row_[3]) * ld + sc_[3],
             *p4 = gbase + (mbeg + row_[4]) * ld + sc_[4], *p5 = gbase + (mbeg + row_[5]) * ld + sc_[5];
  const int64_t inc = (int64_t)T2_BR * ld;
  int is_step = 0, is_stage = 0;
  auto issue = [&]() {
#ifdef MAE_DBG_TN_NO_LOAD   // phase ablation builds (tools/build_dbg_lib.sh tn_no_load tn_no_mfma tn3_nosched): timing probes, wrong values
    return;
#endif
    char* dst = smem + is_stage * T2_STAGE + (isY ? T2_HALF : 0) + (wave & 3) * (T2_GPW * 1024);
    if (is_step == nsteps - 1 && last_valid < T2_BR) {
      const int64_t mb = mbeg + (int64_t)is_step * T2_BR;
#pragma unroll
      for (int q = 0; q < T2_GPW; ++q) {
        const int64_t m = mb + (row_[q] < last_valid ? row_[q] : last_valid - 1);
        tn_dma<RAW>(gbase + m * ld + sc_[q], dst + q * 1024);
      }
    } else {
      tn_dma<RAW>(p0, dst); tn_dma<RAW>(p1, dst + 1024); tn_dma<RAW>(p2, dst + 2048);
      tn_dma<RAW>(p3, dst + 3072); tn_dma<RAW>(p4, dst + 4096); tn_dma<RAW>(p5, dst + 5120);
      p0 += inc; p1 += inc; p2 += inc; p3 += inc; p4 += inc; p5 += inc;
    }
    is_stage = is_stage == T2_NSTAGE - 1 ? 0 : is_stage + 1;
    ++is_step;
  };

  f32x4 acc[T2_KI][T2_NI], accb[T2_NI];
#pragma unroll
  for (int i = 0; i < T2_KI; ++i)
#pragma unroll
    for (int j = 0; j < T2_NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < T2_NI; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = db != nullptr && k0 == 0 && wk == 0;
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

  const int g = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3;
  const int sw = ((g & 1) << 1) | (q4 >> 1);
  const int lane_off = (4 * g + q4) * T2_RS + p * 8;
  int yo[T2_NI], xo[T2_KI];
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) yo[ni] = T2_HALF + lane_off + (((wn * T2_NI + ni) ^ sw) * 32);
#pragma unroll
  for (int ki = 0; ki < T2_KI; ++ki) xo[ki] = lane_off + (((wk * T2_KI + ki) ^ sw) * 32);

  bf16x8 yf[2][T2_NI], xf[2][T2_KI];
#define TN3_READ(sb, h)                                                                                 \
  {                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni) {                                              \
      const bf16x4 lo = lds_read_tr((sb) + yo[ni] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + yo[ni] + (32 * (h) + 16) * T2_RS);                           \
      yf[h][ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
    _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki) {                                              \
      const bf16x4 lo = lds_read_tr((sb) + xo[ki] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + xo[ki] + (32 * (h) + 16) * T2_RS);                           \
      xf[h][ki] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
  }
#ifdef MAE_DBG_TN_NO_MFMA
#define TN3_MFMA(h) { asm volatile("" :: "v"(xf[h][0]), "v"(xf[h][1]), "v"(xf[h][2]), "v"(xf[h][3]), "v"(xf[h][4]), "v"(xf[h][5]), "v"(yf[h][0]), "v"(yf[h][1]), "v"(yf[h][2])); }
#else
#define TN3_MFMA(h)                                                                                     \
  {                                                                                                     \
    _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki)                                                \
      _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                              \
        acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ki], yf[h][ni], acc[ki][ni], 0, 0, 0); \
  }
#endif
#define TN3_BIAS(h)                                                                                     \
  if (do_bias) {                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                                \
      accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[h][ni], accb[ni], 0, 0, 0);           \
  }
#if defined(MAE_DBG_TN3_NOSCHED) || defined(MAE_DBG_TN_NO_MFMA)
#define TN3_INTERLEAVE() {}
#else
#define TN3_INTERLEAVE()                                                                                \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < T2_KI * T2_NI; ++i) {                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
    }                                                                                                   \
  }
#endif
#define TN3_STEP_TOP(st)                                                                                \
    if ((st) + 1 < nsteps) tn_wait_vm<T2_GPW>(); else tn_wait_vm<0>();                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of the stage refilled below have landed */ \
    __builtin_amdgcn_s_barrier();                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    asm volatile("" ::: "memory");                                                                      \
    const char* sb = smem + cs * T2_STAGE;                                                              \
    if ((st) == nsteps - 1 && last_valid < T2_BR) {                                                     \
      for (int i = tid; i < (T2_BR - last_valid) * T2_CPR; i += 512)                                    \
        *reinterpret_cast<uint4*>(smem + cs * T2_STAGE + T2_HALF + (last_valid + i / T2_CPR) * T2_RS + (i % T2_CPR) * 16) = uint4{0, 0, 0, 0}; \
      __syncthreads();                                                                                  \
    }                                                                                                   \
    if ((st) + 2 < nsteps) issue();                                                                     \
    __builtin_amdgcn_sched_barrier(0);
  int cs = 0;
  if (nsteps > 0) {
    issue();
    if (nsteps > 1) issue();
    {  // step 0: nothing to multiply yet while the first half is read
      TN3_STEP_TOP(0)
      TN3_READ(sb, 0)
      __builtin_amdgcn_sched_barrier(0);
      TN3_READ(sb, 1)
      TN3_MFMA(0)
      TN3_INTERLEAVE()
      __builtin_amdgcn_sched_barrier(0);
      TN3_BIAS(0)
      cs = 1;
    }
#ifdef MAE_DBG_TN3_TRACE   // cycle breakdown of a step (tools/build_dbg_lib.sh tn3_trace): prints from workgroup 0, perturbs the timing a little
    uint64_t tr_bar = 0, tr_dma = 0, tr_p1 = 0, tr_p2 = 0, tr_prev = 0;
#define TN3_TS(var) { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#endif
    for (int st = 1; st < nsteps; ++st) {
#ifdef MAE_DBG_TN3_TRACE
      uint64_t ta, tb, tc, td, te;
      TN3_TS(ta)
      if ((st) + 1 < nsteps) tn_wait_vm<T2_GPW>(); else tn_wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      const char* sb = smem + cs * T2_STAGE;
      TN3_TS(tb)
      if ((st) + 2 < nsteps) issue();
      TN3_TS(tc)
#else
      TN3_STEP_TOP(st)
#endif
      TN3_READ(sb, 0)
      TN3_MFMA(1)
      TN3_INTERLEAVE()
      __builtin_amdgcn_sched_barrier(0);
      TN3_BIAS(1)
      __builtin_amdgcn_sched_barrier(0);
#ifdef MAE_DBG_TN3_TRACE
      TN3_TS(td)
#endif
      TN3_READ(sb, 1)
      TN3_MFMA(0)
      TN3_INTERLEAVE()
      __builtin_amdgcn_sched_barrier(0);
      TN3_BIAS(0)
#ifdef MAE_DBG_TN3_TRACE
      TN3_TS(te)
      if (st >= 4 && st + 4 < nsteps) { tr_bar += tb - ta; tr_dma += tc - tb; tr_p1 += td - tc; tr_p2 += te - td; tr_prev += 1; }
#endif
      cs = cs == T2_NSTAGE - 1 ? 0 : cs + 1;
    }
#ifdef MAE_DBG_TN3_TRACE
    if (blockIdx.x == 0 && lane == 0 && tr_prev)
      printf("tn3 trace wave %d steps %d | wait+barrier %d | dma issue %d | phase1 (18 mfma + 18 reads) %d | phase2 %d cycles/step\n", wave, (int)tr_prev,
             (int)(tr_bar / tr_prev), (int)(tr_dma / tr_prev), (int)(tr_p1 / tr_prev), (int)(tr_p2 / tr_prev));
#endif
    TN3_MFMA(1)
    TN3_BIAS(1)
  }
#undef TN3_STEP_TOP
#undef TN3_INTERLEAVE
#undef TN3_BIAS
#undef TN3_MFMA
#undef TN3_READ
  float* o = out + (int64_t)split * split_stride;
  if (do_bias && lane < 16) {
#pragma unroll
    for (int ni = 0; ni < T2_NI; ++ni) {
      const int n = n0 + wn * (T2_NI * 16) + ni * 16 + lane;
      if (n < N) db[(int64_t)split * split_stride + n] = accb[ni][0];
    }
  }
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) {
    const int n = n0 + wn * (T2_NI * 16) + ni * 16 + (lane & 15);
#pragma unroll
    for (int ki = 0; ki < T2_KI; ++ki) {
      const int k = k0 + wk * (T2_KI * 16) + ki * 16 + (lane >> 4) * 4;
      if (n < N && k < K) store4(o + (int64_t)n * K + k, acc[ki][ni]);
    }
  }
}

// ---- round 3: the same kernel with buffer DMA (scalar step offset, range-checked rows), the six pieces of a step issued in pairs
// between thirds of phase A's MFMAs instead of in one burst at the top of the step, and no special cases in the step (MAE_WGRAD=v4)
template <bool BURST>
__global__ void __launch_bounds__(512, 2) gemm_tn4_kernel(TnGroup grp, int64_t M, int64_t split_stride, int64_t m_chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int vb = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int gtile = vb % grp.total_tiles;
  const int split = vb / grp.total_tiles;
  const bool second = grp.nprob > 1 && gtile >= grp.p[1].tile_begin;   // workgroup-uniform
  const bf16* __restrict__ dY = second ? grp.p[1].dY : grp.p[0].dY;
  const bf16* __restrict__ X = second ? grp.p[1].X : grp.p[0].X;
  float* __restrict__ out = second ? grp.p[1].out : grp.p[0].out;
  float* __restrict__ db = second ? grp.p[1].db : grp.p[0].db;
  const int N = second ? grp.p[1].N : grp.p[0].N, K = second ? grp.p[1].K : grp.p[0].K;
  const int tiles_k = second ? grp.p[1].tiles_k : grp.p[0].tiles_k;
  const int tile = gtile - (second ? grp.p[1].tile_begin : 0);
  const int n0 = (tile / tiles_k) * T2, k0 = (tile % tiles_k) * T2;
  const int64_t mbeg = (int64_t)split * m_chunk;
  const int64_t mend = mbeg + m_chunk < M ? mbeg + m_chunk : M;
  const int nsteps = mend > mbeg ? (int)((mend - mbeg + T2_BR - 1) / T2_BR) : 0;

  // ---- producer side (round 3): buffer DMA with one scalar offset per step.  Waves 0-3 fetch X rows, waves 4-7 dY rows; a lane's six
  // (row, chunk) offsets never change; rows past the split's end are out of the descriptor's range and arrive as zeros (no clamp, no
  // zero-fill pass for a ragged last step); the stream is never switched off (phantom pieces past the last step land in a stage
  // nobody reads any more), so every step waits with the same count.
  const bool isY = wave >= 4;
  const int64_t ld = isY ? N : K;
  const uint32_t ldb = (uint32_t)ld * 2u;
  typedef __attribute__((ext_vector_type(4))) int i32x4_;
  const uint64_t gaddr = (uint64_t)(uintptr_t)(isY ? dY : X);
  const i32x4_ rsrc = i32x4_{(int)(uint32_t)gaddr, (int)(uint32_t)((gaddr >> 32) & 0xffffu), (int)((uint32_t)mend * ldb), 0x00020000};
  uint32_t vo[T2_GPW];
#pragma unroll
  for (int q = 0; q < T2_GPW; ++q) {
    const int c = ((wave & 3) * T2_GPW + q) * 64 + lane;
    const int row = c / T2_CPR, slot = c % T2_CPR;
    int sc = ((((slot >> 1) ^ ((row >> 1) & 3)) << 1) | (slot & 1)) * 8;
    // a last tile column that sticks out of the matrix (widths that are not multiples of 192): its chunks are fetched from the
    // tile's first column instead (valid memory); they only ever reach accumulators whose stores are guarded out below
    if (sc >= (isY ? N - n0 : K - k0)) sc = 0;
    vo[q] = (uint32_t)row * ldb + (uint32_t)((isY ? n0 : k0) + sc) * 2u;
  }
  uint32_t soff = (uint32_t)mbeg * ldb;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (uint32_t)((isY ? T2_HALF : 0) + (wave & 3) * (T2_GPW * 1024));
  uint32_t ldst = lds0;
  auto issue_piece = [&](int q) {
#ifndef MAE_DBG_TN_NO_LOAD
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldst + (uint32_t)(q * 1024)), "v"(vo[q]), "s"(rsrc), "s"(soff) : "memory");
#endif
  };
  auto issue_next = [&]() {
    soff += (uint32_t)T2_BR * ldb;
    ldst = ldst == lds0 + (uint32_t)((T2_NSTAGE - 1) * T2_STAGE) ? lds0 : ldst + (uint32_t)T2_STAGE;
  };

  f32x4 acc[T2_KI][T2_NI], accb[T2_NI];
#pragma unroll
  for (int i = 0; i < T2_KI; ++i)
#pragma unroll
    for (int j = 0; j < T2_NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < T2_NI; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = db != nullptr && k0 == 0 && wk == 0;
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

  const int g = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3;
  const int sw = ((g & 1) << 1) | (q4 >> 1);
  const int lane_off = (4 * g + q4) * T2_RS + p * 8;
  int yo[T2_NI], xo[T2_KI];
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) yo[ni] = T2_HALF + lane_off + (((wn * T2_NI + ni) ^ sw) * 32);
#pragma unroll
  for (int ki = 0; ki < T2_KI; ++ki) xo[ki] = lane_off + (((wk * T2_KI + ki) ^ sw) * 32);

  bf16x8 yf[2][T2_NI], xf[2][T2_KI];
#define TN3_READ(sb, h)                                                                                 \
  {                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni) {                                              \
      const bf16x4 lo = lds_read_tr((sb) + yo[ni] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + yo[ni] + (32 * (h) + 16) * T2_RS);                           \
      yf[h][ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
    _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki) {                                              \
      const bf16x4 lo = lds_read_tr((sb) + xo[ki] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + xo[ki] + (32 * (h) + 16) * T2_RS);                           \
      xf[h][ki] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
  }
#ifdef MAE_DBG_TN_NO_MFMA
#define TN3_MFMA(h) { asm volatile("" :: "v"(xf[h][0]), "v"(xf[h][1]), "v"(xf[h][2]), "v"(xf[h][3]), "v"(xf[h][4]), "v"(xf[h][5]), "v"(yf[h][0]), "v"(yf[h][1]), "v"(yf[h][2])); }
#else
#define TN3_MFMA(h)                                                                                     \
  {                                                                                                     \
    _Pragma("unroll") for (int ki = 0; ki < T2_KI; ++ki)                                                \
      _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                              \
        acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ki], yf[h][ni], acc[ki][ni], 0, 0, 0); \
  }
#endif
#define TN3_BIAS(h)                                                                                     \
  if (do_bias) {                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                                \
      accb[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[h][ni], accb[ni], 0, 0, 0);           \
  }
#if defined(MAE_DBG_TN3_NOSCHED) || defined(MAE_DBG_TN_NO_MFMA)
#define TN3_INTERLEAVE() {}
#else
#define TN3_INTERLEAVE()                                                                                \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < T2_KI * T2_NI; ++i) {                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
    }                                                                                                   \
  }
#endif
#define TN4_STEP_TOP()                                                                                  \
    tn_wait_vm<T2_GPW>();                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of the stage refilled below have landed */ \
    __builtin_amdgcn_s_barrier();                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    asm volatile("" ::: "memory");                                                                        \
    const char* sb = smem + cs * T2_STAGE;                                                                \
    if (BURST) {                                                                                          \
      _Pragma("unroll") for (int q = 0; q < T2_GPW; ++q) issue_piece(q);                                  \
      issue_next();                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
  // a third of phase A: three fragments of half 0 are read beside six MFMAs of the previous step's half 1, then two DMA pieces go out
#define TN4_READ_Y(sb, h)                                                                               \
  {                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni) {                                              \
      const bf16x4 lo = lds_read_tr((sb) + yo[ni] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + yo[ni] + (32 * (h) + 16) * T2_RS);                           \
      yf[h][ni] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
  }
#define TN4_READ_X(sb, h, K0, K1)                                                                       \
  {                                                                                                     \
    _Pragma("unroll") for (int ki = (K0); ki < (K1); ++ki) {                                            \
      const bf16x4 lo = lds_read_tr((sb) + xo[ki] + (32 * (h)) * T2_RS);                                \
      const bf16x4 hi = lds_read_tr((sb) + xo[ki] + (32 * (h) + 16) * T2_RS);                           \
      xf[h][ki] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                       \
    }                                                                                                   \
  }
#define TN4_MFMA(h, K0, K1)                                                                             \
  {                                                                                                     \
    _Pragma("unroll") for (int ki = (K0); ki < (K1); ++ki)                                              \
      _Pragma("unroll") for (int ni = 0; ni < T2_NI; ++ni)                                              \
        acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[h][ki], yf[h][ni], acc[ki][ni], 0, 0, 0); \
  }
#define TN4_IL6()                                                                                       \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
    }                                                                                                   \
  }
#define TN4_PHASE_A(sb)                                                                                 \
    TN4_READ_Y(sb, 0) TN4_MFMA(1, 0, 2) TN4_IL6()                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (!BURST) { issue_piece(0); issue_piece(1); }                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    TN4_READ_X(sb, 0, 0, 3) TN4_MFMA(1, 2, 4) TN4_IL6()                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (!BURST) { issue_piece(2); issue_piece(3); }                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    TN4_READ_X(sb, 0, 3, 6) TN4_MFMA(1, 4, 6) TN4_IL6()                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (!BURST) { issue_piece(4); issue_piece(5); issue_next(); }                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    TN3_BIAS(1)                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
  int cs = 0;
  if (nsteps > 0) {
    // two steps of DMA in flight before the first wait
#pragma unroll
    for (int q = 0; q < T2_GPW; ++q) issue_piece(q);
    issue_next();
#pragma unroll
    for (int q = 0; q < T2_GPW; ++q) issue_piece(q);
    issue_next();
    {  // step 0: nothing to multiply yet while the first half is read (the fragments of "the previous half 1" are zeros)
      const bf16 z = (bf16)0.0f;
      const bf16x8 zz = bf16x8{z, z, z, z, z, z, z, z};
#pragma unroll
      for (int ni = 0; ni < T2_NI; ++ni) yf[1][ni] = zz;
#pragma unroll
      for (int ki = 0; ki < T2_KI; ++ki) xf[1][ki] = zz;
    }
    for (int st = 0; st < nsteps; ++st) {
      TN4_STEP_TOP()
      TN4_PHASE_A(sb)
      TN3_READ(sb, 1)
      TN3_MFMA(0)
      TN3_INTERLEAVE()
      __builtin_amdgcn_sched_barrier(0);
      TN3_BIAS(0)
      cs = cs == T2_NSTAGE - 1 ? 0 : cs + 1;
    }
    TN3_MFMA(1)
    TN3_BIAS(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the phantom pieces issued past the last step land before the LDS is released
  }
#undef TN4_PHASE_A
#undef TN4_IL6
#undef TN4_MFMA
#undef TN4_READ_X
#undef TN4_READ_Y
#undef TN4_STEP_TOP
#undef TN3_INTERLEAVE
#undef TN3_BIAS
#undef TN3_MFMA
#undef TN3_READ
  float* o = out + (int64_t)split * split_stride;
  if (do_bias && lane < 16) {
#pragma unroll
    for (int ni = 0; ni < T2_NI; ++ni) {
      const int n = n0 + wn * (T2_NI * 16) + ni * 16 + lane;
      if (n < N) db[(int64_t)split * split_stride + n] = accb[ni][0];
    }
  }
#pragma unroll
  for (int ni = 0; ni < T2_NI; ++ni) {
    const int n = n0 + wn * (T2_NI * 16) + ni * 16 + (lane & 15);
#pragma unroll
    for (int ki = 0; ki < T2_KI; ++ki) {
      const int k = k0 + wk * (T2_KI * 16) + ki * 16 + (lane >> 4) * 4;
      if (n < N && k < K) store4(o + (int64_t)n * K + k, acc[ki][ni]);
    }
  }
}

// the 192 x 192 ring kernels: every width that is a multiple of 192, and (v3 only, whose loads and stores are guarded by column)
// widths that fill their last tile column well enough -- 1024 = 5.33 tiles, 512 x 2048 = 3 x 11 tiles at 86 % -- to beat the
// 128 x 128 register-staged kernel (measured 0.55-0.70 PF/s at 1024-wide layers against 0.9 PF/s x fill for this one)
static bool wgrad2_shape_ok(int64_t M, int N, int K, bool* ragged = nullptr) {   // no environment in here: the scratch size depends on it
  if (ragged) *ragged = false;
  if (M < 4096) return false;
  if (N % T2 == 0 && K % T2 == 0) return true;
  if (N % 8 != 0 || K % 8 != 0 || N < T2 || K < T2) return false;
  if (ragged) *ragged = true;
  const int64_t tiles = (int64_t)cdiv(N, T2) * cdiv(K, T2);
  return (int64_t)N * K * 100 >= tiles * T2 * T2 * 75;  // >= 75 % of the tile area is real
}
static bool wgrad2_ok(int64_t M, int N, int K) {
  const char* v = getenv("MAE_WGRAD");  // A/B switch, read per call: v1 = register-staged tiles everywhere, v2 = the round-1 ring kernel
  if (v && v[0] == 'v' && v[1] == '1') return false;
  bool ragged = false;
  if (!wgrad2_shape_ok(M, N, K, &ragged)) return false;
  return !(ragged && v && v[0] == 'v' && v[1] == '2');  // only the v3 kernel guards its loads and stores by column
}
// Splits over M for the 192 x 192 ring kernels: one workgroup per CU is resident (144 KiB of LDS), so the launch runs in rounds
// of num_cus() workgroups.  S minimises  rounds(S) x (time of one workgroup at S = 1) / S  +  S x (slab write + read time):
// with few tiles that is the old rule S = CUs / tiles (one full round); with more tiles than half the CUs (1024-wide layers:
// 132 tiles on 256 CUs left 48 % of the chip idle at S = 1) several rounds of shorter workgroups win.
static int wgrad2_splits(int64_t M, int N, int K) {
  const int64_t tiles = (int64_t)cdiv(N, T2) * cdiv(K, T2), cus = num_cus();
  const int64_t smax = std::min<int64_t>(512, std::max<int64_t>(1, M / 256));  // at least 4 reduction steps per block
  const double t1 = (double)M / T2_BR * 1.5;                                    // us: ~1.5 us per 64-row step
  const double slab = (double)N * K * 8.0 / 5.0e6;                              // us per split: fp32 partials written once, read once, ~5 TB/s
  int best = 1;
  double best_cost = 1e30;
  for (int64_t S = 1; S <= smax; ++S) {
    const double cost = (double)cdiv(tiles * S, cus) * t1 / (double)S + (double)S * slab;
    if (cost < best_cost * 0.999) { best_cost = cost; best = (int)S; }
  }
  return best;
}

// ordered (deterministic) sum of the per-split slabs: [S][N*K weight partials | N bias partials].
// 256 threads = 32 outputs (float4) x 8 slices of the split index; slice sl adds slabs sl, sl+8, ... and the 8 partial
// sums are combined through LDS in slice order, so the result does not depend on the grid.  (One thread per output
// walking all S slabs serially took 10 us per launch at S = 56 and dominated the small-tile wgrads.)
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int S, int64_t stride4, int64_t nw4,
                                                          int64_t nb4, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ f32x4 red[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t total = nw4 + nb4;
  for (int64_t base = blockIdx.x * 32ll; base < total; base += (int64_t)gridDim.x * 32) {
    const int64_t i = base + o;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < total)
      for (int s = sl; s < S; s += 8) acc += load4(slabs + ((int64_t)s * stride4 + i) * 4);
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
      f32x4 t = red[0][o];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += red[k][o];
      if (i < nw4) store4(dW + i * 4, t); else store4(db + (i - nw4) * 4, t);
    }
    __syncthreads();
  }
}

// the same ordered sum for the slabs of a pair launch: up to four output segments (dW0, db0, dW1, db1) laid out back to back
// inside every split's slab, in float4 units
struct SlabSegs { int64_t end4[4]; float* dst[4]; };
__global__ void __launch_bounds__(256) slab_reduce_group_kernel(const float* __restrict__ slabs, int S, int64_t stride4, SlabSegs sg) {
  __shared__ f32x4 red[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t total = sg.end4[3];
  for (int64_t base = blockIdx.x * 32ll; base < total; base += (int64_t)gridDim.x * 32) {
    const int64_t i = base + o;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i < total)
      for (int s = sl; s < S; s += 8) acc += load4(slabs + ((int64_t)s * stride4 + i) * 4);
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
      f32x4 t = red[0][o];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += red[k][o];
      const int seg = i < sg.end4[0] ? 0 : (i < sg.end4[1] ? 1 : (i < sg.end4[2] ? 2 : 3));
      const int64_t b4 = seg == 0 ? 0 : sg.end4[seg - 1];
      store4(sg.dst[seg] + (i - b4) * 4, t);
    }
    __syncthreads();
  }
}

// the buffer-DMA kernel addresses a matrix with 32-bit byte offsets (two steps of phantom rows past the end included)
static bool tn4_range_ok(int64_t M, int N, int K) { return (M + 4 * T2_BR) * (int64_t)std::max(N, K) * 2 < ((int64_t)1 << 32); }

static int wgrad_splits(int64_t M, int N, int K) {
  const int tn = N % 128 == 0 || N % 64 != 0 ? 128 : 64, tk = K % 128 == 0 || K % 64 != 0 ? 128 : 64;  // ragged dims take 128-wide tiles
  const int64_t tiles = (int64_t)cdiv(N, tn) * cdiv(K, tk);
  int64_t S = std::max<int64_t>(1, 512 / tiles);
  S = std::min<int64_t>(S, std::max<int64_t>(1, M / 512));  // at least 8 reduction steps per block
  return (int)std::min<int64_t>(S, 64);
}

int64_t mfma_wgrad_scratch_bytes(int64_t M, int N, int K) {
  if (N % 8 != 0 || K % 8 != 0) return 0;
  int S = wgrad_splits(M, N, K);
  if (wgrad2_shape_ok(M, N, K)) S = std::max(S, wgrad2_splits(M, N, K));  // whichever kernel the A/B switch selects at launch time fits
  return S > 1 ? round_up((int64_t)S * ((int64_t)N * K + N) * 4, 256) : 0;
}

template <int NI, int KI, bool RG = false>
static int launch_tn(const bf16* dY, const bf16* X, int64_t M, int N, int K, float* out, float* db, int64_t split_stride, int S,
                     int64_t m_chunk, hipStream_t s) {
  const int tiles_n = (int)cdiv(N, 32 * NI), tiles_k = (int)cdiv(K, 32 * KI);
  const size_t lds = 4 * 64 * TN_RS;
  auto kern = gemm_tn_kernel<NI, KI, RG>;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_n * tiles_k * S)), dim3(256), lds, s, dY, X, M, N, K, out, db, split_stride, tiles_n, tiles_k, m_chunk);
  MAE_LAUNCH_CHECK();
  return 0;
}

int mfma_linear_wgrad(const bf16* dY, const bf16* X, int64_t M, int N, int K, float* dW, float* db, void* slab, hipStream_t s) {
  if (N % 8 != 0 || K % 8 != 0 || N < 16 || K < 16 || M < 1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)db | (uintptr_t)slab) & 15) != 0) return MFMA_UNSUPPORTED;
  const bool v2 = wgrad2_ok(M, N, K);
  const int S = v2 ? wgrad2_splits(M, N, K) : wgrad_splits(M, N, K);
  if (S > 1 && !slab) return MFMA_UNSUPPORTED;
  const int64_t m_chunk = round_up(cdiv(M, S), 64);
  const int64_t stride = S > 1 ? (int64_t)N * K + N : 0;
  float* out = S > 1 ? reinterpret_cast<float*>(slab) : dW;
  float* dbo = !db ? nullptr : (S > 1 ? out + (int64_t)N * K : db);
  const bool n128 = N % 128 == 0, k128 = K % 128 == 0;
  int r;
  if (v2) {
    const int tiles_n = (int)cdiv(N, T2), tiles_k = (int)cdiv(K, T2);
    const int lds = T2_NSTAGE * T2_STAGE;
    // A/B switch (tools/gemm_bench.py --wgrad): v2 | v2r | v3 | v3r | v4 | v4b; default = v4 (buffer DMA), burst issue (v4b) for the 192-wide
    // HBM-bound decoder shapes, v3r where a matrix is beyond 32-bit byte offsets
    const char* ev = getenv("MAE_WGRAD");
    const bool sel = ev && ev[0] == 'v' && (ev[1] == '2' || ev[1] == '3' || ev[1] == '4');
    const bool k4 = (!sel || ev[1] == '4') && tn4_range_ok(M, N, K);
    const bool burst = sel && ev[1] == '4' ? ev[2] == 'b' : std::min(N, K) <= T2;
    const bool k3 = !sel || ev[1] == '3' || k4, raw = !sel || ev[1] == '4' || ev[2] == 'r';
    if (k3) {
      TnGroup g{};
      g.p[0] = TnProb{dY, X, out, dbo, N, K, tiles_k, 0};
      g.nprob = 1; g.total_tiles = tiles_n * tiles_k;
      auto kern = k4 ? (burst ? gemm_tn4_kernel<true> : gemm_tn4_kernel<false>) : (raw ? gemm_tn3_kernel<true> : gemm_tn3_kernel<false>);
      MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, dim3((unsigned)(g.total_tiles * S)), dim3(512), lds, s, g, M, stride, m_chunk);
    } else {
      auto kern = raw ? gemm_tn2_kernel<true> : gemm_tn2_kernel<false>;
      MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_n * tiles_k * S)), dim3(512), lds, s, dY, X, M, N, K, out, dbo, stride, tiles_n, tiles_k, m_chunk);
    }
    MAE_LAUNCH_CHECK();
    r = 0;
  } else if (N % 64 != 0 || K % 64 != 0) {
    r = launch_tn<4, 4, true>(dY, X, M, N, K, out, dbo, stride, S, m_chunk, s);
  } else if (n128 && k128) r = launch_tn<4, 4>(dY, X, M, N, K, out, dbo, stride, S, m_chunk, s);
  else if (n128) r = launch_tn<4, 2>(dY, X, M, N, K, out, dbo, stride, S, m_chunk, s);
  else if (k128) r = launch_tn<2, 4>(dY, X, M, N, K, out, dbo, stride, S, m_chunk, s);
  else r = launch_tn<2, 2>(dY, X, M, N, K, out, dbo, stride, S, m_chunk, s);
  if (r) return r;
  if (S > 1) {
    const int64_t nw4 = (int64_t)N * K / 4, nb4 = db ? N / 4 : 0;
    const int grid = (int)std::min<int64_t>(cdiv(nw4 + nb4, 32), 4096);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)slab, S, stride / 4, nw4, nb4, dW, db);
    MAE_LAUNCH_CHECK();
  }
  return 0;
}


// ---- two weight gradients with the same M in one launch (see TnGroup) -------------------------------------------------
static int wgrad_pair_splits(int64_t M, int64_t tiles, int64_t nk_sum) {
  const int64_t cus = num_cus();
  const int64_t smax = std::min<int64_t>(512, std::max<int64_t>(1, M / 256));
  const double t1 = (double)M / T2_BR * 1.5, slab = (double)nk_sum * 8.0 / 5.0e6;
  int best = 2;
  double best_cost = 1e30;
  for (int64_t S = 2; S <= smax; ++S) {   // at least two splits: the outputs always go through the slab
    const double cost = (double)cdiv(tiles * S, cus) * t1 / (double)S + (double)S * slab;
    if (cost < best_cost * 0.999) { best_cost = cost; best = (int)S; }
  }
  return best;
}
static bool wgrad_pair_ok(int64_t M, int N0, int K0, int N1, int K1) {
  const char* v = getenv("MAE_WGRAD");  // any A/B selection of a single-problem kernel, or MAE_WGRAD_PAIR=0, keeps the launches apart
  const char* pr = getenv("MAE_WGRAD_PAIR");
  if ((v && v[0]) || (pr && pr[0] == '0')) return false;
  return M >= 8192 && wgrad2_shape_ok(M, N0, K0) && wgrad2_shape_ok(M, N1, K1) && N0 % 4 == 0 && N1 % 4 == 0;
}
int64_t mfma_wgrad_pair_scratch_bytes(int64_t M, int N0, int K0, int N1, int K1) {
  if (M < 8192 || !wgrad2_shape_ok(M, N0, K0) || !wgrad2_shape_ok(M, N1, K1)) return 0;
  const int64_t tiles = (int64_t)cdiv(N0, T2) * cdiv(K0, T2) + (int64_t)cdiv(N1, T2) * cdiv(K1, T2);
  const int64_t per = (int64_t)N0 * K0 + N0 + (int64_t)N1 * K1 + N1;
  return round_up((int64_t)wgrad_pair_splits(M, tiles, (int64_t)N0 * K0 + (int64_t)N1 * K1) * per * 4, 256);
}
int mfma_linear_wgrad_pair(const bf16* dY0, const bf16* X0, int N0, int K0, float* dW0, float* db0, const bf16* dY1, const bf16* X1,
                           int N1, int K1, float* dW1, float* db1, int64_t M, void* slab, hipStream_t s) {
  if (!wgrad_pair_ok(M, N0, K0, N1, K1) || !slab || !db0 || !db1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)dY0 | (uintptr_t)X0 | (uintptr_t)dW0 | (uintptr_t)db0 | (uintptr_t)dY1 | (uintptr_t)X1 | (uintptr_t)dW1 | (uintptr_t)db1 |
        (uintptr_t)slab) & 15) != 0)
    return MFMA_UNSUPPORTED;
  const int tn0 = (int)cdiv(N0, T2), tk0 = (int)cdiv(K0, T2), tn1 = (int)cdiv(N1, T2), tk1 = (int)cdiv(K1, T2);
  const int64_t nk0 = (int64_t)N0 * K0, nk1 = (int64_t)N1 * K1;
  const int S = wgrad_pair_splits(M, (int64_t)tn0 * tk0 + (int64_t)tn1 * tk1, nk0 + nk1);
  const int64_t m_chunk = round_up(cdiv(M, S), 64);
  const int64_t stride = nk0 + N0 + nk1 + N1;   // floats per split: [dW0 | db0 | dW1 | db1]
  float* base = reinterpret_cast<float*>(slab);
  TnGroup g{};
  g.p[0] = TnProb{dY0, X0, base, base + nk0, N0, K0, tk0, 0};
  g.p[1] = TnProb{dY1, X1, base + nk0 + N0, base + nk0 + N0 + nk1, N1, K1, tk1, tn0 * tk0};
  g.nprob = 2; g.total_tiles = tn0 * tk0 + tn1 * tk1;
  const int lds = T2_NSTAGE * T2_STAGE;
  const char* pv = getenv("MAE_WGRAD_PAIR");   // A/B: "3" the pointer-DMA kernel, "4" / "4b" the buffer-DMA kernel with spread / burst issue
  const bool k4 = !(pv && pv[0] == '3') && tn4_range_ok(M, N0, K0) && tn4_range_ok(M, N1, K1);
  const bool burst = pv && pv[0] == '4' ? pv[1] == 'b' : std::min(std::min(N0, K0), std::min(N1, K1)) <= T2;
  auto kern = k4 ? (burst ? gemm_tn4_kernel<true> : gemm_tn4_kernel<false>) : gemm_tn3_kernel<true>;
  MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.total_tiles * S)), dim3(512), lds, s, g, M, stride, m_chunk);
  MAE_LAUNCH_CHECK();
  SlabSegs sg;
  sg.end4[0] = nk0 / 4; sg.end4[1] = sg.end4[0] + N0 / 4; sg.end4[2] = sg.end4[1] + nk1 / 4; sg.end4[3] = sg.end4[2] + N1 / 4;
  sg.dst[0] = dW0; sg.dst[1] = db0; sg.dst[2] = dW1; sg.dst[3] = db1;
  const int grid = (int)std::min<int64_t>(cdiv(sg.end4[3], 32), 4096);
  hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(grid), dim3(256), 0, s, (const float*)slab, S, stride / 4, sg);
  MAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace mae
