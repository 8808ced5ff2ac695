// Persistent bf16 MFMA NT GEMM for gfx950, v3 (round 3): the tiles, LDS image and epilogues of gemm_nt2_kernel with a new K-loop.
//
//   out[M,N] = A[M,K] * W[N,K]^T (+bias) with the NONE / GELU_GRAD / GELU_ACT / MUL epilogues the engine's bf16 path uses.
//
// What round 2's kernel lost, read off its ISA and off tools/micro/rates_probe.hip (profiles/r03_rates_probe.txt):
//   * hipcc reused ONE fragment register set for both 32-deep halves of a K-step, so every wave ran
//     read - wait - 24 MFMAs - read - wait - 24 MFMAs, all eight waves in the same phase: the matrix pipe idled while the LDS
//     served the 80 reads of a burst.  The same wave tile with two fragment sets and the reads of the next half issued
//     between the MFMAs of the current one keeps the pipe at 16.5 cycles per MFMA (2.0 PFLOP/s in the probe).
//   * the DMA issue (7 per wave and K-step, tile-switch logic with divisions, clamps and 64-bit pointers inlined per piece)
//     sat between the first reads and the first MFMA.  Here a piece is `s_mov m0` + ONE `buffer_load_dwordx4 ... lds` whose
//     per-lane offset never changes (row-in-group x row pitch + swizzled chunk) and whose tile / group / K-step position is
//     a scalar offset; rows past M need no clamp (the buffer descriptor's range check returns zeros); pieces go out one at a
//     time between pairs of MFMAs.
//   * the activation panel of a tile is first touched in HBM by the DMAs that need it one K-step later (a 2-stage ring cannot
//     look further ahead, 3 x 56 KiB do not fit): every K-step of every workgroup sharing the panel waited for HBM.  Here each
//     wave also issues one 4-byte-per-lane DMA per K-step that touches one line of 32 rows of the NEXT tile's panel: the panel
//     is in L2 a whole tile before its first use (an L2 prefetch; the bytes land in a 256-byte scratch strip).
//   * MFMAs run one half-step BEHIND the reads (the last half of a tile is multiplied at the top of the next tile's first
//     step, before that tile's epilogue), so the wait + barrier at the top of a step is covered by MFMAs already queued.
// Same accumulation order per output element as v2 (K ascending in 32-deep chunks): bit-identical results.
#include "gemm_mfma.h"
#include <cstdlib>

namespace mae {

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int BK3 = 64;
// experiment knobs (alternate builds only; see profiles/r03_nt3_kloop_ab.txt):
//   MAE_NT3_SCHED  0: every DMA piece of a step goes out in phase 1; 1: even pieces in phase 1, odd pieces in phase 2;
//                  2: waves 0-3 issue in phase 1, waves 4-7 (their partners on the SIMDs) in phase 2.  A step that runs an
//                  epilogue always issues everything in phase 1, before the stores (the counted waits rely on that order).
//   MAE_NT3_NOPF   no L2 prefetch of the next tile's activation panel
#ifndef MAE_NT3_SCHED
#define MAE_NT3_SCHED 0
#endif
#ifdef MAE_NT3_NOPF
constexpr int NT3_PF = 0;
#else
constexpr int NT3_PF = 1;
#endif

template <int NI, int MI>
struct Geo3 {
  static constexpr int BN = 32 * NI, BM = 64 * MI;
  static constexpr int STAGE = (BM + BN) * BK3 * 2;                   // A rows first, then W rows, 128 B per row
  static constexpr int TAIL = 2 * BN * 4 + 256;                       // bias strips of two tiles + the L2-prefetch scratch strip
  static constexpr int NSTAGE = (3 * STAGE + TAIL <= 160 * 1024) ? 3 : 2;
  static constexpr int AHEAD = NSTAGE - 1;
  static constexpr int BIAS_OFF = NSTAGE * STAGE;
  static constexpr int PF_OFF = BIAS_OFF + 2 * BN * 4;
  static constexpr int LDS = PF_OFF + 256;
  static constexpr int GPW = (BM + BN) / 64;                          // 1 KiB DMA pieces (8 rows) per wave and stage
  static constexpr int NBIAS = BN / 64;
  static_assert((BM + BN) % 64 == 0, "DMA pieces must divide evenly over the 8 waves");
};

__device__ __forceinline__ void unpack8(const bf16x8& v, f32x4& a, f32x4& b) {
  a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
__device__ __forceinline__ void ld8(const float* p, f32x4& a, f32x4& b) { a = load4(p); b = load4(p + 4); }
__device__ __forceinline__ void ld8(const bf16* p, f32x4& a, f32x4& b) { unpack8(*reinterpret_cast<const bf16x8*>(p), a, b); }
template <class V> __device__ __forceinline__ void stream_store(V v, V* p) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st8(float* p, const f32x4& a, const f32x4& b) {
  stream_store(a, reinterpret_cast<f32x4*>(p));
  stream_store(b, reinterpret_cast<f32x4*>(p + 4));
}
__device__ __forceinline__ bf16x8 pk8(const f32x4& a, const f32x4& b) {
  return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
__device__ __forceinline__ void st8(bf16* p, const f32x4& a, const f32x4& b) { stream_store(pk8(a, b), reinterpret_cast<bf16x8*>(p)); }
// the 16 bytes of the lane 8 places away inside its 16-lane row (DPP row_ror:8; lanes l and l ^ 8 swap)
__device__ __forceinline__ bf16x8 row_swap8(const bf16x8& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  u32x4_ x = __builtin_bit_cast(u32x4_, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0x128, 0xf, 0xf, true);
  return __builtin_bit_cast(bf16x8, x);
}

// LDS-DMA pieces from inline asm (the compiler neither counts nor drains them: every wait below is ours).  M0 is written in
// the statement that uses it.  Raw buffer addressing: byte offset = voff (per lane) + soff (scalar), range-checked against
// the descriptor's num_records (out of range -> zeros, no fault).
__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  return i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

__device__ __forceinline__ int64_t xcd_remap3(int64_t bid, int64_t nb) {
  const int64_t q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Counted waits: vector-memory operations of a wave retire in issue order (LDS-DMA, loads and stores share one counter).
// -DMAE_DBG_VMCNT0 drains every wait; tests/test_gpu_kernels.py compares the two builds bit for bit.
template <int N>
__device__ __forceinline__ void wait_vm3() {
#ifdef MAE_DBG_VMCNT0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

template <int I0, int I1, int MI, int NI>
__device__ __forceinline__ void mfma_range(f32x4 (&acc)[MI][NI], const bf16x8 (&a)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
  for (int i = I0; i < I1; ++i) acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i % NI], a[i / NI], acc[i / NI][i % NI], 0, 0, 0);
}

// MFMAs that follow DMA piece q in phase 1: the MI * NI - (MI + NI) MFMAs left after the read-interleaved ones, spread evenly
template <int MI, int NI, int GPW>
struct Spread {
  static constexpr int NR = MI + NI, REST = MI * NI - NR;
  static constexpr int lo(int q) { return NR + q * REST / GPW; }
  static constexpr int hi(int q) { return NR + (q + 1) * REST / GPW; }
};

}  // namespace

template <int MODE, class TO, bool HAS_BIAS, int NI, int MI>
__global__ void __launch_bounds__(512, 2) gemm_nt3_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int64_t M, int N, int K,
                                                          const float* __restrict__ bias, const void* __restrict__ aux, TO* __restrict__ out,
                                                          TO* __restrict__ out2, int tiles_m, int tiles_n) {
  using G_ = Geo3<NI, MI>;
  using SP = Spread<MI, NI, G_::GPW>;
  constexpr int BM = G_::BM, BN = G_::BN, STAGE = G_::STAGE, NSTAGE = G_::NSTAGE, AHEAD = G_::AHEAD, GPW = G_::GPW;
  constexpr int NB = HAS_BIAS ? G_::NBIAS : 0;
  constexpr int NJ = NI / 2, NR = MI + NI;
  constexpr bool TWO = MODE == MAE_EPI_GELU_GRAD;
  constexpr int STORE8 = sizeof(TO) == 2 ? 1 : 2;          // store instructions per 8 outputs
  constexpr int E = MI * NJ * STORE8 * (TWO ? 2 : 1);      // epilogue stores per wave (full tile)
  constexpr int GRP = GPW + NT3_PF;                        // vector-memory operations per issued stage: the pieces + the L2 prefetch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WROWS = 16 * MI;
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x, T = tiles_m * tiles_n;
  const int vb = (int)xcd_remap3(blockIdx.x, G);
  const int ntile = (T - vb + G - 1) / G;
  const int nk = K / BK3;
  const int nsteps = ntile * nk;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint32_t rowbytes = (uint32_t)K * 2u;

  // ---- producer side (all scalar but three per-lane offsets that never change)
  const i32x4 rsA = make_rsrc(A, (uint32_t)((uint64_t)M * rowbytes));
  const i32x4 rsW = make_rsrc(W, (uint32_t)N * rowbytes);
  const i32x4 rsB = make_rsrc(bias, HAS_BIAS ? (uint32_t)N * 4u : 0u);
  const uint32_t r8 = (uint32_t)lane >> 3;
  const uint32_t voff = r8 * rowbytes + ((((uint32_t)lane & 7u) ^ r8) << 4);   // row r8 of an 8-row piece, swizzled source chunk for LDS slot lane & 7
  const uint32_t voff_pf = (uint32_t)(wave * (BM / 8) + (lane % (BM / 8))) * rowbytes;  // one line of each of BM / 8 rows of the next tile's panel
  const uint32_t voff_b = (uint32_t)lane * 4u;
  const int dq = G / tiles_n, dr = G % tiles_n;            // tile id + G in (row, column) form, without a division per tile
  int i_tm = vb / tiles_n, i_tn = vb % tiles_n;            // tile the DMA stream is in
  int n_tm = i_tm + dq, n_tn = i_tn + dr;                  // the tile after it
  if (n_tn >= tiles_n) { n_tn -= tiles_n; ++n_tm; }
  int i_tile = 0, i_k = 0, i_stage = 0;
  uint32_t sA = (uint32_t)i_tm * BM * rowbytes, sW = (uint32_t)i_tn * BN * rowbytes;
  uint32_t sPF = ntile > 1 ? (uint32_t)n_tm * BM * rowbytes : 0xffffff00u;     // no next tile: out of range, fetches nothing
  auto issue_piece = [&](int q) {
    const int g = wave * GPW + q;
    const bool isA = g < BM / 8;
    const uint32_t soff = (isA ? sA + (uint32_t)(g * 8) * rowbytes : sW + (uint32_t)((g - BM / 8) * 8) * rowbytes) + (uint32_t)i_k * 128u;
#ifndef MAE_DBG_NO_DMA
    dma16(isA ? rsA : rsW, lds0 + (uint32_t)(i_stage * STAGE + g * 1024), voff, soff);
#endif
  };
  auto issue_tail = [&]() {   // the tile's bias strip rides with its first K-step; the L2 prefetch; advance the stream
#ifndef MAE_DBG_NO_DMA
    if (HAS_BIAS && i_k == 0) {
#pragma unroll
      for (int i = 0; i < G_::NBIAS; ++i) dma4(rsB, lds0 + (uint32_t)(G_::BIAS_OFF + (i_tile & 1) * (BN * 4) + 256 * i), voff_b, (uint32_t)(i_tn * BN * 4 + 256 * i));
    }
    if (NT3_PF) dma4(rsA, lds0 + (uint32_t)G_::PF_OFF, voff_pf, sPF + (uint32_t)i_k * 128u);
#endif
    i_stage = i_stage == NSTAGE - 1 ? 0 : i_stage + 1;
    if (++i_k == nk) {
      i_k = 0;
      ++i_tile;
      i_tm = n_tm; i_tn = n_tn;
      n_tm += dq; n_tn += dr;
      if (n_tn >= tiles_n) { n_tn -= tiles_n; ++n_tm; }
      sA = (uint32_t)i_tm * BM * rowbytes; sW = (uint32_t)i_tn * BN * rowbytes;
      sPF = i_tile + 1 < ntile ? (uint32_t)n_tm * BM * rowbytes : 0xffffff00u;
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];   // fragment set 0 = the 32-deep half 0 of a K-step, set 1 = half 1
  {
    const bf16 z = (bf16)0.0f;
    const bf16x8 zz = bf16x8{z, z, z, z, z, z, z, z};
#pragma unroll
    for (int i = 0; i < MI; ++i) af1[i] = zz;
#pragma unroll
    for (int j = 0; j < NI; ++j) bf1[j] = zz;
  }

#pragma unroll
  for (int q = 0; q < GPW; ++q) issue_piece(q);
  issue_tail();
  if (AHEAD > 1 && nsteps > 1) {
#pragma unroll
    for (int q = 0; q < GPW; ++q) issue_piece(q);
    issue_tail();
  }

  // ---- consumer side
  const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
  const int gb = (fq & 1) ? 3 + fq : fq;
  const int a_lane = (wm * WROWS + fr) * 128, b_lane = BM * 128 + (wn * (NI * 16) + fr) * 128;
  int c_tm = vb / tiles_n, c_tn = vb % tiles_n;   // tile being multiplied
  int64_t p_m0 = 0;                               // tile whose epilogue is pending
  int p_n0 = 0, p_strip = 0;
  int ct = 0, ck = 0, cs = 0;
  bool prev_full = false;

  auto epilogue = [&](int64_t m0, int n0, int strip) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mi][2 * j][r]), __float_as_uint(acc[mi][2 * j + 1][r]), false, false);
          acc[mi][2 * j][r] = __uint_as_float(sw[0]);
          acc[mi][2 * j + 1][r] = __uint_as_float(sw[1]);
        }
    const float* sbias = reinterpret_cast<const float*>(smem + G_::BIAS_OFF + strip * (BN * 4)) + wn * (NI * 16);
    const int colw = n0 + wn * (NI * 16) + 4 * gb;      // this lane's column inside group 0
#ifndef MAE_DBG_NO_EPI
    if constexpr (sizeof(TO) == 2) {
      // bf16 outputs, whole-line stores (see gemm_nt2_kernel): a lane holds 8 consecutive columns of one row; two neighbouring
      // 32-column groups that form one aligned 128-byte line go out together after the lanes of rows 0-7 and rows 8-15 swapped
      // one group's 16 bytes (DPP row_ror:8)
      const bool lo8 = fr < 8;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int64_t mrow = m0 + wm * WROWS + mi * 16;
        const int64_t m = mrow + fr;
        bf16x8 pa[NJ], pb[TWO ? NJ : 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = 32 * j + 4 * gb;
          f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
          if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
          f32x4 v0 = acc[mi][2 * j] + b0, v1 = acc[mi][2 * j + 1] + b1;
          if (MODE == MAE_EPI_GELU_GRAD || MODE == MAE_EPI_GELU_ACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v0[r] = to_f(from_f<TO>(v0[r]));
              v1[r] = to_f(from_f<TO>(v1[r]));
            }
            f32x4 a0, a1, g0, g1;
            gelu_fast_pair(v0, a0, g0);
            gelu_fast_pair(v1, a1, g1);
            if (MODE == MAE_EPI_GELU_GRAD) { pa[j] = pk8(g0, g1); pb[TWO ? j : 0] = pk8(a0, a1); }
            else pa[j] = pk8(a0, a1);
          } else if (MODE == MAE_EPI_MUL) {
            f32x4 q0, q1;
            const int64_t mc = m < M ? m : M - 1;
            ld8(reinterpret_cast<const TO*>(aux) + mc * N + colw + 32 * j, q0, q1);
            pa[j] = pk8(v0 * q0, v1 * q1);
          } else {
            pa[j] = pk8(v0, v1);
          }
          acc[mi][2 * j] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mi][2 * j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto store_rows = [&](TO* __restrict__ dst, const bf16x8* pk, int ja) {
          const bf16x8 Av = pk[ja], Bv = pk[ja + 1];
          const bf16x8 R = row_swap8(lo8 ? Bv : Av);
          const int col = colw + 32 * (lo8 ? ja : ja + 1);
          const int64_t r1 = mrow + (fr & 7), r2 = r1 + 8;
          if (r1 < M) stream_store(lo8 ? Av : R, reinterpret_cast<bf16x8*>(dst + r1 * N + col));
          if (r2 < M) stream_store(lo8 ? R : Bv, reinterpret_cast<bf16x8*>(dst + r2 * N + col));
        };
        auto store_single = [&](TO* __restrict__ dst, const bf16x8& v, int j) {
          if (m < M) stream_store(v, reinterpret_cast<bf16x8*>(dst + m * N + colw + 32 * j));
        };
        if (NJ == 2) {
          store_rows(out, pa, 0);
          if (TWO) store_rows(out2, pb, 0);
        } else if (wn == 0) {
          store_rows(out, pa, 0); store_single(out, pa[NJ - 1], NJ - 1);
          if (TWO) { store_rows(out2, pb, 0); store_single(out2, pb[TWO ? NJ - 1 : 0], NJ - 1); }
        } else {
          store_single(out, pa[0], 0); store_rows(out, pa, NJ - 2);
          if (TWO) { store_single(out2, pb[0], 0); store_rows(out2, pb, TWO ? NJ - 2 : 0); }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = 32 * j + 4 * gb;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int64_t m = m0 + wm * WROWS + mi * 16 + fr;
          if (m < M) st8(reinterpret_cast<float*>(out) + m * N + colw + 32 * j, acc[mi][2 * j] + b0, acc[mi][2 * j + 1] + b1);
          acc[mi][2 * j] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mi][2 * j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (acc[mi][ni][0] == 1.2345e30f) out[0] = (TO)0.f;
        acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#endif
    prev_full = m0 + BM <= M;
  };

#define NT3_READ(AF, BF, SW)                                                                                       \
  {                                                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) AF[mi] = *reinterpret_cast<const bf16x8*>(stg + a_lane + mi * 2048 + (SW)); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) BF[ni] = *reinterpret_cast<const bf16x8*>(stg + b_lane + ni * 2048 + (SW)); \
  }
#define NT3_IL(n)                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
  }

  for (int step = 0; step < nsteps; ++step) {
    // ---- wait for this step's stage.  Outstanding afterwards may be: the next stage (AHEAD == 2; + the bias pieces when it opens
    // a tile) and the stores of an epilogue that ran after this stage had been issued (one or two steps ago; only counted when
    // that tile was full, so that every store was issued)
    {
      const bool last = step + 1 >= nsteps;
      const bool epi1 = ck == 1 && ct > 0 && prev_full;                      // the epilogue ran in the previous iteration
      const bool epi2 = AHEAD == 2 && ck == 2 && ct > 0 && prev_full;        // ... two iterations ago
      if (AHEAD == 2) {
        const bool opens = ck == nk - 1, ep = epi1 || epi2;   // the stage after this one opens a tile (it carried the bias pieces)
        if (last) wait_vm3<0>();
        else if (opens && ep) wait_vm3<GRP + NB + E>();
        else if (opens) wait_vm3<GRP + NB>();
        else if (ep) wait_vm3<GRP + E>();
        else wait_vm3<GRP>();
      } else {
        if (epi1) wait_vm3<E>();
        else wait_vm3<0>();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage refilled below have landed in its registers
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    const char* stg = smem + cs * STAGE;
    const bool do_issue = step + AHEAD < nsteps;

    // ---- phase 1: reads of half 0 beside the MFMAs of the previous step's half 1, then the DMA pieces between the other MFMAs
    NT3_READ(af0, bf0, sw0)
    mfma_range<0, NR>(acc, af1, bf1);
    NT3_IL(NR)
    __builtin_amdgcn_sched_barrier(0);
    const bool epi_iter = ck == 0 && step > 0;
    const bool all_p1 = MAE_NT3_SCHED == 0 || epi_iter || (MAE_NT3_SCHED == 2 && wave < 4);
    const bool none_p1 = MAE_NT3_SCHED == 2 && !epi_iter && wave >= 4;
#pragma unroll
    for (int q = 0; q < GPW; ++q) {
      if (do_issue && (all_p1 || (!none_p1 && (q & 1) == 0))) issue_piece(q);
#pragma unroll
      for (int i = SP::lo(q); i < SP::hi(q); ++i) acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[i % NI], af1[i / NI], acc[i / NI][i % NI], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_issue && all_p1) issue_tail();
    __builtin_amdgcn_sched_barrier(0);

    // ---- a tile's first step: the previous tile is complete now
    if (ck == 0 && step > 0) {
      epilogue(p_m0, p_n0, p_strip);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- phase 2: reads of half 1 beside the MFMAs of half 0
    NT3_READ(af1, bf1, sw1)
    mfma_range<0, NR>(acc, af0, bf0);
    NT3_IL(NR)
    if (MAE_NT3_SCHED == 0) {
      mfma_range<NR, MI * NI>(acc, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GPW; ++q) {
        if (do_issue && !all_p1 && (none_p1 || (q & 1) == 1)) issue_piece(q);
#pragma unroll
        for (int i = SP::lo(q); i < SP::hi(q); ++i) acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[i % NI], af0[i / NI], acc[i / NI][i % NI], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (do_issue && !all_p1) issue_tail();
      __builtin_amdgcn_sched_barrier(0);
    }

    cs = cs == NSTAGE - 1 ? 0 : cs + 1;
    if (++ck == nk) {
      ck = 0;
      p_m0 = (int64_t)c_tm * BM; p_n0 = c_tn * BN; p_strip = ct & 1;
      ++ct;
      c_tm += dq; c_tn += dr;
      if (c_tn >= tiles_n) { c_tn -= tiles_n; ++c_tm; }
    }
  }
  mfma_range<0, MI * NI>(acc, af1, bf1);
  epilogue(p_m0, p_n0, p_strip);
#undef NT3_READ
#undef NT3_IL
}

template <int MODE, class TO, int NI, int MI>
static int launch_nt3(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  using G_ = Geo3<NI, MI>;
  const int64_t T = cdiv(M, G_::BM) * (N / G_::BN);
  MAE_REQUIRE(T < (1ll << 30), "gemm: too many tiles");
  const int tiles_m = (int)cdiv(M, G_::BM), tiles_n = N / G_::BN;
  const int grid = (int)std::min<int64_t>(T, num_cus());  // one persistent workgroup per CU
  if (e.bias) {
    auto kern = gemm_nt3_kernel<MODE, TO, true, NI, MI>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n);
  } else {
    auto kern = gemm_nt3_kernel<MODE, TO, false, NI, MI>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n);
  }
  MAE_LAUNCH_CHECK();
  return 0;
}

// rounds of tiles on the CUs x rows per tile = time proxy; the 192-row tile must win by a margin (15 % more staged bytes per flop)
static bool prefer_bm192_3(int64_t M, int N) {
  static const int force = [] { const char* v = getenv("MAE_NT_BM"); return v ? atoi(v) : 0; }();
  if (force == 192) return true;
  if (force == 256) return false;
  const int64_t t256 = cdiv(M, 256) * (N / 192), t192 = cdiv(M, 192) * (N / 192);
  const int64_t c256 = cdiv(t256, num_cus()) * 256, c192 = cdiv(t192, num_cus()) * 192;
  return c192 * 100 < c256 * 95;
}

template <int MODE, class TO>
static int launch_nt3_ni(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (N % 192 == 0) {
    if (prefer_bm192_3(M, N)) return launch_nt3<MODE, TO, 6, 3>(A, W, M, N, K, e, s);
    return launch_nt3<MODE, TO, 6, 4>(A, W, M, N, K, e, s);
  }
  return launch_nt3<MODE, TO, 4, 4>(A, W, M, N, K, e, s);
}

int mfma_linear_fwd_v3(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (K % 64 != 0 || K < 192 || (N % 128 != 0 && N % 192 != 0) || M < 1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)e.out | (uintptr_t)e.out2 | (uintptr_t)e.bias | (uintptr_t)e.aux) & 15) != 0) return MFMA_UNSUPPORTED;
  // 32-bit byte offsets inside the buffer descriptors (a tile may start up to 255 rows before the end and reach 256 rows past it)
  if ((uint64_t)(M + 512) * (uint64_t)K * 2u >= (1ull << 32) || (uint64_t)N * (uint64_t)K * 2u >= (1ull << 32)) return MFMA_UNSUPPORTED;
  const bool f32out = e.out_dt == MAE_F32;
  switch (e.mode) {
    case MAE_EPI_NONE: return f32out ? launch_nt3_ni<MAE_EPI_NONE, float>(A, W, M, N, K, e, s) : launch_nt3_ni<MAE_EPI_NONE, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_GRAD: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_GELU_GRAD, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_ACT: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_GELU_ACT, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_MUL: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_MUL, bf16>(A, W, M, N, K, e, s);
    default: return MFMA_UNSUPPORTED;
  }
}

}  // namespace mae
