// Persistent bf16 MFMA NT GEMM for gfx950 with an LDS-DMA ring (v2 of gemm_nt_kernel).
//
//   out[M,N] = A[M,K] * W[N,K]^T (+bias) with the NONE / GELU epilogues, M = batch*tokens (huge), K = 192..1536.
//
// Why: with K = 384 a 128x128 tile lives for only 6 K-steps, so a one-tile register prefetch leaves every CU waiting
// on memory latency (v1 measured ~5 TB/s of L2->CU traffic, 0.3-0.7 PF).  Here one 512-thread workgroup per CU walks a
// list of 256x128 tiles and keeps a 3-stage ring (3 x 48 KiB of LDS) filled by `global_load_lds_dwordx4` two K-steps
// ahead, ACROSS tile boundaries: 96 KiB in flight per CU, no pipeline drain between tiles, half the operand traffic per
// flop of the 128x128 tile.
//   * LDS image is lane-linear per DMA (8 rows x 128 B per wave instruction); the bank-conflict swizzle is applied
//     on the per-lane SOURCE address and again on the fragment read (same involution, chunk ^ (row & 7)).
//   * one raw s_barrier per K-step; waits are counted (`s_waitcnt vmcnt(N)`, never 0 in steady state): N = the vector
//     memory operations issued after the stage being waited for = the next stage's DMAs (+ the previous tile's
//     epilogue stores for the first two K-steps of a tile, when that tile was full so all its stores were issued).
//   * the bias slice of the tile rides along as one 4-byte DMA per K-step into a double-buffered LDS strip, so the
//     epilogue needs no ordinary global load (hipcc drains the DMA queue with vmcnt(0) before using one).
//   * blocks are XCD-remapped so the 32 CUs of an XCD work on adjacent tiles (shared A panels / W tiles in L2).
#include "gemm_mfma.h"
#include <cstdlib>

namespace mae {

namespace {

constexpr int BK2 = 64;

// NI = 16-column MFMA tiles per wave along N (2 waves along N): BN = 32 NI = 128 or 192
// MI = 16-row MFMA tiles per wave along M (4 waves along M):    BM = 64 MI = 256 or 192.  The 192-row tile exists for
//      load balance: with N = 384 a 72 000-row GEMM has 564 tiles of 256x192 (2.2 per CU -> 3 rounds, 73 % busy) but
//      750 tiles of 192x192 (2.93 per CU -> 3 shorter rounds, 98 % busy).
template <int NI, int MI>
struct Geo {
  static constexpr int BN = 32 * NI, BM = 64 * MI;
  static constexpr int STAGE = (BM + BN) * BK2 * 2;               // A rows first, then W rows, 128 B per row
  static constexpr int NSTAGE = (3 * STAGE + 2 * BN * 4 <= 160 * 1024) ? 3 : 2;
  static constexpr int AHEAD = NSTAGE - 1;                         // K-steps of DMA in flight during a compute phase
  static constexpr int BIAS_OFF = NSTAGE * STAGE;                  // [2 tiles][BN] floats
  static constexpr int LDS = BIAS_OFF + 2 * BN * 4;
  static constexpr int GPW = (BM + BN) / 8 / 8;                    // 1 KiB DMA groups (8 rows) per wave and stage: 5..7
  static constexpr int NBIAS = BN / 64;                            // 256-byte bias DMAs per wave at the first K-step of a tile
  static_assert((BM + BN) % 64 == 0, "DMA groups must divide evenly over the 8 waves");
};

__device__ __forceinline__ void ld8(const float* p, f32x4& a, f32x4& b) { a = load4(p); b = load4(p + 4); }
__device__ __forceinline__ void unpack8(const bf16x8& v, f32x4& a, f32x4& b) {
  a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
__device__ __forceinline__ void ld8(const bf16* p, f32x4& a, f32x4& b) { unpack8(*reinterpret_cast<const bf16x8*>(p), a, b); }
// Outputs are written once and read by a LATER kernel: non-temporal stores stream them to HBM instead of parking the
// lines in L2 until eviction, which both frees L2 for the A / W re-reads and spreads the write traffic over the tile loop
// (measured: fc1 + GELU forward 220 -> 170 us, decoder fc1 332 -> 242 us).
// (a per-launch runtime choice between the two store forms was tried: the uniform branch per store cost 1.1 ms per step)
#ifdef MAE_DBG_EPI_PLAINSTORE
template <class V> __device__ __forceinline__ void stream_store(V v, V* p) { *p = v; }
#else
template <class V> __device__ __forceinline__ void stream_store(V v, V* p) { __builtin_nontemporal_store(v, p); }
#endif
__device__ __forceinline__ void st8(float* p, const f32x4& a, const f32x4& b) {
  stream_store(a, reinterpret_cast<f32x4*>(p));
  stream_store(b, reinterpret_cast<f32x4*>(p + 4));
}
__device__ __forceinline__ void st8(bf16* p, const f32x4& a, const f32x4& b) {
  stream_store(bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]}, reinterpret_cast<bf16x8*>(p));
}
__device__ __forceinline__ bf16x8 pk8(const f32x4& a, const f32x4& b) {
  return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
// the 16 bytes of the lane 8 places away inside its 16-lane row (DPP row_ror:8; lanes l and l ^ 8 swap)
__device__ __forceinline__ bf16x8 row_swap8(const bf16x8& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  u32x4_ x = __builtin_bit_cast(u32x4_, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0x128, 0xf, 0xf, true);
  return __builtin_bit_cast(bf16x8, x);
}
__device__ __forceinline__ void glds16(const bf16* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
}

__device__ __forceinline__ int64_t xcd_remap2(int64_t bid, int64_t nb) {
  const int64_t q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// The counted waits below assume (i) vector-memory operations of a wave retire in issue order (LDS-DMA loads and the
// epilogue's stores share one counter, MI355X_MICROARCH: "loads, stores, atomics and LDS-DMA count together, in issue
// order") and (ii) E equals the number of store instructions the compiler emits per wave and tile.  -DMAE_DBG_VMCNT0
// builds the same kernel with every wait drained to zero: tests/test_gpu_kernels.py compares the two builds bit for bit
// (tools/build_dbg_lib.sh vmcnt0), so a miscounted wait shows up as a difference instead of a rare wrong tile.
template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef MAE_DBG_VMCNT0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

}  // namespace

// MODE: any MAE_EPI_*.  RESID / DGELU read their (M,N) side input with ordinary loads in the epilogue.
template <int MODE, class TO, bool HAS_BIAS, int NI, int MI>
__global__ void __launch_bounds__(512, 2) gemm_nt2_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int64_t M, int N,
                                                          int K, const float* __restrict__ bias, const void* __restrict__ aux,
                                                          TO* __restrict__ out, TO* __restrict__ out2, int tiles_m, int tiles_n) {
  using G_ = Geo<NI, MI>;
  constexpr int BM2 = G_::BM;
  constexpr int BN = G_::BN, STAGE = G_::STAGE, NSTAGE = G_::NSTAGE, AHEAD = G_::AHEAD, GPW = G_::GPW;
  constexpr int NB = HAS_BIAS ? G_::NBIAS : 0;
  constexpr int STORE8 = sizeof(TO) == 2 ? 1 : 2;                                // store instructions per 8 outputs
  constexpr int E = MI * (NI / 2) * STORE8 * ((MODE == MAE_EPI_GELU || MODE == MAE_EPI_GELU_GRAD) ? 2 : 1);      // epilogue stores per wave (full tile)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int WROWS = 16 * MI;  // rows of the tile owned by one wave
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x, T = tiles_m * tiles_n;
  const int vb = (int)xcd_remap2(blockIdx.x, G);
  const int ntile = (T - vb + G - 1) / G;
  const int nk = K / BK2;
  const int nsteps = ntile * nk;

  // ---- producer side: this wave's DMA groups of a stage (group g < 32: A rows 8g.., else W rows 8(g-32)..)
  const int r8 = lane >> 3;
  const int src_chunk = ((lane & 7) ^ r8) * 8;  // swizzled source chunk (elements) for LDS slot lane&7 of row r8
  const bf16 *p0, *p1, *p2, *p3, *p4, *p5 = nullptr, *p6 = nullptr;
  const float* pbias = bias;
  int is_tile = 0, is_k = 0, is_stage = 0;
  auto src_ptr = [&](int q, int64_t m0, int n0) -> const bf16* {
    const int g = wave * GPW + q;
    if (g < BM2 / 8) {
      int64_t row = m0 + g * 8 + r8;
      row = row < M ? row : M - 1;  // clamp: rows past M are loaded (valid memory) and never stored
      return A + row * K + src_chunk;
    }
    return W + (int64_t)(n0 + (g - BM2 / 8) * 8 + r8) * K + src_chunk;
  };
  auto set_tile = [&](int ord) {
    const int t = vb + ord * G;
    const int64_t m0 = (int64_t)(t / tiles_n) * BM2;
    const int n0 = (t % tiles_n) * BN;
    p0 = src_ptr(0, m0, n0); p1 = src_ptr(1, m0, n0); p2 = src_ptr(2, m0, n0);
    p3 = src_ptr(3, m0, n0); p4 = src_ptr(4, m0, n0);
    if (GPW > 5) p5 = src_ptr(5, m0, n0);
    if (GPW > 6) p6 = src_ptr(6, m0, n0);
    if (HAS_BIAS) pbias = bias + n0 + lane;
  };
  auto issue = [&]() {
#ifdef MAE_DBG_NO_DMA
    return;
#endif
    char* dst = smem + is_stage * STAGE + wave * (GPW * 1024);
    const int ko = is_k * BK2;
    glds16(p0 + ko, dst);
    glds16(p1 + ko, dst + 1024);
    glds16(p2 + ko, dst + 2048);
    glds16(p3 + ko, dst + 3072);
    glds16(p4 + ko, dst + 4096);
    if (GPW > 5) glds16(p5 + ko, dst + 5120);
    if (GPW > 6) glds16(p6 + ko, dst + 6144);
    if (HAS_BIAS && is_k == 0) {  // the tile's bias strip rides with its first K-step (every wave writes the same bytes)
      char* bdst = smem + G_::BIAS_OFF + (is_tile & 1) * (BN * 4);
#pragma unroll
      for (int i = 0; i < G_::NBIAS; ++i) glds4(pbias + 64 * i, bdst + 256 * i);
    }
    is_stage = is_stage == NSTAGE - 1 ? 0 : is_stage + 1;
    if (++is_k == nk) {
      is_k = 0;
      if (++is_tile < ntile) set_tile(is_tile);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  set_tile(0);
  issue();
  if (AHEAD > 1 && nsteps > 1) issue();

  int ct = 0, ck = 0, cs = 0;
  bool prev_full = false;
  // MUL / DGELU with a bf16 side input: the tile's (M,N) side input is fetched at the top of the tile's LAST K-step, so its
  // HBM latency hides under that step's MFMAs instead of stalling every (j, mi) unit of the epilogue
  constexpr bool PREF = (MODE == MAE_EPI_MUL || MODE == MAE_EPI_DGELU) && sizeof(TO) == 2;
  bf16x8 qa[PREF ? NI / 2 : 1][PREF ? MI : 1];
  const int gb = (fq & 1) ? 3 + fq : fq;
  for (int step = 0; step < nsteps; ++step) {
    // ---- wait for this step's DMAs: allowed outstanding = vector-memory ops issued after them
    //      = the next AHEAD-1 steps' DMAs (+ the bias DMAs when that step opens a tile)
    //        (+ the previous tile's epilogue stores, when that tile was full so every store was issued)
    const bool after_epi = ct > 0 && prev_full && ck < AHEAD;
#ifdef MAE_DBG_NO_DMA
    if (false) {
    } else
#endif
    if (AHEAD == 2) {
      if (step + 1 >= nsteps) wait_vm<0>();
      else if (ck == nk - 1) wait_vm<GPW + NB>();   // the next step is a tile's first: it carried the bias DMAs
      else if (after_epi) wait_vm<GPW + E>();
      else wait_vm<GPW>();
    } else {
      if (after_epi) wait_vm<E>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (PREF && ck == nk - 1) {
      const int t = vb + ct * G;
      const int64_t m0 = (int64_t)(t / tiles_n) * BM2;
      const int n0 = (t % tiles_n) * BN;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int64_t m = m0 + wm * WROWS + mi * 16 + fr;
        m = m < M ? m : M - 1;
        const bf16* q = reinterpret_cast<const bf16*>(aux) + m * N + n0 + wn * (NI * 16) + 4 * gb;
#pragma unroll
        for (int j = 0; j < NI / 2; ++j) qa[PREF ? j : 0][PREF ? mi : 0] = *reinterpret_cast<const bf16x8*>(q + 32 * j);  // (a non-temporal load measured 0.1 ms per step slower)
      }
    }
    const char* a_base = smem + cs * STAGE + (wm * WROWS + fr) * 128;
    const char* b_base = smem + cs * STAGE + BM2 * 128 + (wn * (NI * 16) + fr) * 128;
    // fragments of both 32-deep halves are read up front, the DMA refill is issued between the two read bursts
#ifndef MAE_DBG_NO_MFMA
    bf16x8 af[2][MI], bfr[2][NI];
    const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 16 * 128 + sw0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[0][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 16 * 128 + sw0);
    if (step + AHEAD < nsteps) issue();  // refills the stage every wave finished reading in the previous iteration
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[1][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 16 * 128 + sw1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[1][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 16 * 128 + sw1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0);
#else
    if (step + AHEAD < nsteps) issue();
#endif
    cs = cs == NSTAGE - 1 ? 0 : cs + 1;

    if (++ck == nk) {
      // ---- epilogue of tile `ct`: regroup to 8 consecutive columns per lane (see gemm_nt_kernel), store
      const int t = vb + ct * G;
      const int64_t m0 = (int64_t)(t / tiles_n) * BM2;
      const int n0 = (t % tiles_n) * BN;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < NI / 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mi][2 * j][r]), __float_as_uint(acc[mi][2 * j + 1][r]), false, false);
            acc[mi][2 * j][r] = __uint_as_float(sw[0]);
            acc[mi][2 * j + 1][r] = __uint_as_float(sw[1]);
          }
      const float* sbias = reinterpret_cast<const float*>(smem + G_::BIAS_OFF + (ct & 1) * (BN * 4)) + wn * (NI * 16);
#if !defined(MAE_DBG_NT_HALF_LINES) && !defined(MAE_DBG_NO_EPI) && !defined(MAE_DBG_EPI_NOGELU) && \
    !defined(MAE_DBG_EPI_NOSTORE) && !defined(MAE_DBG_EPI_ONESTORE)
      if constexpr (sizeof(TO) == 2) {
        // bf16 outputs, WHOLE-LINE stores.  A lane holds 8 consecutive columns (16 B) of one row and the 4 lanes of a row 64 B:
        // written unit by unit, every store instruction covers 16 rows x half a line, and the other half of each line arrives
        // with a later instruction (measured: whole-line stores are worth 13 us of 180 on fc1 + GELU, 28 us of 162 without the
        // GELU arithmetic, r02_nt2_epilogue_ablation.txt).  Here two neighbouring 32-column groups whose 128 bytes form one
        // aligned line are stored together: the lanes of rows 0-7 and of rows 8-15 swap one group's 16 bytes (DPP row_ror:8),
        // so that one instruction writes rows 0-7 x 128 B and the next rows 8-15 x 128 B.  The wave's 96 (NI = 6) columns are
        // a line and a half: the wave with the even column offset pairs groups (0, 1), the other one (1, 2); the group left
        // over shares its line with the neighbouring wave and goes out as before.  NI = 4: one pair, nothing left over.
        constexpr int NJ = NI / 2;
        constexpr bool TWO = MODE == MAE_EPI_GELU || MODE == MAE_EPI_GELU_GRAD;
        const bool lo8 = fr < 8;
        const int colw = n0 + wn * (NI * 16) + 4 * gb;      // this lane's column inside group 0
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int64_t mrow = m0 + wm * WROWS + mi * 16;   // first row of the 16-row group
          const int64_t m = mrow + fr;
          bf16x8 pa[NJ], pb[TWO ? NJ : 1];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int nl = 32 * j + 4 * gb;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
            f32x4 v0 = acc[mi][2 * j] + b0, v1 = acc[mi][2 * j + 1] + b1;
            if (MODE == MAE_EPI_GELU || MODE == MAE_EPI_GELU_GRAD || MODE == MAE_EPI_GELU_ACT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[r] = to_f(from_f<TO>(v0[r]));
                v1[r] = to_f(from_f<TO>(v1[r]));
              }
              f32x4 a0, a1, g0, g1;
              gelu_fast_pair(v0, a0, g0);
              gelu_fast_pair(v1, a1, g1);
              if (MODE == MAE_EPI_GELU) { pa[j] = pk8(v0, v1); pb[TWO ? j : 0] = pk8(a0, a1); }
              else if (MODE == MAE_EPI_GELU_GRAD) { pa[j] = pk8(g0, g1); pb[TWO ? j : 0] = pk8(a0, a1); }
              else pa[j] = pk8(a0, a1);
            } else if (MODE == MAE_EPI_DGELU || MODE == MAE_EPI_MUL) {
              f32x4 q0, q1;
              if (PREF) unpack8(qa[PREF ? j : 0][PREF ? mi : 0], q0, q1);
              else {
                const int64_t mc = m < M ? m : M - 1;
                ld8(reinterpret_cast<const TO*>(aux) + mc * N + colw + 32 * j, q0, q1);
              }
              if (MODE == MAE_EPI_DGELU) { f32x4 a_, g_; gelu_fast_pair(q0, a_, g_); v0 *= g_; gelu_fast_pair(q1, a_, g_); v1 *= g_; }
              else { v0 *= q0; v1 *= q1; }
              pa[j] = pk8(v0, v1);
            } else {
              pa[j] = pk8(v0, v1);
            }
            acc[mi][2 * j] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[mi][2 * j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          // ja = first group of the aligned pair (wave-uniform; NI = 6: 0 for the wave at column offset 0, 1 for the other)
          auto store_rows = [&](TO* __restrict__ dst, const bf16x8* pk, int ja) {
            const bf16x8 A = pk[ja], Bv = pk[ja + 1];
            const bf16x8 R = row_swap8(lo8 ? Bv : A);                      // rows 0-7 hand over group ja + 1, rows 8-15 group ja
            const int col = colw + 32 * (lo8 ? ja : ja + 1);
            const int64_t r1 = mrow + (fr & 7), r2 = r1 + 8;
            if (r1 < M) stream_store(lo8 ? A : R, reinterpret_cast<bf16x8*>(dst + r1 * N + col));   // rows 0-7: 128 B each
            if (r2 < M) stream_store(lo8 ? R : Bv, reinterpret_cast<bf16x8*>(dst + r2 * N + col));  // rows 8-15
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
      } else
#endif
#pragma unroll
      for (int j = 0; j < NI / 2; ++j) {
        const int nl = 32 * j + 4 * gb;  // column inside the wave's NI*16
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int64_t m = m0 + wm * WROWS + mi * 16 + fr;
#ifdef MAE_DBG_NO_EPI
          if (m < M && acc[mi][2 * j][0] == 1.2345e30f) {
#else
          if (m < M) {
#endif
            const int64_t o = m * N + n0 + wn * (NI * 16) + nl;
            f32x4 v0 = acc[mi][2 * j] + b0, v1 = acc[mi][2 * j + 1] + b1;
            if (MODE == MAE_EPI_GELU) {
              f32x4 a0, a1;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[r] = to_f(from_f<TO>(v0[r]));
                v1[r] = to_f(from_f<TO>(v1[r]));
              }
              { f32x4 g_; gelu_fast_pair(v0, a0, g_); gelu_fast_pair(v1, a1, g_); }
              st8(out + o, v0, v1);
              st8(out2 + o, a0, a1);
            } else if (MODE == MAE_EPI_RESID) {
              v0 += load4(reinterpret_cast<const float*>(aux) + o);
              v1 += load4(reinterpret_cast<const float*>(aux) + o + 4);
              st8(out + o, v0, v1);
            } else if (MODE == MAE_EPI_DGELU) {
              f32x4 q0, q1;
              if (PREF) unpack8(qa[PREF ? j : 0][PREF ? mi : 0], q0, q1);
              else ld8(reinterpret_cast<const TO*>(aux) + o, q0, q1);
              { f32x4 a_, g_; gelu_fast_pair(q0, a_, g_); v0 *= g_; gelu_fast_pair(q1, a_, g_); v1 *= g_; }
              st8(out + o, v0, v1);
            } else if (MODE == MAE_EPI_GELU_GRAD) {
              f32x4 a0, a1, g0, g1;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[r] = to_f(from_f<TO>(v0[r]));
                v1[r] = to_f(from_f<TO>(v1[r]));
              }
#if defined(MAE_DBG_EPI_NOGELU)   // epilogue ablation builds (tools/run_epi_ablation.sh): timing probes, wrong values
              a0 = v0; a1 = v1; g0 = v0 * 0.5f; g1 = v1 * 0.5f;
#else
              gelu_fast_pair(v0, a0, g0);
              gelu_fast_pair(v1, a1, g1);
#endif
#if defined(MAE_DBG_EPI_NOSTORE)
              asm volatile("" ::"v"(g0), "v"(g1), "v"(a0), "v"(a1));
#elif defined(MAE_DBG_EPI_ONESTORE)
              asm volatile("" ::"v"(g0), "v"(g1));
              st8(out2 + o, a0, a1);
#else
              st8(out + o, g0, g1);
              st8(out2 + o, a0, a1);
#endif
            } else if (MODE == MAE_EPI_GELU_ACT) {
              f32x4 a0, a1, g_;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[r] = to_f(from_f<TO>(v0[r]));
                v1[r] = to_f(from_f<TO>(v1[r]));
              }
              gelu_fast_pair(v0, a0, g_);
              gelu_fast_pair(v1, a1, g_);
              st8(out + o, a0, a1);
            } else if (MODE == MAE_EPI_MUL) {
              f32x4 q0, q1;
              if (PREF) unpack8(qa[PREF ? j : 0][PREF ? mi : 0], q0, q1);
              else ld8(reinterpret_cast<const TO*>(aux) + o, q0, q1);
              st8(out + o, v0 * q0, v1 * q1);
            } else {
              st8(out + o, v0, v1);
            }
          }
          acc[mi][2 * j] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mi][2 * j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      prev_full = m0 + BM2 <= M;
      ck = 0;
      ++ct;
    }
  }
}

template <int MODE, class TO, int NI, int MI>
static int launch_nt2(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  using G_ = Geo<NI, MI>;
  const int64_t T = cdiv(M, G_::BM) * (N / G_::BN);
  MAE_REQUIRE(T < (1ll << 30), "gemm: too many tiles");
  const int tiles_m = (int)cdiv(M, G_::BM), tiles_n = N / G_::BN;
  const int grid = (int)std::min<int64_t>(T, num_cus());  // one persistent workgroup per CU
  if (e.bias) {
    auto kern = gemm_nt2_kernel<MODE, TO, true, NI, MI>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n);
  } else {
    auto kern = gemm_nt2_kernel<MODE, TO, false, NI, MI>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n);
  }
  MAE_LAUNCH_CHECK();
  return 0;
}

// rounds of tiles on the 256 CUs x rows per tile = time proxy; the 192-row tile must win by a margin because it stages
// 15 % more operand bytes per flop
static bool prefer_bm192(int64_t M, int N) {
  static const int force = [] { const char* v = getenv("MAE_NT_BM"); return v ? atoi(v) : 0; }();
  if (force == 192) return true;
  if (force == 256) return false;
  const int64_t t256 = cdiv(M, 256) * (N / 192), t192 = cdiv(M, 192) * (N / 192);
  const int64_t c256 = cdiv(t256, num_cus()) * 256, c192 = cdiv(t192, num_cus()) * 192;
  return c192 * 100 < c256 * 95;  // measured: 192-row tiles win whenever they save a round (decoder fc2 130 vs 143 us, pred head 69 vs 79 us)
}

template <int MODE, class TO>
static int launch_nt2_ni(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  static const bool force_ni4 = [] { const char* v = getenv("MAE_NT_NI"); return v && atoi(v) == 4; }();
  if (N % 192 == 0 && !(force_ni4 && N % 128 == 0)) {
    if (prefer_bm192(M, N)) return launch_nt2<MODE, TO, 6, 3>(A, W, M, N, K, e, s);
    return launch_nt2<MODE, TO, 6, 4>(A, W, M, N, K, e, s);
  }
  return launch_nt2<MODE, TO, 4, 4>(A, W, M, N, K, e, s);
}

int mfma_linear_fwd_v2(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (K % 64 != 0 || K < 192 || (N % 128 != 0 && N % 192 != 0) || M < 1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)e.out | (uintptr_t)e.out2 | (uintptr_t)e.bias | (uintptr_t)e.aux) & 15) != 0) return MFMA_UNSUPPORTED;
  const bool f32out = e.out_dt == MAE_F32;
  switch (e.mode) {
    case MAE_EPI_NONE: return f32out ? launch_nt2_ni<MAE_EPI_NONE, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_NONE, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU: return f32out ? launch_nt2_ni<MAE_EPI_GELU, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_GELU, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_RESID: return f32out ? launch_nt2_ni<MAE_EPI_RESID, float>(A, W, M, N, K, e, s) : MFMA_UNSUPPORTED;
    case MAE_EPI_DGELU: return f32out ? launch_nt2_ni<MAE_EPI_DGELU, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_DGELU, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_GRAD: return f32out ? launch_nt2_ni<MAE_EPI_GELU_GRAD, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_GELU_GRAD, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_MUL: return f32out ? launch_nt2_ni<MAE_EPI_MUL, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_MUL, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_ACT: return f32out ? launch_nt2_ni<MAE_EPI_GELU_ACT, float>(A, W, M, N, K, e, s) : launch_nt2_ni<MAE_EPI_GELU_ACT, bf16>(A, W, M, N, K, e, s);
    default: return MFMA_UNSUPPORTED;
  }
}

}  // namespace mae
