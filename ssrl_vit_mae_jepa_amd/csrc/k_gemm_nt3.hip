// Persistent bf16 MFMA NT GEMM for gfx950, v3 (round 3): the tiles, fragment layout and epilogues of gemm_nt2_kernel with a new K-loop.
//
//   out[M,N] = A[M,K] * W[N,K]^T (+bias) with the NONE / GELU_GRAD / GELU_ACT / MUL epilogues the engine's bf16 path uses.
//
// What round 2's kernel lost, read off its ISA and off tools/micro/{rates,dma_mfma}_probe.hip (profiles/r03_nt3_kloop.txt):
//   * hipcc reused ONE fragment register set for both 32-deep halves of a K-step, so every wave ran
//     read - wait - 24 MFMAs - read - wait - 24 MFMAs, all eight waves in the same phase.  Here: two fragment sets, the reads of
//     the next half issued between the MFMAs of the current one, MFMAs one half-step BEHIND the reads (the last half of a tile
//     is multiplied at the top of the next tile's first step, before that tile's epilogue).  Rebuilt in the probe, this K-step
//     with its 7 DMA pieces per wave, counted wait and barrier runs at 760 ns = 2.1 PFLOP/s when every piece hits L2.
//   * it does not when the activation rows are first touched in HBM one step before they are needed: 1300 ns.  A tile's row
//     panel is new to the chip, the tiles_n workgroups that share it (same XCD) ask for the same slice at the same time, and a
//     two-stage ring has one step of lookahead: every step of every sharer waits for HBM (probe: 1300 -> 960 ns with):
//       - SPLIT RINGS: weight rows always hit L2 (one step of lookahead is enough: 2 stages), so the LDS the third stage of a
//         uniform ring would need goes to the activation rows alone: 3 x 32 KiB (256-row tiles) or 4 x 24 KiB (192-row tiles),
//         two / three steps of lookahead.  A wave issues its weight pieces first and its activation pieces after them, so
//         that the in-order vmcnt can wait for the young weight pieces while the younger activation pieces stay in flight.
//   * a DMA piece is `s_mov m0` + ONE `buffer_load_dwordx4 ... lds` whose per-lane offset never changes (row-in-group x row pitch
//     + swizzled chunk) and whose tile / group / K-step position is a scalar offset; rows past M need no clamp (the buffer
//     descriptor's range check returns zeros); pieces go out one at a time between pairs of MFMAs, all of them in the first
//     half of a step (before the epilogue's stores, which the counted waits then allow to stay in flight).
//   * the scalar bookkeeping of a step (stream positions, stage pointers, which wait applies) used to sit between the barrier and
//     the first MFMA, on every wave's critical path: now a tile's K-steps are an inner loop whose waits are fixed, the tile switch
//     happens in a peeled first step, and the position updates sit in the shadow of phase 2's MFMAs.
//   Tried and dropped (same file): a K order rotated by the tile column so that the sharers of a row panel ask for different
//   slices (880 ns in the probe, nothing in the kernel, and the summation order would depend on the column), an L2 prefetch of the NEXT tile's panel by 4-byte-per-lane DMAs (a whole tile ahead it thrashes
//   L2 on the narrow-N shapes and costs as many tag lookups as half the pieces: 5-10 % slower), pieces spread over both halves of
//   a step or issued by the two waves of a SIMD in different halves (2-5 % slower).
#include "gemm_mfma.h"
#include <cstdlib>
#include <cstring>

namespace mae {

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int BK3 = 64;
// WM = waves along M (2 waves along N).  WM = 4: one 512-thread workgroup per CU (256- or 192-row tiles, all 160 KiB of LDS).
// WM = 2: 256-thread workgroups, two per CU (128-row tiles, 80 KiB each): two independent wave groups share every SIMD, so one
// group's epilogue (VALU + stores) can run beside the other group's K-loop (matrix pipe); no LDS left for the bias strips
// (the epilogue reads the bias from L2).
template <int NI, int MI, int WM>
struct Geo3 {
  static constexpr int WN = WM == 8 ? 1 : 2;                          // waves along N (WM = 8: eight waves along M, each owning whole rows of the tile)
  static constexpr int NWV = WN * WM;                                 // waves per workgroup
  static constexpr int BN = 16 * NI * WN, BM = 16 * MI * WM;
  static constexpr int ASTG = BM * BK3 * 2, WSTG = BN * BK3 * 2;      // one K-step of activation rows / of weight rows, 128 B per row
  static constexpr int CAP = NWV == 8 ? 160 * 1024 : 80 * 1024;
  static constexpr bool STRIP = NWV == 8;                              // bias strips of two tiles in LDS
  static constexpr int TAIL = STRIP ? 2 * BN * 4 : 0;
  static constexpr int SW = 2;                                        // weight ring: L2 hits, one step ahead
  static constexpr int SA_ = (CAP - TAIL - SW * WSTG) / ASTG;
  static constexpr int SA = SA_ >= 4 ? 4 : SA_;                       // activation ring: 1 to 3 steps ahead
  static constexpr int A_OFF = 0, W_OFF = SA * ASTG, BIAS_OFF = W_OFF + SW * WSTG;
  static constexpr int LDS = BIAS_OFF + TAIL;
  static constexpr int NA = BM / 8 / NWV, NW = BN / 8 / NWV;          // 1 KiB DMA pieces (8 rows) per wave and step: activation, weight
  static constexpr int GPW = NA + NW;
  static constexpr int NBIAS = BN / 64;
  static_assert(SA >= 2 && LDS <= CAP && BM % (8 * NWV) == 0 && BN % (8 * NWV) == 0, "LDS budget / even deal of the pieces over the waves");
};

__device__ __forceinline__ void unpack8(const bf16x8& v, f32x4& a, f32x4& b) {
  a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
__device__ __forceinline__ void ld8(const float* p, f32x4& a, f32x4& b) { a = load4(p); b = load4(p + 4); }
__device__ __forceinline__ void ld8(const bf16* p, f32x4& a, f32x4& b) { unpack8(*reinterpret_cast<const bf16x8*>(p), a, b); }
// Outputs are written once and read by a LATER kernel: non-temporal stores (see gemm_nt2_kernel).  -DMAE_NT3_STORE=1 / 2 / 3 build
// the plain / sc1 / sc0 sc1 forms for A/B runs of the whole step (profiles/r03_nt3_kloop.txt).
#ifndef MAE_NT3_STORE
#define MAE_NT3_STORE 0
#endif
template <class V> __device__ __forceinline__ void stream_store(V v, V* p) {
  static_assert(sizeof(V) == 16, "16-byte stores only");
#if MAE_NT3_STORE == 0
  __builtin_nontemporal_store(v, p);
#elif MAE_NT3_STORE == 1
  *p = v;
#elif MAE_NT3_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
// Stores that cover only PART of a 128-byte line: the 32-column group a wave shares with its neighbour (a third of every bf16 output) and
// fp32 rows, whose two 16-byte halves per lane go out in two instructions.  Non-temporal partial lines are not merged in L2 and reach HBM
// piece by piece: PMC WRITE_SIZE shows +10-11 % on every bf16 launch and +25-33 % on the fp32 ones, 2.1 GB per step; with ordinary stores
// for those pieces (-DMAE_NT3_PART_STORE=1) the counter equals the algorithmic bytes exactly (step 98.6 -> 96.5 GB) but the step is 0.07-0.13 ms
// SLOWER (linear_nt 10.12 -> 10.22 ms, 2 x A/B on one box: the dirty lines now wait in L2 for their other half).  Time wins: nt for the bf16
// half-lines.  The fp32 rows are different: ordinary stores take the two fp32-output launches of a step from 55.2 to 42.9 us (= 2, the default).
#ifndef MAE_NT3_PART_STORE
#define MAE_NT3_PART_STORE 2
#endif
template <class V> __device__ __forceinline__ void part_store(V v, V* p) {
#if MAE_NT3_PART_STORE == 1
  *p = v;
#else
  stream_store(v, p);
#endif
}
__device__ __forceinline__ void st8(float* p, const f32x4& a, const f32x4& b) {
#if MAE_NT3_PART_STORE == 2   // ordinary stores for the fp32 rows only
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
#else
  part_store(a, reinterpret_cast<f32x4*>(p));
  part_store(b, reinterpret_cast<f32x4*>(p + 4));
#endif
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

// Epilogue stores and side-input loads go through raw buffer descriptors: the address is a per-lane 32-bit offset that never changes (row in the
// wave tile x N + column) plus a SCALAR offset (tile, 16-row group, column group), rows past M fall outside the descriptor's range and are dropped
// (loads return zeros; the scalar offset takes part in the range check, tools/micro/buffer_range_probe.hip) -- no 64-bit address arithmetic, no
// row compares and no exec masking in the epilogue.  Cache policy: 2 = nt, 0 = ordinary, 16 = sc1, 17 = sc0 sc1.
constexpr int kStreamPolicy = MAE_NT3_STORE == 0 ? 2 : (MAE_NT3_STORE == 1 ? 0 : (MAE_NT3_STORE == 2 ? 16 : 17));
template <int POLICY, class V>
__device__ __forceinline__ void bstore16(const V& v, __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  static_assert(sizeof(V) == 16, "16-byte stores only");
  // The scalar offset is ADDED INTO the vector offset (one v_add per store) and the instruction's soffset field left at zero: with an SGPR in that
  // field hipcc's hazard recognizer assumes that a VALU may overwrite the data registers of a 128-bit store right behind it (true on older parts),
  // and on gfx950 the DPP move that builds the next store's data then corrupts the first dword of this one (seen in the GELU + slope epilogue:
  // timing-dependent wrong values in dword 0 of scattered rows).
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rs, (int)(voff + soff), 0, POLICY);
}
__device__ __forceinline__ bf16x8 bload16(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
}

// keep `keep` in the lanes whose 4-lane bank is NOT in BANKS and take `from`'s value of the lane 8 places away (row_ror:8) in the others
template <int BANKS>
__device__ __forceinline__ bf16x8 row_merge8(const bf16x8& keep, const bf16x8& from) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  u32x4_ k = __builtin_bit_cast(u32x4_, keep);
  const u32x4_ f = __builtin_bit_cast(u32x4_, from);
#pragma unroll
  for (int i = 0; i < 4; ++i) k[i] = (unsigned)__builtin_amdgcn_update_dpp((int)k[i], (int)f[i], 0x128, 0xf, BANKS, false);
  return __builtin_bit_cast(bf16x8, k);
}

// LDS-DMA pieces from inline asm (the compiler neither counts nor drains them: every wait below is ours).  M0 is written in
// the statement that uses it.  Raw buffer addressing: byte offset = voff (per lane) + soff (scalar), range-checked against
// the descriptor's num_records (out of range -> zeros, no fault).
__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// the same with the non-temporal cache policy (streamed-once rows: the lines are not kept in L2)
__device__ __forceinline__ void dma16_nt(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
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

// the first 32-deep half of a tile: C = 0 inside the instruction instead of 96 zeroed accumulator registers per wave and tile
template <int I0, int I1, int MI, int NI>
__device__ __forceinline__ void mfma_range_first(f32x4 (&acc)[MI][NI], const bf16x8 (&a)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
  for (int i = I0; i < I1; ++i) acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i % NI], a[i / NI], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

// MFMAs that follow DMA piece q in phase 1: the MI * NI - (MI + NI) MFMAs left after the read-interleaved ones, spread evenly
template <int MI, int NI, int GPW>
struct Spread {
  static constexpr int NR = MI + NI, REST = MI * NI - NR;
  static constexpr int lo(int q) { return NR + q * REST / GPW; }
  static constexpr int hi(int q) { return NR + (q + 1) * REST / GPW; }
};

}  // namespace

template <int MODE, class TO, bool HAS_BIAS, int NI, int MI, int WM>
__global__ void __launch_bounds__((64 * Geo3<NI, MI, WM>::NWV), 2) gemm_nt3_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int64_t M, int N, int K,
                                                               const float* __restrict__ bias, const void* __restrict__ aux, TO* __restrict__ out,
                                                               TO* __restrict__ out2, int tiles_m, int tiles_n, int a_nt) {
  using G_ = Geo3<NI, MI, WM>;
  using SP = Spread<MI, NI, G_::GPW>;
  constexpr int BM = G_::BM, BN = G_::BN, ASTG = G_::ASTG, WSTG = G_::WSTG, SA = G_::SA, SW = G_::SW, NA = G_::NA, NW = G_::NW, GPW = G_::GPW;
  constexpr int NJ = NI / 2, NR = MI + NI;
  constexpr bool TWO = MODE == MAE_EPI_GELU_GRAD;
  constexpr bool KEEP = MODE == 77;   // NONE epilogue with ordinary stores: the output is read by the very next kernel and fits the memory-side cache
  constexpr int STORE8 = sizeof(TO) == 2 ? 1 : 2;          // store instructions per 8 outputs
  constexpr int E = MI * NJ * STORE8 * (TWO ? 2 : 1);      // epilogue stores per wave (full tile)
  static_assert(SW == 2, "the waits below assume weight pieces one step ahead");
  constexpr int NAF = SA > 2 ? NA : 0;                      // activation pieces that stay in flight across a step's wait
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G_::WN, wn = wave % G_::WN;
  constexpr int WROWS = 16 * MI;
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x, T = tiles_m * tiles_n;
  const int vb = (int)xcd_remap3(blockIdx.x, G);
  const int ntile = (T - vb + G - 1) / G;
  const int nk = K / BK3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint32_t rowbytes = (uint32_t)K * 2u;

  // ---- producer side: two DMA streams (weight rows one step ahead, activation rows SA - 1 steps ahead), all scalar but two
  // per-lane offsets that never change.  Every wave owns NA activation pieces and NW weight pieces of a step.  The streams are
  // never switched off: past the last tile they fetch rows that do not exist (the descriptor's range check turns those into
  // zeros, or the 32-bit offset wraps onto valid rows) into stages nobody reads any more, which keeps every step's count of
  // vector-memory operations, and with it every wait, the same.
  const i32x4 rsA = make_rsrc(A, (uint32_t)((uint64_t)M * rowbytes));
  const i32x4 rsW = make_rsrc(W, (uint32_t)N * rowbytes);
  const i32x4 rsB = make_rsrc(bias, HAS_BIAS ? (uint32_t)N * 4u : 0u);
  const uint32_t r8 = (uint32_t)lane >> 3;
  const uint32_t voff = r8 * rowbytes + ((((uint32_t)lane & 7u) ^ r8) << 4);   // row r8 of an 8-row piece, swizzled source chunk for LDS slot lane & 7
  const uint32_t voff_b = (uint32_t)lane * 4u;
  const int dq = G / tiles_n, dr = G % tiles_n;            // tile id + G in (row, column) form, without a division per tile
  struct Stream { int tm, tn, k, tile; uint32_t base, lds; };   // tile (row, column), K-step inside it, tile ordinal; scalar source offset and LDS address of the step
  Stream sa, sw;
  sa.tm = sw.tm = vb / tiles_n; sa.tn = sw.tn = vb % tiles_n; sa.k = sw.k = 0; sa.tile = sw.tile = 0;
  sa.base = ((uint32_t)sa.tm * BM + (uint32_t)(wave * NA * 8)) * rowbytes;   // this wave's first activation piece of the step
  sw.base = ((uint32_t)sw.tn * BN + (uint32_t)(wave * NW * 8)) * rowbytes;
  sa.lds = lds0 + (uint32_t)(G_::A_OFF + wave * NA * 1024);
  sw.lds = lds0 + (uint32_t)(G_::W_OFF + wave * NW * 1024);
  const uint32_t piece = 8u * rowbytes;
  auto issue_a = [&](int q) {
#ifndef MAE_DBG_NO_DMA
    if (a_nt) dma16_nt(rsA, sa.lds + (uint32_t)(q * 1024), voff, sa.base + (uint32_t)q * piece);
    else dma16(rsA, sa.lds + (uint32_t)(q * 1024), voff, sa.base + (uint32_t)q * piece);
#endif
  };
  auto issue_w = [&](int q) {
#ifndef MAE_DBG_NO_DMA
    dma16(rsW, sw.lds + (uint32_t)(q * 1024), voff, sw.base + (uint32_t)q * piece);
#endif
  };
  auto issue_bias = [&]() {     // the tile's bias strip rides with the weight pieces of its first step (every wave writes the same bytes)
#ifndef MAE_DBG_NO_DMA
    if (HAS_BIAS && G_::STRIP && sw.k == 0) {
#pragma unroll
      for (int i = 0; i < G_::NBIAS; ++i) dma4(rsB, lds0 + (uint32_t)(G_::BIAS_OFF + (sw.tile & 1) * (BN * 4) + 256 * i), voff_b, (uint32_t)(sw.tn * BN * 4 + 256 * i));
    }
#endif
  };
  auto next_a = [&]() {
    sa.base += 128u;
    sa.lds = sa.lds == lds0 + (uint32_t)(G_::A_OFF + (SA - 1) * ASTG + wave * NA * 1024) ? lds0 + (uint32_t)(G_::A_OFF + wave * NA * 1024) : sa.lds + (uint32_t)ASTG;
    if (++sa.k == nk) {
      sa.k = 0; ++sa.tile;
      sa.tm += dq; sa.tn += dr;
      if (sa.tn >= tiles_n) { sa.tn -= tiles_n; ++sa.tm; }
      sa.base = ((uint32_t)sa.tm * BM + (uint32_t)(wave * NA * 8)) * rowbytes;
    }
  };
  auto next_w = [&]() {
    sw.base += 128u;
    sw.lds = sw.lds == lds0 + (uint32_t)(G_::W_OFF + (SW - 1) * WSTG + wave * NW * 1024) ? lds0 + (uint32_t)(G_::W_OFF + wave * NW * 1024) : sw.lds + (uint32_t)WSTG;
    if (++sw.k == nk) {
      sw.k = 0; ++sw.tile;
      sw.tm += dq; sw.tn += dr;
      if (sw.tn >= tiles_n) { sw.tn -= tiles_n; ++sw.tm; }
      sw.base = ((uint32_t)sw.tn * BN + (uint32_t)(wave * NW * 8)) * rowbytes;
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

  // prologue: activation steps 0 .. SA - 2, then weight step 0 (the youngest: the first wait drains everything)
#pragma unroll
  for (int a = 0; a < SA - 1; ++a) {
#pragma unroll
    for (int q = 0; q < NA; ++q) issue_a(q);
    next_a();
  }
#pragma unroll
  for (int q = 0; q < NW; ++q) issue_w(q);
  issue_bias();
  next_w();

  // ---- consumer side
  const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
  const int gb = (fq & 1) ? 3 + fq : fq;
  const char* rd_a = smem + G_::A_OFF + (wm * WROWS + fr) * 128;            // this lane's fragment rows in the current stages
  const char* rd_w = smem + G_::W_OFF + (wn * (NI * 16) + fr) * 128;
  const char* const rd_a_last = smem + G_::A_OFF + (SA - 1) * ASTG + (wm * WROWS + fr) * 128;
  const char* const rd_w_last = smem + G_::W_OFF + (SW - 1) * WSTG + (wn * (NI * 16) + fr) * 128;
  int c_tm = vb / tiles_n, c_tn = vb % tiles_n;   // tile being multiplied
  bool prev_full = false;
  // epilogue addressing: descriptors over the whole (M, N) outputs / side input, two per-lane offsets that never change
  const uint32_t out_bytes = (uint32_t)((uint64_t)M * (uint64_t)N * sizeof(TO));
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == MAE_EPI_GELU_GRAD ? out2 : out), 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(MODE == MAE_EPI_MUL ? aux : (const void*)out), 0, (int)out_bytes, 0x00020000);
  const uint32_t vo_one = ((uint32_t)(wm * WROWS + fr) * (uint32_t)N + (uint32_t)(wn * (NI * 16) + 4 * gb)) * (uint32_t)sizeof(TO);
  const uint32_t vo_pair = ((uint32_t)(wm * WROWS + (fr & 7)) * (uint32_t)N + (uint32_t)(wn * (NI * 16) + 4 * gb + (fr < 8 ? 0 : 32))) * (uint32_t)sizeof(TO);

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
    const float* sbias = G_::STRIP ? reinterpret_cast<const float*>(smem + G_::BIAS_OFF + strip * (BN * 4)) + wn * (NI * 16) : bias + n0 + wn * (NI * 16);
    constexpr uint32_t OSZ = sizeof(TO);
    const uint32_t rowb = (uint32_t)N * OSZ;                                        // bytes per output row
    const uint32_t so_tile = ((uint32_t)m0 * (uint32_t)N + (uint32_t)n0) * OSZ;      // scalar: the tile's first element (32-bit: checked by the launcher)
#ifndef MAE_DBG_NO_EPI
    if constexpr (sizeof(TO) == 2) {
      // bf16 outputs, whole-line stores (see gemm_nt2_kernel): a lane holds 8 consecutive columns of one row; two neighbouring
      // 32-column groups that form one aligned 128-byte line go out together after the lanes of rows 0-7 and rows 8-15 swapped
      // one group's 16 bytes (DPP row_ror:8)
      constexpr int POL = KEEP ? 0 : kStreamPolicy, POL1 = KEEP ? 0 : (MAE_NT3_PART_STORE == 1 ? 0 : kStreamPolicy);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const uint32_t so = so_tile + (uint32_t)(mi * 16) * rowb;                    // scalar: this 16-row group
        bf16x8 pa[NJ], pb[TWO ? NJ : 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = 32 * j + 4 * gb;
          f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
          if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
          f32x4 v0 = acc[mi][2 * j], v1 = acc[mi][2 * j + 1];
          if (HAS_BIAS) { v0 += b0; v1 += b1; }   // (without a bias no add at all: x + 0.0f is not an identity the compiler may drop)
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
            unpack8(bload16(rsX, vo_one, so + (uint32_t)(64 * j)), q0, q1);        // rows past M read zeros (their products are never stored)
            pa[j] = pk8(v0 * q0, v1 * q1);
          } else {
            pa[j] = pk8(v0, v1);
          }
        }
        auto store_rows = [&](__amdgpu_buffer_rsrc_t rs, const bf16x8* pk, int ja) {
          // S1 = (rows 0-7: own group ja | rows 8-15: the partner's group ja + 1), S2 = (rows 0-7: the partner's group ja | rows 8-15: own group ja + 1):
          // one DPP move per dword with a bank mask, no selects
          const bf16x8 S1 = row_merge8<0xc>(pk[ja], pk[ja + 1]), S2 = row_merge8<0x3>(pk[ja + 1], pk[ja]);
          bstore16<POL>(S1, rs, vo_pair, so + (uint32_t)(64 * ja));
          bstore16<POL>(S2, rs, vo_pair, so + (uint32_t)(64 * ja) + 8u * rowb);
        };
        auto store_single = [&](__amdgpu_buffer_rsrc_t rs, const bf16x8& v, int j) { bstore16<POL1>(v, rs, vo_one, so + (uint32_t)(64 * j)); };
        if (NJ % 2 == 0) {   // the wave's columns are whole 128-byte lines (128- and 256-wide tiles)
#pragma unroll
          for (int ja = 0; ja < NJ; ja += 2) {
            store_rows(rsO, pa, ja);
            if (TWO) store_rows(rsO2, pb, TWO ? ja : 0);
          }
        } else if (wn == 0) {
          store_rows(rsO, pa, 0); store_single(rsO, pa[NJ - 1], NJ - 1);
          if (TWO) { store_rows(rsO2, pb, 0); store_single(rsO2, pb[TWO ? NJ - 1 : 0], NJ - 1); }
        } else {
          store_single(rsO, pa[0], 0); store_rows(rsO, pa, NJ - 2);
          if (TWO) { store_single(rsO2, pb[0], 0); store_rows(rsO2, pb, TWO ? NJ - 2 : 0); }
        }
      }
    } else {
      constexpr int POLF = MAE_NT3_PART_STORE >= 1 ? 0 : kStreamPolicy;   // fp32 rows: two 16-byte halves per lane in two instructions (partial lines)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = 32 * j + 4 * gb;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (HAS_BIAS) { b0 = load4(sbias + nl); b1 = load4(sbias + nl + 4); }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const uint32_t so = so_tile + (uint32_t)(mi * 16) * rowb + (uint32_t)(128 * j);
          f32x4 v0 = acc[mi][2 * j], v1 = acc[mi][2 * j + 1];
          if (HAS_BIAS) { v0 += b0; v1 += b1; }
          bstore16<POLF>(v0, rsO, vo_one, so);
          bstore16<POLF>(v1, rsO, vo_one, so + 16u);
        }
      }
    }
#else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (acc[mi][ni][0] == 1.2345e30f) out[0] = (TO)0.f;
      }
#endif
    prev_full = m0 + BM <= M;
  };



#define NT3_READ(AF, BF, SWZ)                                                                                      \
  {                                                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) AF[mi] = *reinterpret_cast<const bf16x8*>(rd_a + mi * 2048 + (SWZ)); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) BF[ni] = *reinterpret_cast<const bf16x8*>(rd_w + ni * 2048 + (SWZ)); \
  }
#define NT3_IL(n)                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
  }
  // One K-step.  Top: wait for this step's weight pieces (issued in the previous step; everything older, the step's activation
  // pieces included, has retired by then), make every wave's reads of the stages refilled below land, barrier.
  // Phase 1: reads of half 0 beside the MFMAs of the previous step's half 1, then the DMA pieces between the other MFMAs: the
  // next step's weight pieces first, the activation pieces of step + SA - 1 after them.  Phase 2: reads of half 1 beside the
  // MFMAs of half 0; the scalar bookkeeping of the next step sits in its shadow.
#define NT3_STEP_TOP(NVM)                                                                                          \
    wait_vm3<NVM>();                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("" ::: "memory");
#define NT3_PHASE1()                                                                                               \
    NT3_READ(af0, bf0, sw0)                                                                                        \
    mfma_range<0, NR>(acc, af1, bf1);                                                                              \
    NT3_IL(NR)                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int q = 0; q < GPW; ++q) {                                                              \
      if (q < NW) issue_w(q); else issue_a(q - NW);                                                                \
      if (q == NW - 1) issue_bias();                                                                               \
      _Pragma("unroll") for (int i = SP::lo(q); i < SP::hi(q); ++i)                                                \
        acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[i % NI], af1[i / NI], acc[i / NI][i % NI], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
#define NT3_PHASE2_(RANGE)                                                                                         \
    NT3_READ(af1, bf1, sw1)                                                                                        \
    RANGE<0, NR>(acc, af0, bf0);                                                                                   \
    NT3_IL(NR)                                                                                                     \
    next_w(); next_a();                                                                                            \
    rd_a = rd_a == rd_a_last ? rd_a - (SA - 1) * ASTG : rd_a + ASTG;                                               \
    rd_w = rd_w == rd_w_last ? rd_w - (SW - 1) * WSTG : rd_w + WSTG;                                               \
    RANGE<NR, MI * NI>(acc, af0, bf0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define NT3_PHASE2() NT3_PHASE2_(mfma_range)
#define NT3_PHASE2_FIRST() NT3_PHASE2_(mfma_range_first)   /* a tile's first half: the accumulators start from zero */

#ifdef MAE_DBG_NT3_CLOCK   // alternate build: the in-kernel clock and cycles per K-step of one wave (printf perturbs the launch a little)
  const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int tile = 0; tile < ntile; ++tile) {
    // ---- a tile's first step: the previous tile is completed by phase 1 and stored before phase 2
    // (outstanding at the wait: the activation pieces issued after the awaited weight pieces in the previous step)
    if (tile == 0) { NT3_STEP_TOP(0) } else { NT3_STEP_TOP(NAF) }
    NT3_PHASE1()
    if (tile > 0) {
      const int p_tm = c_tm - dq - (c_tn - dr < 0 ? 1 : 0), p_tn = c_tn - dr < 0 ? c_tn - dr + tiles_n : c_tn - dr;
      epilogue((int64_t)p_tm * BM, p_tn * BN, (tile - 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    NT3_PHASE2_FIRST()
    // ---- the other steps; at the second one the epilogue's stores (counted only when that tile was full: every store was issued)
    // may stay in flight too
    const bool stores = tile > 0 && prev_full;
    for (int k = 1; k < nk; ++k) {
#ifdef MAE_DBG_NT3_LOOSE   // TIMING PROBE, WRONG VALUES: the epilogue's stores may stay in flight for LOOSE steps instead of one
      if (k <= MAE_DBG_NT3_LOOSE && stores) { NT3_STEP_TOP(NAF + E) } else { NT3_STEP_TOP(NAF) }
#else
      if (k == 1 && stores) { NT3_STEP_TOP(NAF + E) } else { NT3_STEP_TOP(NAF) }
#endif
      NT3_PHASE1()
      NT3_PHASE2()
    }
    c_tm += dq; c_tn += dr;
    if (c_tn >= tiles_n) { c_tn -= tiles_n; ++c_tm; }
  }
  mfma_range<0, MI * NI>(acc, af1, bf1);
  {
    const int p_tm = c_tm - dq - (c_tn - dr < 0 ? 1 : 0), p_tn = c_tn - dr < 0 ? c_tn - dr + tiles_n : c_tn - dr;
    epilogue((int64_t)p_tm * BM, p_tn * BN, (ntile - 1) & 1);
  }
#ifdef MAE_DBG_NT3_CLOCK
  if (blockIdx.x == 100 && tid == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    printf("nt3 clock: M %d N %d K %d mode %d tile %dx%d | %d tiles x %d steps | %.2f GHz | %d cycles per step | %.1f us\n", (int)M, N, K, MODE, BM, BN, ntile, nk,
           (double)(t1 - ck_t0) / (double)(r1 - ck_r0) * 0.1, (int)((t1 - ck_t0) / (unsigned long long)(ntile * nk)), (double)(r1 - ck_r0) * 0.01);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the phantom pieces issued past the last tile land before the LDS is released
#undef NT3_PHASE2_FIRST
#undef NT3_PHASE2
#undef NT3_PHASE2_
#undef NT3_PHASE1
#undef NT3_STEP_TOP
#undef NT3_READ
#undef NT3_IL
}

template <int MODE, class TO, int NI, int MI, int WM>
static int launch_nt3(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  using G_ = Geo3<NI, MI, WM>;
  const int64_t T = cdiv(M, G_::BM) * (N / G_::BN);
  MAE_REQUIRE(T < (1ll << 30), "gemm: too many tiles");
  const int tiles_m = (int)cdiv(M, G_::BM), tiles_n = N / G_::BN;
  const int grid = (int)std::min<int64_t>(T, (int64_t)num_cus() * (G_::NWV == 8 ? 1 : 2));  // persistent: one 8-wave or two 4-wave workgroups per CU
  // activation rows that a single workgroup reads (tiles_n == 1: the 192-wide decoder outputs) are streamed with the non-temporal policy:
  // cold, those shapes run 15-21 % faster (profiles/r03_nt3_kloop.txt); panels shared by several tile columns must stay in L2.
  // MAE_NT3_ANT = largest tiles_n that gets the policy (0 = never; A/B)
  static const int ant_max = [] { const char* v = getenv("MAE_NT3_ANT"); return v ? atoi(v) : 1; }();
  const int a_nt = tiles_n <= ant_max ? 1 : 0;
  if (e.bias) {
    auto kern = gemm_nt3_kernel<MODE, TO, true, NI, MI, WM>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * G_::NWV), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n, a_nt);
  } else {
    auto kern = gemm_nt3_kernel<MODE, TO, false, NI, MI, WM>;
    MAE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_::LDS));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * G_::NWV), G_::LDS, s, A, W, M, N, K, e.bias, e.aux, (TO*)e.out, (TO*)e.out2, tiles_m, tiles_n, a_nt);
  }
  MAE_LAUNCH_CHECK();
  return 0;
}

// rounds of tiles on the CUs x rows per tile = time proxy; the 192-row tile must win by a margin: it stages 15 % more bytes per flop and, with two
// waves across its 192 columns, stores a third of every row as half lines, which the eight-waves-along-M layout of the 256-row tile does not
// (decoder launches, M = 290000, N = 192: 1152 against 1280 row-rounds, yet 117.6 us on 256-row tiles against 124.2 us; encoder N = 384:
// 576 against 768 row-rounds, 86.0 against 99.4 us)
static bool prefer_bm192_3(int64_t M, int N) {
  static const int force = [] { const char* v = getenv("MAE_NT_BM"); return v ? atoi(v) : 0; }();
  if (force == 192) return true;
  if (force == 256) return false;
  const int64_t t256 = cdiv(M, 256) * (N / 192), t192 = cdiv(M, 192) * (N / 192);
  const int64_t c256 = cdiv(t256, num_cus()) * 256, c192 = cdiv(t192, num_cus()) * 192;
  return c192 * 100 < c256 * 85;   // (95 and 85 measure the same in the step: 10.03 ms of NT GEMMs either way, ViT-B/16 21.83)
}

template <int MODE, class TO>
static int launch_nt3_ni(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  const char* var = getenv("MAE_GEMM_NT");   // A/B: "v3w2" = two 4-wave workgroups per CU on 128-row tiles
  const bool w2 = var && strstr(var, "w2") != nullptr;
  // Widths that are multiples of 256 take 192 x 256 tiles -- the 256 x 192 tile transposed: same staged bytes per flop and the same 24 MFMA tiles per
  // wave (48 x 128), but a wave then owns 128 columns = two whole 128-byte lines of every output row (no partial-line stores) and a row panel is
  // shared by N / 256 workgroups instead of N / 192.  gemm_bench: -4 ... -11 % on the fc1 / fc2-dgrad / 512- and 1024-wide launches, bit-identical;
  // in the step: default -0.08 ms, ViT-B/16 -0.5 ms, ViT-L/14 -3.7 ms.  MAE_NT_N256=0 keeps the 192-wide tiles (A/B).
  static const int n256 = [] { const char* v = getenv("MAE_NT_N256"); return v ? atoi(v) : 1; }();
  if (n256 && N % 256 == 0 && !w2) return launch_nt3<MODE, TO, 8, 3, 4>(A, W, M, N, K, e, s);
  // (128 x 384 tiles -- wave tile 32 x 192, three whole lines per row, for 384 / 1152 -- stage 14 % more bytes per flop: +1.0 ms per step, not kept)
  if (N % 192 == 0) {
    if (w2) return launch_nt3<MODE, TO, 6, 4, 2>(A, W, M, N, K, e, s);
    if (prefer_bm192_3(M, N)) return launch_nt3<MODE, TO, 6, 3, 4>(A, W, M, N, K, e, s);
    // bf16 outputs: the 256 x 192 tile with EIGHT waves along M, each owning 32 whole rows (three full lines per row: no partial-line stores; 14 instead
    // of 10 fragment reads per half-step, which the LDS absorbs).  gemm_bench -1 ... -4 % on the 192- / 576- / 1152-wide launches, K = 4096 1278 -> 1285 TF/s,
    // step -0.02 ... -0.05 ms; fp32 outputs (two half-row stores per lane) lose 3 % and keep the 4 x 2 layout.  MAE_NT_WN1=0: 4 x 2 everywhere (A/B).
    static const int wn1 = [] { const char* v = getenv("MAE_NT_WN1"); return v ? atoi(v) : 1; }();
    if (wn1 && sizeof(TO) == 2) return launch_nt3<MODE, TO, 12, 2, 8>(A, W, M, N, K, e, s);
    return launch_nt3<MODE, TO, 6, 4, 4>(A, W, M, N, K, e, s);
  }
  if (w2) return launch_nt3<MODE, TO, 4, 4, 2>(A, W, M, N, K, e, s);
  return launch_nt3<MODE, TO, 4, 4, 4>(A, W, M, N, K, e, s);
}

int mfma_linear_fwd_v3(const bf16* A, const bf16* W, int64_t M, int N, int K, const Epi& e, hipStream_t s) {
  if (K % 64 != 0 || K < 192 || (N % 128 != 0 && N % 192 != 0) || M < 1) return MFMA_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)e.out | (uintptr_t)e.out2 | (uintptr_t)e.bias | (uintptr_t)e.aux) & 15) != 0) return MFMA_UNSUPPORTED;
  // 32-bit byte offsets inside the buffer descriptors (a tile may start up to 255 rows before the end and reach 256 rows past it)
  if ((uint64_t)(M + 512) * (uint64_t)K * 2u >= (1ull << 32) || (uint64_t)N * (uint64_t)K * 2u >= (1ull << 32)) return MFMA_UNSUPPORTED;
  if ((uint64_t)(M + 512) * (uint64_t)N * (e.out_dt == MAE_F32 ? 4u : 2u) >= (1ull << 32)) return MFMA_UNSUPPORTED;   // the epilogue's buffer offsets are 32-bit too
  const bool f32out = e.out_dt == MAE_F32;
  switch (e.mode) {
    case MAE_EPI_NONE: {
      // bf16 outputs of at most 64 MB (MAE_NT_KEEP=<bytes>; 0 = never) are written with ordinary stores: they stay in the 256 MB memory-side cache for the
      // LayerNorm / attention kernel that reads them next (proj, fc2 and the dgrads of the 384-wide encoder: LayerNorm forward 1.81 -> 1.78 ms, backward
      // 2.54 -> 2.50 ms per step, the GEMMs unchanged; at 400 MB the GEMMs lose more than the readers win)
      static const int64_t keep = [] { const char* v = getenv("MAE_NT_KEEP"); return v ? atoll(v) : 64ll << 20; }();
      if (!f32out && keep > 0 && M * (int64_t)N * 2 <= keep && N % 256 != 0) return launch_nt3_ni<77, bf16>(A, W, M, N, K, e, s);   // (the 192 x 256 tiles lose 0.6 % with it: ViT-B/16)
      return f32out ? launch_nt3_ni<MAE_EPI_NONE, float>(A, W, M, N, K, e, s) : launch_nt3_ni<MAE_EPI_NONE, bf16>(A, W, M, N, K, e, s);
    }
    case MAE_EPI_GELU_GRAD: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_GELU_GRAD, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_GELU_ACT: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_GELU_ACT, bf16>(A, W, M, N, K, e, s);
    case MAE_EPI_MUL: return f32out ? MFMA_UNSUPPORTED : launch_nt3_ni<MAE_EPI_MUL, bf16>(A, W, M, N, K, e, s);
    default: return MFMA_UNSUPPORTED;
  }
}

}  // namespace mae
