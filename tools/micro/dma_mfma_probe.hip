// Micro-probe (round 3): does issuing LDS-DMA pieces cost the matrix pipe anything, and whose time?
// One 512-thread workgroup per CU (waves w and w + 4 share a SIMD).  Per "step" a wave runs 48 bf16 MFMAs 16x16x32 (the NT GEMM's
// 64 x 96 wave tile, operands in registers) and / or issues LDS-DMA pieces of 1 KiB from an L2-resident window:
//   mode 0  waves 0-3: MFMAs, waves 4-7: nothing                      (one wave per SIMD)
//   mode 1  waves 0-3: MFMAs, waves 4-7: 14 DMA pieces per step       (loader waves beside MFMA waves, 56 KiB per step and CU)
//   mode 2  all 8 waves: MFMAs + 7 pieces per step, one piece after every 6th MFMA   (what gemm_nt3_kernel does, spread out)
//   mode 3  all 8 waves: MFMAs only
//   mode 4  all 8 waves: 7 pieces per step only
//   mode 5  as mode 2 with global_load_lds (64-bit per-lane address) instead of buffer_load ... lds
// Reported: time per step, MFMA TFLOP/s, DMA GB/s per CU, the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_mfma_probe.hip -o /tmp/dma_mfma_probe && /tmp/dma_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void gdma16(const char* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) probe(const u32x4* __restrict__ src, const char* __restrict__ win, unsigned win_bytes, float* __restrict__ out,
                                                unsigned long long* __restrict__ ticks, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint64_t a = (uint64_t)(uintptr_t)win;
  const i32x4 rs = i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)win_bytes, 0x00020000};
  bf16x8 af[4], bfr[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 4095]);
#pragma unroll
  for (int i = 0; i < 6; ++i) bfr[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 4 + i) & 4095]);
  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_mfma = MODE == 2 || MODE == 3 || MODE == 5 || ((MODE == 0 || MODE == 1) && wave < 4);
  const bool do_dma = MODE == 2 || MODE == 4 || MODE == 5 || (MODE == 1 && wave >= 4);
  const int pieces = MODE == 1 ? 14 : 7;
  const uint32_t voff = lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t pos = (uint32_t)wave * 14336u;
  for (int s = 0; s < steps; ++s) {
    if (MODE == 2 || MODE == 5) {
#pragma unroll
      for (int q = 0; q < 7; ++q) {
#pragma unroll
        for (int i = q * 48 / 7; i < (q + 1) * 48 / 7; ++i) {
          const int ii = i % 24;
          acc[ii / 6][ii % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ii % 6], af[ii / 6], acc[ii / 6][ii % 6], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2) dma16(rs, lds0 + (uint32_t)((s & 1) * 57344 + wave * 7168 + q * 1024), voff, pos);
        else gdma16(win + pos + voff, lds0 + (uint32_t)((s & 1) * 57344 + wave * 7168 + q * 1024));
        pos += 1024u;
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      pos += 57344u - 7168u;
      if (pos + 8192u > win_bytes) pos = (uint32_t)wave * 14336u;
    } else {
      if (do_mfma) {
#pragma unroll
        for (int i = 0; i < 48; ++i) {
          const int ii = i % 24;
          acc[ii / 6][ii % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ii % 6], af[ii / 6], acc[ii / 6][ii % 6], 0, 0, 0);
        }
      }
      if (do_dma) {
        for (int q = 0; q < pieces; ++q) {
          dma16(rs, lds0 + (uint32_t)((s & 1) * 57344 + (wave & 3) * 14336 + q * 1024), voff, pos);
          pos += 1024u;
        }
        if (pieces == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        pos += 57344u;
        if (pos + 16384u > win_bytes) pos = (uint32_t)wave * 14336u;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) v += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = v;
  if (lane == 0) { ticks[(blockIdx.x * 8 + wave) * 2] = t1 - t0; ticks[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

// ---- modes 6-9: the K-step of gemm_nt3_kernel rebuilt piece by piece (64 x 96 wave tile, two fragment sets, reads of the next half
// beside the MFMAs of the current one, 7 DMA pieces between pairs of MFMAs):
//   mode 6  reads + MFMAs + pieces, no barrier (vmcnt(7) at the end of a step)
//   mode 7  + s_waitcnt vmcnt(0), lgkmcnt(0), s_barrier at the top of every step (a 2-stage ring: one step in flight)
//   mode 8  mode 7 without the pieces (reads + MFMAs + barrier)
//   mode 10 mode 9 with a 3-deep activation ring: a step issues its 3 weight pieces (for the next step) first, then the 4 activation pieces of the
//           step after next, and waits with vmcnt(4) at the top: the activation pieces have two steps to arrive from HBM
//   mode 9  mode 7 with the activation pieces (4 of 7 per wave) streamed ONCE from a large buffer, each 32 KiB slice shared by 8
//           workgroups of one XCD (equal blockIdx % 8) (first touch: HBM), weight pieces from the L2-resident window
//   mode 13 mode 7 with the GEMM's source pattern, everything L2-resident: a piece is 8 rows x 128 B of a 448-row block at a 768-byte
//           row pitch (K = 384), swizzled chunk order inside a row, the K-step selects the 128-byte column; 344 KiB block shared by all
//   mode 11 / 12  modes 10 / 9 with the K order of a 6-step tile rotated by the workgroup's position among the 8 that share its slices:
//           at any time the sharers ask for 6 different slices, a slice's first request pays HBM, the others find it in L2
template <int MODE>
__global__ void __launch_bounds__(512, 2) kstep(const char* __restrict__ win, unsigned win_bytes, const char* __restrict__ big, unsigned big_bytes, float* __restrict__ out,
                                                unsigned long long* __restrict__ ticks, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = 4, NI = 6, NR = 10, STAGE = 57344;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  for (int i = tid; i < ((MODE == 10 || MODE == 11 || MODE == 14) ? 147456 : 2 * STAGE) / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u, 0x3c00bc00u, 0x3c003c00u, 0xbc003c00u};
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint64_t a = (uint64_t)(uintptr_t)win, b = (uint64_t)(uintptr_t)big;
  const i32x4 rs = i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)win_bytes, 0x00020000};
  const i32x4 rb = i32x4{(int)(uint32_t)b, (int)(uint32_t)((b >> 32) & 0xffffu), (int)big_bytes, 0x00020000};
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) af1[i] = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
#pragma unroll
  for (int j = 0; j < NI; ++j) bf1[j] = __builtin_bit_cast(bf16x8, u32x4{0, 0, 0, 0});
  const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
  const int a_lane = (wm * 64 + fr) * 128, b_lane = 256 * 128 + (wn * 96 + fr) * 128;
  const uint32_t r8_ = (uint32_t)lane >> 3;
  const uint32_t voff = MODE == 13 ? r8_ * 768u + ((((uint32_t)lane & 7u) ^ r8_) << 4) : lane * 16;
  uint32_t pos = (uint32_t)wave * 7168u;
  // streamed activation slices: group = blockIdx / 8 ... 32 groups, each walks its own part of `big`
  const uint32_t big_part = big_bytes / 32u;
  const uint32_t grp = (blockIdx.x & 7u) | ((blockIdx.x >> 6) << 3);   // 8 workgroups of ONE XCD (equal blockIdx % 8) stream the same slices
  uint32_t bpos = grp * big_part + (uint32_t)wave * 4096u;
  int cs = 0;
#define RD(AF, BF, SW)                                                                                             \
  {                                                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) AF[mi] = *reinterpret_cast<const bf16x8*>(stgA + a_lane + mi * 2048 + (SW)); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) BF[ni] = *reinterpret_cast<const bf16x8*>(stgB + b_lane + ni * 2048 + (SW)); \
  }
#define IL(n) _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#define MM(AF, BF, I0, I1) _Pragma("unroll") for (int i = (I0); i < (I1); ++i) acc[i / NI][i % NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[i % NI], AF[i / NI], acc[i / NI][i % NI], 0, 0, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
    if (MODE == 10 || MODE == 11 || MODE == 14) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE >= 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::: "memory");
    const char* stgA = (MODE == 10 || MODE == 11 || MODE == 14) ? smem + (s % 3) * 32768 : smem + cs * STAGE;
    const char* stgB = (MODE == 10 || MODE == 11 || MODE == 14) ? smem + 98304 + cs * 24576 - 32768 : smem + cs * STAGE;
    RD(af0, bf0, sw0)
    MM(af1, bf1, 0, NR)
    IL(NR)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      if (MODE != 8) {
        const uint32_t dst = lds0 + (uint32_t)((cs ^ 1) * STAGE + wave * 7168 + q * 1024);
        if (MODE == 10 || MODE == 11 || MODE == 14) {   // pieces 0-2: weights of the next step; pieces 3-6: activations of the step after next (ring of 3 x 32 KiB at 112 KiB)
          if (q < 3) dma16(rs, lds0 + (uint32_t)(98304 + (cs ^ 1) * 24576 + (wave * 3 + q) * 1024), voff, pos + q * 1024u);
          else if (MODE == 14) dma16(rs, lds0 + (uint32_t)(((s + 2) % 3) * 32768 + (wave * 4 + q - 3) * 1024), voff, pos + q * 1024u);
          else dma16(rb, lds0 + (uint32_t)(((s + 2) % 3) * 32768 + (wave * 4 + q - 3) * 1024), voff, bpos + (q - 3) * 1024u);
        } else if ((MODE == 9 || MODE == 12) && q < 4) dma16(rb, dst, voff, bpos + q * 1024u);
        else if (MODE == 15) { if (s + 1 < steps) dma16(rs, dst, voff, pos + q * 1024u); }
        else if (MODE == 13) dma16(rs, dst, voff, (uint32_t)((wave * 7 + q) * 8) * 768u + (uint32_t)(s % 6) * 128u);
        else dma16(rs, dst, voff, pos + q * 1024u);
      }
      MM(af1, bf1, NR + 2 * q, NR + 2 * q + 2)
      __builtin_amdgcn_sched_barrier(0);
    }
    RD(af1, bf1, sw1)
    MM(af0, bf0, 0, NR)
    IL(NR)
    MM(af0, bf0, NR, MI * NI)
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 6) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    pos += 57344u;
    if (pos + 65536u > win_bytes) pos = (uint32_t)wave * 7168u;
    if (MODE == 11 || MODE == 12) {   // slice of the step being fetched next: tile * 6 + (k + position) % 6
      const int sn = s + 1, tile = sn / 6, k = sn % 6, j = (blockIdx.x >> 3) & 7;
      bpos = grp * big_part + (uint32_t)((tile % 160) * 6 + (k + j) % 6) * 32768u + (uint32_t)wave * 4096u;
    } else {
      bpos += 32768u;
      if (bpos + 32768u > (grp + 1) * big_part) bpos = grp * big_part + (uint32_t)wave * 4096u;
    }
    cs ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) v += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = v + (float)af1[0][0];
  if (lane == 0) { ticks[(blockIdx.x * 8 + wave) * 2] = t1 - t0; ticks[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
#undef RD
#undef IL
#undef MM
}

int main() {
  const int steps = 4000, wgs = 256;
  std::vector<unsigned> h(4096 * 4);
  unsigned x = 12345;
  for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x & 0xbf80bf80u) | 0x3c003c00u; }
  u32x4* src; float* out; char* win; unsigned long long* ticks;
  const unsigned win_bytes = 2u << 20;   // 2 MiB: stays in every XCD's L2
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, wgs * 512 * 4); hipMalloc(&win, win_bytes); hipMalloc(&ticks, wgs * 8 * 2 * 8);
  {
    std::vector<unsigned> hw(win_bytes / 4);
    unsigned y = 777;
    const bool rnd = getenv("PROBE_RANDOM") != nullptr;
    for (auto& w : hw) { y = y * 1664525u + 1013904223u; w = rnd ? ((y & 0xbf80bf80u) | 0x3c003c00u) : 0x3c3c3c3cu; }
    hipMemcpy(win, hw.data(), win_bytes, hipMemcpyHostToDevice);
    printf("window data: %s\n", rnd ? "random bf16" : "constant");
  }
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  auto run = [&](int mode) {
    auto launch = [&] {
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
        default: hipLaunchKernelGGL(probe<5>, dim3(wgs), dim3(512), 2 * 57344, 0, src, win, win_bytes, out, ticks, steps); break;
      }
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(wgs * 16);
    hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
    double mlo = 0, mhi = 0, rlo = 0, rhi = 0;   // waves 0-3 / waves 4-7: s_memtime ticks, s_memrealtime ticks (100 MHz)
    for (int b = 0; b < wgs; ++b)
      for (int w = 0; w < 8; ++w) {
        (w < 4 ? mlo : mhi) += (double)t[(b * 8 + w) * 2];
        (w < 4 ? rlo : rhi) += (double)t[(b * 8 + w) * 2 + 1];
      }
    const double us_lo = rlo / (wgs * 4) * 0.01, us_hi = rhi / (wgs * 4) * 0.01, ghz = (mlo + mhi) / (rlo + rhi) * 0.1;
    const int mfma_waves = mode == 0 || mode == 1 ? 4 : (mode == 4 ? 0 : 8);
    const double fl = 48.0 * 16384 * mfma_waves * steps * wgs;
    const double bytes = (mode == 1 ? 4 * 14 : (mode == 2 || mode == 4 || mode == 5 ? 8 * 7 : 0)) * 1024.0 * steps;
    const char* names[6] = {"4 MFMA waves (one per SIMD)", "4 MFMA waves + 4 loader waves", "8 waves: MFMAs + buffer DMA pieces", "8 MFMA waves", "8 waves: DMA pieces only",
                            "8 waves: MFMAs + global DMA pieces"};
    printf("mode %d %-36s: %8.3f ms | waves 0-3 %7.1f us, waves 4-7 %7.1f us | %6.0f TFLOP/s | DMA %6.1f GB/s per CU | %.2f GHz | %5.0f ns per step\n", mode, names[mode], ms, us_lo, us_hi,
           fl / (us_lo * 1e-6) / 1e12, bytes / ((mode == 1 ? us_hi : us_lo) * 1e-6) / 1e9, ghz, ms * 1e6 / steps);
  };
  for (int m : {3, 0, 1, 4, 2, 5}) run(m);
  char* big; const unsigned big_bytes = 1u << 30;   // 1 GiB streamed once per launch (steps x 32 KiB x 32 groups x ... wraps inside each group's 32 MiB part)
  hipMalloc(&big, big_bytes); hipMemset(big, 0x3c, big_bytes);
  auto run2 = [&](int mode) {
    auto launch = [&] {
      switch (mode) {
        case 6: hipLaunchKernelGGL(kstep<6>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 7: hipLaunchKernelGGL(kstep<7>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 8: hipLaunchKernelGGL(kstep<8>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 9: hipLaunchKernelGGL(kstep<9>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 11: hipFuncSetAttribute((const void*)kstep<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
                 hipLaunchKernelGGL(kstep<11>, dim3(wgs), dim3(512), 147456, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 14: hipFuncSetAttribute((const void*)kstep<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
                 hipLaunchKernelGGL(kstep<14>, dim3(wgs), dim3(512), 147456, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 15: hipLaunchKernelGGL(kstep<15>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 13: hipLaunchKernelGGL(kstep<13>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        case 12: hipLaunchKernelGGL(kstep<12>, dim3(wgs), dim3(512), 2 * 57344, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
        default: hipFuncSetAttribute((const void*)kstep<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
                 hipLaunchKernelGGL(kstep<10>, dim3(wgs), dim3(512), 147456, 0, win, win_bytes, big, big_bytes, out, ticks, steps); break;
      }
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(wgs * 16);
    hipMemcpy(t.data(), ticks, t.size() * 8, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0;
    for (size_t i = 0; i < t.size(); i += 2) { mt += (double)t[i]; rt += (double)t[i + 1]; }
    const char* names[10] = {"reads + MFMAs + pieces, no barrier", "+ vmcnt(0) + barrier per step", "reads + MFMAs + barrier, no pieces", "as 7, activation pieces first-touch (HBM)", "as 9, activation ring 3 deep (2 steps ahead)", "as 10 + K order rotated among the sharers", "as 9 + K order rotated among the sharers", "as 7, GEMM source pattern (768-B pitch), L2-resident", "as 10, every piece from the L2 window", "as 7, a uniform branch around every piece"};
    printf("mode %d %-42s: %8.3f ms | %5.0f ns per step | %6.0f TFLOP/s | %.2f GHz\n", mode, names[mode - 6], ms, ms * 1e6 / steps,
           2.0 * 256 * 192 * 64 * (double)steps * wgs / ms / 1e9, mt / rt * 0.1);
  };
  for (int m : {8, 7, 15, 14, 7, 15, 14}) run2(m);
  return 0;
}
