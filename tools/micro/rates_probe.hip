// Micro-probe (round 3): the per-CU rates the NT GEMM's K-step is made of, each with the in-kernel clock beside it
// (s_memtime ticks per s_memrealtime tick x 100 MHz), so that "cycles" and "nanoseconds" can be told apart:
//   1. LDS fragment reads (ds_read_b128 / ds_read_b64) by 4 / 8 / 16 waves per CU, drained every 16 reads or kept in flight
//   2. bf16 MFMA issue (16x16x32 and 32x32x16), 4 or 8 waves per CU, random operands in registers
//   3. LDS-DMA fill (global_load_lds_dwordx4) from an L2-resident window, 8 waves, one or two steps in flight, no consumer
//   4. reads + MFMAs of one wave, software-pipelined (two fragment register sets), 4 / 8 waves: does the LDS keep the matrix pipe fed?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rates_probe.hip -o /tmp/rates_probe && /tmp/rates_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct Stamp { unsigned long long mt, rt; };
__device__ __forceinline__ Stamp stamp() { return Stamp{__builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime()}; }

__device__ __forceinline__ void fill_lds(char* smem, int bytes, int tid, int nthreads) {
  for (int i = tid; i < bytes / 16; i += nthreads) {
    unsigned x = i * 2654435761u;
    reinterpret_cast<u32x4*>(smem)[i] = u32x4{(x & 0x3f803f80u) | 0x3c003c00u, ((x >> 3) & 0x3f803f80u) | 0x3c003c00u, ((x >> 5) & 0x3f803f80u) | 0xbc003c00u, ((x >> 7) & 0x3f803f80u) | 0x3c00bc00u};
  }
}

// ---- 1. LDS read rate.  KIND 0 = ds_read_b128, 1 = ds_read_b64.  MODE 0 = 16 reads, lgkmcnt(0); 1 = two sets of 8, lgkmcnt(8) between
template <int KIND, int MODE>
__global__ void lds_read(unsigned* __restrict__ out, unsigned long long* __restrict__ ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fill_lds(smem, 128 * 1024, tid, blockDim.x);
  __syncthreads();
  unsigned acc = 0;
  const unsigned base0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + lane * (KIND == 0 ? 16 : 8) + (wave & 7) * 64;
  const Stamp s0 = stamp();
  u32x4 a4[8], b4[8];
  u32x2 a2[8], b2[8];
#define RD(dst4, dst2, r, it) \
  if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst4[r]) : "v"(base0 + (((it) & 3) * 24576) + (r) * 1024)); \
  else asm volatile("ds_read_b64 %0, %1" : "=v"(dst2[r]) : "v"(base0 + (((it) & 3) * 24576) + (r) * 512));
#define USE(dst4, dst2) _Pragma("unroll") for (int r = 0; r < 8; ++r) acc ^= KIND == 0 ? (dst4[r][0] ^ dst4[r][3]) : (dst2[r][0] ^ dst2[r][1]);
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { RD(a4, a2, r, it) }
#pragma unroll
      for (int r = 0; r < 8; ++r) { RD(b4, b2, r, it + 1) }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      USE(a4, a2) USE(b4, b2)
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) { RD(a4, a2, r, 0) }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { RD(b4, b2, r, it + 1) }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      USE(a4, a2)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 8; ++r) { RD(a4, a2, r, it + 2) }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      USE(b4, b2)
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    USE(a4, a2)
  }
  const Stamp s1 = stamp();
  out[blockIdx.x * blockDim.x + tid] = acc;
  if (tid == 0) { ticks[2 * blockIdx.x] = s1.mt - s0.mt; ticks[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
#undef RD
#undef USE
}

// ---- 2. MFMA issue rate.  SHAPE 0 = 16x16x32 (16 accumulators of 4), 1 = 32x32x16 (4 accumulators of 16)
template <int SHAPE>
__global__ void mfma_rate(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 4095]);
    b[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 4095]);
  }
  float v = 0.f;
  const Stamp s0 = stamp();
  if (SHAPE == 0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 1], acc[i & 3], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v += acc[i][0] + acc[i][15];
  }
  const Stamp s1 = stamp();
  out[blockIdx.x * blockDim.x + tid] = v;
  if (tid == 0) { ticks[2 * blockIdx.x] = s1.mt - s0.mt; ticks[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
}

// ---- 3. LDS-DMA fill rate: 8 waves x 7 x 1 KiB per step (56 KiB, the 256 x 192 stage) from a window of `win` bytes per `share` workgroups
__device__ __forceinline__ void glds16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int AHEAD>
__global__ void __launch_bounds__(512) dma_rate(const char* __restrict__ stream, unsigned long long* __restrict__ ticks, int steps, int share, int win_steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* my = stream + (size_t)(blockIdx.x / share) * ((size_t)win_steps * 57344) + wave * 7168 + lane * 16;
  const Stamp s0 = stamp();
  int slot = 0;
  for (int s = 0; s < steps; ++s) {
    const char* g = my + (size_t)(s % win_steps) * 57344;
    char* d = smem + slot * 57344 + wave * 7168;
#pragma unroll
    for (int q = 0; q < 7; ++q) glds16(g + q * 1024, d + q * 1024);
    if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    slot = slot == 1 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const Stamp s1 = stamp();
  if (tid == 0) { ticks[2 * blockIdx.x] = s1.mt - s0.mt; ticks[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
}

// ---- 4. one wave's K-step, software-pipelined: reads of chunk c+1 (MI + NI ds_read_b128, set B) interleaved with the MI x NI MFMAs of chunk c (set A)
template <int MI, int NI, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) pipe_rate(float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  fill_lds(smem, 128 * 1024, tid, blockDim.x);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const char* a_base = smem + ((wave & 3) * 16 * MI + fr) * 128 + ((fq ^ (fr & 7)) << 4);
  const char* b_base = smem + 65536 + ((wave >> 2) * 16 * NI + fr) * 128 + ((fq ^ (fr & 7)) << 4);
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][MI], bfr[2][NI];
#define RD(set, it)                                                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) af[set][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + (((it) & 1) << 6)); \
  _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) bfr[set][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + (((it) & 1) << 6));
#define MM(set)                                                                                                           \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                    \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
#define IL()                                                                                                              \
  _Pragma("unroll") for (int i = 0; i < MI + NI; ++i) {                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, (MI * NI) / (MI + NI), 0);                                                \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                    \
  }                                                                                                                       \
  __builtin_amdgcn_sched_group_barrier(0x008, (MI * NI) % (MI + NI), 0);
  RD(0, 0)
  const Stamp s0 = stamp();
  for (int it = 0; it < iters; it += 2) {
    asm volatile("" ::: "memory");
    RD(1, it + 1) MM(0) IL()
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    RD(0, it + 2) MM(1) IL()
    __builtin_amdgcn_sched_barrier(0);
  }
  const Stamp s1 = stamp();
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) v += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + tid] = v + (float)af[0][0][0];
  if (tid == 0) { ticks[2 * blockIdx.x] = s1.mt - s0.mt; ticks[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
#undef RD
#undef MM
#undef IL
}

static unsigned long long* g_ticks;
static void* g_out;
static double clock_ghz() {
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), g_ticks, 512 * 8, hipMemcpyDeviceToHost);
  double mt = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { mt += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  return mt / rt * 0.1;
}
template <class F>
static float timed(F&& launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms;
}

template <int KIND, int MODE>
static void run_lds(int waves) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)lds_read<KIND, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const float ms = timed([&] { hipLaunchKernelGGL((lds_read<KIND, MODE>), dim3(256), dim3(64 * waves), 128 * 1024, 0, (unsigned*)g_out, g_ticks, iters); });
  const double reads = (double)waves * 16 * iters + (MODE ? 8.0 * waves : 0);
  const double ns = ms * 1e6 / reads, ghz = clock_ghz();
  printf("lds %-12s %2d waves/CU  %-22s: %6.2f ns per read instruction per CU = %6.1f GB/s per CU = %5.1f B/clk at the in-kernel clock %.2f GHz\n",
         KIND == 0 ? "ds_read_b128" : "ds_read_b64", waves, MODE ? "8 kept in flight" : "16 then lgkmcnt(0)", ns, (KIND == 0 ? 1024 : 512) / ns,
         (KIND == 0 ? 1024 : 512) / ns / ghz, ghz);
}
template <int SHAPE>
static void run_mfma(const u32x4* src, int waves) {
  const int iters = 20000;
  const float ms = timed([&] { hipLaunchKernelGGL((mfma_rate<SHAPE>), dim3(256), dim3(64 * waves), 0, 0, src, (float*)g_out, g_ticks, iters); });
  const double n = (double)waves * (SHAPE == 0 ? 16 : 8) * iters;  // MFMAs per CU
  const double fl = n * 256 * 32768.0 / (SHAPE == 0 ? 2 : 1);
  const double ghz = clock_ghz();
  printf("mfma %-9s %2d waves/CU: %7.3f ms  %6.0f TFLOP/s  %5.2f cycles per MFMA per SIMD at %.2f GHz\n", SHAPE == 0 ? "16x16x32" : "32x32x16", waves, ms, fl / ms / 1e9,
         ms * 1e6 * ghz / (n / 4), ghz);
}
template <int AHEAD>
static void run_dma(const char* stream, int share, int win_steps, const char* what) {
  const int steps = 4000;
  hipFuncSetAttribute((const void*)dma_rate<AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 57344);
  const float ms = timed([&] { hipLaunchKernelGGL((dma_rate<AHEAD>), dim3(256), dim3(512), 2 * 57344, 0, stream, g_ticks, steps, share, win_steps); });
  const double ghz = clock_ghz();
  printf("dma  %d step(s) in flight, %-44s: %6.0f ns per 56 KiB step = %5.1f GB/s per CU = %5.2f TB/s chip  (%.2f GHz)\n", AHEAD, what, ms * 1e6 / steps,
         57344.0 / (ms * 1e6 / steps), 57344.0 * 256 / (ms * 1e6 / steps) / 1e3, ghz);
}
template <int MI, int NI, int waves>
static void run_pipe() {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)pipe_rate<MI, NI, waves>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const float ms = timed([&] { hipLaunchKernelGGL((pipe_rate<MI, NI, waves>), dim3(256), dim3(64 * waves), 128 * 1024, 0, (float*)g_out, g_ticks, iters); });
  const double n = (double)waves * MI * NI * iters, ghz = clock_ghz();
  printf("pipe wave tile %3d x %3d, %2d waves/CU: %7.3f ms  %6.0f TFLOP/s  %5.2f cycles per MFMA per SIMD at %.2f GHz  (LDS reads %5.1f GB/s per CU)\n", 16 * MI, 16 * NI, waves, ms,
         n * 256 * 16384.0 / ms / 1e9, ms * 1e6 * ghz / (n / 4), ghz, (double)waves * (MI + NI) * iters * 1024 / (ms * 1e6));
}

int main() {
  hipMalloc(&g_ticks, 512 * 8);
  hipMalloc(&g_out, 256 * 1024 * 4);
  std::vector<unsigned> h(4096 * 4);
  unsigned x = 12345;
  for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x & 0xbf80bf80u) | 0x3c003c00u; }  // random bf16 pairs, both signs, magnitude around 1
  u32x4* src; hipMalloc(&src, h.size() * 4); hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int waves : {4, 8, 16}) { run_lds<0, 0>(waves); run_lds<0, 1>(waves); run_lds<1, 0>(waves); run_lds<1, 1>(waves); }
  for (int waves : {4, 8}) { run_mfma<0>(src, waves); run_mfma<1>(src, waves); }
  char* stream; const size_t sbytes = (size_t)256 * 16 * 57344 + (1 << 20);
  hipMalloc(&stream, sbytes); hipMemset(stream, 0x3c, sbytes);
  run_dma<1>(stream, 256, 16, "0.9 MB window shared by all (L2 hits)");
  run_dma<2>(stream, 256, 16, "0.9 MB window shared by all (L2 hits)");
  run_dma<1>(stream, 8, 16, "0.9 MB windows shared by 8 (L2 / MALL)");
  run_dma<2>(stream, 8, 16, "0.9 MB windows shared by 8 (L2 / MALL)");
  run_dma<1>(stream, 1, 16, "0.9 MB window each, 235 MB in all (MALL/HBM)");
  run_dma<2>(stream, 1, 16, "0.9 MB window each, 235 MB in all (MALL/HBM)");
  run_pipe<4, 6, 8>(); run_pipe<4, 6, 4>(); run_pipe<8, 6, 4>(); run_pipe<8, 8, 4>(); run_pipe<4, 4, 8>(); run_pipe<2, 6, 8>(); run_pipe<4, 8, 8>();
  return 0;
}
