// Micro-probe: how much would a ping-pong K-step (two barriers; waves 0-3 read fragments while waves 4-7 run MFMAs, then swap)
// buy over the shipped lock-step K-step (one barrier; every wave reads, then every wave runs its MFMAs) for the NT GEMM's
// wave tile (64 x 96: 4 + 6 fragments per 32-deep half, 48 MFMAs per 64-deep step)?  No DMA, no epilogue: LDS holds static data.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pingpong_probe.hip -o /tmp/pingpong_probe && /tmp/pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int MI = 4, NI = 6, STAGE = (256 + 192) * 128;

__device__ __forceinline__ void glds16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
// MODE 0 = lock-step, 1 = ping-pong (no DMA); 2 = lock-step + the ring refill (7 x 1 KiB LDS-DMA per wave and step, issued
// after the first fragment reads, waited for at the top of the next step: the shipped 256 x 192 structure); 3 = ping-pong + refill
template <int MODE>
__global__ void __launch_bounds__(512, 2) probe(const uint4* __restrict__ src, float* __restrict__ out, int steps, const char* __restrict__ stream, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * STAGE / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[(i + blockIdx.x * 7) & 4095];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  f32x4 acc[MI][NI];
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][MI], bfr[2][NI];
  const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
#define READ(cs)                                                                                          \
  {                                                                                                       \
    const char* a_base = smem + (cs) * STAGE + (wm * 64 + fr) * 128;                                      \
    const char* b_base = smem + (cs) * STAGE + 256 * 128 + (wn * 96 + fr) * 128;                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw0); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) bfr[0][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw0); \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) af[1][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw1); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) bfr[1][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw1); \
  }
#define MFMA()                                                                                            \
  {                                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                   \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                 \
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0); \
  }
  int cs = 0;
  // refill source: 56 KiB per step from this workgroup's own 896 KiB window (L2-resident after the first pass)
  const char* my = stream + (size_t)(blockIdx.x / share) * (16 * STAGE) + wave * 7168 + lane * 16;  // `share` workgroups stream the same window
#define REFILL(s_, cs_)                                                                                   \
  {                                                                                                       \
    const char* g_ = my + ((s_) & 15) * STAGE;                                                            \
    char* d_ = smem + ((cs_) ^ 1) * STAGE + wave * 7168;                                                  \
    _Pragma("unroll") for (int q = 0; q < 7; ++q) glds16(g_ + q * 1024, d_ + q * 1024);                   \
  }
  if (MODE == 2) {
    for (int s = 0; s < steps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* a_base = smem + cs * STAGE + (wm * 64 + fr) * 128;
      const char* b_base = smem + cs * STAGE + 256 * 128 + (wn * 96 + fr) * 128;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[0][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw0);
      REFILL(s, cs)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[1][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw1);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bfr[1][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw1);
      MFMA()
      cs ^= 1;
    }
  } else if (MODE == 3 && wave < 4) {
    for (int s = 0; s < steps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      READ(cs)
      REFILL(s, cs)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      MFMA()
      __builtin_amdgcn_sched_barrier(0);
      cs ^= 1;
    }
  } else if (MODE == 3) {
    for (int s = 0; s < steps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s > 0) MFMA()
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      READ(cs)
      REFILL(s, cs)
      __builtin_amdgcn_sched_barrier(0);
      cs ^= 1;
    }
    MFMA()
  } else if (MODE == 0) {
    for (int s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      READ(cs)
      MFMA()
      cs ^= 1;
    }
  } else if (wave < 4) {
    for (int s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      READ(cs)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      MFMA()
      __builtin_amdgcn_sched_barrier(0);
      cs ^= 1;
    }
  } else {
    for (int s = 0; s < steps; ++s) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s > 0) MFMA()
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      READ(cs)
      __builtin_amdgcn_sched_barrier(0);
      cs ^= 1;
    }
    MFMA()
  }
  float v = 0.f;
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) v += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = v;
}

int main() {
  const int steps = 2000, wgs = 256;
  std::vector<unsigned> h(4096 * 4);
  unsigned x = 12345;
  for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x & 0x3f803f80u) | 0x3c003c00u; }  // random bf16 pairs around 1
  uint4* src; float* out; char* stream;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, wgs * 512 * 4);
  hipMalloc(&stream, (size_t)wgs * 16 * STAGE + (1 << 20)); hipMemset(stream, 0x3c, (size_t)wgs * 16 * STAGE + (1 << 20));
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int share : {1, 8, 64})
    for (int mode = 0; mode < 4; ++mode) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), 2 * STAGE, 0, src, out, steps, stream, share);
      else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(512), 2 * STAGE, 0, src, out, steps, stream, share);
      else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(512), 2 * STAGE, 0, src, out, steps, stream, share);
      else hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(512), 2 * STAGE, 0, src, out, steps, stream, share);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = 2.0 * 256 * 192 * 64 * steps * wgs;
      const char* names[4] = {"lock-step", "ping-pong", "lock-step + DMA refill", "ping-pong + DMA refill"};
      printf("share %2d  %-24s: %.3f ms, %.0f ns per K-step, %.0f TFLOP/s\n", share, names[mode], ms, ms * 1e6 / steps, fl / ms / 1e9);
    }
  return 0;
}
