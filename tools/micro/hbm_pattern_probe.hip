// Micro-probe (round 3): does the ORDER in which the NT GEMM's DMA stream walks the activation matrix cost HBM bandwidth?
// The GEMM reads a panel of R rows one K-step at a time: 128 bytes of every row (row pitch = 2K bytes), then the next 128 bytes
// of every row one step later -- from DRAM's point of view, R scattered half-lines per step, each page revisited K / 64 times.
// Same bytes, same LDS-DMA pieces (1 KiB per wave instruction, three steps of 24 KiB in flight per workgroup), two orders:
//   mode 0  a piece = 8 rows x 128 B (the GEMM's piece), step s covers columns [128 s, 128 s + 128) of the panel's 192 rows
//   mode 1  a piece = 1 row x 1 KiB contiguous, step s covers rows [24 (s mod 8), +24) x columns [1 KiB (s / 8), +1 KiB)
// 256 workgroups of 8 waves, each panel read by the two workgroups with equal blockIdx / 2 ... on one XCD (as tiles_n = 2),
// the matrix (M x K bf16, 72000 x 1536 = 221 MB) evicted from the Infinity Cache before every launch by a 512 MiB fill.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_pattern_probe.hip -o /tmp/hbm_pattern_probe && /tmp/hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int NT>
__device__ __forceinline__ void dma16(const i32x4& rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int MODE, int DEPTH, int NT, int BAR>
__global__ void __launch_bounds__(512) stream(const char* __restrict__ A, unsigned bytes, int K, int panels, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint64_t a = (uint64_t)(uintptr_t)A;
  const i32x4 rs = i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  const uint32_t rowbytes = (uint32_t)K * 2u;
  const int nk = K / 64;
  // workgroups b and b + 8 share a panel when share == 2 (same XCD under round-robin placement)
  const int G = gridDim.x;
  const int vb = share == 2 ? ((blockIdx.x & 7) | ((blockIdx.x >> 4) << 3)) : (int)blockIdx.x;   // panel stream id
  const int nstream = share == 2 ? G / 2 : G;
  const uint32_t r8 = (uint32_t)lane >> 3;
  const uint32_t voff0 = r8 * rowbytes + (((uint32_t)lane & 7u) << 4);
  const uint32_t voff1 = (uint32_t)lane * 16u;
  int stage = 0;
  for (int p = vb; p < panels; p += nstream) {
    const uint32_t base = (uint32_t)p * 192u * rowbytes;
    for (int s = 0; s < nk; ++s) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int g = wave * 3 + q;   // piece 0..23 of the step
        if (MODE == 0) dma16<NT>(rs, lds0 + (uint32_t)(stage * 24576 + g * 1024), voff0, base + (uint32_t)(g * 8) * rowbytes + (uint32_t)s * 128u);
        else dma16<NT>(rs, lds0 + (uint32_t)(stage * 24576 + g * 1024), voff1, base + (uint32_t)((s & 7) * 24 + g) * rowbytes + (uint32_t)(s >> 3) * 1024u);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DEPTH) : "memory");   // DEPTH steps in flight
      if (BAR) __builtin_amdgcn_s_barrier();
      stage = stage == DEPTH ? 0 : stage + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__global__ void fill(int* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1; }

template <int MODE, int DEPTH, int NT, int BAR>
static void run(const char* A, size_t bytes, int K, int share, int* flush) {
  const int M = 72000, wgs = 256, panels = M / 192;
  float best = 1e9f;
  hipFuncSetAttribute((const void*)stream<MODE, DEPTH, NT, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (DEPTH + 1) * 24576);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, flush, ((size_t)512 << 20) / 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, DEPTH, NT, BAR>), dim3(wgs), dim3(512), (DEPTH + 1) * 24576, 0, A, (unsigned)bytes, K, panels, share);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("K %4d  sharers %d  %-24s depth %d (%3d KiB in flight per CU) %s %s: %7.1f us  HBM %5.2f TB/s\n", K, share, MODE == 0 ? "8 rows x 128 B per piece" : "1 row x 1 KiB per piece", DEPTH,
         DEPTH * 24, NT ? "nt     " : "default", BAR ? "barrier   " : "no barrier", best * 1e3, (double)bytes / best / 1e9);
}

int main() {
  const int M = 72000;
  char* A; int* flush;
  hipMalloc(&flush, (size_t)512 << 20); hipMemset(flush, 0, (size_t)512 << 20);
  for (int K : {384, 1536}) {
    const size_t bytes = (size_t)M * K * 2;
    hipMalloc(&A, bytes); hipMemset(A, 1, bytes);
    run<0, 1, 0, 1>(A, bytes, K, 1, flush); run<0, 2, 0, 1>(A, bytes, K, 1, flush); run<0, 3, 0, 1>(A, bytes, K, 1, flush); run<0, 5, 0, 1>(A, bytes, K, 1, flush);
    run<0, 3, 1, 1>(A, bytes, K, 1, flush); run<0, 3, 0, 0>(A, bytes, K, 1, flush); run<0, 5, 1, 0>(A, bytes, K, 1, flush);
    run<0, 3, 0, 1>(A, bytes, K, 2, flush); run<0, 5, 0, 1>(A, bytes, K, 2, flush);
    if (K == 1536) { run<1, 3, 0, 1>(A, bytes, K, 1, flush); run<1, 5, 1, 0>(A, bytes, K, 1, flush); }
    hipFree(A);
  }
  return 0;
}
