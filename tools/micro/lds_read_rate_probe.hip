// Micro-probe: how long does the LDS pipe of a CU take per read instruction, for the three read kinds the GEMM kernels use?
//   ds_read_b128 (NT kernel fragments), ds_read_b64 (plain), ds_read_b64_tr_b16 (wgrad: transposed fragments)
// Eight waves (one workgroup per CU, as the GEMMs run) issue R independent reads each per iteration from a conflict-free address
// pattern and consume them with one cheap VALU op; cycles per read instruction = s_memtime ticks / (8 waves x R x iterations).
// The wgrad K-step issues 288 transposed reads per 64-row step; at the rate printed here that alone is the step time.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_read_rate_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int R = 16;

template <int KIND>
__global__ void __launch_bounds__(512, 2) probe(unsigned* __restrict__ out, long long* __restrict__ ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 144 * 1024 / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  unsigned acc = 0;
  // addresses: the wgrad image (384-byte rows, 32-byte granule swizzle) for the transposed reads; lane-linear 16 / 8 bytes otherwise
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3;
  const int tr_off = (4 * g + q4) * 384 + (((wave & 7) ^ (((g & 1) << 1) | (q4 >> 1))) * 32) + p * 8;
  const long long t0 = __builtin_readcyclecounter();
  // the reads are inline asm so that exactly R instructions of the probed kind are issued per iteration (hipcc merges plain
  // 8-byte loads into ds_read2st64_b64 and folds address arithmetic); one lgkmcnt(0) per iteration
  for (int it = 0; it < iters; ++it) {
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + ((it & 3) * 24576);
    u32x4 v4[R];
    u32x2 v2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v4[r]) : "v"(base + r * 1024 + lane * 16 + wave * 64));
      else if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(v2[r]) : "v"(base + r * 512 + lane * 8 + wave * 64));
      else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v2[r]) : "v"(base + tr_off + (r & 1) * 16 * 384 + (r >> 1) * 3072));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) acc ^= KIND == 0 ? (v4[r][0] ^ v4[r][3]) : (v2[r][0] ^ v2[r][1]);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int bytes_per_lane) {
  unsigned* out; long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);
  hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<KIND><<<256, 512, 144 * 1024>>>(out, ticks, iters);
  hipEventRecord(a); probe<KIND><<<256, 512, 144 * 1024>>>(out, ticks, iters); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  const double reads = 8.0 * R * iters;
  printf("%-22s %7.2f s_memtime ticks / read instruction / CU   (%5.1f B/tick/CU;  kernel %.1f us -> %.2f ns per read instruction per CU)\n", name,
         avg / reads, 64.0 * bytes_per_lane * reads / avg, ms * 1e3, ms * 1e6 / reads);
  hipFree(out); hipFree(ticks);
}

int main() {
  run<0>("ds_read_b128", 16);
  run<1>("ds_read_b64", 8);
  run<2>("ds_read_b64_tr_b16", 8);
  return 0;
}
