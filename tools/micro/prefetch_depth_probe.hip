// Micro-probe: is the NT GEMM's K-step bound by the LATENCY of the activation rows' first touch (HBM) behind a one-step-deep
// prefetch?  Wave tile / fragments / MFMAs as the shipped 256 x 192 kernel (pingpong_probe.hip), lock-step, per K-step:
//   A part: 32 KiB (256 rows x 128 B) streamed ONCE from a window shared by the 8 workgroups of an XCD group (first touch = HBM,
//           the seven others hit L2 or merge), ring depth DA (2 = one step ahead, the shipped structure; 3 = two steps ahead)
//   W part: 24 KiB from a small window every workgroup re-reads (L2 hits), ring depth 2
//   hipcc -O3 --offload-arch=gfx950 tools/micro/prefetch_depth_probe.hip -o tools/micro/prefetch_depth_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int MI = 4, NI = 6, ASTG = 256 * 128, WSTG = 192 * 128;
__device__ __forceinline__ void glds16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DA>
__global__ void __launch_bounds__(512, 2) probe(float* __restrict__ out, int steps, const char* __restrict__ a_stream, const char* __restrict__ w_win, int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                 // [DA][ASTG]
  char* sW = smem + DA * ASTG;     // [2][WSTG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  // group of `share` blocks with equal blockIdx % 8 (= one XCD under round-robin placement) streams the same A rows
  const int grp = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * share));
  const char* ap = a_stream + (size_t)grp * steps * ASTG + wave * 4096 + lane * 16;  // 4 x 1 KiB pieces of A per wave and step
  const char* wp = w_win + wave * 3072 + lane * 16;                                  // 3 x 1 KiB pieces of W
  f32x4 acc[MI][NI];
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][MI], bfr[2][NI];
  const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
  int ia = 0, iw = 0, sa_issue = 0;
  auto issue_a = [&]() {
    char* d = sA + sa_issue * ASTG + wave * 4096;
    const char* g = ap + (size_t)ia * ASTG;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(g + q * 1024, d + q * 1024);
    sa_issue = sa_issue == DA - 1 ? 0 : sa_issue + 1; ++ia;
  };
  auto issue_w = [&]() {
    char* d = sW + (iw & 1) * WSTG + wave * 3072;
    const char* g = wp + (size_t)(iw & 7) * WSTG;
#pragma unroll
    for (int q = 0; q < 3; ++q) glds16(g + q * 1024, d + q * 1024);
    ++iw;
  };
  // prologue: A DA-1 steps ahead, W one step ahead
  for (int i = 0; i < DA - 1; ++i) issue_a();
  issue_w();
  int ca = 0;
  for (int s = 0; s < steps; ++s) {
    // outstanding, in issue order (oldest first): [A(s) 4] [W(s) 3] ... ; everything of step s must have landed:
    // DA == 2: nothing younger than step s -> vmcnt(0);  DA == 3: A(s+1) (4 pieces) was issued before W(s) ... order per step is
    // [A(s+DA-1) then W(s+1)], so what is younger than {A(s), W(s)} is A(s+1) for DA == 3: but W(s) was issued AFTER A(s+1) -> vmcnt(0) too
    // unless the order inside a step is W first: issue order per step = W(s+1), then A(s+DA-1): then younger than W(s),A(s) is: A(s+1) [4] -> vmcnt(4)
    if (DA == 3) wait_vm<4>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* a_base = sA + ca * ASTG + (wm * 64 + fr) * 128;
    const char* b_base = sW + (s & 1) * WSTG + (wn * 96 + fr) * 128;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[0][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw0);
    if (s + 1 < steps) issue_w();            // into the W stage read in the previous step
    if (s + DA - 1 < steps) issue_a();       // into the A stage read in the previous step
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[1][mi] = *reinterpret_cast<const bf16x8*>(a_base + mi * 2048 + sw1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bfr[1][ni] = *reinterpret_cast<const bf16x8*>(b_base + ni * 2048 + sw1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0);
    ca = ca == DA - 1 ? 0 : ca + 1;
  }
  float v = 0.f;
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) v += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = v;
}

int main() {
  const int steps = 1000, wgs = 256;
  float* out; char *a_stream, *w_win;
  const size_t a_bytes = (size_t)wgs * steps * ASTG + (1 << 20);   // enough for share = 1
  hipMalloc(&out, wgs * 512 * 4); hipMalloc(&a_stream, a_bytes); hipMalloc(&w_win, 8 * WSTG + (1 << 20));
  hipMemset(a_stream, 0x3c, a_bytes); hipMemset(w_win, 0x3c, 8 * WSTG + (1 << 20));
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ASTG + 2 * WSTG);
  hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ASTG + 2 * WSTG);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (int share : {1, 2, 8})
      for (int da : {2, 3}) {
        hipEventRecord(e0);
        if (da == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(512), 2 * ASTG + 2 * WSTG, 0, out, steps, a_stream, w_win, share);
        else hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(512), 3 * ASTG + 2 * WSTG, 0, out, steps, a_stream, w_win, share);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 256 * 192 * 64 * steps * wgs;
        printf("A rows shared by %d workgroups of an XCD, A ring depth %d: %.0f ns per K-step, %.0f TFLOP/s, HBM read %.2f TB/s\n", share, da,
               ms * 1e6 / steps, fl / ms / 1e9, (double)wgs / share * steps * ASTG / ms / 1e9);
      }
  return 0;
}
