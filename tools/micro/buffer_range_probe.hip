// Does the range check of a raw buffer descriptor (stride 0) include the scalar offset?  buffer_store_dwordx4 v, voff, rsrc, soff offen with
// num_records = 1024 B: lanes write 16 B at voff = 16 * lane (+ soff).  If soff takes part in the check, nothing lands at or beyond byte 1024 of the
// buffer for any soff; if it does not, soff = 512 writes up to byte 1536.   hipcc -O3 --offload-arch=gfx950 tools/micro/buffer_range_probe.hip -o /tmp/brp && /tmp/brp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void k(char* base, unsigned soff, int nt) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  const i32x4 rs = i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), 1024, 0x00020000};
  const u32x4 v = u32x4{0xabcdabcdu, 0xabcdabcdu, 0xabcdabcdu, 0xabcdabcdu};
  const unsigned voff = threadIdx.x * 16u;
  if (nt) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
int main() {
  char* d; hipMalloc(&d, 4096);
  unsigned char h[4096];
  for (int nt = 0; nt < 2; ++nt)
    for (unsigned soff : {0u, 512u, 1008u, 1024u, 2048u}) {
      hipMemset(d, 0x11, 4096);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, soff, nt);
      hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
      int first = -1, last = -1;
      for (int i = 0; i < 4096; ++i) if (h[i] != 0x11) { if (first < 0) first = i; last = i; }
      printf("nt %d soff %4u: bytes written [%d, %d]  -> %s\n", nt, soff, first, last, last < 1024 ? "scalar offset IS range-checked" : "WRITES PAST num_records");
    }
  return 0;
}
