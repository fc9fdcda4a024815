// Probe: semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 (run on the GPU box: hipcc tools/probe/dma_probe.hip && ./a.out)
//  (1) the LDS destination is M0 + inst_offset + lane * 16;  (2) the memory address is base + voffset + soffset + inst_offset;
//  (3) an out-of-range lane writes zeros to its LDS slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out, unsigned bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = 0xDEADBEEFu;
  __syncthreads();
  u32x4 r;
  unsigned long long a = (unsigned long long)src;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  // lanes read in REVERSED order from memory (per-lane source), lane 5 out of range
  unsigned voff = (63 - threadIdx.x) * 16;
  if (threadIdx.x == 5) voff = 0xFFFFFFFFu;
  unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + 2048);
  unsigned keep;
  unsigned soff = __builtin_amdgcn_readfirstlane(4096u);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, %4 offen offset:2048 lds\n\t"
               "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(r), "s"(lds), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = smem[i];
}
int main() {
  const int N = 8192;   // dwords
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, o, (unsigned)(N * 4));
  std::vector<unsigned> r(4096);
  hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 4096; ++i) {
    unsigned want = 0xDEADBEEFu;
    if (i >= 512 && i < 512 + 768) {           // LDS bytes 2048 .. 2048 + 3072
      const int piece = (i - 512) / 256, lane = ((i - 512) % 256) / 4, w = i % 4;
      want = lane == 5 ? 0u : (4096 + piece * 1024 + (63 - lane) * 16) / 4 + w;
    }
    if (r[i] != want) { if (bad < 10) printf("dword %d: got %u want %u\n", i, r[i], want); ++bad; }
  }
  printf("dma_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
