// Diagnostic: the shader clock the chip sustains while every SIMD runs back-to-back fp32 (or bf16) MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o tools/bin/clock_probe ; clock_probe [waves_per_simd] [iters] [bf16|bf16z|f16]
//   (bf16: v_mfma_f32_32x32x16_bf16 on pseudo-random operands; bf16z: the same on all-zero operands -- no data toggling)
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz reference: their ratio over the loop is the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(unsigned long long* out, float* sink, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* d = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    d[0] = t1 - t0; d[1] = r1 - r0;
  }
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__global__ __launch_bounds__(256) void spin_bf16(unsigned long long* out, float* sink, int iters, unsigned mask) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  u32x4 ua, ub;
  for (int i = 0; i < 4; ++i) {
    h = h * 1664525u + 1013904223u; ua[i] = (h & 0x3fff3fffu) & mask;      // bf16 pairs of moderate magnitude
    h = h * 1664525u + 1013904223u; ub[i] = (h & 0x3fff3fffu) & mask;
  }
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  const f16x8 ah = __builtin_bit_cast(f16x8, ua), bh = __builtin_bit_cast(f16x8, ub);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (F16) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* d = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    d[0] = t1 - t0; d[1] = r1 - r0;
  }
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 20000;
  const bool f16 = argc > 3 && argv[3][0] == 'f';
  const bool bf = f16 || (argc > 3 && argv[3][0] == 'b'), bfz = bf && !f16 && argv[3][4] == 'z';
  const int blocks = 256 * wps;           // 4 waves per block: one per SIMD; `wps` blocks per CU
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, (size_t)blocks * 4 * 16); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    if (f16) hipLaunchKernelGGL(spin_bf16<true>, dim3(blocks), dim3(256), 0, 0, out, sink, iters, 0xffffffffu);
    else if (bf) hipLaunchKernelGGL(spin_bf16<false>, dim3(blocks), dim3(256), 0, 0, out, sink, iters, bfz ? 0u : 0xffffffffu);
    else hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cpm;
    for (size_t i = 0; i < h.size() / 2; ++i) {
      mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
      cpm.push_back((double)h[2 * i] / (4.0 * iters));
    }
    std::sort(mhz.begin(), mhz.end()); std::sort(cpm.begin(), cpm.end());
    const double flops = (bf ? 32768.0 : 4096.0) * 4.0 * iters * blocks * 4;
    printf("%d wave(s)/SIMD, %d x 4 MFMAs per wave: %.3f ms = %.1f TFLOP/s; shader clock median %.0f MHz (min %.0f, max %.0f); "
           "clocks per MFMA per wave %.1f\n", wps, iters, ms, flops / ms * 1e-9, mhz[mhz.size() / 2], mhz.front(), mhz.back(),
           cpm[cpm.size() / 2]);
  }
  return 0;
}
