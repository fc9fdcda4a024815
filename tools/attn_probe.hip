// Diagnostic: phases of the flash self-attention key-tile loop (s_memtime stamps, AT_T in csrc/attention.hip).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/attn_probe.hip -o tools/bin/attn_probe ; attn_probe n tokens heads
#define LDMK_AT_STAMPS 1
#include "../dsml_thesis_amd/csrc/attention.hip"
#include <algorithm>
#include <vector>
namespace ldmk { void set_error(const char*, ...) {} }
int main(int argc, char** argv) {
  const int n = atoi(argv[1]), tokens = atoi(argv[2]), heads = atoi(argv[3]);
  const long long rows = (long long)n * tokens, C = heads * 32;
  float *qkv, *out;
  unsigned long long* st;
  hipMalloc(&qkv, rows * 3 * C * 4); hipMalloc(&out, rows * C * 4);
  std::vector<float> h(rows * 3 * C);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const long long waves = (long long)((tokens + 127) / 128) * heads * n * 4;
  hipMalloc(&st, waves * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    ldmk_attn_self_lse(qkv, out, (float*)st, n, tokens, heads, 0.1767767f, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> s(waves * 8);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"load-issue", "barrier-1", "lds-store", "barrier-2", "K-read + S mfma", "softmax", "PV mfma"};
  std::vector<double> ph[7];
  for (long long i = 0; i < waves; ++i)
    for (int q = 0; q < 7; ++q) ph[q].push_back((double)s[i * 8 + q] / (double)(s[i * 8 + 7] & 0xFFFFFFFFull));
  std::vector<double> mhz;
  for (long long i = 0; i < waves; ++i) {
    double ticks = 0;
    for (int q = 0; q < 7; ++q) ticks += (double)s[i * 8 + q];
    mhz.push_back(100.0 * ticks / (double)(s[i * 8 + 7] >> 32));
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  double tot = 0;
  printf("attention n=%d tokens=%d heads=%d: %.1f us (stamped build) = %.1f TFLOP/s; median cycles per 128-key tile per wave:\n", n, tokens,
         heads, best * 1e3, 4.0 * tokens * tokens * 32 * heads * n / best * 1e-9);
  for (int q = 0; q < 7; ++q) { const double m = med(ph[q]); tot += m; printf("   %-16s %7.0f\n", names[q], m); }
  printf("   shader clock in the loop (s_memtime / s_memrealtime): median %.0f MHz\n", med(mhz));
  printf("   total            %7.0f   (128 MFMAs of 64 cycles = 8192 when the pipe is all this wave's)\n", tot);
  return 0;
}
