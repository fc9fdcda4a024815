// Diagnostic: where a workgroup of the LDS-tiled conv kernel spends its cycles (s_memtime stamps, IG_T in csrc/igemm.hip).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/igemm_probe.hip -o tools/bin/igemm_probe ; igemm_probe n cin cout hw
#ifndef LDMK_IG_STAMPS
#define LDMK_IG_STAMPS 1
#endif
#include "../dsml_thesis_amd/csrc/igemm.hip"
#include <algorithm>
#include <vector>
namespace ldmk {
void set_error(const char*, ...) {}
const char* rgemm_unsupported(const ldmk_igemm_args&, int) { return "n/a"; }
int rgemm_dispatch(const ldmk_igemm_args&, int, hipStream_t) { return -1; }
const char* sgemm_unsupported(const ldmk_igemm_args&, int, int) { return "n/a"; }
int sgemm_dispatch(const ldmk_igemm_args&, int, int, float*, hipStream_t) { return -1; }
const char* igemm_ws_unsupported(const ldmk_igemm_args&, int, int) { return "n/a"; }
int igemm_ws_dispatch(const ldmk_igemm_args&, int, int, float*, hipStream_t) { return -1; }
}
int main(int argc, char** argv) {
  const int n = atoi(argv[1]), cin = atoi(argv[2]), cout = atoi(argv[3]), hw = atoi(argv[4]);
  const int compute = argc > 5 ? atoi(argv[5]) : 0;          // 2: LDMK_COMPUTE_BF16X3
  const long long M = (long long)n * hw * hw, K = 9LL * cin;
  float *x, *w, *out;
  unsigned long long* st;
  hipMalloc(&x, M * cin * 4); hipMalloc(&w, K * cout * 4); hipMalloc(&out, M * cout * 4);
  std::vector<float> h(M * cin);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  h.resize(K * cout);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int tiles = (int)((M + 127) / 128) * ((cout + 159) / 160);
  hipMalloc(&st, (size_t)tiles * 4 * 8 * 8);
  ldmk_igemm_args a = {};
  a.M = (int)M; a.N = cout; a.K = (int)K; a.a0 = x; a.c0 = cin; a.a_mode = LDMK_A_CONV3X3; a.in_h = a.in_w = a.out_h = a.out_w = hw;
  a.stride = 1; a.pad_lo = 1; a.rows_per_sample = hw * hw; a.w = w; a.ldb = cout; a.out = out; a.ldc = cout; a.alpha = 1.f;
  a.tile_cfg = 5; a.splitk = 1; a.splitk_ws = (float*)st; a.splitk_ws_elems = 1;
  if (compute == 2) {
    void* wsplit;
    hipMalloc(&wsplit, 3 * (size_t)cout * K * 2);
    ldmk_pack_wsplit(w, (int)K, cout, cout, 1, 0, wsplit, (int)K, nullptr);
    a.compute = LDMK_COMPUTE_BF16X3; a.w_split = wsplit; a.w_split_ld = (int)K;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 8; ++it) {
    hipEventRecord(e0);
    ldmk_igemm(&a, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> s((size_t)tiles * 4 * 8);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  const char* names[5] = {"barrier-1", "lds-store", "barrier-2", "load-issue", "mfma-block"};
  std::vector<double> ph[5], tot, mhz;
  for (int i = 0; i < tiles * 4; ++i) {
    const double iters = (double)(s[i * 8 + 7] & 0xFFFFFFFFull);
    mhz.push_back(100.0 * (double)(s[i * 8 + 6] - s[i * 8 + 5]) / (double)(s[i * 8 + 7] >> 32));
    for (int q = 0; q < 5; ++q) ph[q].push_back((double)s[i * 8 + q] / iters);
    tot.push_back((double)(s[i * 8 + 6] - s[i * 8 + 5]) / iters);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("conv %d->%d @%dx%d n=%d: tiles %d, %.1f us (stamped build), median cycles per 32-deep slice per wave: total %.0f\n", cin, cout, hw,
         hw, n, tiles, best * 1e3, med(tot));
  for (int q = 0; q < 5; ++q) printf("   %-11s %7.0f\n", names[q], med(ph[q]));
  printf(compute == 2 ? "   (60 bf16 MFMAs of 32 cycles = 1920 when the pipe is all this wave's)\n"
                      : "   (80 MFMAs of 64 cycles = 5120 when the pipe is all this wave's)\n");
  printf("   shader clock in the loop (s_memtime / s_memrealtime): median %.0f MHz\n", med(mhz));
  return 0;
}
