// Diagnostic: where the producer and consumer waves of the warp-specialised bf16x3 tile spend their cycles.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DLDMK_WS_STAMPS tools/ws_probe.hip -o tools/bin/ws_probe ; ws_probe n cin cout hw [conv=1]
#define LDMK_WS_STAMPS 1
#include "../dsml_thesis_amd/csrc/igemm.hip"
#include "../dsml_thesis_amd/csrc/igemm_ws.hip"
#include <algorithm>
#include <vector>
namespace ldmk {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char* rgemm_unsupported(const ldmk_igemm_args&, int) { return "n/a"; }
int rgemm_dispatch(const ldmk_igemm_args&, int, hipStream_t) { return -1; }
const char* sgemm_unsupported(const ldmk_igemm_args&, int, int) { return "n/a"; }
int sgemm_dispatch(const ldmk_igemm_args&, int, int, float*, hipStream_t) { return -1; }
}
int main(int argc, char** argv) {
  const int n = atoi(argv[1]), cin = atoi(argv[2]), cout = atoi(argv[3]), hw = atoi(argv[4]);
  const int conv = argc > 5 ? atoi(argv[5]) : 1;
  const long long M = (long long)n * hw * hw, K = (conv ? 9LL : 1LL) * cin;
  float *x, *w, *out;
  unsigned long long* st;
  hipMalloc(&x, M * cin * 4); hipMalloc(&w, K * cout * 4); hipMalloc(&out, M * cout * 4);
  std::vector<float> h(M * cin);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  h.resize(K * cout);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int tiles = (int)((M + 255) / 256) * ((cout + 159) / 160);
  hipMalloc(&st, (size_t)tiles * 8 * 4 * 8);
  void* wsplit;
  hipMalloc(&wsplit, 3 * (size_t)cout * K * 2);
  ldmk_pack_wsplit(w, (int)K, cout, cout, 1, 0, wsplit, (int)K, nullptr);
  ldmk_igemm_args a = {};
  a.M = (int)M; a.N = cout; a.K = (int)K; a.a0 = x; a.c0 = cin; a.a_mode = conv ? LDMK_A_CONV3X3 : LDMK_A_ROWS;
  a.in_h = a.in_w = a.out_h = a.out_w = hw; a.stride = 1; a.pad_lo = 1; a.rows_per_sample = hw * hw; a.w = w; a.ldb = cout; a.out = out;
  a.ldc = cout; a.alpha = 1.f; a.tile_cfg = 21; a.splitk = 1; a.compute = LDMK_COMPUTE_BF16X3; a.w_split = wsplit; a.w_split_ld = (int)K;
  a.splitk_counters = (int*)st; a.splitk_counters_len = 1 << 30;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 8; ++it) {
    hipEventRecord(e0);
    if (ldmk_igemm(&a, nullptr) != 0) return 1;
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> s((size_t)tiles * 8 * 4);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  const double stages = 2.0 * (K / 32);
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  std::vector<double> ps, pl, pb, pt, cc, cb, ct;
  for (int t = 0; t < tiles; ++t)
    for (int wv = 0; wv < 8; ++wv) {
      const unsigned long long* d = &s[((size_t)t * 8 + wv) * 4];
      if (wv >= 4) { ps.push_back(d[0] / stages); pl.push_back(d[1] / stages); pb.push_back(d[2] / stages); pt.push_back(d[3] / stages); }
      else { cc.push_back(d[0] / stages); cb.push_back(d[2] / stages); ct.push_back(d[3] / stages); }
    }
  printf("%s %d->%d @%dx%d n=%d: %d workgroups, %.1f us (stamped build); median cycles per 16-deep stage per wave\n", conv ? "conv" : "rows", cin,
         cout, hw, hw, n, tiles, best * 1e3);
  printf("   producers: split+store %6.0f   load issue %6.0f   barrier wait %6.0f   total %6.0f\n", med(ps), med(pl), med(pb), med(pt));
  printf("   consumers: products    %6.0f   (60 MFMAs of 32 cycles = 1920)   barrier wait %6.0f   total %6.0f\n", med(cc), med(cb), med(ct));
  return 0;
}
