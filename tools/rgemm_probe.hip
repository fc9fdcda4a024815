// Diagnostic: where a row-GEMM wave spends its cycles (s_memtime stamps; see RG_STAMP in csrc/rgemm.hip).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/rgemm_probe.hip -o /tmp/rgemm_probe
//   /tmp/rgemm_probe M K N cfg [flags]   (cfg 0..5 = wave tiles 1x5 2x5 1x4 2x4 1x2 1x1; flags: 1 residual (default), 2 GEGLU,
//                                         4 folded LayerNorm, 0 plain bias)
#define LDMK_RG_STAMPS 1
#include "../dsml_thesis_amd/csrc/rgemm.hip"
#include <algorithm>
#include <vector>
namespace ldmk { void set_error(const char*, ...) {} }
int main(int argc, char** argv) {
  int M = atoi(argv[1]), K = atoi(argv[2]), N = atoi(argv[3]), cfg = atoi(argv[4]);
  const int flags = argc > 5 ? atoi(argv[5]) : 1;
  float *x, *w, *wf, *out, *bias, *res;
  unsigned long long* st;
  hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&w, (size_t)K * N * 4); hipMalloc(&wf, (size_t)K * N * 4);
  hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&res, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(x, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  h.resize((size_t)K * N);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(w, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4); hipMemset(res, 0, (size_t)M * N * 4);
  ldmk_pack_wfrag(w, N, K, N, wf, nullptr);
  const ldmk::RTile t = ldmk::kRTiles[cfg];
  const int waves = ((M + 32 * t.tm - 1) / (32 * t.tm)) * (N / 32 / t.tn);
  hipMalloc(&st, (size_t)(waves + 4) * 4 * 8);
  ldmk_igemm_args a = {};
  a.M = M; a.N = N; a.K = K; a.a0 = x; a.c0 = K; a.rows_per_sample = M; a.w = w; a.ldb = N; a.bias = bias; a.residual = res;
  a.out = out; a.ldc = N; a.alpha = 1.f; a.w_frag = wf; a.splitk_ws = (float*)st;
  if (!(flags & 1)) a.residual = nullptr;
  if (flags & 2) { a.epi = LDMK_EPI_GEGLU; a.ldc = N / 2; a.residual = nullptr; }
  if (flags & 4) {
    float* rs; hipMalloc(&rs, (size_t)M * 8); hipMemset(rs, 0, (size_t)M * 8);
    a.a_tf = LDMK_TF_LAYERNORM_FOLDED; a.row_stats = rs; a.ln_colsum = bias;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 10; ++it) {
    hipEventRecord(e0);
    ldmk::rgemm_dispatch(a, cfg, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  std::vector<unsigned long long> s((size_t)waves * 4);
  hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> d[3], start, end;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int i = 0; i < waves; ++i) {
    for (int k = 0; k < 3; ++k) d[k].push_back((double)(s[i * 4 + k + 1] - s[i * 4 + k]));
    t0 = std::min(t0, s[i * 4]); t1 = std::max(t1, s[i * 4 + 3]);
  }
  for (int i = 0; i < waves; ++i) start.push_back((double)(s[i * 4] - t0));
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
  printf("M=%d K=%d N=%d tile %dx%d waves=%d  best %.1f us = %.1f TFLOP/s\n", M, K, N, t.tm, t.tn, waves, best * 1e3,
         2.0 * M * N * K / best * 1e-9);
  printf("  ticks (100 MHz s_memtime? or shader clock): prologue med %.0f  loop med %.0f (max %.0f)  epilogue med %.0f ; span %.0f ; last wave start %.0f\n",
         med(d[0]), med(d[1]), mx(d[1]), med(d[2]), (double)(t1 - t0), mx(start));
  const int mf = (K / 8) * 4 * t.tm * t.tn;
  printf("  MFMAs per wave %d -> loop ticks per MFMA %.1f\n", mf, med(d[1]) / mf);
  return 0;
}
