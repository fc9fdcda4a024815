// What-if probe of the pre-split attention kernel (csrc/attention_bf16.hip: attn_x3p_fwd_kernel<QB, DBG>): the same kernel with
// one ingredient of its key loop removed at a time, timed on the 4096-token level.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -I dsml_thesis_amd/csrc -I include \
//         tools/probe/attn_x3p_probe.hip -o tools/bin/attn_x3p_probe && tools/bin/attn_x3p_probe
#include "../../dsml_thesis_amd/csrc/attention_bf16.hip"
#include <stdio.h>
#include <vector>

namespace ldmk {
void set_error(const char*, ...) {}
void attn_rowdot_launch(const float*, const float*, float*, int, int, long long, hipStream_t) {}
}

template <int QB, int DBG>
static float run(const float* qkv, unsigned char* kv, float* out, int n, int tokens, int heads) {
  using namespace ldmk;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((attn_x3p_fwd_kernel<QB, DBG>), dim3((tokens + 128 * QB - 1) / (128 * QB), heads, n), dim3(256), 0, 0, qkv, kv, out,
                       (unsigned char*)nullptr, tokens, heads, 0.17677669f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const int n = 16, tokens = 4096, heads = 5, C = heads * 32;
  std::vector<float> h((size_t)n * tokens * 3 * C);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1.5e-3f; }
  float *qkv, *out; unsigned char* kv;
  hipMalloc(&qkv, h.size() * 4); hipMalloc(&out, (size_t)n * tokens * C * 4);
  const long long kvb = ldmk_attn_kv_split_bytes(n, tokens, heads);
  hipMalloc(&kv, kvb);
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(ldmk::attn_kv_split_kernel, dim3((tokens + 63) / 64, heads, n), dim3(256), 0, 0, qkv, kv, tokens, heads);
  hipDeviceSynchronize();
#define ROW(QB, DBG, what) printf("QB=%d  %-58s %8.1f us\n", QB, what, run<QB, DBG>(qkv, kv, out, n, tokens, heads)); fflush(stdout);
  ROW(1, 0, "the kernel")
  ROW(1, 2, "no exponentials")
  ROW(1, 4, "no split of the probabilities")
  ROW(1, 6, "no exponentials, no split")
  ROW(1, 32 + 4, "no softmax, no split (products + LDS reads + DMA only)")
  ROW(1, 16, "no LDS operand reads")
  ROW(1, 32 + 4 + 16, "products + DMA only")
  ROW(1, 1 + 8, "no products (softmax + split + LDS reads + DMA)")
  ROW(1, 1, "no S^T products")
  ROW(1, 8, "no O^T products")
  ROW(2, 0, "the kernel")
  ROW(2, 32 + 4, "no softmax, no split (products + LDS reads + DMA only)")
  ROW(2, 32 + 4 + 16, "products + DMA only")
  ROW(2, 1 + 8, "no products (softmax + split + LDS reads + DMA)")
  return 0;
}
