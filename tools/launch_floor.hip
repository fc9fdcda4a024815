// Diagnostic: what a DEPENDENT phase costs on this chip, as a kernel boundary inside a hipGraph and as a grid barrier inside
// one persistent kernel.  The batch-1 autoregressive DDIM step is a chain of ~400 dependent launches; this decides whether
// to spend the effort on fewer launches, or on grid barriers inside fewer, longer kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/bin/launch_floor ; launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel() {}

// every thread moves `per` float4 from src to dst (+1): a dependent elementwise pass
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int per) {
  size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
  for (int k = 0; k < per; ++k) {
    float4 v = src[i + (size_t)k * 256];
    v.x += 1.f;
    dst[i + (size_t)k * 256] = v;
  }
}

// persistent kernel: `phases` copy passes separated by grid barriers (monotonic counter, bounded spin so that it always exits)
__global__ __launch_bounds__(256) void persistent_kernel(float4* a, float4* b, int per, int phases, unsigned* counter, unsigned* fail) {
  const unsigned nblk = gridDim.x;
  float4* src = a;
  float4* dst = b;
  for (int p = 0; p < phases; ++p) {
    // every block reads what OTHER blocks wrote in the previous phase (rotate the block index)
    const size_t rb = (blockIdx.x + 17 * p) % nblk;
    size_t i = rb * 256 * per + threadIdx.x;
    for (int k = 0; k < per; ++k) {
      float4 v = src[i + (size_t)k * 256];
      v.x += 1.f;
      dst[i + (size_t)k * 256] = v;
    }
    // grid barrier
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = nblk * (unsigned)(p + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 2000000u) { *fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float4* t = src; src = dst; dst = t;
  }
}

static float time_graph(hipGraphExec_t ge, hipStream_t st, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int N = 400;
  const size_t maxf4 = (size_t)2048 * 256 * 8;
  float4 *a, *b; CK(hipMalloc(&a, maxf4 * 16)); CK(hipMalloc(&b, maxf4 * 16));
  CK(hipMemset(a, 0, maxf4 * 16)); CK(hipMemset(b, 0, maxf4 * 16));
  unsigned *counter, *fail; CK(hipMalloc(&counter, 4)); CK(hipMalloc(&fail, 4));
  // 1) chain of empty kernels in a graph
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph of %d empty kernels: %.2f us per kernel\n", N, time_graph(ge, st, 20) * 1e3f / N);
  }
  // 2) chain of dependent copy kernels
  const int grids[] = {32, 256, 1024, 2048};
  const int pers[] = {1, 8};
  for (int gi = 0; gi < 4; ++gi)
    for (int pi = 0; pi < 2; ++pi) {
      const int grid = grids[gi], per = pers[pi];
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, per);
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      printf("graph of %d dependent copy kernels, %4d blocks x %d float4/thread (%7.1f KB): %.2f us per kernel\n", N, grid, per,
             grid * 256.0 * per * 16 / 1024, time_graph(ge, st, 10) * 1e3f / N);
    }
  // 3) eager (stream) launches of the same chain
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, 1);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream of %d dependent copy kernels (256 blocks): %.2f us per kernel\n", N, ms * 1e3f / N);
  }
  // 4) persistent kernel with grid barriers
  for (int gi = 0; gi < 2; ++gi)
    for (int pi = 0; pi < 2; ++pi) {
      const int grid = gi == 0 ? 256 : 512, per = pers[pi], phases = 200;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float ms = 0;
      for (int w = 0; w < 3; ++w) {
        CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(fail, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(256), 0, st, a, b, per, phases, counter, fail);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      unsigned hf = 0; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
      printf("persistent kernel, %d blocks x %d float4/thread, %d phases with grid barriers: %.2f us per phase%s\n", grid, per, phases,
             ms * 1e3f / phases, hf ? "  (SPIN LIMIT HIT)" : "");
    }
  return 0;
}
