// Microbenchmark: fp32 GEMM emulated on the bf16 matrix pipe by 3-way operand splitting (x = h + m + l, each bf16).
//   hipcc --offload-arch=gfx950 -O3 tools/splitgemm_bench.hip -o tools/bin/splitgemm_bench
//   tools/bin/splitgemm_bench M N K
// Reports time / effective TFLOP/s (2MNK) and the error against an fp64 product for 1, 3, 6 and 9 bf16 products
// per fp32 multiply, next to the error of an ordinary fp32 dot product of the same data.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline int swz(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }

// NS splits staged; NP products issued (1: hh; 3: +hm,mh; 6: +mm,hl,lh; 9: all)
template <int NS, int NP>
__global__ __launch_bounds__(256, 2) void splitgemm(const float* __restrict__ A, const uint4* __restrict__ Wp,
                                                    const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char As[NS][128 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[NS][128 * 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, r32 = lane & 31, kh = lane >> 5;
  const int ntn = N / 128;
  const int m0 = (blockIdx.x / ntn) * 128, n0 = (blockIdx.x % ntn) * 128;
  const int KS = K / 32;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  float4 ra[2][2];
  uint4 rb[NS][2];
  auto gload = [&](int ks) {
    for (int i = 0; i < 2; i++) {
      int idx = t + 256 * i, row = idx >> 2, kslot = idx & 3;
      const float4* p = reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + ks * 32 + kslot * 8);
      ra[i][0] = p[0]; ra[i][1] = p[1];
    }
    for (int s = 0; s < NS; s++)
      for (int i = 0; i < 2; i++) {
        int idx = t + 256 * i;
        rb[s][i] = Wp[((size_t)(s * KS + ks) * N + n0) * 4 + idx];
      }
  };
  auto lstore = [&]() {
    for (int i = 0; i < 2; i++) {
      int idx = t + 256 * i, row = idx >> 2, kslot = idx & 3;
      float x[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w, ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
      for (int s = 0; s < NS; s++) {
        bf16x8 v;
        for (int e = 0; e < 8; e++) { v[e] = (__bf16)x[e]; x[e] -= (float)v[e]; }
        *reinterpret_cast<bf16x8*>(&As[s][swz(row, kslot)]) = v;
      }
    }
    for (int s = 0; s < NS; s++)
      for (int i = 0; i < 2; i++) {
        int idx = t + 256 * i, n = idx >> 2, kslot = idx & 3;
        *reinterpret_cast<uint4*>(&Bs[s][swz(n, kslot)]) = rb[s][i];
      }
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int ks = 0; ks < KS; ks++) {
    if (ks + 1 < KS) gload(ks + 1);
    for (int kk = 0; kk < 2; kk++) {
      bf16x8 a[2][NS], b[2][NS];
      const int kslot = kk * 2 + kh;
      for (int i = 0; i < 2; i++)
        for (int s = 0; s < NS; s++) {
          a[i][s] = *reinterpret_cast<const bf16x8*>(&As[s][swz(wm * 64 + i * 32 + r32, kslot)]);
          b[i][s] = *reinterpret_cast<const bf16x8*>(&Bs[s][swz(wn * 64 + i * 32 + r32, kslot)]);
        }
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
          f32x16 c = acc[i][j];
          if (NP >= 9) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][2], c, 0, 0, 0);
          }
          if (NP >= 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          }
          if (NP >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __syncthreads();
    if (ks + 1 < KS) {
      lstore();
      __syncthreads();
    }
  }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + r32;
      const float bv = bias[col];
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        C[(size_t)row * N + col] = acc[i][j][r] + bv;
      }
    }
}


// Both operands pre-split into bf16 planes in memory (what producer kernels would write): A planes [3][M][K] bf16,
// W planes [3][K/32][N][32] bf16.  Staging is a pure 16-byte copy; WM = row groups of 64 per workgroup (BM = 64*WM).
template <int NP, int WMG>
__global__ __launch_bounds__(128 * WMG, 1) void splitgemm_pre(const uint4* __restrict__ Ap, const uint4* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 64 * WMG, NT = 128 * WMG;
  __shared__ __attribute__((aligned(16))) unsigned char As[3][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[3][128 * 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, r32 = lane & 31, kh = lane >> 5;
  const int ntn = N / 128;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * 128;
  const int KS = K / 32;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  constexpr int NA = BM * 4 / NT;      // uint4 per thread per plane (A): BM rows x 4 kslots
  constexpr int NB = 128 * 4 / NT;
  uint4 ra[3][NA], rb[3][NB];
  auto gload = [&](int ks) {
    for (int s = 0; s < 3; s++) {
      for (int i = 0; i < NA; i++) {
        int idx = t + NT * i, row = idx >> 2, kslot = idx & 3;
        ra[s][i] = Ap[((size_t)s * M + m0 + row) * (K / 8) + ks * 4 + kslot];
      }
      for (int i = 0; i < NB; i++) {
        int idx = t + NT * i;
        rb[s][i] = Wp[((size_t)(s * KS + ks) * N + n0) * 4 + idx];
      }
    }
  };
  auto lstore = [&]() {
    for (int s = 0; s < 3; s++) {
      for (int i = 0; i < NA; i++) { int idx = t + NT * i; *reinterpret_cast<uint4*>(&As[s][swz(idx >> 2, idx & 3)]) = ra[s][i]; }
      for (int i = 0; i < NB; i++) { int idx = t + NT * i; *reinterpret_cast<uint4*>(&Bs[s][swz(idx >> 2, idx & 3)]) = rb[s][i]; }
    }
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int ks = 0; ks < KS; ks++) {
    if (ks + 1 < KS) gload(ks + 1);
    for (int kk = 0; kk < 2; kk++) {
      bf16x8 a[2][3], b[2][3];
      const int kslot = kk * 2 + kh;
      for (int i = 0; i < 2; i++)
        for (int s = 0; s < 3; s++) {
          a[i][s] = *reinterpret_cast<const bf16x8*>(&As[s][swz(wm * 64 + i * 32 + r32, kslot)]);
          b[i][s] = *reinterpret_cast<const bf16x8*>(&Bs[s][swz(wn * 64 + i * 32 + r32, kslot)]);
        }
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
          f32x16 c = acc[i][j];
          if (NP >= 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          }
          if (NP >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    __syncthreads();
    if (ks + 1 < KS) {
      lstore();
      __syncthreads();
    }
  }
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + r32;
      const float bv = bias[col];
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        C[(size_t)row * N + col] = acc[i][j][r] + bv;
      }
    }
}

static unsigned short bf16_rn(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <typename F>
static void run_any(const char* name, F launch, float* dC, int M, int N, int K,
                    const std::vector<float>& A, const std::vector<float>& W, const std::vector<float>& bias) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) launch();
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  std::vector<float> C((size_t)M * N);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, erms = 0, fmax = 0, frms = 0; int cnt = 0;
  srand(7);
  for (int sidx = 0; sidx < 2000; sidx++) {
    int r = rand() % M, c = rand() % N;
    double ref = bias[c], mag = fabs(bias[c]); float f32 = 0.f;
    for (int k = 0; k < K; k++) { double p = (double)A[(size_t)r * K + k] * W[(size_t)k * N + c]; ref += p; mag += fabs(p); f32 = fmaf(A[(size_t)r * K + k], W[(size_t)k * N + c], f32); }
    f32 += bias[c];
    double e = fabs(C[(size_t)r * N + c] - ref) / mag, f = fabs(f32 - ref) / mag;
    emax = fmax > 0 || true ? (e > emax ? e : emax) : emax; erms += e * e; fmax = f > fmax ? f : fmax; frms += f * f; cnt++;
  }
  printf("%-10s M=%d N=%d K=%d  %8.1f us  %7.1f TFLOP/s  err/sum|ab|: max %.3e rms %.3e   (fp32 fma chain: max %.3e rms %.3e)\n", name, M, N, K,
         ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, emax, sqrt(erms / cnt), fmax, sqrt(frms / cnt));
}

template <int NS, int NP>
static void run(const char* name, const float* dA, const uint4* dW, const float* dB, float* dC, int M, int N, int K,
                const std::vector<float>& A, const std::vector<float>& W, const std::vector<float>& bias) {
  dim3 grid((M / 128) * (N / 128));
  run_any(name, [&]() { splitgemm<NS, NP><<<grid, 256>>>(dA, dW, dB, dC, M, N, K); }, dC, M, N, K, A, W, bias);
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 640, K = argc > 3 ? atoi(argv[3]) : 2880;
  if (M % 128 || N % 128 || K % 32) { printf("M,N multiples of 128, K of 32\n"); return 1; }
  std::vector<float> A((size_t)M * K), W((size_t)K * N), bias(N);
  srand(1);
  auto rnd = []() { float u = 0; for (int i = 0; i < 4; i++) u += (float)rand() / RAND_MAX - 0.5f; return u; };
  for (auto& v : A) v = rnd() * 2.f;
  for (auto& v : W) v = rnd() * 0.05f;
  for (auto& v : bias) v = rnd();
  const int KS = K / 32;
  std::vector<unsigned short> Wp((size_t)3 * K * N);
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) {
      float x = W[(size_t)k * N + n];
      for (int s = 0; s < 3; s++) {
        unsigned short h = bf16_rn(x); x -= bf16_f(h);
        Wp[(((size_t)s * KS + k / 32) * N + n) * 32 + (k % 32)] = h;
      }
    }
  float *dA, *dB, *dC; uint4* dW;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dW, Wp.size() * 2)); CK(hipMalloc(&dB, N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
  run<1, 1>("bf16x1", dA, dW, dB, dC, M, N, K, A, W, bias);
  run<3, 1>("s3p1", dA, dW, dB, dC, M, N, K, A, W, bias);
  run<3, 3>("s3p3", dA, dW, dB, dC, M, N, K, A, W, bias);
  run<2, 3>("bf16x3", dA, dW, dB, dC, M, N, K, A, W, bias);
  run<3, 6>("bf16x6", dA, dW, dB, dC, M, N, K, A, W, bias);
  run<3, 9>("bf16x9", dA, dW, dB, dC, M, N, K, A, W, bias);
  // operands pre-split in memory (A planes [3][M][K] bf16)
  std::vector<unsigned short> Apl((size_t)3 * M * K);
  for (size_t i = 0; i < (size_t)M * K; i++) {
    float x = A[i];
    for (int sp = 0; sp < 3; sp++) { unsigned short h = bf16_rn(x); x -= bf16_f(h); Apl[(size_t)sp * M * K + i] = h; }
  }
  uint4* dAp;
  CK(hipMalloc(&dAp, Apl.size() * 2));
  CK(hipMemcpy(dAp, Apl.data(), Apl.size() * 2, hipMemcpyHostToDevice));
  run_any("pre x6 128x128", [&]() { splitgemm_pre<6, 2><<<dim3((M / 128) * (N / 128)), 256>>>(dAp, dW, dB, dC, M, N, K); }, dC, M, N, K, A, W, bias);
  if (M % 256 == 0)
    run_any("pre x6 256x128", [&]() { splitgemm_pre<6, 4><<<dim3((M / 256) * (N / 128)), 512>>>(dAp, dW, dB, dC, M, N, K); }, dC, M, N, K, A, W, bias);
  run_any("pre x1 128x128", [&]() { splitgemm_pre<1, 2><<<dim3((M / 128) * (N / 128)), 256>>>(dAp, dW, dB, dC, M, N, K); }, dC, M, N, K, A, W, bias);
  return 0;
}
