// ldmk_attn_self_small: self attention (d_head = 32) for SMALL problems -- batch 1-2 of the small-batch route.
//
// attn_self_kernel (attention.hip) gives a workgroup 128 queries of one (sample, head) and walks ALL keys in staged
// 128-key tiles: at batch 1 that is 8 x 5 = 40 workgroups at 32x32 tokens (1024 keys each, 8 dependent
// load -> barrier -> LDS -> barrier -> MFMA rounds) on a chip with 256 CUs: 21 us per call, 16 calls per step.
// Here the keys are split instead:
//   * a workgroup owns ONE 32-query MFMA tile of one (sample, head)                 -> 160 / 80 / 40 workgroups at batch 1;
//   * its NW waves (4 or 8) each take a contiguous 1/NW of the keys and stream them straight from global memory in
//     32-key sub-tiles -- K rows as 16 contiguous floats per lane, V as sixteen coalesced 128-B row reads -- no LDS
//     staging, no barrier inside the loop, the next sub-tile's loads in flight under the current one's MFMAs;
//   * same register algebra as the big kernel: S^T = K Q^T puts a softmax row into one lane (+1 cross-half shuffle),
//     and the P registers are the B operand of O^T += V^T P^T as they are; scores in the log2 domain;
//   * the NW partial (max, sum, O) triples are merged through LDS in wave order (fixed: bitwise reproducible), normalised,
//     transposed and written as 128-B rows by the whole workgroup.
// `qkv` may be the raw split-K slabs of the fused QKV projection (ldmk_igemm raw_slabs): every operand load then sums the
// slabs in slab order, which removes the reduce launch in front of the attention (attention.py:170-193).
#include "ldmk_common.h"

namespace ldmk {

constexpr int AS_D = 32;

template <int NW, bool SLABS>
__global__ __launch_bounds__(64 * NW) void attn_small_kernel(const float* __restrict__ qkv, const int nslab,
                                                             const long long slab_stride, float* __restrict__ out,
                                                             const int tokens, const int heads, const float scale) {
  __shared__ float part_o[NW][16 * 64];          // O^T partials in accumulator order [r][lane]
  __shared__ float part_ml[NW][2][64];
  __shared__ float tr[32 * 36];                   // final [query][d] tile, padded rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int C = heads * AS_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  constexpr float LOG2E = 1.4426950408889634f;

  // Q fragment (B operand of S^T = K Q^T): MFMA step s pairs d = s (lanes 0-31) with d = 16 + s (lanes 32-63)
  float qf[16];
  {
    const bool q_valid = q0 + l31 < tokens;
    const float* qp = base + (long long)(q_valid ? q0 + l31 : 0) * ld + h * AS_D + 16 * half;
    const float qs = q_valid ? scale * LOG2E : 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 q4 = *reinterpret_cast<const float4*>(qp + 4 * s4);
      if constexpr (SLABS) {
        for (int sl = 1; sl < nslab; sl += 2) {
          const float4 t = *reinterpret_cast<const float4*>(qp + (long long)sl * slab_stride + 4 * s4);
          const float4 u = *reinterpret_cast<const float4*>(qp + (long long)min(sl + 1, nslab - 1) * slab_stride + 4 * s4);
          const bool two = sl + 1 < nslab;
          q4.x += t.x; q4.y += t.y; q4.z += t.z; q4.w += t.w;
          q4.x += two ? u.x : 0.f; q4.y += two ? u.y : 0.f; q4.z += two ? u.z : 0.f; q4.w += two ? u.w : 0.f;
        }
      }
      qf[4 * s4] = q4.x * qs; qf[4 * s4 + 1] = q4.y * qs; qf[4 * s4 + 2] = q4.z * qs; qf[4 * s4 + 3] = q4.w * qs;
    }
  }
  // this wave's keys: [kbeg, kend), whole 32-key sub-tiles
  const int per = ((tokens + NW - 1) / NW + 31) & ~31;
  const int kbeg = min(tokens, wave * per), kend = min(tokens, kbeg + per);

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  float kreg[16], vreg[16];
  auto load_kv = [&](int key0) {
    // K: key key0 + l31, d = 16 half .. + 15;  V^T operand of step s: key key0 + (s&3) + 8 (s>>2) + 4 half, d = l31.
    // All 20 loads of a slab are issued before anything is added: one memory round trip per slab, not per load.
    const float* kp = base + (long long)min(key0 + l31, tokens - 1) * ld + C + h * AS_D + 16 * half;
    const float* vp[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
      vp[s] = base + (long long)min(key0 + (s & 3) + 8 * (s >> 2) + 4 * half, tokens - 1) * ld + 2 * C + h * AS_D + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * j);
      kreg[4 * j] = k4.x; kreg[4 * j + 1] = k4.y; kreg[4 * j + 2] = k4.z; kreg[4 * j + 3] = k4.w;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) vreg[s] = *vp[s];
    if constexpr (SLABS) {
      // two more slabs per round trip (clamped addresses, a slab past the end contributes +0)
      for (int sl = 1; sl < nslab; sl += 2) {
        const long long o0 = (long long)sl * slab_stride, o1 = (long long)min(sl + 1, nslab - 1) * slab_stride;
        const bool two = sl + 1 < nslab;
        float4 kt[2][4];
        float vt[2][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kt[0][j] = *reinterpret_cast<const float4*>(kp + o0 + 4 * j);
          kt[1][j] = *reinterpret_cast<const float4*>(kp + o1 + 4 * j);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) { vt[0][s] = vp[s][o0]; vt[1][s] = vp[s][o1]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kreg[4 * j] += kt[0][j].x; kreg[4 * j + 1] += kt[0][j].y; kreg[4 * j + 2] += kt[0][j].z; kreg[4 * j + 3] += kt[0][j].w;
          kreg[4 * j] += two ? kt[1][j].x : 0.f; kreg[4 * j + 1] += two ? kt[1][j].y : 0.f;
          kreg[4 * j + 2] += two ? kt[1][j].z : 0.f; kreg[4 * j + 3] += two ? kt[1][j].w : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) { vreg[s] += vt[0][s]; vreg[s] += two ? vt[1][s] : 0.f; }
      }
    }
  };
  if (kbeg < kend) load_kv(kbeg);
  for (int key0 = kbeg; key0 < kend; key0 += 32) {
    f32x16 s_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], qf[s], s_acc, 0, 0, 0);
    float vcur[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) vcur[s] = vreg[s];
    if (key0 + 32 < kend) load_kv(key0 + 32);                  // next sub-tile in flight under the softmax and the P.V product
    // s_acc[r] = log2(e) * scale * S[query l31][key = key0 + (r&3) + 8 (r>>2) + 4 half]
    if (key0 + 32 > kend) {                                     // ragged last sub-tile: keys past the range get -inf
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= kend) s_acc[r] = -INFINITY;
    }
    float mx = fmaxf(s_acc[0], s_acc[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s_acc[r], s_acc[r + 1]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                        // finite: every sub-tile holds at least one valid key
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s_acc[r] = __builtin_amdgcn_exp2f(s_acc[r] - m_new);
      psum += s_acc[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);    // 0 on the first sub-tile (m_run = -inf)
    l_run = l_run * corr + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[r], s_acc[r], o, 0, 0, 0);
  }
  // ---- merge the NW key ranges (wave order), normalise, transpose, store
#pragma unroll
  for (int r = 0; r < 16; ++r) part_o[wave][r * 64 + lane] = o[r];
  part_ml[wave][0][lane] = m_run;
  part_ml[wave][1][lane] = l_run;
  __syncthreads();
  if (wave == 0) {
    float m = part_ml[0][0][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, part_ml[w][0][lane]);
    float l = 0.f;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = __builtin_amdgcn_exp2f(part_ml[w][0][lane] - m);     // 0 for a wave without keys (its max is -inf)
      l = fmaf(part_ml[w][1][lane], f, l);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaf(part_o[w][r * 64 + lane], f, acc[r]);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[l31 * 36 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[r] * inv;
  }
  __syncthreads();
  for (int i = tid; i < 32 * 8; i += 64 * NW) {
    const int q = i >> 3, d4 = (i & 7) * 4;
    if (q0 + q < tokens)
      *reinterpret_cast<float4*>(out + ((long long)b * tokens + q0 + q) * C + h * AS_D + d4) =
          *reinterpret_cast<const float4*>(tr + q * 36 + d4);
  }
}

}  // namespace ldmk

extern "C" int ldmk_attn_self_small(const float* qkv, int nslab, long long slab_stride, float* out, int n, int tokens,
                                    int heads, float scale, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && out && n > 0 && tokens > 0 && heads > 0 && nslab >= 1, "ldmk_attn_self_small: bad args");
  LDMK_REQUIRE(nslab == 1 || slab_stride >= (long long)n * tokens * 3 * heads * AS_D, "ldmk_attn_self_small: slab_stride smaller than one slab");
  LDMK_REQUIRE(heads <= 65535 && n <= 65535, "ldmk_attn_self_small: grid limits");
  const dim3 grid((tokens + 31) / 32, heads, n);
  hipStream_t st = (hipStream_t)stream;
  if (tokens >= 512) {
    if (nslab > 1) hipLaunchKernelGGL((attn_small_kernel<8, true>), grid, dim3(512), 0, st, qkv, nslab, slab_stride, out, tokens, heads, scale);
    else hipLaunchKernelGGL((attn_small_kernel<8, false>), grid, dim3(512), 0, st, qkv, nslab, slab_stride, out, tokens, heads, scale);
  } else {
    if (nslab > 1) hipLaunchKernelGGL((attn_small_kernel<4, true>), grid, dim3(256), 0, st, qkv, nslab, slab_stride, out, tokens, heads, scale);
    else hipLaunchKernelGGL((attn_small_kernel<4, false>), grid, dim3(256), 0, st, qkv, nslab, slab_stride, out, tokens, heads, scale);
  }
  return check_launch("ldmk_attn_self_small");
}
