// Backward of the d_head = 32 self attention (attention.py:178-192) on the f32 matrix cores, flash style: the score
// matrix is recomputed tile by tile from Q, K and the forward's per-row log-sum-exp, never stored.
//   P = exp(scale*QK^T - L),  dV = P^T dO,  dP = dO V^T,  dS = P o (dP - D),  D = rowsum(dO o O),
//   dQ = scale * dS K,  dK = scale * dS^T Q.
// Two kernels, both deterministic (no atomics):
//   attn_bwd_dq_kernel : a wave owns 32 queries and walks the keys   -> dQ
//   attn_bwd_dkv_kernel: a wave owns 32 keys    and walks the queries -> dK, dV
// Same operand trick as the forward kernel: a 32x32 score tile comes out of the MFMA with (column = the wave's own
// index on the lane, 16 registers x 2 half-waves = the other index), which is exactly the k-pair layout of the B
// operand of the next product, so P and dS feed the following MFMAs straight from registers.
#include "ldmk_common.h"

namespace ldmk {

constexpr int AB_D = 32;
constexpr int AB_T = 64;          // rows per staged tile
constexpr int AB_STR = AB_D + 1;  // padded rows: one LDS image serves the row-wise and the column-wise operand reads

// D[b][h][q] = sum_d dO[q][h][d] * O[q][h][d]
__global__ void attn_rowdot_kernel(const float* __restrict__ dout, const float* __restrict__ out, float* __restrict__ dsum,
                                   int tokens, int heads, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long row = i / heads;
  const int h = (int)(i - row * heads);
  const long long b = row / tokens;
  const int q = (int)(row - b * tokens);
  const float4* a = reinterpret_cast<const float4*>(dout + (row * heads + h) * AB_D);
  const float4* o = reinterpret_cast<const float4*>(out + (row * heads + h) * AB_D);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 x = a[j], y = o[j];
    s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
  }
  dsum[(b * heads + h) * tokens + q] = s;
}

void attn_rowdot_launch(const float* dout, const float* out, float* dsum, int tokens, int heads, long long total, hipStream_t st) {
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dout, out, dsum, tokens, heads, total);
}

// stage 64 rows x 32 floats of a [rows][ld] matrix (row r0.., column offset coff) into a stride-33 LDS image
__device__ __forceinline__ void stage_tile(float* dst, const float* __restrict__ src, long long ld, int r0, int rows, int tid) {
  const int rr = tid >> 3, d4 = (tid & 7) * 4;
  float4 v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + rr + 32 * i;
    v[i] = r < rows ? *reinterpret_cast<const float4*>(src + (long long)r * ld + d4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float* d = dst + (rr + 32 * i) * AB_STR + d4;
    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
  }
}

// write a wave's 32x32 accumulator held as acc[r] = X^T[d = (r&3)+8*(r>>2)+4*half][row = l31] as rows of 128 B
__device__ __forceinline__ void store_rows(float* ts, const f32x16& acc, float mul, float* __restrict__ dst, long long ld,
                                           int row0, int rows, int l31, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) ts[l31 * AB_STR + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[r] * mul;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 32; q += 2)
    if (row0 + q + half < rows) dst[(long long)(row0 + q + half) * ld + l31] = ts[(q + half) * AB_STR + l31];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          float* __restrict__ dqkv, int tokens, int heads, float scale) {
  __shared__ float Ks[AB_T * AB_STR];
  __shared__ float Vs[AB_T * AB_STR];
  __shared__ float Ts[4][32 * AB_STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * AB_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  const bool q_valid = q0 + l31 < tokens;
  const int qq = q_valid ? q0 + l31 : 0;
  float qf[16], dof[16];
  {
    const float* qp = base + (long long)qq * ld + h * AB_D;
    const float* dp = dout + ((long long)b * tokens + qq) * C + h * AB_D;
    const float qs = q_valid ? scale : 0.f, ds_ = q_valid ? 1.f : 0.f;     // clamped row pointers: unconditional loads
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      qf[s] = qp[2 * s + half] * qs;
      dof[s] = dp[2 * s + half] * ds_;
    }
  }
  const float Lq = q_valid ? lse[((long long)b * heads + h) * tokens + qq] : INFINITY;
  const float Dq = q_valid ? dsum[((long long)b * heads + h) * tokens + qq] : 0.f;
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;

  const int ntiles = (tokens + AB_T - 1) / AB_T;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage_tile(Ks, base + C + h * AB_D, ld, kt * AB_T, tokens, tid);
    stage_tile(Vs, base + 2 * C + h * AB_D, ld, kt * AB_T, tokens, tid);
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * AB_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
      const float* kb = Ks + (sub * 32 + l31) * AB_STR + half;
      const float* vb = Vs + (sub * 32 + l31) * AB_STR + half;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[2 * s], qf[s], sa, 0, 0, 0);     // S^T[key][q]
        da = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[2 * s], dof[s], da, 0, 0, 0);    // dP^T[key][q]
      }
      const bool ragged = key0 + 32 > tokens;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = __expf(sa[r] - Lq);
      if (ragged) {                    // last sub-tile of a ragged sequence only (kept a real branch, see attention.hip)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) sa[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] *= da[r] - Dq;                                 // dS^T[key][q]
      const float* kc = Ks + (sub * 32 + 4 * half) * AB_STR + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[((r & 3) + 8 * (r >> 2)) * AB_STR], sa[r], dq, 0, 0, 0);   // dQ^T[d][q]
    }
  }
  if (!wave_active) return;
  store_rows(Ts[wave], dq, scale, dqkv + (long long)b * tokens * ld + h * AB_D, ld, q0, tokens, l31, half);
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           float* __restrict__ dqkv, int tokens, int heads, float scale) {
  __shared__ float Qs[AB_T * AB_STR];
  __shared__ float Os[AB_T * AB_STR];       // dO tile
  __shared__ float Ls[AB_T], Ds[AB_T];
  __shared__ float Ts[4][32 * AB_STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * AB_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const float* dbase = dout + (long long)b * tokens * C;
  const bool wave_active = k0 < tokens;
  const bool k_valid = k0 + l31 < tokens;
  const int kk = k_valid ? k0 + l31 : 0;
  float kf[16], vf[16];
  {
    const float* kp = base + (long long)kk * ld + C + h * AB_D;
    const float ks_ = k_valid ? scale : 0.f, vs_ = k_valid ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      kf[s] = kp[2 * s + half] * ks_;
      vf[s] = kp[C + 2 * s + half] * vs_;
    }
  }
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  const float* lrow = lse + ((long long)b * heads + h) * tokens;
  const float* drow = dsum + ((long long)b * heads + h) * tokens;

  const int ntiles = (tokens + AB_T - 1) / AB_T;
  for (int qt = 0; qt < ntiles; ++qt) {
    __syncthreads();
    stage_tile(Qs, base + h * AB_D, ld, qt * AB_T, tokens, tid);
    stage_tile(Os, dbase + h * AB_D, C, qt * AB_T, tokens, tid);
    if (tid < AB_T) {
      const int q = qt * AB_T + tid;
      Ls[tid] = q < tokens ? lrow[q] : INFINITY;       // exp(s - inf) = 0: rows past the end contribute nothing
      Ds[tid] = q < tokens ? drow[q] : 0.f;
    }
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int qbase = qt * AB_T + sub * 32;
      if (qbase >= tokens) break;
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
      const float* qb = Qs + (sub * 32 + l31) * AB_STR + half;
      const float* ob = Os + (sub * 32 + l31) * AB_STR + half;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[2 * s], kf[s], sa, 0, 0, 0);     // S[q][key]
        da = __builtin_amdgcn_mfma_f32_32x32x2f32(ob[2 * s], vf[s], da, 0, 0, 0);     // dP[q][key]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float p = __expf(sa[r] - Ls[ql]);
        da[r] = p * (da[r] - Ds[ql]);      // dS[q][key]
        sa[r] = p;                         // P[q][key]
      }
      const float* oc = Os + (sub * 32 + 4 * half) * AB_STR + l31;
      const float* qc = Qs + (sub * 32 + 4 * half) * AB_STR + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ((r & 3) + 8 * (r >> 2)) * AB_STR;
        dv = __builtin_amdgcn_mfma_f32_32x32x2f32(oc[o], sa[r], dv, 0, 0, 0);          // dV^T[d][key] += dO^T P
        dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[o], da[r], dk, 0, 0, 0);          // dK^T[d][key] += Q^T dS
      }
    }
  }
  if (!wave_active) return;
  float* obase = dqkv + (long long)b * tokens * ld + h * AB_D;
  store_rows(Ts[wave], dk, scale, obase + C, ld, k0, tokens, l31, half);
  store_rows(Ts[wave], dv, 1.0f, obase + 2 * C, ld, k0, tokens, l31, half);
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the short-context cross attention (ldmk_attn_cross, L <= 128 keys; attention.py:170-193 with a context).
// Pass 1, one thread per (row, head): recompute p over the L keys, dP_j = dO.v_j, dS_j = p_j (dP_j - sum_i p_i dP_i)
// * scale, dQ = sum_j dS_j k_j; p and dS are kept ([rows][heads][L]) for pass 2.
__global__ __launch_bounds__(256) void attn_cross_bwd_q_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                               const float* __restrict__ v, int ldkv,
                                                               const float* __restrict__ dout, int ldo, float* __restrict__ dq,
                                                               float* __restrict__ pbuf, float* __restrict__ dsbuf, int tokens,
                                                               int L, int heads, float scale, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (sample, token, head)
  if (idx >= total) return;
  const int h = (int)(idx % heads);
  const long long row = idx / heads;
  const int b = (int)(row / tokens);
  float qv[AB_D], dov[AB_D], dqv[AB_D];
  const float* qp = q + row * ldq + h * AB_D;
  const float* dp_ = dout + row * ldo + h * AB_D;
#pragma unroll
  for (int d = 0; d < AB_D; ++d) { qv[d] = qp[d]; dov[d] = dp_[d]; dqv[d] = 0.f; }
  const float* kb = k + (long long)b * L * ldkv + h * AB_D;
  const float* vb = v + (long long)b * L * ldkv + h * AB_D;
  float* pr = pbuf + idx * L;
  float* dsr = dsbuf + idx * L;
  float m = -INFINITY;
  for (int j = 0; j < L; ++j) {                        // scores (kept in pbuf), running max
    const float* kp = kb + (long long)j * ldkv;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < AB_D; ++d) s = fmaf(qv[d], kp[d], s);
    s *= scale;
    pr[j] = s;
    m = fmaxf(m, s);
  }
  float l = 0.f;
  for (int j = 0; j < L; ++j) { const float e = __expf(pr[j] - m); pr[j] = e; l += e; }
  const float inv = 1.0f / l;
  float dsum = 0.f;
  for (int j = 0; j < L; ++j) {                        // p, dP (kept in dsbuf), D = sum p dP
    const float* vp = vb + (long long)j * ldkv;
    float dpj = 0.f;
#pragma unroll
    for (int d = 0; d < AB_D; ++d) dpj = fmaf(dov[d], vp[d], dpj);
    const float pj = pr[j] * inv;
    pr[j] = pj;
    dsr[j] = dpj;
    dsum = fmaf(pj, dpj, dsum);
  }
  for (int j = 0; j < L; ++j) {
    const float ds = pr[j] * (dsr[j] - dsum) * scale;
    dsr[j] = ds;
    const float* kp = kb + (long long)j * ldkv;
#pragma unroll
    for (int d = 0; d < AB_D; ++d) dqv[d] = fmaf(ds, kp[d], dqv[d]);
  }
  float* dqp = dq + row * ldq + h * AB_D;
#pragma unroll
  for (int d = 0; d < AB_D; ++d) dqp[d] = dqv[d];
}

// Pass 2, one wave per (sample, key, head): dK[j] = sum_q dS[q][j] q[q], dV[j] = sum_q p[q][j] dO[q]; lanes split the
// queries, d = 32 values per lane folded with a wave reduction (fixed order).
__global__ __launch_bounds__(256) void attn_cross_bwd_kv_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ dout,
                                                                int ldo, const float* __restrict__ pbuf,
                                                                const float* __restrict__ dsbuf, float* __restrict__ dk,
                                                                float* __restrict__ dv, int ldkv, int tokens, int L, int heads,
                                                                long long total) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (sample, key, head)
  if (w >= total) return;
  const int lane = threadIdx.x & 63;
  const int h = (int)(w % heads);
  const long long r = w / heads;
  const int j = (int)(r % L), b = (int)(r / L);
  float ak[AB_D], av[AB_D];
#pragma unroll
  for (int d = 0; d < AB_D; ++d) { ak[d] = 0.f; av[d] = 0.f; }
  for (int t = lane; t < tokens; t += 64) {
    const long long row = (long long)b * tokens + t;
    const float ds = dsbuf[(row * heads + h) * L + j], pj = pbuf[(row * heads + h) * L + j];
    const float* qp = q + row * ldq + h * AB_D;
    const float* dp_ = dout + row * ldo + h * AB_D;
#pragma unroll
    for (int d = 0; d < AB_D; ++d) { ak[d] = fmaf(ds, qp[d], ak[d]); av[d] = fmaf(pj, dp_[d], av[d]); }
  }
  float* dkp = dk + ((long long)b * L + j) * ldkv + h * AB_D;
  float* dvp = dv + ((long long)b * L + j) * ldkv + h * AB_D;
#pragma unroll
  for (int d = 0; d < AB_D; ++d) {
    const float sk = wave_sum(ak[d]), sv = wave_sum(av[d]);
    if (lane == 0) { dkp[d] = sk; dvp[d] = sv; }
  }
}

}  // namespace ldmk

extern "C" int ldmk_attn_cross_bwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* dout, int ldo,
                                   float* dq, float* dk, float* dv, float* scratch, int n, int tokens, int ctx_len, int heads,
                                   float scale, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(q && k && v && dout && dq && dk && dv && scratch, "ldmk_attn_cross_bwd: null buffer");
  LDMK_REQUIRE(n > 0 && tokens > 0 && heads > 0 && ctx_len >= 1 && ctx_len <= 128, "ldmk_attn_cross_bwd: bad shape (ctx_len in [1,128])");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)n * tokens * heads;
  float* pbuf = scratch;                              // [n*tokens][heads][L]
  float* dsbuf = scratch + total * ctx_len;
  hipLaunchKernelGGL(attn_cross_bwd_q_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q, ldq, k, v, ldkv, dout,
                     ldo, dq, pbuf, dsbuf, tokens, ctx_len, heads, scale, total);
  const long long waves = (long long)n * ctx_len * heads;
  hipLaunchKernelGGL(attn_cross_bwd_kv_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, q, ldq, dout, ldo, pbuf, dsbuf,
                     dk, dv, ldkv, tokens, ctx_len, heads, waves);
  return check_launch("ldmk_attn_cross_bwd");
}

extern "C" int ldmk_attn_self_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                  float* dsum, int n, int tokens, int heads, float scale, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && out && dout && lse && dqkv && dsum, "ldmk_attn_self_bwd: null buffer");
  LDMK_REQUIRE(n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535, "ldmk_attn_self_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)n * tokens * heads;
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dout, out, dsum, tokens, heads, total);
  dim3 grid((tokens + 127) / 128, heads, n);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dsum, dqkv, tokens, heads, scale);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dsum, dqkv, tokens, heads, scale);
  return check_launch("ldmk_attn_self_bwd");
}
