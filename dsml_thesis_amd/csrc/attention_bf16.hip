// Self attention (d_head = 32) forward + backward with bf16 matrix-core products, for the bf16 TRAINING step
// (BASELINE configs[4]; the reference trains under Lightning's mixed precision, main.py:532 / ddpm.py:1014-1047).
//
// Round 2 ran the training step's GEMMs on v_mfma_f32_32x32x16_bf16 but kept attention on the fp32 pipe: at 64x64x4,
// batch 16, the three attention kernels were then 35 of the step's 88 ms (profiles/r03_train_bf16_kernel_stats.txt:
// forward 8.3, dQ 11.7, dK/dV 15.1 ms).  Same split as torch.autocast(bfloat16) makes for scaled-dot-product attention:
// Q K^T, P V and the five backward products take bf16 operands (rounded to nearest even) and accumulate in fp32; the
// softmax, its statistics (log-sum-exp, D = rowsum(dO o O)) and everything stored stay fp32.
//
// Algorithm = attention.hip / attention_bwd.hip (flash style, scores recomputed from the forward's log-sum-exp, two
// deterministic backward kernels), re-cut for the 32x32x16 instruction:
//   * a 32-deep contraction is TWO instructions (k = 16 each) instead of sixteen; operands are 8 bf16 per lane
//     (lanes 0-31: k 0..7, lanes 32-63: k 8..15);
//   * tiles of 64 rows are staged once as TWO bf16 LDS images: row-major [row][d] (16-byte operand reads for products
//     that contract over d) and transposed [d][row] (two 8-byte reads for products that contract over rows);
//   * the accumulator trick carries over: a 32x32 score tile holds, in registers 8t .. 8t+7 of lane (column, half h), rows
//     16t + 4h + (j&3) + 8(j>>2) -- taken as they are (packed to bf16) they are the B operand of step t of the next
//     product, provided the A operand enumerates the contraction index in the same order; the transposed image serves that
//     order as two runs of four.
#include "ldmk_common.h"
#include <type_traits>

namespace ldmk {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BA_D = 32;
constexpr int BA_T = 64;            // rows per staged tile
constexpr int BA_RS = 40;           // row-major image: bf16 per row (80 B: 16-byte aligned operand reads, odd multiple of 16 B)
constexpr int BA_TS = 72;           // transposed image: bf16 per d row (144 B)
constexpr int BA_FS = 33;           // fp32 transpose buffer stride (output rows)

__device__ __forceinline__ bf16x8_t pack8(const float* v) {
  return bf16x8_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
}
__device__ __forceinline__ bf16x8_t pack8(const f32x16& a, int t) {
  return bf16x8_t{(__bf16)a[8 * t], (__bf16)a[8 * t + 1], (__bf16)a[8 * t + 2], (__bf16)a[8 * t + 3],
                  (__bf16)a[8 * t + 4], (__bf16)a[8 * t + 5], (__bf16)a[8 * t + 6], (__bf16)a[8 * t + 7]};
}

// stage 64 rows x 32 floats of a [rows][ld] fp32 matrix as bf16: R[row][d] (if R) and T[d][row] (if T); rows past the end are 0
__device__ __forceinline__ void stage_bf16(__bf16* R, __bf16* T, const float* __restrict__ src, long long ld, int r0, int rows, int tid) {
  const int rr = tid >> 3, d4 = (tid & 7) * 4;
  float4 v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + rr + 32 * i;
    v[i] = r < rows ? *reinterpret_cast<const float4*>(src + (long long)r * ld + d4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = rr + 32 * i;
    const bf16x4_t b = {(__bf16)v[i].x, (__bf16)v[i].y, (__bf16)v[i].z, (__bf16)v[i].w};
    if (R) *reinterpret_cast<bf16x4_t*>(R + row * BA_RS + d4) = b;
    if (T) { T[(d4 + 0) * BA_TS + row] = b[0]; T[(d4 + 1) * BA_TS + row] = b[1]; T[(d4 + 2) * BA_TS + row] = b[2]; T[(d4 + 3) * BA_TS + row] = b[3]; }
  }
}

// A operand from the row-major image: rows sub*32 + l31, contraction index d = 16 t + 8 half .. + 7
__device__ __forceinline__ bf16x8_t op_rows(const __bf16* R, int sub, int l31, int half, int t) {
  return *reinterpret_cast<const bf16x8_t*>(R + (sub * 32 + l31) * BA_RS + 16 * t + 8 * half);
}
// A operand from the transposed image: row d = l31, contraction index = tile rows sub*32 + 16 t + 4 half + (j&3) + 8 (j>>2)
__device__ __forceinline__ bf16x8_t op_cols(const __bf16* T, int sub, int l31, int half, int t) {
  const __bf16* p = T + l31 * BA_TS + sub * 32 + 16 * t + 4 * half;
  const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(p), hi = *reinterpret_cast<const bf16x4_t*>(p + 8);
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// register fragment (B operand): row `rowp` of a [rows][ld] fp32 matrix, d = 16 t + 8 half .. + 7, scaled
__device__ __forceinline__ void frag_rows(bf16x8_t (&f)[2], const float* __restrict__ rowp, int half, float mul) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half), b = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half + 4);
    const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, b.x * mul, b.y * mul, b.z * mul, b.w * mul};
    f[t] = pack8(v);
  }
}
__device__ __forceinline__ f32x16 mm(const bf16x8_t a, const bf16x8_t b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// write a wave's accumulator acc[r] = X^T[d = (r&3)+8(r>>2)+4 half][row = l31] as fp32 rows of 128 B
__device__ __forceinline__ void store_rows_bf(float* ts, const f32x16& acc, float mul, float* __restrict__ dst, long long ld,
                                              int row0, int rows, int l31, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) ts[l31 * BA_FS + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[r] * mul;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 32; q += 2)
    if (row0 + q + half < rows) dst[(long long)(row0 + q + half) * ld + l31] = ts[(q + half) * BA_FS + l31];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
}

// ---- forward: out = softmax(scale Q K^T) V, lse = log-sum-exp of the scaled scores (natural log), both fp32
__global__ __launch_bounds__(256) void attn_bf16_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ lse, int tokens, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) __bf16 Kr[BA_T * BA_RS];
  __shared__ __attribute__((aligned(16))) __bf16 Vt[BA_D * BA_TS];
  __shared__ float Ts[4][32 * BA_FS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  const bool q_valid = q0 + l31 < tokens;
  bf16x8_t qf[2];
  frag_rows(qf, base + (long long)(q_valid ? q0 + l31 : 0) * ld + h * BA_D, half, q_valid ? scale : 0.f);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage_bf16(Kr, nullptr, base + C + h * BA_D, ld, kt * BA_T, tokens, tid);
    stage_bf16(nullptr, Vt, base + 2 * C + h * BA_D, ld, kt * BA_T, tokens, tid);
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * BA_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) s = mm(op_rows(Kr, sub, l31, half, t), qf[t], s);        // S^T[key][q]
      if (key0 + 32 > tokens) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) s[r] = -INFINITY;
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_new); psum += s[r]; }
      psum += __shfl_xor(psum, 32, 64);
      const float corr = __expf(m_run - m_new);
      l_run = l_run * corr + psum;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= corr;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < 2; ++t) o = mm(op_cols(Vt, sub, l31, half, t), pack8(s, t), o);   // O^T[d][q] += V^T P^T
    }
  }
  if (!wave_active) return;
  if (lse != nullptr && half == 0 && q_valid) lse[((long long)b * heads + h) * tokens + q0 + l31] = m_run + __logf(l_run);
  store_rows_bf(Ts[wave], o, 1.0f / l_run, out + (long long)b * tokens * C + h * BA_D, C, q0, tokens, l31, half);
}

// ---- backward, dQ: a wave owns 32 queries and walks the keys
__global__ __launch_bounds__(256) void attn_bf16_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           float* __restrict__ dqkv, int tokens, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) __bf16 Kr[BA_T * BA_RS];
  __shared__ __attribute__((aligned(16))) __bf16 Kt[BA_D * BA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 Vr[BA_T * BA_RS];
  __shared__ float Ts[4][32 * BA_FS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  const bool q_valid = q0 + l31 < tokens;
  const int qq = q_valid ? q0 + l31 : 0;
  bf16x8_t qf[2], dof[2];
  frag_rows(qf, base + (long long)qq * ld + h * BA_D, half, q_valid ? scale : 0.f);
  frag_rows(dof, dout + ((long long)b * tokens + qq) * C + h * BA_D, half, q_valid ? 1.f : 0.f);
  const float Lq = q_valid ? lse[((long long)b * heads + h) * tokens + qq] : INFINITY;
  const float Dq = q_valid ? dsum[((long long)b * heads + h) * tokens + qq] : 0.f;
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage_bf16(Kr, Kt, base + C + h * BA_D, ld, kt * BA_T, tokens, tid);
    stage_bf16(Vr, nullptr, base + 2 * C + h * BA_D, ld, kt * BA_T, tokens, tid);
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * BA_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sa = mm(op_rows(Kr, sub, l31, half, t), qf[t], sa);      // S^T[key][q]
        da = mm(op_rows(Vr, sub, l31, half, t), dof[t], da);     // dP^T[key][q]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __expf(sa[r] - Lq);
        if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) p = 0.f;
        sa[r] = p * (da[r] - Dq);                                 // dS^T[key][q]
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) dq = mm(op_cols(Kt, sub, l31, half, t), pack8(sa, t), dq);   // dQ^T[d][q] += K^T dS^T
    }
  }
  if (!wave_active) return;
  store_rows_bf(Ts[wave], dq, scale, dqkv + (long long)b * tokens * ld + h * BA_D, ld, q0, tokens, l31, half);
}

// ---- backward, dK / dV: a wave owns 32 keys and walks the queries
__global__ __launch_bounds__(256) void attn_bf16_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ dsum,
                                                            float* __restrict__ dqkv, int tokens, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) __bf16 Qr[BA_T * BA_RS];
  __shared__ __attribute__((aligned(16))) __bf16 Qt[BA_D * BA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 Or[BA_T * BA_RS];      // dO tile
  __shared__ __attribute__((aligned(16))) __bf16 Ot[BA_D * BA_TS];
  __shared__ float Ls[BA_T], Ds[BA_T];
  __shared__ float Ts[4][32 * BA_FS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const float* dbase = dout + (long long)b * tokens * C;
  const bool wave_active = k0 < tokens;
  const bool k_valid = k0 + l31 < tokens;
  const int kk = k_valid ? k0 + l31 : 0;
  bf16x8_t kf[2], vf[2];
  frag_rows(kf, base + (long long)kk * ld + C + h * BA_D, half, k_valid ? scale : 0.f);
  frag_rows(vf, base + (long long)kk * ld + 2 * C + h * BA_D, half, k_valid ? 1.f : 0.f);
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  const float* lrow = lse + ((long long)b * heads + h) * tokens;
  const float* drow = dsum + ((long long)b * heads + h) * tokens;
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  for (int qt = 0; qt < ntiles; ++qt) {
    __syncthreads();
    stage_bf16(Qr, Qt, base + h * BA_D, ld, qt * BA_T, tokens, tid);
    stage_bf16(Or, Ot, dbase + h * BA_D, C, qt * BA_T, tokens, tid);
    if (tid < BA_T) {
      const int q = qt * BA_T + tid;
      Ls[tid] = q < tokens ? lrow[q] : INFINITY;       // exp(s - inf) = 0: rows past the end contribute nothing
      Ds[tid] = q < tokens ? drow[q] : 0.f;
    }
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int qbase = qt * BA_T + sub * 32;
      if (qbase >= tokens) break;
      f32x16 sa, da;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; da[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sa = mm(op_rows(Qr, sub, l31, half, t), kf[t], sa);      // S[q][key]
        da = mm(op_rows(Or, sub, l31, half, t), vf[t], da);      // dP[q][key]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float p = __expf(sa[r] - Ls[ql]);
        da[r] = p * (da[r] - Ds[ql]);      // dS[q][key]
        sa[r] = p;                         // P[q][key]
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dv = mm(op_cols(Ot, sub, l31, half, t), pack8(sa, t), dv);          // dV^T[d][key] += dO^T P
        dk = mm(op_cols(Qt, sub, l31, half, t), pack8(da, t), dk);          // dK^T[d][key] += Q^T dS
      }
    }
  }
  if (!wave_active) return;
  float* obase = dqkv + (long long)b * tokens * ld + h * BA_D;
  store_rows_bf(Ts[wave], dk, scale, obase + C, ld, k0, tokens, l31, half);
  store_rows_bf(Ts[wave], dv, 1.0f, obase + 2 * C, ld, k0, tokens, l31, half);
}

void attn_rowdot_launch(const float* dout, const float* out, float* dsum, int tokens, int heads, long long total, hipStream_t st);   // attention_bwd.hip

// ---------------------------------------------------------------------------------------------------------------------
// fp32-ACCURATE self attention on the bf16 matrix cores (the sampling path; LDMK_COMPUTE_BF16X3 of include/ldmk.h applied
// to the two attention products).  Every fp32 operand -- Q (pre-scaled), K, V and the probabilities P -- is the exact sum
// of three bf16 values; each product accumulates the six partial products that are not below fp32 resolution.  A 32-key
// step is 24 bf16 MFMAs of 32 cycles (768) where the fp32 form issues 32 of 64 (2048); softmax, running statistics and the
// output stay fp32.  Same tile walk and operand orders as attn_bf16_fwd_kernel above; K / V tiles hold three images each.
__device__ __forceinline__ float bfu(__bf16 b) { return (float)b; }
__device__ __forceinline__ void split4(const float4& v, bf16x4_t& h, bf16x4_t& m, bf16x4_t& l) {
  h = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  const float r0 = v.x - bfu(h[0]), r1 = v.y - bfu(h[1]), r2 = v.z - bfu(h[2]), r3 = v.w - bfu(h[3]);
  m = bf16x4_t{(__bf16)r0, (__bf16)r1, (__bf16)r2, (__bf16)r3};
  l = bf16x4_t{(__bf16)(r0 - bfu(m[0])), (__bf16)(r1 - bfu(m[1])), (__bf16)(r2 - bfu(m[2])), (__bf16)(r3 - bfu(m[3]))};
}
__device__ __forceinline__ void split8(const float* v, bf16x8_t (&o)[3]) {
  float r[8];
  o[0] = pack8(v);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = v[i] - bfu(o[0][i]);
  o[1] = pack8(r);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] -= bfu(o[1][i]);
  o[2] = pack8(r);
}
// the six partial products, smallest first (images: 0 = hi, 1 = mid, 2 = lo)
__device__ __forceinline__ f32x16 mm6(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16 c) {
  c = mm(a[2], b[0], c);
  c = mm(a[0], b[2], c);
  c = mm(a[1], b[1], c);
  c = mm(a[1], b[0], c);
  c = mm(a[0], b[1], c);
  return mm(a[0], b[0], c);
}

// ---- the softmax step of one 32-key block, written for the instruction count: the key loop of these kernels is bound by its
// vector work next to the matrix cores (s_memtime / PMC), so
//  * the two 32-lane halves of a wave exchange maxima and sums with v_permlane32_swap (one VALU operation) instead of a
//    ds_bpermute round trip through the LDS,
//  * score - max, the row sums and the residuals of the split are packed fp32 operations (v_pk_add_f32: two elements each),
//  * the maxima are v_max3 chains (the file is compiled with -fno-honor-nans: no canonicalising v_max x, x per input; a
//    query column always sees at least one unmasked key, so no NaN can arise),
//  * the bf16 images of a pair of probabilities are one v_cvt_pk_bf16_f32 each, widened back with a shift / a mask.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void both_halves(float x, float& a, float& b) {       // a, b = the value of lane l31 / of lane l31 + 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2 p) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t v = {(__bf16)p.x, (__bf16)p.y};
  unsigned u = __builtin_bit_cast(unsigned, v);
  asm("" : "+v"(u));                 // (opaque: otherwise the low half is converted a second time on its own)
  return u;
}
__device__ __forceinline__ f32x2 widen_pk_bf16(unsigned u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
template <bool NO_EXP = false>
__device__ __forceinline__ void x3_softmax(f32x16& s, f32x16& o, float& m_run, float& l_run) {
  float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
  mx = fmaxf(mx, s[15]);
  float a, b;
  both_halves(mx, a, b);
  const float m_new = fmaxf(fmaxf(m_run, a), b);
  const f32x2 mm2 = {m_new, m_new};
  f32x2 ps = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2 d = f32x2{s[2 * i], s[2 * i + 1]} - mm2;
    if constexpr (!NO_EXP) {
      d.x = __builtin_amdgcn_exp2f(d.x);
      d.y = __builtin_amdgcn_exp2f(d.y);
    }
    s[2 * i] = d.x;
    s[2 * i + 1] = d.y;
    ps += d;
  }
  both_halves(ps.x + ps.y, a, b);
  if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {      // rescale only when some lane's maximum moved
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
    m_run = m_new;
  }
  l_run += a + b;
}
// probabilities s[8 t .. 8 t + 7] as their three bf16 images (the exact split of split8, two elements per operation)
__device__ __forceinline__ void split_p8(const f32x16& s, int t, bf16x8_t (&o)[3]) {
  u32x4a h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 p = {s[8 * t + 2 * j], s[8 * t + 2 * j + 1]};
    h[j] = cvt_pk_bf16(p);
    const f32x2 r = p - widen_pk_bf16(h[j]);
    m[j] = cvt_pk_bf16(r);
    l[j] = cvt_pk_bf16(r - widen_pk_bf16(m[j]));
  }
  o[0] = __builtin_bit_cast(bf16x8_t, h);
  o[1] = __builtin_bit_cast(bf16x8_t, m);
  o[2] = __builtin_bit_cast(bf16x8_t, l);
}

constexpr int X3_KIMG = BA_T * BA_RS, X3_VIMG = BA_D * BA_TS;       // bf16 elements per K / V image

__global__ __launch_bounds__(256) void attn_x3_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int tokens,
                                                          int heads, float scale) {
  // one array: three K images, three V^T images; after the key loop the same memory is the four output transpose buffers
  __shared__ __attribute__((aligned(16))) __bf16 smem_x3[3 * X3_KIMG + 3 * X3_VIMG];
  static_assert((3 * X3_KIMG + 3 * X3_VIMG) * 2 >= 4 * 32 * BA_FS * 4, "transpose buffers alias the staging images");
  __bf16* Kr = smem_x3;
  __bf16* Vt = smem_x3 + 3 * X3_KIMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  const bool q_valid = q0 + l31 < tokens;
  constexpr float LOG2E = 1.4426950408889634f;
  // Q fragments (B operand of S^T = K Q^T), pre-scaled by scale * log2(e) in fp32 (scores live in the log2 domain), then split
  bf16x8_t qf[2][3];
  {
    const float* rowp = base + (long long)(q_valid ? q0 + l31 : 0) * ld + h * BA_D;
    const float mul = q_valid ? scale * LOG2E : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half), c = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half + 4);
      const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, c.x * mul, c.y * mul, c.z * mul, c.w * mul};
      split8(v, qf[t]);
    }
  }
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  // staging map: thread -> rows rr, rr + 32 of the tile, 4 consecutive d
  const int rr = tid >> 3, d4 = (tid & 7) * 4;
  const float* kp = base + C + h * BA_D + d4;
  float4 kreg[2], vreg[2];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = kt * BA_T + rr + 32 * i;
      const bool ok = r < tokens;
      const float* src = kp + (long long)(ok ? r : 0) * ld;
      kreg[i] = *reinterpret_cast<const float4*>(src);
      vreg[i] = *reinterpret_cast<const float4*>(src + C);
      if (!ok) { kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f); vreg[i] = kreg[i]; }
    }
  };
  fetch(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = rr + 32 * i;
      bf16x4_t g[3];
      split4(kreg[i], g[0], g[1], g[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4_t*>(Kr + q * X3_KIMG + row * BA_RS + d4) = g[q];
      split4(vreg[i], g[0], g[1], g[2]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        __bf16* T = Vt + q * X3_VIMG + d4 * BA_TS + row;
        T[0] = g[q][0]; T[BA_TS] = g[q][1]; T[2 * BA_TS] = g[q][2]; T[3 * BA_TS] = g[q][3];
      }
    }
    __syncthreads();
    if (kt + 1 < ntiles) fetch(kt + 1);               // in flight under this tile's products
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * BA_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t ka[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) ka[q] = op_rows(Kr + q * X3_KIMG, sub, l31, half, t);
        s = mm6(ka, qf[t], s);                                                          // S^T[key][q], log2 domain
      }
      if (key0 + 32 > tokens) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) s[r] = -INFINITY;
      }
      x3_softmax(s, o, m_run, l_run);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t pb[3], va[3];
        split_p8(s, t, pb);
#pragma unroll
        for (int q = 0; q < 3; ++q) va[q] = op_cols(Vt + q * X3_VIMG, sub, l31, half, t);
        o = mm6(va, pb, o);                                                             // O^T[d][q] += V^T P^T
      }
    }
  }
  __syncthreads();                     // every wave is done with the staging images: their memory becomes the transpose buffers
  if (!wave_active) return;
  float* ts = reinterpret_cast<float*>(smem_x3) + wave * (32 * BA_FS);
  store_rows_bf(ts, o, 1.0f / l_run, out + (long long)b * tokens * C + h * BA_D, C, q0, tokens, l31, half);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel with K and V PRE-SPLIT (round 4).  attn_x3_fwd_kernel re-splits every K / V element once per 128-query
// workgroup -- tokens / 128 times (32 x at 4096 tokens) -- and stores the V^T images with 2-byte LDS writes: ~44 of its ~250
// vector operations per 32 keys and wave, plus 15 LDS stores.  Here a pre-pass (attn_kv_split_kernel) splits K and V ONCE
// per (sample, head, 64-key tile) and writes the six images in MFMA-operand order -- per tile 8 fragments x 3 planes of 1 KiB:
//   fragment 2 sub + t      : K  rows (keys)  sub 32 + l31, d = 16 t + 8 half .. + 7                  (A operand of S^T = K Q^T)
//   fragment 4 + 2 sub + t  : V^T rows d = l31, keys sub 32 + 16 t + 4 half + (j & 3) + 8 (j >> 2)    (A operand of O^T += V^T P^T:
//                             the order in which the probabilities leave the first product's accumulator)
// and the attention kernel moves a tile memory -> LDS with LDS-DMA (`buffer_load_dwordx4 ... lds`, 6 pieces per wave), double
// buffered, and reads every operand with one conflict-free ds_read_b128.  Same split values, same instruction sequence per
// accumulator: bitwise the results of attn_x3_fwd_kernel.
typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));
constexpr int X3P_QB2_MIN_TOKENS = 2048;
constexpr int X3P_TILE = 24 * 1024;           // bytes per 64-key tile: (4 K + 4 V^T fragments) x 3 planes x 1 KiB

__global__ __launch_bounds__(256) void attn_kv_split_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ kv, int tokens, int heads) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ntiles = gridDim.x;
  const float* base = qkv + (long long)b * tokens * ld + h * BA_D;
  unsigned char* dst = kv + (((long long)b * heads + h) * ntiles + kt) * X3P_TILE;
  const int sub = wave >> 1, t = wave & 1;
  // K fragment (sub, t): this lane's key row, 8 consecutive d
  {
    const int key = kt * BA_T + sub * 32 + l31;
    float v[8];
    if (key < tokens) {
      const float* p = base + C + (long long)key * ld + 16 * t + 8 * half;
      const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    bf16x8_t g[3];
    split8(v, g);
    unsigned char* d = dst + (2 * sub + t) * 3072 + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8_t*>(d + q * 1024) = g[q];
  }
  // V^T fragment (sub, t): this lane's d, 8 keys in accumulator order
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kt * BA_T + sub * 32 + 16 * t + 4 * half + (j & 3) + 8 * (j >> 2);
      v[j] = key < tokens ? base[2 * C + (long long)key * ld + l31] : 0.f;
    }
    bf16x8_t g[3];
    split8(v, g);
    unsigned char* d = dst + (4 + 2 * sub + t) * 3072 + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8_t*>(d + q * 1024) = g[q];
  }
}

__device__ __forceinline__ void x3p_dma3(unsigned voff, const au32x4& rs, unsigned lds_dst) {     // 3 x 1 KiB, contiguous both sides
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// a * b rounded to fp32 on its own (no contraction into a following subtraction: hipcc's default is -ffp-contract=fast)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// QB = query blocks of 32 per wave.  With QB = 2 a wave owns 64 queries: every K / V^T fragment read from the LDS feeds two
// products, the two blocks' accumulator chains are independent (the matrix pipe never waits for the previous product of the
// same accumulator) and one block's softmax issues next to the other block's products.  The LDS (2 x 24 KiB) allows three
// workgroups per CU either way, i.e. 168 registers per lane.
// (DBG, probe builds only -- tools/probe/attn_x3p_probe.hip: bit 0 = no S^T products, 1 = no exponentials, 2 = no split of the
//  probabilities, 3 = no O^T products, 4 = no LDS operand reads, 5 = no softmax at all.  Results are garbage then; the product
//  library instantiates DBG = 0 only.)
template <int QB, int DBG = 0>
__global__ __launch_bounds__(256, 3) void attn_x3p_fwd_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ kv,
                                                              float* __restrict__ out, unsigned char* __restrict__ out_ps, int tokens, int heads,
                                                              float scale) {
  // two tile buffers; after the key loop the same memory is the four output transpose buffers
  __shared__ __attribute__((aligned(1024))) unsigned char smem_p[2 * X3P_TILE];
  static_assert(2 * X3P_TILE >= 4 * 32 * BA_FS * 4, "transpose buffers alias the tile buffers");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB);
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  constexpr float LOG2E = 1.4426950408889634f;
  bf16x8_t qf[QB][2][3];
  f32x16 o[QB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const bool q_valid = q0 + 32 * j + l31 < tokens;
    const float* rowp = base + (long long)(q_valid ? q0 + 32 * j + l31 : 0) * ld + h * BA_D;
    const float mul = q_valid ? scale * LOG2E : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half), c = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half + 4);
      const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, c.x * mul, c.y * mul, c.z * mul, c.w * mul};
      split8(v, qf[j][t]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    m_run[j] = -INFINITY;
    l_run[j] = 0.f;
  }
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  // this (sample, head)'s tiles as one buffer; wave w moves fragments 2w, 2w + 1 of a tile (6 KiB contiguous)
  const unsigned char* kvh = kv + ((long long)b * heads + h) * ntiles * X3P_TILE;
  au32x4 rs;
  {
    const unsigned long long a = (unsigned long long)kvh;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    rs.z = __builtin_amdgcn_readfirstlane((unsigned)ntiles * (unsigned)X3P_TILE);
    rs.w = 0x00020000u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem_p;
  const unsigned woff = (unsigned)wave * 6144u + (unsigned)lane * 16u;
  auto fetch = [&](int kt) {                      // LDS-DMA of tile kt into buffer kt & 1 (tiles past the end: out of range, zeros)
    const unsigned dstb = lds0 + (unsigned)(kt & 1) * X3P_TILE + (unsigned)wave * 6144u;
    const unsigned src = (unsigned)kt * (unsigned)X3P_TILE + woff;
    x3p_dma3(src, rs, dstb);
    x3p_dma3(src + 3072u, rs, dstb + 3072u);
  };
  fetch(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed
    asm volatile("s_barrier" ::: "memory");               // ... everyone's have, and the other buffer is no longer read
    fetch(kt + 1);                                        // in flight under this tile's products
    if (!wave_active) continue;
    const unsigned char* tb = smem_p + (kt & 1) * X3P_TILE + lane * 16;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * BA_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 s[QB];
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t ka[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if constexpr (DBG & 16) { ka[q] = qf[0][t][q]; asm volatile("" : "+v"(ka[q])); }
          else ka[q] = *reinterpret_cast<const bf16x8_t*>(tb + ((2 * sub + t) * 3 + q) * 1024);
        }
        if constexpr (DBG & 1) {
#pragma unroll
          for (int j = 0; j < QB; ++j) asm volatile("" : "+v"(s[j]) : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]));
        } else if constexpr (QB == 1) {
          s[0] = mm6(ka, qf[0][t], s[0]);                                                 // S^T[key][q], log2 domain
        } else {                                 // the two blocks' chains interleaved, each in its own (bitwise) order
          s[0] = mm(ka[2], qf[0][t][0], s[0]); s[1] = mm(ka[2], qf[1][t][0], s[1]);
          s[0] = mm(ka[0], qf[0][t][2], s[0]); s[1] = mm(ka[0], qf[1][t][2], s[1]);
          s[0] = mm(ka[1], qf[0][t][1], s[0]); s[1] = mm(ka[1], qf[1][t][1], s[1]);
          s[0] = mm(ka[1], qf[0][t][0], s[0]); s[1] = mm(ka[1], qf[1][t][0], s[1]);
          s[0] = mm(ka[0], qf[0][t][1], s[0]); s[1] = mm(ka[0], qf[1][t][1], s[1]);
          s[0] = mm(ka[0], qf[0][t][0], s[0]); s[1] = mm(ka[0], qf[1][t][0], s[1]);
        }
      }
      if (key0 + 32 > tokens) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) s[j][r] = -INFINITY;
      }
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        if constexpr (DBG & 32) asm volatile("" : "+v"(s[j]));
        else x3_softmax<(DBG & 2) != 0>(s[j], o[j], m_run[j], l_run[j]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t va[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if constexpr (DBG & 16) { va[q] = qf[0][t][q]; asm volatile("" : "+v"(va[q])); }
          else va[q] = *reinterpret_cast<const bf16x8_t*>(tb + ((4 + 2 * sub + t) * 3 + q) * 1024);
        }
        if constexpr ((DBG & 12) != 0) {
#pragma unroll
          for (int j = 0; j < QB; ++j) {
            bf16x8_t pb[3];
            if constexpr (DBG & 4) {
#pragma unroll
              for (int q = 0; q < 3; ++q) pb[q] = __builtin_bit_cast(bf16x8_t, u32x4a{__float_as_uint(s[j][8 * t + q]), __float_as_uint(s[j][8 * t + q + 1]), __float_as_uint(s[j][8 * t + q + 2]), __float_as_uint(s[j][8 * t + q + 3])});
            } else split_p8(s[j], t, pb);
            if constexpr (DBG & 8) asm volatile("" : "+v"(o[j]) : "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(va[0]), "v"(va[1]), "v"(va[2]));
            else o[j] = mm6(va, pb, o[j]);
          }
        } else if constexpr (QB == 1) {
          bf16x8_t pb[3];
          split_p8(s[0], t, pb);
          o[0] = mm6(va, pb, o[0]);                                                       // O^T[d][q] += V^T P^T
        } else {
          bf16x8_t pb[2][3];
          split_p8(s[0], t, pb[0]);
          split_p8(s[1], t, pb[1]);
          o[0] = mm(va[2], pb[0][0], o[0]); o[1] = mm(va[2], pb[1][0], o[1]);
          o[0] = mm(va[0], pb[0][2], o[0]); o[1] = mm(va[0], pb[1][2], o[1]);
          o[0] = mm(va[1], pb[0][1], o[0]); o[1] = mm(va[1], pb[1][1], o[1]);
          o[0] = mm(va[1], pb[0][0], o[0]); o[1] = mm(va[1], pb[1][0], o[1]);
          o[0] = mm(va[0], pb[0][1], o[0]); o[1] = mm(va[0], pb[1][1], o[1]);
          o[0] = mm(va[0], pb[0][0], o[0]); o[1] = mm(va[0], pb[1][0], o[1]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the trailing out-of-range fetch)
  __syncthreads();                     // every wave is done with the tile buffers: their memory becomes the transpose buffers
  if (!wave_active) return;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qb0 = q0 + 32 * j;
    if (qb0 >= tokens) break;
    const bool q_valid = qb0 + l31 < tokens;
    if (out_ps) {
      // the result in the PS layout (include/ldmk.h) of the [n tokens][C] matrix -- the pre-split A operand of attn1.to_out on the
      // pre-split GEMM tiles (csrc/igemm_ps.hip).  The accumulator already has the layout its transposed epilogue writes from:
      // lane = (query, half), registers 4 g .. 4 g + 3 = d 8 g + 4 half .. + 3: one 8-byte store per (g, plane), no LDS pass.
      // (tokens % 32 == 0: a wave's 32 queries are one row block)
      const float inv = 1.0f / l_run[j];
      unsigned char* d0 = out_ps + ((((long long)b * tokens + qb0) >> 5) * (C / 16) + 2 * h) * 3072 + l31 * 16 + half * 8;
      if (q_valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4_t hh, mm_, ll;
          // (the fp32 product is rounded BEFORE it is split -- no fused multiply-subtract into the residuals -- so the planes sum
          //  to exactly the value the fp32 output holds)
          float4 v4 = make_float4(mul_rounded(o[j][4 * g], inv), mul_rounded(o[j][4 * g + 1], inv), mul_rounded(o[j][4 * g + 2], inv), mul_rounded(o[j][4 * g + 3], inv));
          asm volatile("" : "+v"(v4.x), "+v"(v4.y), "+v"(v4.z), "+v"(v4.w));
          split4(v4, hh, mm_, ll);
          unsigned char* d = d0 + (g >> 1) * 3072 + (g & 1) * 512;
          *reinterpret_cast<bf16x4_t*>(d) = hh;
          *reinterpret_cast<bf16x4_t*>(d + 1024) = mm_;
          *reinterpret_cast<bf16x4_t*>(d + 2048) = ll;
        }
      }
    }
    if (out) {
      float* ts = reinterpret_cast<float*>(smem_p) + wave * (32 * BA_FS);
      store_rows_bf(ts, o[j], 1.0f / l_run[j], out + (long long)b * tokens * C + h * BA_D, C, qb0, tokens, l31, half);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32-accurate self attention from THREE fp16 products per term (LDMK_COMPUTE_F16X2 of include/ldmk.h applied to the two
// attention products).  gfx950 sustains only 0.60-0.69 of its nominal bf16 / fp16 matrix rate on real operands (power:
// tools/clock_probe.hip, profiles/r04_clock_bf16.txt), so the lever left is the NUMBER of matrix instructions per product.
// An operand scaled by a power of two into fp16's range is written x' = hi + lo with hi = fp16(x'), lo = fp16(x' - hi)
// (round-to-nearest-even; x' - hi is exact in fp32): 2 x 11 significand bits, |x' - hi - lo| <= 2^-23 |x'| -- one fp32 ulp, the
// size of an fp32 rounding error, where the three-way bf16 split is exact.  hi hi, hi lo, lo hi accumulate in one fp32 accumulator
// (fp16 x fp16 products are exact in fp32), smallest first; lo lo (<= 2^-22 of the product) is dropped.  Measured against
// float64 the error is 1.0-1.7 x that of an fp32 dot product (the larger figure at K = 160; tests/test_f16x2_gpu.py).
//   scales: K, V and the pre-scaled Q by 2^6 (|x| < 1000 required: the pre-pass raises *range_flag otherwise and the caller
//   re-runs in the bf16x3 arithmetic; an element below 2^-9 keeps an ABSOLUTE precision of 2^-31), the probabilities by 2^14
//   (the exponent argument is offset by 14).  Scores live in the domain scaled by 2^12; the output is rescaled at the end.
// Tiles: per 64 keys 8 fragments x 2 planes of 1 KiB (16 KiB), same fragment order as the bf16x3 tiles above.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int H2_TILE = 16 * 1024;
constexpr float H2_S = 64.f;                  // 2^6: K, V, Q
constexpr float H2_RANGE = 1000.f;            // |K|, |V|, |scale log2(e) Q| below this (x 64 < 65504, fp16's largest finite value)
constexpr float H2_PEXP = 14.f;               // probabilities scaled by 2^14

__device__ __forceinline__ unsigned cvt_pk_f16(f32x2 p) {
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f16x2_t v = {(_Float16)p.x, (_Float16)p.y};
  unsigned u = __builtin_bit_cast(unsigned, v);
  asm("" : "+v"(u));
  return u;
}
__device__ __forceinline__ f32x2 widen_pk_f16(unsigned u) {
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const f16x2_t v = __builtin_bit_cast(f16x2_t, u);
  return f32x2{(float)v.x, (float)v.y};
}
__device__ __forceinline__ unsigned split_lo_pk_f16(f32x2 p, unsigned h) { return h2_lo_pair(h, p.x, p.y); }      // (ldmk_common.h)
// 8 (already scaled) values as their two fp16 images
__device__ __forceinline__ void split8_h2(const float* v, f16x8_t (&o)[2]) {
  u32x4a h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 p = {v[2 * j], v[2 * j + 1]};
    h[j] = cvt_pk_f16(p);
    l[j] = split_lo_pk_f16(p, h[j]);
  }
  o[0] = __builtin_bit_cast(f16x8_t, h);
  o[1] = __builtin_bit_cast(f16x8_t, l);
}
__device__ __forceinline__ bool out_of_h2_range(float v) { return (__float_as_uint(v) & 0x7fffffffu) >= __float_as_uint(H2_RANGE); }   // (inf, NaN too)
__device__ __forceinline__ f32x16 mmh(const f16x8_t a, const f16x8_t b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mmh3(const f16x8_t (&a)[2], const f16x8_t (&b)[2], f32x16 c) {
  c = mmh(a[1], b[0], c);
  c = mmh(a[0], b[1], c);
  return mmh(a[0], b[0], c);
}

__global__ __launch_bounds__(256) void attn_kv_split_h2_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ kv, int tokens, int heads,
                                                               int* __restrict__ range_flag) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int ntiles = gridDim.x;
  const float* base = qkv + (long long)b * tokens * ld + h * BA_D;
  unsigned char* dst = kv + (((long long)b * heads + h) * ntiles + kt) * H2_TILE;
  const int sub = wave >> 1, t = wave & 1;
  bool bad = false;
  {
    const int key = kt * BA_T + sub * 32 + l31;
    float v[8];
    if (key < tokens) {
      const float* p = base + C + (long long)key * ld + 16 * t + 8 * half;
      const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { bad |= out_of_h2_range(v[j]); v[j] = h2_clamp(v[j]) * H2_S; }
    f16x8_t g[2];
    split8_h2(v, g);
    unsigned char* d = dst + (2 * sub + t) * 2048 + lane * 16;
    *reinterpret_cast<f16x8_t*>(d) = g[0];
    *reinterpret_cast<f16x8_t*>(d + 1024) = g[1];
  }
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kt * BA_T + sub * 32 + 16 * t + 4 * half + (j & 3) + 8 * (j >> 2);
      v[j] = key < tokens ? base[2 * C + (long long)key * ld + l31] : 0.f;
      bad |= out_of_h2_range(v[j]);
      v[j] = h2_clamp(v[j]) * H2_S;
    }
    f16x8_t g[2];
    split8_h2(v, g);
    unsigned char* d = dst + (4 + 2 * sub + t) * 2048 + lane * 16;
    *reinterpret_cast<f16x8_t*>(d) = g[0];
    *reinterpret_cast<f16x8_t*>(d + 1024) = g[1];
  }
  if (bad) *range_flag = 1;
}

__device__ __forceinline__ void h2_dma4(unsigned voff, const au32x4& rs, unsigned lds_dst) {     // 4 x 1 KiB, contiguous both sides
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:3072 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// the softmax step of one 32-key block in the scaled domains: s holds 2^12 x the log2-domain scores on entry, 2^14 x the
// probabilities on exit; l_run sums the scaled probabilities.  The running maximum m_run lives on the INTEGER grid of the
// unscaled log2 domain (round 5): m = ceil(2^-12 max s) >= the true maximum, so the probabilities stay <= 2^14 (the largest of a
// row > 2^13: the fp16 images keep their precision) and
//   - the exponent is ONE packed fma per pair, 2^-12 s + (14 - m), rounded once (14 - m is a small integer: exact) -- the
//     subtract-then-scale form was two packed instructions and two roundings;
//   - the rescale factor 2^(m_old - m_new) is an exact power of two (no rounding in O or l), and it is needed only when a
//     row's maximum crosses an integer.
// -DLDMK_H2_SOFTMAX_R4: the round-4 form (A/B)
__device__ __forceinline__ void h2_softmax(f32x16& s, f32x16& o, float& m_run, float& l_run) {
  constexpr float INV = 1.0f / (H2_S * H2_S);
  float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
  mx = fmaxf(mx, s[15]);
  float a, b;
  both_halves(mx, a, b);
#ifdef LDMK_H2_SOFTMAX_R4
  const float m_new = fmaxf(fmaxf(m_run, a), b);
  const f32x2 mm2 = {m_new, m_new}, inv2 = {INV, INV}, off2 = {H2_PEXP, H2_PEXP};
#else
  const float m_new = fmaxf(m_run, __builtin_ceilf(fmaxf(a, b) * INV));
  const f32x2 inv2 = {INV, INV}, off2 = {H2_PEXP - m_new, H2_PEXP - m_new};
#endif
  f32x2 ps = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#ifdef LDMK_H2_SOFTMAX_R4
    f32x2 d = (f32x2{s[2 * i], s[2 * i + 1]} - mm2) * inv2 + off2;      // (s - m) exact, x 2^-12 exact, + 14: one rounding
#else
    f32x2 d = __builtin_elementwise_fma(f32x2{s[2 * i], s[2 * i + 1]}, inv2, off2);
#endif
    d.x = __builtin_amdgcn_exp2f(d.x);
    d.y = __builtin_amdgcn_exp2f(d.y);
    s[2 * i] = d.x;
    s[2 * i + 1] = d.y;
    ps += d;
  }
  both_halves(ps.x + ps.y, a, b);
  if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#ifdef LDMK_H2_SOFTMAX_R4
    const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * INV);
#else
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
#endif
    l_run *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;
    m_run = m_new;
  }
  l_run += a + b;
}

// LAZY running maximum (the pipelined loop, from the second 64-key tile of a row on): the probabilities of a block are taken against the
// maximum the row ALREADY has -- no maximum over the block, no wait for it before the first exponential -- and the row sum, which
// the loop needs anyway, says whether that was good enough.  Probabilities are scaled by 2^OFF with OFF = 8 (also in the exact
// block that sets the maxima): a block whose scores stay under the running maximum sums to <= 32 x 2^8 = 2^13, and only a row sum
// >= 2^15 -- a score more than two binary orders above everything seen so far -- sends the wave down h2_lazy_raise: the row's
// maximum goes up by the excess exponent and this block's probabilities, O and l are scaled by that power of two -- exact.
// fp16's range is safe either way (every probability < its row sum < 2^15) and the two fp16 images keep 22 bits of anything within
// 2^-9 of the row's largest.  13 of ~80 VALU instructions per block less, and the head of the dependent chain (maximum ->
// exponent) is gone.  A score >= 120 binary orders above the row's maximum overflows the fp32 exponential itself: such a block is
// outside this kernel's range like an operand outside fp16's -- the launch's range flag goes up (the caller computes the site again
// in bf16x3, whose kernel takes the exact maximum first) and the probabilities are clamped so that the result, wrong either way,
// stays finite (ldmk_common.h, h2_clamp).
constexpr int H2_LAZY_OFF = 8;
__device__ __forceinline__ void h2_lazy_raise(f32x16& s, f32x16& o, float& m_run, float& l_run, float& rs, bool& bad) {
  float t = 0.f, a, b;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bad |= s[r] > 0x1p100f;
    s[r] = fminf(s[r], 0x1p100f);
    t += s[r];
  }
  both_halves(t, a, b);
  rs = a + b;
  const int e = (int)((__float_as_uint(rs) >> 23) & 0xffu) - 127;           // rs in [2^e, 2^(e + 1))
  const int up = rs < 32768.f ? 0 : e - 12;                                 // rows under the bound stay; the others land in [2^12, 2^13)
  const float corr = __builtin_ldexpf(1.0f, -up);
#pragma unroll
  for (int r = 0; r < 16; ++r) { s[r] *= corr; o[r] *= corr; }
  l_run *= corr;
  rs *= corr;
  m_run += (float)up;
}
// One step of the PIPELINED key loop (attn_h2_fwd_kernel<2, true>): the VALU side of 32-key block b -- softmax, then the fp16
// images of its probabilities -- with the MFMAs of its neighbours issued between the slices: first the six of P V for block b - 1
// (operands pb / vf, into the OTHER query block's accumulator op), then the six of Q K^T for block b + 1 (into sn).  What decides
// the order (tools/probe/valu_rate.hip, profiles/r05_valu_rate.txt): a wave's own non-packed VALU instructions run under its
// MFMA (1 MFMA + 8 v_fma_f32: 48 clocks = the VALU side alone), a packed fp32 instruction waits for the MFMA to drain (1 MFMA +
// 8 v_pk_fma_f32: 89 clocks), and the phase-separated loop left the matrix pipe 45 % and the VALU 61 % busy with three waves per
// SIMD -- in sum 106 %: nothing overlapped.  Every slice is fenced (sched_barrier): the instruction order below IS the schedule.
// The arithmetic, operation by operation and in the same order per accumulator, is that of h2_softmax / mmh3 / split8_h2: the
// results are the same bits as the phase-separated kernel's.
#define H2P_FENCE() __builtin_amdgcn_sched_barrier(0)
#define H2P_SPLIT_PAIR(H, L, I, K)                         \
  {                                                      \
    const f32x2 p_ = {s[2 * (K)], s[2 * (K) + 1]};       \
    const unsigned h_ = cvt_pk_f16(p_);                  \
    H[I] = h_;                                           \
    L[I] = split_lo_pk_f16(p_, h_);                      \
  }
// (packed fp32 instructions sit at the END of a slice, behind >= 27 clocks of other work: by then the slice's MFMA has left the pipe)
#define H2P_EXP2(K)                                        \
  {                                                        \
    s[2 * (K)] = __builtin_amdgcn_exp2f(d[K].x);           \
    s[2 * (K) + 1] = __builtin_amdgcn_exp2f(d[K].y);       \
  }
#define H2P_SUM(K) ps += f32x2{s[2 * (K)], s[2 * (K) + 1]};
template <bool LOADK, bool LOADV, int OFF>
__device__ __forceinline__ void h2p_step(f32x16& s, f32x16& sn, f32x16& op, f32x16& oc, float& m_run, float& l_run, const f16x8_t (&qn)[2][2],
                                         f16x8_t (&kf)[2][2], f16x8_t (&vf)[2][2], f16x8_t (&pb)[2][2], const unsigned char* tk,
                                         const unsigned char* tv) {
  constexpr float INV = 1.0f / (H2_S * H2_S);
  if constexpr (LOADK) {                     // K of block b + 1 (its Q K^T starts at slice 6; the last reader of kf was the previous step)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) kf[t][q] = *reinterpret_cast<const f16x8_t*>(tk + (t * 2 + q) * 1024);
  }
  H2P_FENCE();
  op = mmh(vf[0][1], pb[0][0], op);
  H2P_FENCE();
  float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
  mx = fmaxf(fmaxf(mx, s[3]), s[4]);
  mx = fmaxf(fmaxf(mx, s[5]), s[6]);
  mx = fmaxf(fmaxf(mx, s[7]), s[8]);
  H2P_FENCE();
  op = mmh(vf[0][0], pb[0][1], op);
  H2P_FENCE();
  mx = fmaxf(fmaxf(mx, s[9]), s[10]);
  mx = fmaxf(fmaxf(mx, s[11]), s[12]);
  mx = fmaxf(fmaxf(mx, s[13]), s[14]);
  mx = fmaxf(mx, s[15]);
  float a, b;
  both_halves(mx, a, b);
  H2P_FENCE();
  op = mmh(vf[0][0], pb[0][0], op);
  H2P_FENCE();
  const float m_new = fmaxf(m_run, __builtin_ceilf(fmaxf(a, b) * INV));
  const bool moved = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0;
  const f32x2 inv2 = {INV, INV}, off2 = {(float)OFF - m_new, (float)OFF - m_new};
  f32x2 d[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = __builtin_elementwise_fma(f32x2{s[2 * k], s[2 * k + 1]}, inv2, off2);
  H2P_FENCE();
  op = mmh(vf[1][1], pb[1][0], op);
  H2P_FENCE();
  H2P_EXP2(0) H2P_EXP2(1)
#pragma unroll
  for (int k = 4; k < 6; ++k) d[k] = __builtin_elementwise_fma(f32x2{s[2 * k], s[2 * k + 1]}, inv2, off2);
  H2P_FENCE();
  op = mmh(vf[1][0], pb[1][1], op);
  H2P_FENCE();
  H2P_EXP2(2) H2P_EXP2(3)
#pragma unroll
  for (int k = 6; k < 8; ++k) d[k] = __builtin_elementwise_fma(f32x2{s[2 * k], s[2 * k + 1]}, inv2, off2);
  f32x2 ps = {0.f, 0.f};
  H2P_SUM(0) H2P_SUM(1)
  H2P_FENCE();
  op = mmh(vf[1][0], pb[1][0], op);
  H2P_FENCE();
  H2P_EXP2(4) H2P_EXP2(5)
  H2P_SUM(2) H2P_SUM(3)
  H2P_FENCE();
  if constexpr (LOADV) {                     // V of block b (its P V runs in the next step; the six MFMAs above were vf's last readers)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) vf[t][q] = *reinterpret_cast<const f16x8_t*>(tv + (t * 2 + q) * 1024);
  }
  {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sn = mmh(kf[0][1], qn[0][0], zero);
  }
  H2P_FENCE();
  H2P_EXP2(6) H2P_EXP2(7)
  H2P_SUM(4) H2P_SUM(5)
  H2P_FENCE();
  u32x4a h0, l0, h1, l1;
  sn = mmh(kf[0][0], qn[0][1], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h0, l0, 0, 0)
  H2P_SUM(6) H2P_SUM(7)
  both_halves(ps.x + ps.y, a, b);
  H2P_FENCE();
  sn = mmh(kf[0][0], qn[0][0], sn);
  H2P_FENCE();
  if (moved) {
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
    l_run *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[r] *= corr;
    m_run = m_new;
  }
  l_run += a + b;
  H2P_SPLIT_PAIR(h0, l0, 1, 1)
  H2P_FENCE();
  sn = mmh(kf[1][1], qn[1][0], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h0, l0, 2, 2)
  H2P_SPLIT_PAIR(h0, l0, 3, 3)
  H2P_FENCE();
  sn = mmh(kf[1][0], qn[1][1], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 0, 4)
  H2P_SPLIT_PAIR(h1, l1, 1, 5)
  H2P_FENCE();
  sn = mmh(kf[1][0], qn[1][0], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 2, 6)
  H2P_SPLIT_PAIR(h1, l1, 3, 7)
  pb[0][0] = __builtin_bit_cast(f16x8_t, h0);
  pb[0][1] = __builtin_bit_cast(f16x8_t, l0);
  pb[1][0] = __builtin_bit_cast(f16x8_t, h1);
  pb[1][1] = __builtin_bit_cast(f16x8_t, l1);
  H2P_FENCE();
}

// The same step with the lazy running maximum (above)
template <bool LOADK, bool LOADV, int OFF>
__device__ __forceinline__ void h2p_step_lazy(f32x16& s, f32x16& sn, f32x16& op, f32x16& oc, float& m_run, float& l_run, const f16x8_t (&qn)[2][2],
                                              f16x8_t (&kf)[2][2], f16x8_t (&vf)[2][2], f16x8_t (&pb)[2][2], const unsigned char* tk,
                                              const unsigned char* tv, bool& bad) {
  constexpr float INV = 1.0f / (H2_S * H2_S);
  if constexpr (LOADK) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) kf[t][q] = *reinterpret_cast<const f16x8_t*>(tk + (t * 2 + q) * 1024);
  }
  const float c = (float)OFF - m_run;
  const f32x2 inv2 = {INV, INV}, off2 = {c, c};
  f32x2 d[8];
  H2P_FENCE();
  op = mmh(vf[0][1], pb[0][0], op);
  H2P_FENCE();
  d[0] = f32x2{__builtin_fmaf(s[0], INV, c), __builtin_fmaf(s[1], INV, c)};          // (not packed: these run under the MFMA)
  d[1] = f32x2{__builtin_fmaf(s[2], INV, c), __builtin_fmaf(s[3], INV, c)};
  H2P_FENCE();
  op = mmh(vf[0][0], pb[0][1], op);
  H2P_FENCE();
  H2P_EXP2(0) H2P_EXP2(1)
#pragma unroll
  for (int k = 2; k < 8; ++k) d[k] = __builtin_elementwise_fma(f32x2{s[2 * k], s[2 * k + 1]}, inv2, off2);
  H2P_FENCE();
  op = mmh(vf[0][0], pb[0][0], op);
  H2P_FENCE();
  H2P_EXP2(2) H2P_EXP2(3)
  f32x2 ps = {0.f, 0.f};
  H2P_SUM(0) H2P_SUM(1)
  H2P_FENCE();
  op = mmh(vf[1][1], pb[1][0], op);
  H2P_FENCE();
  H2P_EXP2(4) H2P_EXP2(5)
  H2P_SUM(2) H2P_SUM(3)
  H2P_FENCE();
  op = mmh(vf[1][0], pb[1][1], op);
  H2P_FENCE();
  H2P_EXP2(6) H2P_EXP2(7)
  H2P_SUM(4) H2P_SUM(5)
  H2P_FENCE();
  op = mmh(vf[1][0], pb[1][0], op);
  H2P_FENCE();
  H2P_SUM(6) H2P_SUM(7)
  float a, b;
  both_halves(ps.x + ps.y, a, b);
  float rs = a + b;                                               // the row's sum over this block's 32 keys (both half-lanes hold it)
  if (__builtin_amdgcn_ballot_w64(!(rs < 32768.f)) != 0) h2_lazy_raise(s, oc, m_run, l_run, rs, bad);      // (rare)
  l_run += rs;
  H2P_FENCE();
  if constexpr (LOADV) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) vf[t][q] = *reinterpret_cast<const f16x8_t*>(tv + (t * 2 + q) * 1024);
  }
  {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sn = mmh(kf[0][1], qn[0][0], zero);
  }
  H2P_FENCE();
  u32x4a h0, l0, h1, l1;
  H2P_SPLIT_PAIR(h0, l0, 0, 0)
  H2P_SPLIT_PAIR(h0, l0, 1, 1)
  H2P_FENCE();
  sn = mmh(kf[0][0], qn[0][1], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h0, l0, 2, 2)
  H2P_SPLIT_PAIR(h0, l0, 3, 3)
  H2P_FENCE();
  sn = mmh(kf[0][0], qn[0][0], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 0, 4)
  H2P_FENCE();
  sn = mmh(kf[1][1], qn[1][0], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 1, 5)
  H2P_FENCE();
  sn = mmh(kf[1][0], qn[1][1], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 2, 6)
  H2P_FENCE();
  sn = mmh(kf[1][0], qn[1][0], sn);
  H2P_FENCE();
  H2P_SPLIT_PAIR(h1, l1, 3, 7)
  pb[0][0] = __builtin_bit_cast(f16x8_t, h0);
  pb[0][1] = __builtin_bit_cast(f16x8_t, l0);
  pb[1][0] = __builtin_bit_cast(f16x8_t, h1);
  pb[1][1] = __builtin_bit_cast(f16x8_t, l1);
  H2P_FENCE();
}

#undef H2P_SUM
#undef H2P_EXP2
#undef H2P_SPLIT_PAIR
#undef H2P_FENCE

// QB = query blocks of 32 per wave; PIPE (QB = 2, tokens a multiple of 256): the pipelined key loop, h2p_step; LAZY: the lazy
// running maximum (h2p_step_lazy) behind the exact first tile of every row -- pipelined loop only: in the phase-separated loop it
// measured neutral (profiles/r05_ab_attn_mix.txt) and both forms inlined cost a wave per SIMD
template <int QB, int PIPE, bool LAZY>
__global__ __launch_bounds__(256) void attn_h2_fwd_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ kv, float* __restrict__ out,
                                                          unsigned char* __restrict__ out_ps, int tokens, int heads, float scale,
                                                          int* __restrict__ range_flag, const int gx, const int remap) {
  static_assert((PIPE == 0 && !LAZY) || QB == 2, "the pipelined loop alternates two query blocks");
  __shared__ __attribute__((aligned(1024))) unsigned char smem_h[2 * H2_TILE];
  static_assert(2 * H2_TILE >= 4 * 32 * BA_FS * 4, "transpose buffers alias the tile buffers");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
  const int C = heads * BA_D, ld = 3 * C;
  // 1-D grid, XCD-aware (round 5): workgroups are dealt round-robin over the 8 XCDs, so the query tiles of one (sample, head) --
  // which all stream the same K / V tiles -- used to land on 8 different L2s and each fetched the stream from the fabric
  // (5.7 GB per 64x64x4 step by FETCH_SIZE, profiles/traffic_r05.json).  xcd_remap puts consecutive ids on ONE XCD: the
  // tiles of a head share that L2.  (remap = 0: the old order, A/B only)
  const int nwg = gx * heads * (int)gridDim.y;
  const int bid = remap ? xcd_remap(blockIdx.x + gx * heads * blockIdx.y, nwg) : (int)(blockIdx.x + gx * heads * blockIdx.y);
  const int qt = bid % gx, h = (bid / gx) % heads, b = bid / (gx * heads);
  const int q0 = qt * (128 * QB) + wave * (32 * QB);
  const float* base = qkv + (long long)b * tokens * ld;
  const bool wave_active = q0 < tokens;
  constexpr float LOG2E = 1.4426950408889634f;
  f16x8_t qf[QB][2][2];
  f32x16 o[QB];
  float m_run[QB], l_run[QB];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const bool q_valid = q0 + 32 * j + l31 < tokens;
    const float* rowp = base + (long long)(q_valid ? q0 + 32 * j + l31 : 0) * ld + h * BA_D;
    const float mul = q_valid ? scale * LOG2E : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half), c = *reinterpret_cast<const float4*>(rowp + 16 * t + 8 * half + 4);
      float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, c.x * mul, c.y * mul, c.z * mul, c.w * mul};
#pragma unroll
      for (int i = 0; i < 8; ++i) { bad |= out_of_h2_range(v[i]); v[i] = h2_clamp(v[i]) * H2_S; }
      split8_h2(v, qf[j][t]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    m_run[j] = -INFINITY;
    l_run[j] = 0.f;
  }
  if (bad) *range_flag = 1;
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  const unsigned char* kvh = kv + ((long long)b * heads + h) * ntiles * H2_TILE;
  au32x4 rs;
  {
    const unsigned long long a = (unsigned long long)kvh;
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    rs.z = __builtin_amdgcn_readfirstlane((unsigned)ntiles * (unsigned)H2_TILE);
    rs.w = 0x00020000u;
  }
  const unsigned lds0 = (unsigned)(size_t)smem_h;
  const unsigned woff = (unsigned)wave * 4096u + (unsigned)lane * 16u;
  auto fetch = [&](int kt) {                      // LDS-DMA of tile kt into buffer kt & 1 (tiles past the end: out of range, zeros)
    h2_dma4((unsigned)kt * (unsigned)H2_TILE + woff, rs, lds0 + (unsigned)(kt & 1) * H2_TILE + (unsigned)wave * 4096u);
  };
  if constexpr (PIPE != 0) {
    // block order (sub 0, q 0), (sub 0, q 1), (sub 1, q 0), (sub 1, q 1) per 64-key tile; step b: VALU of block b, MFMAs of P V (b - 1)
    // and Q K^T (b + 1).  K / V fragments are read from LDS once per 32 keys and kept for both query blocks.  Tile kt + 1 must
    // have landed before the last step of tile kt (its Q K^T reads K of tile kt + 1): the fetch of tile kt + 2 is issued there,
    // behind the barrier that also says every wave has read the last fragment of tile kt -- two buffers, as before.
    f32x16 sa, sb;
    f16x8_t kf[2][2], vf[2][2], pb[2][2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    fetch(1);
    {
      const unsigned char* tb = smem_h + lane * 16;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          kf[t][q] = *reinterpret_cast<const f16x8_t*>(tb + (t * 2 + q) * 1024);
          vf[t][q] = __builtin_bit_cast(f16x8_t, u32x4a{0u, 0u, 0u, 0u});        // (the first step's P V: zeros times zeros onto zeros)
          pb[t][q] = vf[t][q];
        }
      sa = mmh3(kf[0], qf[0][0], zero);
      sa = mmh3(kf[1], qf[0][1], sa);
    }
    constexpr int OFF = LAZY ? H2_LAZY_OFF : (int)H2_PEXP;
    auto tile = [&](const int kt, auto lazy) {
      constexpr bool LZ = decltype(lazy)::value;
      const unsigned char* cur = smem_h + (kt & 1) * H2_TILE + lane * 16;
      const unsigned char* nxt = smem_h + ((kt + 1) & 1) * H2_TILE + lane * 16;
      if constexpr (LZ) {
        h2p_step_lazy<false, true, OFF>(sa, sb, o[1], o[0], m_run[0], l_run[0], qf[1], kf, vf, pb, nullptr, cur + 8 * 1024, bad);
        h2p_step_lazy<true, false, OFF>(sb, sa, o[0], o[1], m_run[1], l_run[1], qf[0], kf, vf, pb, cur + 4 * 1024, nullptr, bad);
        h2p_step_lazy<false, true, OFF>(sa, sb, o[1], o[0], m_run[0], l_run[0], qf[1], kf, vf, pb, nullptr, cur + 12 * 1024, bad);
      } else {
        h2p_step<false, true, OFF>(sa, sb, o[1], o[0], m_run[0], l_run[0], qf[1], kf, vf, pb, nullptr, cur + 8 * 1024);          // V, keys 0-31
        h2p_step<true, false, OFF>(sb, sa, o[0], o[1], m_run[1], l_run[1], qf[0], kf, vf, pb, cur + 4 * 1024, nullptr);          // K, keys 32-63
        h2p_step<false, true, OFF>(sa, sb, o[1], o[0], m_run[0], l_run[0], qf[1], kf, vf, pb, nullptr, cur + 12 * 1024);         // V, keys 32-63
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      fetch(kt + 2);                                                                  // (past the end: out of range, zeros)
      if constexpr (LZ) h2p_step_lazy<true, false, OFF>(sb, sa, o[0], o[1], m_run[1], l_run[1], qf[0], kf, vf, pb, nxt, nullptr, bad);
      else h2p_step<true, false, OFF>(sb, sa, o[0], o[1], m_run[1], l_run[1], qf[0], kf, vf, pb, nxt, nullptr);              // K, keys 0-31 of tile kt + 1
    };
    if constexpr (LAZY) {
      tile(0, std::false_type{});                            // the first tile sets the running maxima
      for (int kt = 1; kt < ntiles; ++kt) tile(kt, std::true_type{});
    } else {
      for (int kt = 0; kt < ntiles; ++kt) tile(kt, std::false_type{});
    }
    o[1] = mmh3(vf[0], pb[0], o[1]);           // P V of the last block
    o[1] = mmh3(vf[1], pb[1], o[1]);
    if constexpr (LAZY) {
      if (bad) *range_flag = 1;                // (h2_lazy_raise: a block overflowed the fp32 exponential)
    }
  } else {
  fetch(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    fetch(kt + 1);
    if (!wave_active) continue;
    const unsigned char* tb = smem_h + (kt & 1) * H2_TILE + lane * 16;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int key0 = kt * BA_T + sub * 32;
      if (key0 >= tokens) break;
      f32x16 s[QB];
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8_t ka[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) ka[q] = *reinterpret_cast<const f16x8_t*>(tb + ((2 * sub + t) * 2 + q) * 1024);
#pragma unroll
        for (int j = 0; j < QB; ++j) s[j] = mmh3(ka, qf[j][t], s[j]);                      // 2^12 S^T[key][q], log2 domain
      }
      if (key0 + 32 > tokens) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) s[j][r] = -INFINITY;
      }
#pragma unroll
      for (int j = 0; j < QB; ++j) h2_softmax(s[j], o[j], m_run[j], l_run[j]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8_t va[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) va[q] = *reinterpret_cast<const f16x8_t*>(tb + ((4 + 2 * sub + t) * 2 + q) * 1024);
#pragma unroll
        for (int j = 0; j < QB; ++j) {
          float pv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) pv[i] = s[j][8 * t + i];
          f16x8_t pb[2];
          split8_h2(pv, pb);
          o[j] = mmh3(va, pb, o[j]);                                                       // 2^20 O^T[d][q] += V^T P^T
        }
      }
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!wave_active) return;
#pragma unroll
  for (int j = 0; j < QB; ++j) {
    const int qb0 = q0 + 32 * j;
    if (qb0 >= tokens) break;
    const float mul = (1.0f / H2_S) / l_run[j];
    if (out_ps) {
      // the result as the pre-split A operand of attn1.to_out in the F16X2 form of the PS layout (two fp16 planes of 2^6 x, 2-KiB
      // units; csrc/igemm_ps.hip): written from the accumulators as in attn_x3p_fwd_kernel; tokens % 32 == 0
      unsigned char* d0 = out_ps + ((((long long)b * tokens + qb0) >> 5) * (C / 16) + 2 * h) * 2048 + l31 * 16 + half * 8;
      if (qb0 + l31 < tokens) {
        bool bad2 = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4] = {mul_rounded(o[j][4 * g], mul), mul_rounded(o[j][4 * g + 1], mul), mul_rounded(o[j][4 * g + 2], mul), mul_rounded(o[j][4 * g + 3], mul)};
          asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
          typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
          f16x4_t hh, ll;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bad2 |= out_of_h2_range(v[e]);
            const float sv = h2_clamp(v[e]) * H2_S;
            hh[e] = (_Float16)sv;
            ll[e] = (_Float16)(sv - (float)hh[e]);
          }
          unsigned char* d = d0 + (g >> 1) * 2048 + (g & 1) * 512;
          *reinterpret_cast<f16x4_t*>(d) = hh;
          *reinterpret_cast<f16x4_t*>(d + 1024) = ll;
        }
        if (bad2) *range_flag = 1;
      }
    }
    if (out) {
      float* ts = reinterpret_cast<float*>(smem_h) + wave * (32 * BA_FS);
      store_rows_bf(ts, o[j], mul, out + (long long)b * tokens * C + h * BA_D, C, qb0, tokens, l31, half);
    }
  }
}

}  // namespace ldmk

extern "C" int ldmk_attn_self_x3(const float* qkv, float* out, int n, int tokens, int heads, float scale, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && out && n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535, "ldmk_attn_self_x3: bad args");
  dim3 grid((tokens + 127) / 128, heads, n);
  hipLaunchKernelGGL(attn_x3_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, tokens, heads, scale);
  return check_launch("ldmk_attn_self_x3");
}

extern "C" long long ldmk_attn_kv_split_bytes(int n, int tokens, int heads) {
  if (n <= 0 || tokens <= 0 || heads <= 0) return -1;
  return (long long)n * heads * ((tokens + ldmk::BA_T - 1) / ldmk::BA_T) * ldmk::X3P_TILE;
}

extern "C" int ldmk_attn_self_x3p(const float* qkv, void* kv_scratch, float* out, int n, int tokens, int heads, float scale, void* stream) {
  return ldmk_attn_self_x3p_ps(qkv, kv_scratch, out, nullptr, n, tokens, heads, scale, stream);
}

extern "C" int ldmk_attn_self_x3p_ps(const float* qkv, void* kv_scratch, float* out, void* out_ps, int n, int tokens, int heads, float scale,
                                     void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && kv_scratch && (out || out_ps) && n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535, "ldmk_attn_self_x3p: bad args");
  LDMK_REQUIRE(!out_ps || tokens % 32 == 0, "ldmk_attn_self_x3p_ps: out_ps needs tokens %% 32 == 0 (%d)", tokens);
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  LDMK_REQUIRE((long long)ntiles * X3P_TILE < (1LL << 31), "ldmk_attn_self_x3p: %d tokens: a head's pre-split K / V exceeds 2 GiB", tokens);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_kv_split_kernel, dim3(ntiles, heads, n), dim3(256), 0, st, qkv, reinterpret_cast<unsigned char*>(kv_scratch), tokens, heads);
  // 64 queries per wave from X3P_QB2_MIN_TOKENS tokens (below, the grid of 256-query workgroups no longer fills the chip);
  // LDMK_ATTN_QB = 1 | 2 pins the form (measurements).  The result does not depend on it, bit for bit.
  static const int qb_env = [] { const char* e = getenv("LDMK_ATTN_QB"); return e ? atoi(e) : 0; }();
  const int qb = qb_env == 1 || qb_env == 2 ? qb_env : (tokens >= X3P_QB2_MIN_TOKENS ? 2 : 1);
  if (qb == 2)
    hipLaunchKernelGGL(attn_x3p_fwd_kernel<2>, dim3((tokens + 255) / 256, heads, n), dim3(256), 0, st, qkv,
                       reinterpret_cast<const unsigned char*>(kv_scratch), out, reinterpret_cast<unsigned char*>(out_ps), tokens, heads, scale);
  else
    hipLaunchKernelGGL(attn_x3p_fwd_kernel<1>, dim3((tokens + 127) / 128, heads, n), dim3(256), 0, st, qkv,
                       reinterpret_cast<const unsigned char*>(kv_scratch), out, reinterpret_cast<unsigned char*>(out_ps), tokens, heads, scale);
  return check_launch("ldmk_attn_self_x3p");
}

extern "C" long long ldmk_attn_kv_split_h2_bytes(int n, int tokens, int heads) {
  if (n <= 0 || tokens <= 0 || heads <= 0) return -1;
  return (long long)n * heads * ((tokens + ldmk::BA_T - 1) / ldmk::BA_T) * ldmk::H2_TILE;
}

extern "C" int ldmk_attn_self_h2_ps(const float* qkv, void* kv_scratch, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                                    float scale, void* stream);

extern "C" int ldmk_attn_self_h2(const float* qkv, void* kv_scratch, float* out, int* range_flag, int n, int tokens, int heads, float scale,
                                 void* stream) {
  return ldmk_attn_self_h2_ps(qkv, kv_scratch, out, nullptr, range_flag, n, tokens, heads, scale, stream);
}

static int attn_self_h2_any(const float* qkv, void* kv_scratch, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                            float scale, bool prepass, void* stream);

extern "C" int ldmk_attn_self_h2_ps(const float* qkv, void* kv_scratch, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                                    float scale, void* stream) {
  return attn_self_h2_any(qkv, kv_scratch, out, out_ps, range_flag, n, tokens, heads, scale, true, stream);
}

extern "C" int ldmk_attn_self_h2_tiles(const float* qkv, const void* kv_tiles, float* out, void* out_ps, int* range_flag, int n, int tokens,
                                       int heads, float scale, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(tokens % 64 == 0, "ldmk_attn_self_h2_tiles: tokens=%d must be a multiple of 64 (whole K / V tiles)", tokens);
  return attn_self_h2_any(qkv, const_cast<void*>(kv_tiles), out, out_ps, range_flag, n, tokens, heads, scale, false, stream);
}

static int attn_self_h2_any(const float* qkv, void* kv_scratch, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                            float scale, bool prepass, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && kv_scratch && (out || out_ps) && range_flag && n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535,
               "ldmk_attn_self_h2: bad args");
  LDMK_REQUIRE(!out_ps || tokens % 32 == 0, "ldmk_attn_self_h2_ps: out_ps needs tokens %% 32 == 0 (%d)", tokens);
  const int ntiles = (tokens + BA_T - 1) / BA_T;
  LDMK_REQUIRE((long long)ntiles * H2_TILE < (1LL << 31), "ldmk_attn_self_h2: %d tokens: a head's pre-split K / V exceeds 2 GiB", tokens);
  hipStream_t st = (hipStream_t)stream;
  if (prepass)
    hipLaunchKernelGGL(attn_kv_split_h2_kernel, dim3(ntiles, heads, n), dim3(256), 0, st, qkv, reinterpret_cast<unsigned char*>(kv_scratch), tokens,
                       heads, range_flag);
  static const int qb_env = [] { const char* e = getenv("LDMK_ATTN_QB"); return e ? atoi(e) : 0; }();
  const int qb = qb_env == 1 || qb_env == 2 ? qb_env : (tokens >= X3P_QB2_MIN_TOKENS ? 2 : 1);
  static const int remap = [] { const char* e = getenv("LDMK_ATTN_XCD"); return e ? atoi(e) : 1; }();
  const int gx = qb == 2 ? (tokens + 255) / 256 : (tokens + 127) / 128;
  // (grid.x = query tiles x heads, grid.y = samples: the kernel linearises and re-deals the ids over the XCDs itself)
  // the pipelined key loop with the lazy running maximum wherever every wave has whole tiles (two query blocks per wave, tokens a
  // multiple of 256).  LDMK_ATTN_PIPE=1: pipelined with the exact maximum per block (the same bits as the phase-separated loop),
  // LDMK_ATTN_PIPE=0: the phase-separated loop (A/B: profiles/r05_ab_attn_pipe.txt)
  static const int pipe_env = [] { const char* e = getenv("LDMK_ATTN_PIPE"); return e ? atoi(e) : 2; }();
#define LDMK_ATTN_H2_LAUNCH(QB_, PIPE_, LAZY_)                                                                                                    \
  hipLaunchKernelGGL((attn_h2_fwd_kernel<QB_, PIPE_, LAZY_>), dim3(gx * heads, n), dim3(256), 0, st, qkv, reinterpret_cast<const unsigned char*>(kv_scratch), \
                     out, reinterpret_cast<unsigned char*>(out_ps), tokens, heads, scale, range_flag, gx, remap)
  const bool piped = qb == 2 && pipe_env != 0 && tokens % 256 == 0;
  if (piped && pipe_env == 2) LDMK_ATTN_H2_LAUNCH(2, 1, true);
  else if (piped) LDMK_ATTN_H2_LAUNCH(2, 1, false);
  else if (qb == 2) LDMK_ATTN_H2_LAUNCH(2, 0, false);
  else LDMK_ATTN_H2_LAUNCH(1, 0, false);
#undef LDMK_ATTN_H2_LAUNCH
  return check_launch("ldmk_attn_self_h2");
}

extern "C" int ldmk_attn_self_lse_bf16(const float* qkv, float* out, float* lse, int n, int tokens, int heads, float scale,
                                       void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && out && n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535, "ldmk_attn_self_lse_bf16: bad args");
  dim3 grid((tokens + 127) / 128, heads, n);
  hipLaunchKernelGGL(attn_bf16_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, lse, tokens, heads, scale);
  return check_launch("ldmk_attn_self_lse_bf16");
}

extern "C" int ldmk_attn_self_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                       float* dsum, int n, int tokens, int heads, float scale, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(qkv && out && dout && lse && dqkv && dsum, "ldmk_attn_self_bwd_bf16: null buffer");
  LDMK_REQUIRE(n > 0 && tokens > 0 && heads > 0 && heads <= 65535 && n <= 65535, "ldmk_attn_self_bwd_bf16: bad shape");
  hipStream_t st = (hipStream_t)stream;
  attn_rowdot_launch(dout, out, dsum, tokens, heads, (long long)n * tokens * heads, st);
  dim3 grid((tokens + 127) / 128, heads, n);
  hipLaunchKernelGGL(attn_bf16_dq_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dsum, dqkv, tokens, heads, scale);
  hipLaunchKernelGGL(attn_bf16_dkv_kernel, grid, dim3(256), 0, st, qkv, dout, lse, dsum, dqkv, tokens, heads, scale);
  return check_launch("ldmk_attn_self_bwd_bf16");
}
