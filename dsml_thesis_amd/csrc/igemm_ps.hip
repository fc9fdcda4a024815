// Implicit GEMM of the fp32-accurate bf16x3 arithmetic (LDMK_COMPUTE_BF16X3, include/ldmk.h) on PRE-SPLIT operands:
// tile_cfg 23..30.
//
// Round 3 found the bf16x3 kernels (igemm.hip BF = 3, igemm_ws.hip) bound by what surrounds their matrix instructions: every
// A element is loaded to registers, split three ways (~5.5 vector operations) and stored to LDS once per N-tile, next to waves
// that try to issue MFMAs on the same SIMD (matrix pipe 34 % busy over a step).  Here NO operand passes through a register on
// its way to LDS and no arithmetic is done on it:
//   * both operands arrive already split, in the "PS" layout (include/ldmk.h): for a matrix X[R][K], the three bf16 planes of
//     each (32-row block, 16-deep k-slab) are 3 x 1 KiB contiguous, each KiB in the lane order of the MFMA operand
//     (lane = 32 (k / 8 % 2) + row % 32 holds 8 consecutive k: 16 bytes).  Producers write it: ldmk_pack_ps (weights, once),
//     ldmk_ln_stats_ps (the LayerNorm statistics pass), the transposed epilogue of this kernel (out_ps), the Winograd /
//     GroupNorm-apply passes;
//   * a stage (16 k of a BM x BN tile) is filled by `buffer_load_dwordx4 ... lds` (LDS-DMA): one wave instruction moves one
//     plane of one block, 1 KiB, memory -> LDS; a wave issues 2-3 groups of three per stage and nothing else;
//   * fragments are read back with one conflict-free ds_read_b128 per operand (the LDS image IS the fragment order);
//   * NS-deep ring of stages, the DMA of stage s + NS - 1 issued right after the barrier that frees its buffer, completion
//     counted with s_waitcnt vmcnt(N) (the loads are inline asm: the compiler neither counts nor drains them), one raw
//     s_barrier per stage.
// The products, their order inside an accumulator (six bf16 MFMAs per 16 k, smallest partial product first) and the split-K
// partition are those of igemm_kernel<BF = 3>: results are bitwise equal to tile_cfg 1 / 5 at equal splitk
// (tests/test_ps_gpu.py).  Epilogues: the lane = column form of igemm_ws.hip (bias, per-sample vector, residual, folded
// LayerNorm, GEGLU, GroupNorm records) or, TR = true, with the accumulators TRANSPOSED (operands swapped in the MFMA: lane =
// row, registers = 4 x 4 consecutive columns), which stores float4 rows and can write the result pre-split in the PS layout
// for the next GEMM (args.out_ps) -- each activation element is then split exactly once, by its producer.
#include "ldmk_common.h"
#include <stdlib.h>

namespace ldmk {

typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

// What-if probes (drop the DMA / the matrix instructions / the epilogue, phase stamps, phase stagger: results are garbage with them)
// exist only in builds made with -DLDMK_PS_PROBES (tools/ps_probe.sh, tools/pw_stamps.py); the shipped library ignores LDMK_PS_DEBUG /
// LDMK_PS_STAGGER and its kernels carry none of the branches.
#ifdef LDMK_PS_PROBES
constexpr bool PS_PROBES = true;
#else
constexpr bool PS_PROBES = false;
#endif
static int ps_probe_bits() {
  if (!PS_PROBES) return 0;
  const int dbg = ps_probe_bits();
  return dbg;
}

constexpr unsigned PS_OOB = 0x80000000u;      // a byte offset beyond every buffer here (< 2 GiB each): reads as zeros, no memory traffic

__device__ __forceinline__ pu32x4 ps_rsrc(const void* ptr, unsigned bytes) {      // raw buffer descriptor, uniform -> SGPRs
  const unsigned long long a = (unsigned long long)ptr;
  pu32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}

// One unit = the three planes of one 32-row block at one k-slab: 3 KiB contiguous in memory and in the LDS stage.
// LDS destination = M0 + instruction offset + 16 lane; memory address = base + voff + instruction offset (probed on gfx950:
// tools/probe/dma_probe.hip); M0 is saved / restored around the group (the compiler owns it outside the statement).
__device__ __forceinline__ void ps_dma3(unsigned voff, const pu32x4& rs, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// PL = planes per operand: 3 (bf16x3: hi, mid, lo bf16) or 2 (F16X2: hi, lo fp16 of the operand scaled into fp16's range; the
// same layout with 2-KiB units).  One unit = the PL planes of one 32-row block at one k-slab.
template <int PL>
__device__ __forceinline__ void ps_dma(unsigned voff, const pu32x4& rs, unsigned lds_dst) {
  if constexpr (PL == 3) {
    ps_dma3(voff, rs, lds_dst);
  } else {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                 "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
  }
}

typedef _Float16 pf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
constexpr float PS_H2_SCALE = 64.f;            // 2^LDMK_F16X2_A_EXP
__device__ __forceinline__ void ps_split2h(const float4& v, pf16x4& h, pf16x4& l) {       // v already scaled; the split of igemm.hip's split2h
  h = pf16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
  const pu32x2 hu = __builtin_bit_cast(pu32x2, h);
  l = __builtin_bit_cast(pf16x4, pu32x2{h2_lo_pair(hu.x, v.x, v.y), h2_lo_pair(hu.y, v.z, v.w)});
}
__device__ __forceinline__ bool ps_h2_out_of_range(const float4& v) {                       // |x| >= LDMK_F16X2_RANGE, inf or NaN
  constexpr unsigned LIM = 0x447a0000u;
  return (__float_as_uint(v.x) & 0x7fffffffu) >= LIM || (__float_as_uint(v.y) & 0x7fffffffu) >= LIM ||
         (__float_as_uint(v.z) & 0x7fffffffu) >= LIM || (__float_as_uint(v.w) & 0x7fffffffu) >= LIM;
}
// (activations, s = 2^6: saturated just inside the range -- h2_clamp, ldmk_common.h; weights, any s: their exponent is chosen at pack time)
__device__ __forceinline__ float4 ps_scaled(const float4& v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }
__device__ __forceinline__ float4 ps_scaled_sat(const float4& v) {
  return make_float4(h2_clamp(v.x) * PS_H2_SCALE, h2_clamp(v.y) * PS_H2_SCALE, h2_clamp(v.z) * PS_H2_SCALE, h2_clamp(v.w) * PS_H2_SCALE);
}

template <int N> __device__ __forceinline__ void ps_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

__device__ __forceinline__ pbf16x4 ps_bf4(const float4& v) { return pbf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; }
__device__ __forceinline__ void ps_split3(const float4& v, pbf16x4& h, pbf16x4& m, pbf16x4& l) {      // exact: h + m + l == v
  h = ps_bf4(v);
  const float4 r = make_float4(v.x - (float)h[0], v.y - (float)h[1], v.z - (float)h[2], v.w - (float)h[3]);
  m = ps_bf4(r);
  l = ps_bf4(make_float4(r.x - (float)m[0], r.y - (float)m[1], r.z - (float)m[2], r.w - (float)m[3]));
}

// Zeros for the optional epilogue operands: a NULL bias / column sum / per-sample vector / residual reads these instead, so the
// epilogue has no branch around its loads (they are all requested before the first use) and adds exact zeros.
__device__ __attribute__((aligned(16))) const float kPsZeros[4] = {0.f, 0.f, 0.f, 0.f};

// The epilogue of one wave: acc[TM][TN] 32x32 tiles at (rowbase, colbase).  TR: transposed accumulators (lane = row).
// KV (F16X2, transposed epilogue, the fused QKV projection): the K and V column tiles of the result are not stored as fp32 but written
// straight as the self attention's pre-split tiles (csrc/attention_bf16.hip: attn_h2_fwd_kernel; per (sample, head, 64 keys) 8
// fragments x 2 fp16 planes of 1 KiB) -- what ldmk_attn_self_h2's pre-pass would make of them, bit for bit, without the fp32 round
// trip.  A K fragment wants lane = (key, half) with 8 consecutive d: the transposed accumulator has 4 + 4 of them in the two
// half-lanes, one v_permlane32_swap per register pairs them up.  A V^T fragment wants lane = d: one pass through a per-wave LDS
// scratch (`kv_ts`, 32 x 33 floats; the ring is free by then).
constexpr int PS_KV_TILE = 16 * 1024;          // = H2_TILE of csrc/attention_bf16.hip
// the folded LayerNorm of one accumulator value + bias: rstd (acc - mean colsum) + b with the roundings of igemm.hip / rgemm.hip --
// one fma, one multiply, one add.  Contraction is switched off here: in the general epilogue a select sits between the multiply
// and the add, in the lean one nothing does, and the compiler would fuse them into a second fma (one rounding less: other bits).
__device__ __forceinline__ float ps_lnf_bias(float acc, float mean, float cs, float rstd, float bias) {
#pragma clang fp contract(off)
  const float t = __builtin_fmaf(-mean, cs, acc) * rstd;
  return t + bias;
}
// the same for a launch without a folded LayerNorm: alpha acc + b (+ 0 for the absent per-sample vector) + residual, each its own
// rounding as in the general epilogue (where the `if (lnf)` select sits between the multiply and the first add)
__device__ __forceinline__ float ps_bias_res(float acc_alpha, float bias, float vec, float res) {
#pragma clang fp contract(off)
  float t = acc_alpha + bias;
  t = t + vec;                 // (the per-sample vector, or +0 where the general form adds its zero)
  return t + res;
}
// lane = column form: alpha acc + bias [+ per-sample vector | + residual], each addition its own rounding as in the general form
__device__ __forceinline__ float ps_col_finish(float acc_alpha, float bias) {
#pragma clang fp contract(off)
  return acc_alpha + bias;
}
__device__ __forceinline__ float ps_col_finish(float acc_alpha, float bias, float extra) {
#pragma clang fp contract(off)
  const float t = acc_alpha + bias;
  return t + extra;
}
__device__ __forceinline__ float ps_col_finish(float acc_alpha, float bias, float vec, float res) {
#pragma clang fp contract(off)
  float t = acc_alpha + bias;
  t = t + vec;
  return t + res;
}
// LEAN != 0 (transposed form; chosen per wave when the whole wave tile is inside M x N, no split-K, no per-sample vector): no row /
// column predicates, no loads of absent operands, no select around the LayerNorm arithmetic -- the operand set is a template
// argument: 1 = folded LayerNorm, no residual (the GEGLU and QKV projections), 2 = residual, no LayerNorm (attn.to_out, ff.net.2),
// 3 = neither (Winograd planes, upsampling phases), 4 = residual + per-sample vector (attn1.to_out carrying the single-token
// cross-attention vector).  Lane = column form (the convolutions: bias, then the time-embedding vector
// or the skip connection, GroupNorm records): 1 = per-sample vector, 2 = residual, 3 = neither, 4 = both -- there every element of the
// general form sits behind its own row predicate and operand branches, so each residual load waited for itself (577 s_waitcnt in
// the 160 -> 160 convolution's kernel); the lean form asks for the 16 residuals of a tile at once.  Same arithmetic, rounding by
// rounding, as the general form (0).  The GEGLU
// epilogue was ~3000 vector instructions per wave for 64 outputs per lane, a third of them addressing and predication.
template <int TM, int TN, bool TR, int PL = 3, bool KV = false, int LEAN = 0>
__device__ __forceinline__ void ps_epilogue(const ldmk_igemm_args& p, f32x16 (&acc)[TM][TN], const int rowbase, const int colbase,
                                            const int splitk, const int ks, const int bz, float* __restrict__ ws, const int lane,
                                            float* __restrict__ kv_ts = nullptr) {
  const int l31 = lane & 31, half = lane >> 5;
  const float alpha = p.alpha;
  constexpr bool L_ANY = LEAN != 0, L_LNF = LEAN == 1, L_RS = LEAN == 2 || LEAN == 4, L_BV = LEAN == 4;
  const bool lnf = L_ANY ? L_LNF : p.a_tf == LDMK_TF_LAYERNORM_FOLDED;      // (lean forms: compile-time -- no select per value)
  const float2* __restrict__ stats2 = reinterpret_cast<const float2*>(p.row_stats);

  if constexpr (TR) {
    // ---- transposed accumulators: acc[i][j][r] = C[rowbase + 32 i + l31][colbase + 32 j + 8 (r >> 2) + 4 half + (r & 3)]
    if (splitk > 1) {
      float* slab = ws + ((long long)bz * splitk + ks) * p.M * p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = rowbase + 32 * i + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = colbase + 32 * j + 8 * q + 4 * half;
            if (col < p.N)
              *reinterpret_cast<float4*>(slab + (long long)row * p.N + col) =
                  make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          }
      }
      return;
    }
    float* __restrict__ outp = p.out ? p.out + (long long)bz * p.out_bstride : nullptr;
    unsigned char* __restrict__ ops_ = reinterpret_cast<unsigned char*>(p.out_ps);
    const bool geglu = p.epi == LDMK_EPI_GEGLU;
    const int Kbo = p.ldc / 16;                     // k-slabs per row block of the PS output (ldc = its column count)
    // optional operands: a missing one reads kPsZeros (column / row offsets masked to 0), so every load below is unconditional
    const float* __restrict__ csp = lnf ? p.ln_colsum : kPsZeros;
    const float* __restrict__ bip = p.bias ? p.bias : kPsZeros;
    const float* __restrict__ rsp = p.residual ? p.residual + (long long)bz * p.out_bstride : kPsZeros;
    const unsigned mcs = lnf ? ~0u : 0u, mbi = p.bias ? ~0u : 0u, mrs = p.residual ? ~0u : 0u, mbv = p.batch_vec ? ~0u : 0u;
    bool bad_ps = false;            // a PS output outside the F16X2 range: one store of the flag at the end, not a branch per 4 values
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = rowbase + 32 * i + l31;
      const bool rok = L_ANY || row < p.M;
      const int rr = rok ? row : p.M - 1;
      float mean = 0.f, rstd = 1.f;
      if (lnf) { const float2 st = stats2[rr]; mean = st.x; rstd = st.y; }
      const float* __restrict__ bvp = p.batch_vec ? p.batch_vec + (long long)(rr / p.rows_per_sample) * p.batch_vec_ld : kPsZeros;
      const unsigned rowoff = (unsigned)rr * (unsigned)p.ldc;
      const long long psrow = ((long long)((rowbase + 32 * i) >> 5) * Kbo) * (PL * 1024) + l31 * 16 + half * 8;
      constexpr int JS = (TN % 2 == 0) ? 2 : 1;      // GEGLU (even TN only): tile j = values, tile j + 1 = gates
#pragma unroll
      for (int j = 0; j < TN; j += JS) {
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
          if (geglu && jj == 1) break;
          const int jt = j + jj;
          const int ctile = colbase + 32 * jt;
          if (!L_ANY && ctile >= p.N) continue;
          const int otile = geglu ? (ctile >> 1) : ctile;            // first output column of this tile
          float kvv[KV ? 16 : 1];                                    // KV: the finished values of a K / V tile (column 8 q + 4 half + e at [4 q + e])
          const bool kv_tile = KV && ctile >= p.N / 3;
#pragma unroll
          for (int qh = 0; qh < 4; qh += 2) {                        // (two column groups at a time: registers)
          float4 cs[4], bi[4], bv[4], rs[4], csg[4], big[4];
#pragma unroll
          for (int q = qh; q < qh + 2; ++q) {                        // their operands are all requested before the first use
            const unsigned c = (unsigned)(ctile + 8 * q + 4 * half), oc = (unsigned)(otile + 8 * q + 4 * half);
            if constexpr (!L_ANY || L_LNF) cs[q] = *reinterpret_cast<const float4*>(csp + (c & mcs));
            bi[q] = *reinterpret_cast<const float4*>(bip + (c & mbi));
            if constexpr (!L_ANY) {
              bv[q] = *reinterpret_cast<const float4*>(bvp + (oc & mbv));
              rs[q] = *reinterpret_cast<const float4*>(rsp + ((rowoff + oc) & mrs));
            } else {
              if constexpr (L_RS) rs[q] = *reinterpret_cast<const float4*>(rsp + (rowoff + oc));
              if constexpr (L_BV) bv[q] = *reinterpret_cast<const float4*>(bvp + oc);
            }
            if constexpr (TN % 2 == 0) {
              if (geglu) {
                if constexpr (!L_ANY || L_LNF) csg[q] = *reinterpret_cast<const float4*>(csp + ((c + 32) & mcs));
                big[q] = *reinterpret_cast<const float4*>(bip + ((c + 32) & mbi));
              }
            }
          }
#pragma unroll
          for (int q = qh; q < qh + 2; ++q) {
            float4 v = make_float4(acc[i][jt][4 * q] * alpha, acc[i][jt][4 * q + 1] * alpha, acc[i][jt][4 * q + 2] * alpha,
                                   acc[i][jt][4 * q + 3] * alpha);
            if constexpr (L_LNF) {
              v.x = ps_lnf_bias(v.x, mean, cs[q].x, rstd, bi[q].x); v.y = ps_lnf_bias(v.y, mean, cs[q].y, rstd, bi[q].y);
              v.z = ps_lnf_bias(v.z, mean, cs[q].z, rstd, bi[q].z); v.w = ps_lnf_bias(v.w, mean, cs[q].w, rstd, bi[q].w);
            } else if constexpr (L_ANY) {        // (bias, +0, residual: in ps_bias_res below, once the GEGLU gate is through)
            } else {
              if (lnf) {      // same arithmetic as igemm.hip / rgemm.hip / igemm_reduce_kernel
                v.x = fmaf(-mean, cs[q].x, v.x) * rstd; v.y = fmaf(-mean, cs[q].y, v.y) * rstd;
                v.z = fmaf(-mean, cs[q].z, v.z) * rstd; v.w = fmaf(-mean, cs[q].w, v.w) * rstd;
              }
              v.x += bi[q].x; v.y += bi[q].y; v.z += bi[q].z; v.w += bi[q].w;
            }
            if constexpr (TN % 2 == 0) {
              if (geglu) {
                constexpr int TNm1 = TN - 1;
                const int jg = jt + 1 < TN ? jt + 1 : TNm1;       // (jt + 1 < TN whenever this branch runs)
                float4 g = make_float4(acc[i][jg][4 * q] * alpha, acc[i][jg][4 * q + 1] * alpha, acc[i][jg][4 * q + 2] * alpha,
                                       acc[i][jg][4 * q + 3] * alpha);
                if constexpr (L_LNF) {
                  g.x = ps_lnf_bias(g.x, mean, csg[q].x, rstd, big[q].x); g.y = ps_lnf_bias(g.y, mean, csg[q].y, rstd, big[q].y);
                  g.z = ps_lnf_bias(g.z, mean, csg[q].z, rstd, big[q].z); g.w = ps_lnf_bias(g.w, mean, csg[q].w, rstd, big[q].w);
                } else {
                  if (lnf) {
                    g.x = fmaf(-mean, csg[q].x, g.x) * rstd; g.y = fmaf(-mean, csg[q].y, g.y) * rstd;
                    g.z = fmaf(-mean, csg[q].z, g.z) * rstd; g.w = fmaf(-mean, csg[q].w, g.w) * rstd;
                  }
                  g.x += big[q].x; g.y += big[q].y; g.z += big[q].z; g.w += big[q].w;
                }
                v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w);
              }
            }
            if constexpr (L_LNF) {      // (the two absent operands add +0 twice: once is the same bits, -0 -> +0 included)
              v.x += 0.f; v.y += 0.f; v.z += 0.f; v.w += 0.f;
            } else if constexpr (L_ANY) {
              const float4 r4 = L_RS ? rs[q] : make_float4(0.f, 0.f, 0.f, 0.f), b4 = L_BV ? bv[q] : make_float4(0.f, 0.f, 0.f, 0.f);
              v.x = ps_bias_res(v.x, bi[q].x, b4.x, r4.x); v.y = ps_bias_res(v.y, bi[q].y, b4.y, r4.y);
              v.z = ps_bias_res(v.z, bi[q].z, b4.z, r4.z); v.w = ps_bias_res(v.w, bi[q].w, b4.w, r4.w);
            } else {
              v.x += bv[q].x; v.y += bv[q].y; v.z += bv[q].z; v.w += bv[q].w;
              v.x += rs[q].x; v.y += rs[q].y; v.z += rs[q].z; v.w += rs[q].w;
            }
            if constexpr (KV) {
              if (kv_tile) { kvv[4 * q] = v.x; kvv[4 * q + 1] = v.y; kvv[4 * q + 2] = v.z; kvv[4 * q + 3] = v.w; }
            }
            if (rok && !kv_tile) {
              const int oc = otile + 8 * q + 4 * half;               // output column
              if (outp) *reinterpret_cast<float4*>(outp + rowoff + oc) = v;
              if (ops_) {
                unsigned char* d = ops_ + psrow + (long long)((otile >> 4) + (q >> 1)) * (PL * 1024) + (q & 1) * 512;
                if constexpr (PL == 3) {
                  pbf16x4 h, m, l;
                  ps_split3(v, h, m, l);
                  *reinterpret_cast<pbf16x4*>(d) = h;
                  *reinterpret_cast<pbf16x4*>(d + 1024) = m;
                  *reinterpret_cast<pbf16x4*>(d + 2048) = l;
                } else {
                  bad_ps |= ps_h2_out_of_range(v);
                  pf16x4 h, l;
                  ps_split2h(ps_scaled_sat(v), h, l);
                  *reinterpret_cast<pf16x4*>(d) = h;
                  *reinterpret_cast<pf16x4*>(d + 1024) = l;
                }
              }
            }
          }
          }
          if constexpr (KV) {
            if (kv_tile && (L_ANY || rowbase + 32 * i < p.M)) {
              const int Cq = p.N / 3;
              const int sec = ctile / Cq;                            // 1 = K, 2 = V
              const int hh = (ctile - sec * Cq) >> 5;
              const int r0 = rowbase + 32 * i;
              const int bsm = r0 / p.attn_tokens, key0 = r0 - bsm * p.attn_tokens;
              const int ntl = p.attn_tokens >> 6;
              unsigned char* dst = reinterpret_cast<unsigned char*>(p.attn_kv_out) +
                                   (((long long)bsm * (Cq >> 5) + hh) * ntl + (key0 >> 6)) * PS_KV_TILE + lane * 16;
              const int sub = (key0 >> 5) & 1;
              bool bad = false;
              if (sec == 1) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  float vals[8];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(kvv[8 * t + e]), __float_as_uint(kvv[8 * t + 4 + e]), false, false);
                    vals[e] = __uint_as_float(r[0]);
                    vals[4 + e] = __uint_as_float(r[1]);
                  }
                  const float4 v0 = make_float4(vals[0], vals[1], vals[2], vals[3]), v1 = make_float4(vals[4], vals[5], vals[6], vals[7]);
                  bad |= ps_h2_out_of_range(v0) || ps_h2_out_of_range(v1);
                  pf16x4 h0, l0, h1, l1;
                  ps_split2h(ps_scaled_sat(v0), h0, l0);
                  ps_split2h(ps_scaled_sat(v1), h1, l1);
                  unsigned char* d = dst + (2 * sub + t) * 2048;
                  *reinterpret_cast<pf16x8*>(d) = pf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                  *reinterpret_cast<pf16x8*>(d + 1024) = pf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
              } else {
                // V^T: [token = l31][d = 8 q + 4 half + e] -> LDS -> lane = d reads 8 keys in the order the probabilities leave
                // the first product's accumulator (keys 16 t + 4 half + (j & 3) + 8 (j >> 2))
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                  for (int e = 0; e < 4; ++e) kv_ts[l31 * 33 + 8 * q + 4 * half + e] = kvv[4 * q + e];
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  float vals[8];
#pragma unroll
                  for (int jx = 0; jx < 8; ++jx) vals[jx] = kv_ts[(16 * t + 4 * half + (jx & 3) + 8 * (jx >> 2)) * 33 + l31];
                  const float4 v0 = make_float4(vals[0], vals[1], vals[2], vals[3]), v1 = make_float4(vals[4], vals[5], vals[6], vals[7]);
                  bad |= ps_h2_out_of_range(v0) || ps_h2_out_of_range(v1);
                  pf16x4 h0, l0, h1, l1;
                  ps_split2h(ps_scaled_sat(v0), h0, l0);
                  ps_split2h(ps_scaled_sat(v1), h1, l1);
                  unsigned char* d = dst + (4 + 2 * sub + t) * 2048;
                  *reinterpret_cast<pf16x8*>(d) = pf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                  *reinterpret_cast<pf16x8*>(d + 1024) = pf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
              }
              if (bad) *p.range_flag = 1;
            }
          }
        }
      }
    }
    if (bad_ps) *p.range_flag = 1;
    return;
  } else {
    // ---- epilogue, lane = column (igemm_ws.hip's).  C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (splitk > 1) {   // raw partial slab [ks][M][N]; igemm_reduce_kernel (or the consumer, raw_slabs) sums them
      float* slab = ws + ((long long)bz * splitk + ks) * p.M * p.N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = colbase + j * 32 + l31;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) slab[(long long)row * p.N + col] = acc[i][j][r];
          }
      }
      return;
    }
    float* __restrict__ outp = p.out + (long long)bz * p.out_bstride;
    const float* resp = p.residual ? p.residual + (long long)bz * p.out_bstride : nullptr;
    if (p.epi == LDMK_EPI_GEGLU) {
      if constexpr (TN % 2 == 0) {
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int cv = colbase + j * 32 + l31;        // packed value column
          const int cg = cv + 32;                       // packed gate column
          if (cv >= p.N) continue;
          const int oc = ((colbase + j * 32) >> 1) + l31;
          const float bv = p.bias ? p.bias[cv] : 0.f, bg = p.bias ? p.bias[cg] : 0.f;
          const float csv = lnf ? p.ln_colsum[cv] : 0.f, csg = lnf ? p.ln_colsum[cg] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float2 st[16];
            if (lnf) {
#pragma unroll
              for (int r = 0; r < 16; ++r) st[r] = stats2[min(rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (row < p.M) {
                float v = acc[i][j][r] * alpha, g = acc[i][j + 1][r] * alpha;
                if (lnf) {
                  v = fmaf(-st[r].x, csv, v) * st[r].y;
                  g = fmaf(-st[r].x, csg, g) * st[r].y;
                }
                v += bv;
                g += bg;
                outp[(long long)row * p.ldc + oc] = v * gelu_erf_f(g);
              }
            }
          }
        }
      }
      return;
    }
    if constexpr (LEAN != 0) {
      // (dispatch: whole wave tile inside M x N, no split-K, no folded LayerNorm, 32-row tiles inside one sample)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = colbase + j * 32 + l31;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r0 = rowbase + i * 32 + 4 * half;
          const unsigned obase = (unsigned)r0 * (unsigned)p.ldc + (unsigned)col;
          float extra[16], vec = 0.f;
          if constexpr (LEAN == 1 || LEAN == 4) vec = p.batch_vec[(long long)((rowbase + i * 32) / p.rows_per_sample) * p.batch_vec_ld + col];
          if constexpr (LEAN == 2 || LEAN == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) extra[r] = resp[obase + (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ldc)];
          }
          float vals[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float t = acc[i][j][r] * alpha;
            if constexpr (LEAN == 3) vals[r] = ps_col_finish(t, bv);
            else if constexpr (LEAN == 1) vals[r] = ps_col_finish(t, bv, vec);
            else if constexpr (LEAN == 2) vals[r] = ps_col_finish(t, bv, extra[r]);
            else vals[r] = ps_col_finish(t, bv, vec, extra[r]);
            outp[obase + (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ldc)] = vals[r];
          }
          if (p.stats_out) {
            const float shift = __shfl(vals[0], l31, 64);      // row 0 of the tile
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = vals[r] - shift;
              sm += d;
              sq = fmaf(d, d, sq);
            }
            sm += __shfl_xor(sm, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (half == 0) {
              float* d = p.stats_out + ((long long)((rowbase + i * 32) >> 5) * p.N + col) * 3;
              d[0] = shift; d[1] = sm; d[2] = sq;
            }
          }
        }
      }
      return;
    }
    const bool tile_in_sample = p.rows_per_sample % 32 == 0;
    int smp[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) smp[i] = p.batch_vec ? min(rowbase + i * 32, p.M - 1) / p.rows_per_sample : 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colbase + j * 32 + l31;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const float cs = lnf ? p.ln_colsum[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float vals[16];
        const int r0 = rowbase + i * 32 + 4 * half;
        if (lnf) {
          float2 st[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) st[r] = stats2[min(r0 + (r & 3) + 8 * (r >> 2), p.M - 1)];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(-st[r].x, cs, acc[i][j][r] * alpha) * st[r].y;
        }
        const unsigned obase = (unsigned)r0 * (unsigned)p.ldc + (unsigned)col;
        const float vec = (p.batch_vec && tile_in_sample) ? p.batch_vec[(long long)smp[i] * p.batch_vec_ld + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          float v = 0.f;
          if (r0 + dr < p.M) {
            v = (lnf ? acc[i][j][r] : acc[i][j][r] * alpha) + bv;
            if (p.batch_vec) v += tile_in_sample ? vec : p.batch_vec[(long long)((r0 + dr) / p.rows_per_sample) * p.batch_vec_ld + col];
            const unsigned o = obase + (unsigned)(dr * p.ldc);
            if (resp) v += resp[o];
            outp[o] = v;
          }
          vals[r] = v;
        }
        if (p.stats_out && rowbase + i * 32 < p.M) {
          // GroupNorm partial record of this 32-row tile x column (the record of gn_partial_kernel / igemm_kernel)
          const float shift = __shfl(vals[0], l31, 64);      // row 0 of the tile
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = vals[r] - shift;
            sm += d;
            sq = fmaf(d, d, sq);
          }
          sm += __shfl_xor(sm, 32, 64);
          sq += __shfl_xor(sq, 32, 64);
          if (half == 0) {
            float* d = p.stats_out + ((long long)((rowbase + i * 32) >> 5) * p.N + col) * 3;
            d[0] = shift; d[1] = sm; d[2] = sq;
          }
        }
      }
    }
  }
}

// which lean epilogue (ps_epilogue's LEAN) a wave tile ending at (row_end, col_end) may take; 0 = the general one
template <bool TR>
__device__ __forceinline__ int ps_lean_form(const ldmk_igemm_args& p, const int splitk, const bool off, const int row_end, const int col_end) {
  if (off || splitk != 1 || row_end > p.M || col_end > p.N) return 0;
  const bool lf = p.a_tf == LDMK_TF_LAYERNORM_FOLDED, geglu = p.epi == LDMK_EPI_GEGLU;
  if (p.batch_vec && p.rows_per_sample % 32 != 0) return 0;      // (a 32-row tile inside one sample)
  if constexpr (TR) {
    if (lf) return p.residual || p.batch_vec ? 0 : 1;
    if (geglu || (p.batch_vec && !p.residual)) return 0;
    return p.batch_vec ? 4 : (p.residual ? 2 : 3);
  } else {
    if (lf || geglu) return 0;
    return p.batch_vec ? (p.residual ? 4 : 1) : (p.residual ? 2 : 3);
  }
}

template <int NWM, int NWN, int TM, int TN, int NS, bool TR, int PL = 3, bool KV = false>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void igemm_ps_kernel(const ldmk_igemm_args p, const int splitk, float* __restrict__ ws, const int dbg_arg, const int nfast) {
  const int dbg = PS_PROBES ? dbg_arg : 0;        // (what-if switches: probe builds only, see PS_PROBES)
  constexpr int NW = NWM * NWN;
  constexpr int BM = 32 * TM * NWM, BN = 32 * TN * NWN;
  constexpr int FA = BM / 32, FB = BN / 32, U = FA + FB;          // units (3-plane blocks) per stage
  constexpr int UHI = (U + NW - 1) / NW, ULO = U / NW;            // units a wave fetches per stage
  constexpr int UB = PL * 1024;                                     // bytes per unit
  constexpr int STAGE = U * UB;                                     // bytes
  static_assert(PL * UHI * (NS - 1) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_ps[];      // [NS][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / NWN, wn = wave - wm * NWN;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  // Which operand the neighbouring workgroups of an XCD share in its L2 (round 5).  These problems are tall (M >> N): the weights
  // are small and L2-resident whatever the order, the A tile is what every column tile re-reads -- with the column tiles of one
  // row tile adjacent (nfast), A comes over the fabric once instead of once per column tile (GEGLU at 64x64: 5x).
  const int m0 = (nfast & 1) ? (bid / tiles_n) * BM : (bid % tiles_m) * BM;
  const int n0 = (nfast & 1) ? (bid % tiles_n) * BN : (bid / tiles_m) * BN;
  const int ks = blockIdx.y, bz = blockIdx.z;

  const int nkc = p.K / 32;
  const int it_per = (nkc + splitk - 1) / splitk;
  const int it_begin = ks * it_per;
  const int it_end = min(nkc, it_begin + it_per);
  const int n16 = it_end > it_begin ? 2 * (it_end - it_begin) : 0;
  const int Kb = p.K / 16;
  const int Mb = (p.M + 31) / 32, Nb = p.N / 32;

  // (dbg, probe runs only -- LDMK_PS_DEBUG: bit 0 = zero-record descriptors: every DMA is issued but dropped by the range check,
  //  no memory traffic; bit 1 = no DMA instructions at all; bit 2 = no matrix instructions.  Results are garbage then.)
  const pu32x4 rs_a = ps_rsrc(reinterpret_cast<const unsigned char*>(p.a_ps) + (long long)bz * p.a_ps_bstride, (dbg & 1) ? 0u : (unsigned)Mb * (unsigned)Kb * (unsigned)UB);
  const pu32x4 rs_b = ps_rsrc(reinterpret_cast<const unsigned char*>(p.w_ps) + (long long)bz * p.w_ps_bstride, (dbg & 1) ? 0u : (unsigned)Nb * (unsigned)Kb * (unsigned)UB);
  // this wave's units u = wave + NW i: byte offset of the block's first k-slab (wave-uniform), or PS_OOB for blocks past the edge
  unsigned ubase[UHI];
#pragma unroll
  for (int i = 0; i < UHI; ++i) {
    const int u = wave + NW * i;
    const int blk = u < FA ? m0 / 32 + u : n0 / 32 + (u - FA);
    const bool ok = u < U && (u < FA ? blk < Mb : blk < Nb);
    ubase[i] = ok ? (unsigned)blk * (unsigned)Kb * (unsigned)UB + (unsigned)(2 * it_begin) * (unsigned)UB : PS_OOB;
  }
  const unsigned lane16 = lane * 16;
  const unsigned lds0 = (unsigned)(size_t)smem_ps;
  auto issue = [&](int s) {                     // the DMA of local stage s into ring buffer s % NS (stages past the end: zeros)
    const unsigned buf = lds0 + (unsigned)(s % NS) * STAGE;
    const bool live = s < n16;
#pragma unroll
    for (int i = 0; i < UHI; ++i) {
      const int u = wave + NW * i;
      if (u < U && !(dbg & 2)) {                 // (wave-uniform)
        const unsigned off = (live && ubase[i] != PS_OOB) ? ubase[i] + (unsigned)s * (unsigned)UB : PS_OOB;
        ps_dma<PL>(lane16 + off, u < FA ? rs_a : rs_b, buf + (unsigned)u * (unsigned)UB);
      }
    }
  };
  const bool hi_wave = wave + NW * (UHI - 1) < U;       // fetches UHI units per stage (else ULO)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Phase stagger (stagger = dbg >> 8, in units of 64 cycles per stage; 0 = off): every workgroup of a launch does the same
  // work, so the co-resident workgroups of a CU -- and all CUs of the chip -- run their main loops together and then their
  // epilogues together: the memory system idles while the matrix cores work and vice versa (probe: GEGLU at K = 160 takes
  // 116 us without its epilogue, 192 us with it).  Waves in an odd hardware wave slot (the second workgroup of a CU) start
  // half a tile period late, and since every tile takes the same time the offset persists over the rounds of the launch.
  if ((dbg >> 8) != 0) {
    const unsigned slot = __builtin_amdgcn_s_getreg(6148);      // HW_REG_HW_ID[3:0]: wave slot within the SIMD
    if (slot & 1) {
      const int n = (dbg >> 8) * n16;                             // 64-cycle units
      for (int i = 0; i < n; i += 64) __builtin_amdgcn_s_sleep(64);
    }
  }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  for (int it = 0; it < n16; ++it) {
    // stage `it` of THIS wave has landed when all but the (NS - 2) younger stages' loads are done; the barrier then makes
    // every wave's part visible and certifies that buffer (it - 1) % NS is no longer read
    if (hi_wave) ps_wait_vm<PL * UHI * (NS - 2)>(); else ps_wait_vm<PL * ULO * (NS - 2)>();
    asm volatile("s_barrier" ::: "memory");
    issue(it + NS - 1);
    const unsigned char* sb = smem_ps + (it % NS) * STAGE + lane16;
    pbf16x8 a8[PL][TM];
#pragma unroll
    for (int g = 0; g < PL; ++g)
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[g][i] = *reinterpret_cast<const pbf16x8*>(sb + ((wm * TM + i) * PL + g) * 1024);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      pbf16x8 b8[PL];
#pragma unroll
      for (int g = 0; g < PL; ++g) b8[g] = *reinterpret_cast<const pbf16x8*>(sb + ((FA + wn * TN + j) * PL + g) * 1024);
      if (dbg & 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g = 0; g < PL; ++g) asm volatile("" :: "v"(a8[g][i]), "v"(b8[g]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (PL == 2) {
          // F16X2: lo hi, hi lo, hi hi (planes: 0 = hi, 1 = lo) -- the order of igemm_kernel<BF = 4>
          const pf16x8 ah = __builtin_bit_cast(pf16x8, a8[0][i]), al = __builtin_bit_cast(pf16x8, a8[1][i]);
          const pf16x8 bh = __builtin_bit_cast(pf16x8, b8[0]), bl = __builtin_bit_cast(pf16x8, b8[1]);
          if constexpr (TR) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
          }
        } else if constexpr (TR) {       // smallest partial products first (planes: 0 = hi, 1 = mid, 2 = lo) -- the order of igemm_kernel<BF = 3>
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0], a8[2][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[2], a8[0][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[1], a8[1][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0], a8[1][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[1], a8[0][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b8[0], a8[0][i], acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][i], b8[0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[0], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  ps_wait_vm<0>();             // (the trailing out-of-range loads still target this workgroup's LDS)
  if (dbg & 16) {              // probe: no epilogue (one never-taken store keeps the accumulators live)
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
    if (keep == 12345.678f && ws) ws[0] = keep;
    return;
  }
  // (wave-uniform: a lean epilogue for a wave tile that lies inside M x N; which one by the launch's operand set)
  const int lean = ps_lean_form<TR>(p, splitk, (nfast & 2) != 0, m0 + (wm + 1) * 32 * TM, n0 + (wn + 1) * 32 * TN);
  if constexpr (KV) {
    __syncthreads();            // every wave is done with the ring: its memory becomes the per-wave V^T transpose scratch
    if (lean == 1)
      ps_epilogue<TM, TN, TR, PL, true, TR ? 1 : 0>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane,
                                                    reinterpret_cast<float*>(smem_ps) + wave * (32 * 33));
    else
      ps_epilogue<TM, TN, TR, PL, true>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane,
                                         reinterpret_cast<float*>(smem_ps) + wave * (32 * 33));
  } else {
    if (lean == 1) ps_epilogue<TM, TN, TR, PL, false, 1>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane);
    else if (lean == 2) ps_epilogue<TM, TN, TR, PL, false, 2>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane);
    else if (lean == 3) ps_epilogue<TM, TN, TR, PL, false, 3>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane);
    else if (lean == 4) ps_epilogue<TM, TN, TR, PL, false, 4>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane);
    else ps_epilogue<TM, TN, TR, PL>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, bz, ws, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same GEMM, warp-specialised (tile_cfg 29 / 30): probes of the kernel above (LDMK_PS_DEBUG) showed its fragment reads and
// its matrix instructions taking TURNS -- every wave reads after the barrier, then multiplies; with the matrix work removed
// a stage still costs what the matrix work alone costs -- and the DMA issue (~100 cycles of stall per piece) sitting in the
// same in-order stream.  Here:
//   * waves 0-3 are CONSUMERS, one per SIMD with the whole 512-register budget: each owns 64 rows x BN columns of a 256 x BN
//     tile (BN = 160 / 128) and runs ONE continuous software pipeline over all stages -- the fragments of MFMA group j + 1
//     (and, at a stage's last group, the first fragments of the NEXT stage) are requested before group j is issued, so no
//     matrix instruction waits for LDS, also not across the stage boundary;
//   * waves 4-7 are PRODUCERS: nothing but the LDS-DMA pieces of the ring (NS = 4 stages of 16 k), the counted wait and the
//     barrier; their issue stalls cost no matrix time;
//   * one barrier per stage, and it certifies the stage AFTER the one about to be multiplied (that is what lets the consumer
//     read ahead across the boundary): at barrier `it` the producers have waited for stage it + 1 and the consumers have
//     finished reading stage it - 1, whose buffer the producers then refill with stage it + 3.
// Same products in the same order into every accumulator as the kernel above: bitwise equal results.
template <int TM, int TN, int NS, bool TR>
__global__ __launch_bounds__(512) void igemm_pw_kernel(const ldmk_igemm_args p, const int splitk, float* __restrict__ ws, const int dbg_arg) {
  const int dbg = PS_PROBES ? dbg_arg : 0;
  constexpr int BM = 128 * TM, BN = 32 * TN;
  constexpr int FA = BM / 32, FB = TN, U = FA + FB;
  constexpr int UHI = (U + 3) / 4, ULO = U / 4;                   // units a producer wave fetches per stage
  constexpr int STAGE = U * 3072;
  static_assert(NS >= 3 && 3 * UHI * (NS - 2) <= 63, "ring depth / vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_ps[];      // [NS][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (bid % tiles_m) * BM;
  const int n0 = (bid / tiles_m) * BN;
  const int ks = blockIdx.y, bz = blockIdx.z;
  const int nkc = p.K / 32;
  const int it_per = (nkc + splitk - 1) / splitk;
  const int it_begin = ks * it_per;
  const int it_end = min(nkc, it_begin + it_per);
  const int n16 = it_end > it_begin ? 2 * (it_end - it_begin) : 0;      // even
  const unsigned lane16 = lane * 16;

  if (wave >= 4) {
    // ---------------------------------------------------------------- producers
    const int pw = wave - 4;
    const int Kb = p.K / 16;
    const int Mb = (p.M + 31) / 32, Nb = p.N / 32;
    const pu32x4 rs_a = ps_rsrc(reinterpret_cast<const unsigned char*>(p.a_ps) + (long long)bz * p.a_ps_bstride, (dbg & 1) ? 0u : (unsigned)Mb * (unsigned)Kb * 3072u);
    const pu32x4 rs_b = ps_rsrc(reinterpret_cast<const unsigned char*>(p.w_ps) + (long long)bz * p.w_ps_bstride, (dbg & 1) ? 0u : (unsigned)Nb * (unsigned)Kb * 3072u);
    unsigned ubase[UHI];
#pragma unroll
    for (int i = 0; i < UHI; ++i) {
      const int u = pw + 4 * i;
      const int blk = u < FA ? m0 / 32 + u : n0 / 32 + (u - FA);
      const bool ok = u < U && (u < FA ? blk < Mb : blk < Nb);
      ubase[i] = ok ? (unsigned)blk * (unsigned)Kb * 3072u + (unsigned)(2 * it_begin) * 3072u : PS_OOB;
    }
    const unsigned lds0 = (unsigned)(size_t)smem_ps;
    auto issue = [&](int s) {
      const unsigned buf = lds0 + (unsigned)(s % NS) * STAGE;
      const bool live = s < n16;
#pragma unroll
      for (int i = 0; i < UHI; ++i) {
        const int u = pw + 4 * i;
        if (u < U && !(dbg & 2)) {
          const unsigned off = (live && ubase[i] != PS_OOB) ? ubase[i] + (unsigned)s * 3072u : PS_OOB;
          ps_dma3(lane16 + off, u < FA ? rs_a : rs_b, buf + (unsigned)u * 3072u);
        }
      }
    };
    const bool hi_wave = pw + 4 * (UHI - 1) < U;
    // (dbg & 8, probe runs: cycle totals of the phases go to args.splitk_counters as [workgroup][wave][4] 64-bit ticks)
    unsigned long long t_acc[3] = {0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_last;
#define PW_T(i) do { if (dbg & 8) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_acc[i] += t_ - t_last; t_last = t_; } } while (0)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    PW_T(0);
    if (hi_wave) ps_wait_vm<3 * UHI * (NS - 2)>(); else ps_wait_vm<3 * ULO * (NS - 2)>();      // stage 0
    PW_T(1);
    asm volatile("s_barrier" ::: "memory");
    PW_T(2);
    for (int it = 0; it < n16; ++it) {
      if (hi_wave) ps_wait_vm<3 * UHI * (NS - 3)>(); else ps_wait_vm<3 * ULO * (NS - 3)>();    // stage it + 1
      PW_T(1);
      asm volatile("s_barrier" ::: "memory");
      PW_T(2);
      issue(it + NS - 1);
      PW_T(0);
    }
    ps_wait_vm<0>();
    if ((dbg & 8) && lane == 0 && p.splitk_counters) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.splitk_counters) + ((long long)blockIdx.x * 8 + wave) * 4;
      d[0] = t_acc[0]; d[1] = t_acc[1]; d[2] = t_acc[2]; d[3] = __builtin_amdgcn_s_memtime() - t_begin;
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wm = wave;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned char* abase = smem_ps + wm * TM * 3072 + lane16;
  const unsigned char* bbase = smem_ps + FA * 3072 + lane16;
  pbf16x8 A2[2][3][TM], B2[2][3];
  auto ldA = [&](pbf16x8 (&a)[3][TM], int buf) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < TM; ++i) a[g][i] = *reinterpret_cast<const pbf16x8*>(abase + buf * STAGE + (i * 3 + g) * 1024);
  };
  auto ldB = [&](pbf16x8 (&b)[3], int buf, int j) {
#pragma unroll
    for (int g = 0; g < 3; ++g) b[g] = *reinterpret_cast<const pbf16x8*>(bbase + buf * STAGE + (j * 3 + g) * 1024);
  };
  unsigned long long t_acc[3] = {0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = t_last;
  asm volatile("s_barrier" ::: "memory");                     // stage 0 is in LDS
  PW_T(2);
  ldA(A2[0], 0);
  ldB(B2[0], 0, 0);
  int buf = 0;
  for (int it = 0; it < n16; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {                             // two stages per trip: the register roles come back
      const int nbuf = buf + 1 == NS ? 0 : buf + 1;
      PW_T(0);
      asm volatile("s_barrier" ::: "memory");                 // stage it + h + 1 is in LDS, stage it + h - 1 may be overwritten
      PW_T(2);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        constexpr int unused = 0;
        (void)unused;
        const int cur = (h * TN + j) & 1;
        // request the next group's fragments before this group's matrix instructions
        if (j + 1 < TN) {
          ldB(B2[cur ^ 1], buf, j + 1);
        } else {
          ldA(A2[h ^ 1], nbuf);
          ldB(B2[cur ^ 1], nbuf, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // smallest partial products first (planes: 0 = hi, 1 = mid, 2 = lo) -- the order of igemm_kernel<BF = 3>
            if constexpr (TR) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][0], A2[h][2][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][2], A2[h][0][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][1], A2[h][1][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][0], A2[h][1][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][1], A2[h][0][i], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B2[cur][0], A2[h][0][i], acc[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][2][i], B2[cur][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][0][i], B2[cur][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][1][i], B2[cur][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][1][i], B2[cur][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][0][i], B2[cur][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[h][0][i], B2[cur][0], acc[i][j], 0, 0, 0);
            }
        }
      }
      buf = nbuf;
    }
  }
  PW_T(0);
  ps_epilogue<TM, TN, TR>(p, acc, m0 + wm * 32 * TM, n0, splitk, ks, bz, ws, lane);
  PW_T(1);
  if ((dbg & 8) && lane == 0 && p.splitk_counters) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(p.splitk_counters) + ((long long)blockIdx.x * 8 + wave) * 4;
    d[0] = t_acc[0]; d[1] = t_acc[1]; d[2] = t_acc[2]; d[3] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#undef PW_T
}

// ------------------------------------------------------------------------------------------------------------------------------
// Producers of the PS layout.
//   element (r, k), plane g of X[R][K] -> bf16 index (((r / 32) (K / 16) + k / 16) 3 + g) 512 + ((k / 8 % 2) 32 + r % 32) 8 + k % 8

// generic packer: X[r][k] = src[r rs + k ks] (weights W[K][N]: rs = 1, ks = ldb; row-major activations: rs = ld, ks = 1).
// Workgroup = one 32-row block x 64 k; thread = (row, 8 consecutive k): 128-byte runs on the store side.
template <int PL>
__global__ __launch_bounds__(256) void pack_ps_kernel(const float* __restrict__ src, int R, int K, long long rs, long long ks,
                                                      long long src_bstride, unsigned char* __restrict__ dst, long long dst_bstride,
                                                      float scale, int* __restrict__ range_flag) {
  const int rb = blockIdx.x, kc = blockIdx.y;
  src += (long long)blockIdx.z * src_bstride;
  dst += (long long)blockIdx.z * dst_bstride;
  const int r = threadIdx.x >> 3, o = threadIdx.x & 7;
  const int row = rb * 32 + r, k0 = kc * 64 + o * 8;
  if (k0 >= K) return;
  float v[8];
  if (row < R) {
    if (ks == 1) {
      const float4 a = *reinterpret_cast<const float4*>(src + row * rs + k0), b = *reinterpret_cast<const float4*>(src + row * rs + k0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[row * rs + (long long)(k0 + e) * ks];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
  unsigned char* d = dst + ((long long)rb * (K / 16) + (k0 >> 4)) * (PL * 1024) + ((o & 1) * 32 + r) * 16;
  if constexpr (PL == 2) {          // F16X2: the two fp16 images of scale x (activations: 2^6, range-checked; weights: 2^w_scale_exp)
    float4 v0 = make_float4(v[0], v[1], v[2], v[3]), v1 = make_float4(v[4], v[5], v[6], v[7]);
    if (range_flag) {               // activations (scale 2^6): range-checked and saturated; weights carry their own exponent
      if (ps_h2_out_of_range(v0) || ps_h2_out_of_range(v1)) *range_flag = 1;
      v0 = h2_clamp4(v0);
      v1 = h2_clamp4(v1);
    }
    pf16x4 h0, l0, h1, l1;
    ps_split2h(ps_scaled(v0, scale), h0, l0);
    ps_split2h(ps_scaled(v1, scale), h1, l1);
    *reinterpret_cast<pf16x8*>(d) = pf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    *reinterpret_cast<pf16x8*>(d + 1024) = pf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  } else {
    pbf16x4 h0, m0, l0, h1, m1, l1;
    ps_split3(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
    ps_split3(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
    *reinterpret_cast<pbf16x8*>(d) = pbf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    *reinterpret_cast<pbf16x8*>(d + 1024) = pbf16x8{m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
    *reinterpret_cast<pbf16x8*>(d + 2048) = pbf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  }
}

// LayerNorm statistics (the two-pass form of ln_stats_kernel, values held in registers) AND the rows in the PS layout: the pass
// reads every element anyway.  Workgroup = one 32-row block; thread = (row, k-octet o of every 64-wide chunk); the 8 threads
// of a row are 8 consecutive lanes (3 shuffle steps).  K % 16 == 0, K <= 64 KCH.
template <int KCH, int PL = 3>
__global__ __launch_bounds__(256) void ln_stats_ps_kernel(const float* __restrict__ x, int rows, int K, float eps, float* __restrict__ stats,
                                                          unsigned char* __restrict__ dst, float guard, int* __restrict__ flag,
                                                          int* __restrict__ range_flag) {
  const int rb = blockIdx.x;
  const int r = threadIdx.x >> 3, o = threadIdx.x & 7;
  const int row = rb * 32 + r;
  const bool rok = row < rows;
  const float* px = x + (long long)(rok ? row : 0) * K;
  float4 v[KCH][2];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k0 = c * 64 + o * 8;
    if (rok && k0 < K) {
      v[c][0] = *reinterpret_cast<const float4*>(px + k0);
      v[c][1] = *reinterpret_cast<const float4*>(px + k0 + 4);
    } else {
      v[c][0] = v[c][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s += ((v[c][0].x + v[c][0].y) + (v[c][0].z + v[c][0].w)) + ((v[c][1].x + v[c][1].y) + (v[c][1].z + v[c][1].w));
  }
#pragma unroll
  for (int d = 4; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  const float mean = s / (float)K;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    if (c * 64 + o * 8 < K) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float a = v[c][e].x - mean, b = v[c][e].y - mean, cc = v[c][e].z - mean, d = v[c][e].w - mean;
        q = fmaf(a, a, q); q = fmaf(b, b, q); q = fmaf(cc, cc, q); q = fmaf(d, d, q);
      }
    }
  }
#pragma unroll
  for (int d = 4; d > 0; d >>= 1) q += __shfl_xor(q, d, 64);
  if (rok && o == 0) {
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
    stats[2 * (long long)row] = mean;
    stats[2 * (long long)row + 1] = rstd;
    if (flag && !(fabsf(mean) * rstd <= guard)) *flag = 1;
  }
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    const int k0 = c * 64 + o * 8;
    if (k0 < K) {
      unsigned char* d = dst + ((long long)rb * (K / 16) + (k0 >> 4)) * (PL * 1024) + ((o & 1) * 32 + r) * 16;
      if constexpr (PL == 2) {
        if (ps_h2_out_of_range(v[c][0]) || ps_h2_out_of_range(v[c][1])) *range_flag = 1;
        pf16x4 h0, l0, h1, l1;
        ps_split2h(ps_scaled_sat(v[c][0]), h0, l0);
        ps_split2h(ps_scaled_sat(v[c][1]), h1, l1);
        *reinterpret_cast<pf16x8*>(d) = pf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<pf16x8*>(d + 1024) = pf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      } else {
        pbf16x4 h0, m0, l0, h1, m1, l1;
        ps_split3(v[c][0], h0, m0, l0);
        ps_split3(v[c][1], h1, m1, l1);
        *reinterpret_cast<pbf16x8*>(d) = pbf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        *reinterpret_cast<pbf16x8*>(d + 1024) = pbf16x8{m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
        *reinterpret_cast<pbf16x8*>(d + 2048) = pbf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
int launch_splitk_reduce(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st);

struct PsCfg { int bm, bn; bool even_tn; };
// pcfg 0: 256x160 (8 x 1 waves of 32 x 160), 1: 256x320 (4 x 2 waves of 64 x 160), 2: 256x256 (4 x 2 waves of 64 x 128: GEGLU pairs),
//      3: 128x320 (4 x 2 waves of 32 x 160), 4: 128x160 (4 x 1 waves of 32 x 160, two workgroups per CU), 5: 128x256 (4 x 2 waves of 32 x 128)
//      6: 256x160, 7: 256x128 (GEGLU pairs): the warp-specialised form (igemm_pw_kernel: 4 consumer waves of 64 x BN + 4 DMA waves)
static const PsCfg kPsCfg[] = {{256, 160, false}, {256, 320, false}, {256, 256, true}, {128, 320, false}, {128, 160, false}, {128, 256, true},
                               {256, 160, false}, {256, 128, true},
                               // 8: 256x160, 9: 256x128, 10: 128x256 with FOUR waves of 64 x 160 / 64 x 128 and a 2-deep ring: two workgroups per
                               // CU, whose DMA issue / barrier / epilogue phases overlap the other one's matrix work
                               {256, 160, false}, {256, 128, true}, {128, 256, true}};

template <int NWM, int NWN, int TM, int TN, int NS, bool TR, int PL = 3, bool KV = false>
static int ps_launch(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  constexpr int BM = 32 * TM * NWM, BN = 32 * TN * NWN;
  constexpr size_t lds = (size_t)NS * (BM / 32 + BN / 32) * PL * 1024;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ps_kernel<NWM, NWN, TM, TN, NS, TR, PL, KV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int dbg = ps_probe_bits();
  static_assert(!KV || (size_t)NWM * NWN * 32 * 33 * 4 <= lds, "V^T transpose scratch fits the ring");
  static const int nfast_env = [] { const char* e = getenv("LDMK_PS_NFAST"); return e ? atoi(e) : 1; }();
  static const int lean_env = [] { const char* e = getenv("LDMK_PS_LEAN"); return e ? atoi(e) : 1; }();     // (0: the general epilogue everywhere, A/B)
  const int nfast = (nfast_env && (a.N + BN - 1) / BN > 1 && (a.M + BM - 1) / BM >= 8 ? 1 : 0) | (lean_env ? 0 : 2);   // (a few row tiles only: the old order)
  hipLaunchKernelGGL((igemm_ps_kernel<NWM, NWN, TM, TN, NS, TR, PL, KV>), dim3(tiles, splitk, a.batch > 1 ? a.batch : 1), dim3(64 * NWM * NWN), lds,
                     st, a, splitk, ws, dbg, nfast);
  if (splitk > 1 && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  return check_launch("ldmk_igemm(ps)");
}

template <int TM, int TN, int NS, bool TR>
static int pw_launch(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  constexpr int BM = 128 * TM, BN = 32 * TN;
  constexpr size_t lds = (size_t)NS * (BM / 32 + TN) * 3072;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_pw_kernel<TM, TN, NS, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int dbg = ps_probe_bits() & 0xff;
  hipLaunchKernelGGL((igemm_pw_kernel<TM, TN, NS, TR>), dim3(tiles, splitk, a.batch > 1 ? a.batch : 1), dim3(512), lds, st, a, splitk, ws, dbg);
  if (splitk > 1 && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  return check_launch("ldmk_igemm(pw)");
}

// ------------------------------------------------------------------------------------------------------------------------------
// The pre-split tile as an IMPLICIT GEMM of a 3x3 convolution (round 5; F16X2, tile_cfg 23 / 24 / 26 / 27 with
// a_mode = LDMK_A_CONV3X3 and a_ps).  The A operand is the GroupNorm-applied activation stored ONCE in the PS layout of the
// [pixels][C] matrix (ldmk_gn_apply_ps_h2: it replaces the fp32 gn_apply pass, same bytes); the nine taps are nine per-lane
// address sets for the same LDS-DMA instruction: lane (half, r) of the unit of output rows 32 b .. 32 b + 31 reads, for tap
// (dy, dx), the 16 bytes of input pixel q = (n, oy stride + dy - pad, ox stride + dx - pad) -- block q / 32, row q % 32 of the
// k-slab -- or, in the halo, an out-of-range offset that the buffer load turns into zeros.  Neighbouring lanes read neighbouring
// pixels, i.e. neighbouring 16-byte pieces of one plane: the gather stays coalesced, no element passes through a register, and
// nothing is split or normalised per N-tile (igemm_kernel<BF = 4> re-reads, scales and splits every input element once per tap and
// column tile).  K order = igemm.hip's (32-channel chunk major, tap minor; ldmk_pack_conv3x3): 18 sixteen-deep stages per
// chunk, unrolled, so the tap of a stage -- hence the VGPR holding its offsets -- is a compile-time index, and 18 % NS == 0
// keeps the ring buffer index static too.  Same products in the same order into every accumulator as igemm_kernel<BF = 4> on
// tile_cfg 5 / 1: bitwise equal at equal splitk (a K split must fall on chunk boundaries: splitk divides C / 32).
template <int NWM, int NWN, int TM, int TN, bool TR>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void igemm_psc_kernel(const ldmk_igemm_args p, const int splitk, float* __restrict__ ws, const int nfast) {
  constexpr int PL = 2, NS = 3;
  constexpr int NW = NWM * NWN;
  constexpr int BM = 32 * TM * NWM, BN = 32 * TN * NWN;
  constexpr int FA = BM / 32, FB = BN / 32, U = FA + FB;
  constexpr int UHI = (U + NW - 1) / NW, ULO = U / NW;
  constexpr int AI = (FA + NW - 1) / NW;                              // A units a wave may own (its first AI units)
  constexpr int UB = PL * 1024;
  constexpr int STAGE = U * UB;
  static_assert(18 % NS == 0 && PL * UHI * (NS - 1) <= 63, "ring / vmcnt");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_ps[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / NWN, wn = wave - wm * NWN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (nfast & 1) ? (bid / tiles_n) * BM : (bid % tiles_m) * BM;       // (see igemm_ps_kernel: column tiles of a row tile adjacent)
  const int n0 = (nfast & 1) ? (bid % tiles_n) * BN : (bid / tiles_m) * BN;
  const int ks = blockIdx.y;

  const int C = p.c0;
  const int nc32 = C / 32;
  const int c_per = (nc32 + splitk - 1) / splitk;
  const int c_begin = ks * c_per;
  const int c_end = min(nc32, c_begin + c_per);
  const int Kb_in = C / 16, Kb_w = p.K / 16;
  const int Nb = p.N / 32;
  const long long samples = ((long long)p.M + p.rows_per_sample - 1) / p.rows_per_sample;
  const unsigned Mb_in = (unsigned)((samples * p.in_h * p.in_w + 31) / 32);
  const pu32x4 rs_a = ps_rsrc(p.a_ps, Mb_in * (unsigned)Kb_in * (unsigned)UB);
  const pu32x4 rs_b = ps_rsrc(p.w_ps, (unsigned)Nb * (unsigned)Kb_w * (unsigned)UB);

  // A units: per tap the byte offset of this lane's input pixel (k-slab 0, plane 0), or PS_OOB in the halo / past the last row
  unsigned toff[AI][9];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int u = wave + NW * i;
    const int row = m0 + 32 * u + l31;
    const bool rok = u < FA && row < p.M;
    const int rr = rok ? row : 0;
    const int n = rr / p.rows_per_sample, rem = rr - n * p.rows_per_sample;
    const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = oy * p.stride + t / 3 - p.pad_lo, ix = ox * p.stride + t % 3 - p.pad_lo;
      const bool ok = rok && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
      const unsigned q = (unsigned)((n * p.in_h + iy) * p.in_w + ix);
      toff[i][t] = ok ? (q >> 5) * (unsigned)Kb_in * (unsigned)UB + ((unsigned)half * 32u + (q & 31u)) * 16u : PS_OOB;
    }
  }
  // B units: byte offset of the column block's first k-slab of this K range (wave-uniform), or PS_OOB
  unsigned ubase[UHI];
#pragma unroll
  for (int i = 0; i < UHI; ++i) {
    const int u = wave + NW * i;
    const int blk = n0 / 32 + (u - FA);
    ubase[i] = (u >= FA && u < U && blk < Nb) ? (unsigned)blk * (unsigned)Kb_w * (unsigned)UB + (unsigned)(18 * c_begin) * (unsigned)UB : PS_OOB;
  }
  const unsigned lane16 = lane * 16;
  const unsigned lds0 = (unsigned)(size_t)smem_ps;
  const bool hi_wave = wave + NW * (UHI - 1) < U;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the DMA of stage (chunk c, position j = 2 tap + h) into ring buffer `buf`; sB = that stage's index within this K range
  auto issue = [&](const int c, const int j, const int buf, const unsigned sB) {
    const bool live = c < c_end;
    const unsigned cs = (unsigned)(2 * c + (j & 1)) * (unsigned)UB;
    const unsigned bdst = lds0 + (unsigned)buf * STAGE;
#pragma unroll
    for (int i = 0; i < UHI; ++i) {
      const int u = wave + NW * i;
      if (u < U) {                                       // (wave-uniform)
        if (i < AI && u < FA) {
          const unsigned t = toff[i < AI ? i : 0][j >> 1];
          ps_dma<PL>((live && t != PS_OOB) ? t + cs : PS_OOB, rs_a, bdst + (unsigned)u * (unsigned)UB);
        } else {
          ps_dma<PL>((live && ubase[i] != PS_OOB) ? lane16 + ubase[i] + sB * (unsigned)UB : PS_OOB, rs_b, bdst + (unsigned)u * (unsigned)UB);
        }
      }
    }
  };

  unsigned sB = 0;
  issue(c_begin, 0, 0, 0);
  issue(c_begin, 1, 1, 1);
  for (int c = c_begin; c < c_end; ++c) {
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      if (hi_wave) ps_wait_vm<PL * UHI * (NS - 2)>(); else ps_wait_vm<PL * ULO * (NS - 2)>();
      asm volatile("s_barrier" ::: "memory");
      issue(j + 2 < 18 ? c : c + 1, (j + 2) % 18, (j + 2) % NS, sB + 2);
      ++sB;
      const unsigned char* sb = smem_ps + (j % NS) * STAGE + lane16;
      pf16x8 a8[PL][TM];
#pragma unroll
      for (int g = 0; g < PL; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i) a8[g][i] = *reinterpret_cast<const pf16x8*>(sb + ((wm * TM + i) * PL + g) * 1024);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        pf16x8 b8[PL];
#pragma unroll
        for (int g = 0; g < PL; ++g) b8[g] = *reinterpret_cast<const pf16x8*>(sb + ((FA + wn * TN + jn) * PL + g) * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          // F16X2: lo hi, hi lo, hi hi (planes: 0 = hi, 1 = lo) -- the order of igemm_kernel<BF = 4>
          if constexpr (TR) {
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[0], a8[1][i], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[1], a8[0][i], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[0], a8[0][i], acc[i][jn], 0, 0, 0);
          } else {
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[1][i], b8[0], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[0][i], b8[1], acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[0][i], b8[0], acc[i][jn], 0, 0, 0);
          }
        }
      }
    }
  }
  ps_wait_vm<0>();
  const int lean = ps_lean_form<TR>(p, splitk, (nfast & 2) != 0, m0 + (wm + 1) * 32 * TM, n0 + (wn + 1) * 32 * TN);
  if (lean == 1) ps_epilogue<TM, TN, TR, PL, false, 1>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, 0, ws, lane);
  else if (lean == 2) ps_epilogue<TM, TN, TR, PL, false, 2>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, 0, ws, lane);
  else if (lean == 3) ps_epilogue<TM, TN, TR, PL, false, 3>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, 0, ws, lane);
  else if (lean == 4) ps_epilogue<TM, TN, TR, PL, false, 4>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, 0, ws, lane);
  else ps_epilogue<TM, TN, TR, PL>(p, acc, m0 + wm * 32 * TM, n0 + wn * 32 * TN, splitk, ks, 0, ws, lane);
}

// GroupNorm scale / shift [+ SiLU] of (the channel concat of) x0 | x1 written ONCE as the [n hw][C] matrix in the F16X2 form of
// the PS layout -- ldmk_gn_apply with the output the conv-mode pre-split tile multiplies.  Workgroup = one 32-row block x 64
// channels, thread = (row, 8 consecutive channels): 16-byte stores, 512-byte runs per plane.
__global__ __launch_bounds__(256) void gn_apply_ps_h2_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1, int c1,
                                                             const float* __restrict__ coef, unsigned char* __restrict__ dst, long long rows,
                                                             int hw, int silu, int* __restrict__ range_flag) {
  const int rb = blockIdx.x, kc = blockIdx.y;
  const int r = threadIdx.x >> 3, o = threadIdx.x & 7;
  const long long row = (long long)rb * 32 + r;
  const int C = c0 + c1;
  const int k0 = kc * 64 + o * 8;
  if (k0 >= C) return;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (row < rows) {
    const float* src = k0 < c0 ? x0 + row * c0 + k0 : x1 + row * c1 + (k0 - c0);
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    const float* cf = coef + ((row / hw) * 2) * C + k0;
    const float4 s0 = *reinterpret_cast<const float4*>(cf), s1 = *reinterpret_cast<const float4*>(cf + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(cf + C), h1 = *reinterpret_cast<const float4*>(cf + C + 4);
    v0 = make_float4(fmaf(a.x, s0.x, h0.x), fmaf(a.y, s0.y, h0.y), fmaf(a.z, s0.z, h0.z), fmaf(a.w, s0.w, h0.w));
    v1 = make_float4(fmaf(b.x, s1.x, h1.x), fmaf(b.y, s1.y, h1.y), fmaf(b.z, s1.z, h1.z), fmaf(b.w, s1.w, h1.w));
    if (silu) {
      v0 = make_float4(silu_f(v0.x), silu_f(v0.y), silu_f(v0.z), silu_f(v0.w));
      v1 = make_float4(silu_f(v1.x), silu_f(v1.y), silu_f(v1.z), silu_f(v1.w));
    }
    if (ps_h2_out_of_range(v0) || ps_h2_out_of_range(v1)) *range_flag = 1;
  }
  unsigned char* d = dst + ((long long)rb * (C / 16) + (k0 >> 4)) * 2048 + ((o & 1) * 32 + r) * 16;
  pf16x4 a0, l0, a1, l1;
  ps_split2h(ps_scaled_sat(v0), a0, l0);
  ps_split2h(ps_scaled_sat(v1), a1, l1);
  *reinterpret_cast<pf16x8*>(d) = pf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  *reinterpret_cast<pf16x8*>(d + 1024) = pf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

template <int NWM, int NWN, int TM, int TN, bool TR>
static int psc_launch(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  constexpr int BM = 32 * TM * NWM, BN = 32 * TN * NWN;
  constexpr size_t lds = (size_t)3 * (BM / 32 + BN / 32) * 2048;
  static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_psc_kernel<NWM, NWN, TM, TN, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  static const int nfast_env = [] { const char* e = getenv("LDMK_PS_NFAST"); return e ? atoi(e) : 1; }();
  static const int lean_env = [] { const char* e = getenv("LDMK_PS_LEAN"); return e ? atoi(e) : 1; }();
  const int nfast = (nfast_env && (a.N + BN - 1) / BN > 1 && (a.M + BM - 1) / BM >= 8 ? 1 : 0) | (lean_env ? 0 : 2);
  hipLaunchKernelGGL((igemm_psc_kernel<NWM, NWN, TM, TN, TR>), dim3(tiles, splitk, 1), dim3(64 * NWM * NWN), lds, st, a, splitk, ws, nfast);
  if (splitk > 1 && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  return check_launch("ldmk_igemm(ps, conv)");
}

const char* igemm_ps_unsupported(const ldmk_igemm_args& a, int pcfg, int splitk) {
  if (a.compute != LDMK_COMPUTE_BF16X3 && a.compute != LDMK_COMPUTE_F16X2) return "the pre-split tiles run the bf16x3 and the f16x2 arithmetic (compute)";
  const int pl = a.compute == LDMK_COMPUTE_F16X2 ? 2 : 3;
  if (pl == 2 && (pcfg == 6 || pcfg == 7)) return "the warp-specialised pre-split tiles (29 / 30) exist in the bf16x3 arithmetic only";
  if (pl == 2 && a.out_ps && !a.range_flag) return "out_ps in the f16x2 arithmetic needs range_flag";
  if (!a.a_ps || !a.w_ps) return "a_ps / w_ps (operands in the PS layout: ldmk_pack_ps, ldmk_ln_stats_ps, out_ps of a producer GEMM)";
  if (a.a_mode == LDMK_A_CONV3X3) {      // the conv-mode tile (igemm_psc_kernel)
    if (pl != 2) return "the conv-mode pre-split tile runs the f16x2 arithmetic only";
    if (pcfg != 0 && pcfg != 1 && pcfg != 3 && pcfg != 4) return "conv mode: tile_cfg 23 / 24 / 26 / 27";
    if (a.c1 || a.a1 || a.upsample || a.a_tf != LDMK_TF_NONE || a.out_ps || a.batch > 1 || a.epi != LDMK_EPI_NONE || a.attn_kv_out)
      return "conv mode: one pre-normalised source (ldmk_gn_apply_ps_h2), no upsampling / prologue / out_ps / batching / GEGLU";
    if (a.c0 % 32) return "conv mode: C_in must be a multiple of 32";
    if ((a.c0 / 32) % splitk) return "conv mode: splitk must divide C_in / 32 (K is split on 32-channel chunk boundaries)";
    const long long smp = ((long long)a.M + a.rows_per_sample - 1) / a.rows_per_sample;
    if (((smp * a.in_h * a.in_w + 31) / 32) * (a.c0 / 16) * 2048 >= (1LL << 31)) return "conv mode: input of 2 GiB or more";
  } else if (a.a_mode != LDMK_A_ROWS) {
    return "rows mode or 3x3 convolution";
  }
  if (a.b_trans || a.upsample || a.skip_a0 || (a.splitk_counters && !(ps_probe_bits() & 8))) return "b_trans / upsample / fused skip / in-launch combine";
  if (a.a_tf != LDMK_TF_NONE && a.a_tf != LDMK_TF_LAYERNORM_FOLDED) return "no staging prologue: a_ps is what gets multiplied";
  if (a.K % 32 || a.N % 32) return "K and N must be multiples of 32";
  const long long kb = a.K / 16;
  if ((a.a_mode == LDMK_A_ROWS && (long long)((a.M + 31) / 32) * kb * pl * 1024 >= (1LL << 31)) || (long long)(a.N / 32) * kb * pl * 1024 >= (1LL << 31))
    return "an operand of 2 GiB or more";
  if (a.epi == LDMK_EPI_GEGLU && (!kPsCfg[pcfg].even_tn || splitk > 1)) return "GEGLU needs a tile of (value, gate) pairs (256x256, 128x256) and no split-K";
  if (splitk > a.K / 32) return "more K slices than 32-deep chunks";
  if (a.out_ps) {
    if (splitk > 1 || a.batch > 1 || a.stats_out) return "out_ps: no split-K, no batching, no GroupNorm records (transposed epilogue)";
    if (a.ldc % 16) return "out_ps: ldc (the PS output's column count) must be a multiple of 16";
  } else if (!a.out && splitk <= 1) {
    return "no output";
  }
  if (a.stats_out && !a.out) return "stats_out needs out";
  if (a.attn_kv_out) {
    if (pl != 2 || (pcfg != 0 && pcfg != 4)) return "attn_kv_out: the F16X2 arithmetic on tile_cfg 23 / 27";
    if (a.stats_out || splitk > 1 || a.batch > 1 || a.epi != LDMK_EPI_NONE || a.out_ps || !a.out || !a.range_flag)
      return "attn_kv_out: the fused QKV projection (no split-K / batching / GEGLU / records / out_ps; out and range_flag required)";
    if (a.attn_heads <= 0 || a.N != 3 * 32 * a.attn_heads) return "attn_kv_out: N must be 3 x 32 x attn_heads ([q | k | v] of d_head 32)";
    if (a.attn_tokens <= 0 || a.attn_tokens % 64 || a.M % a.attn_tokens) return "attn_kv_out: attn_tokens a multiple of 64 that divides M";
  }
  return nullptr;
}

int igemm_ps_dispatch(const ldmk_igemm_args& a, int pcfg, int splitk, float* ws, hipStream_t st) {
  // the transposed epilogue serves every call but those that want GroupNorm records
  const bool tr = !a.stats_out;
  if (a.a_mode == LDMK_A_CONV3X3) {            // (unsupported() admits F16X2 on pcfg 0 / 1 / 3 / 4)
    switch (pcfg) {
      case 0: return tr ? psc_launch<8, 1, 1, 5, true>(a, splitk, ws, st) : psc_launch<8, 1, 1, 5, false>(a, splitk, ws, st);
      case 1: return tr ? psc_launch<4, 2, 2, 5, true>(a, splitk, ws, st) : psc_launch<4, 2, 2, 5, false>(a, splitk, ws, st);
      case 3: return tr ? psc_launch<4, 2, 1, 5, true>(a, splitk, ws, st) : psc_launch<4, 2, 1, 5, false>(a, splitk, ws, st);
      default: return tr ? psc_launch<4, 1, 1, 5, true>(a, splitk, ws, st) : psc_launch<4, 1, 1, 5, false>(a, splitk, ws, st);
    }
  }
  if (a.compute == LDMK_COMPUTE_F16X2 && a.attn_kv_out)      // the fused QKV projection writing the attention's K / V tiles (unsupported() admits 0 / 4)
    return pcfg == 0 ? ps_launch<8, 1, 1, 5, 3, true, 2, true>(a, splitk, ws, st) : ps_launch<4, 1, 1, 5, 3, true, 2, true>(a, splitk, ws, st);
  if (a.compute == LDMK_COMPUTE_F16X2) {       // two fp16 planes per operand, three matrix instructions per product; one more ring stage fits
    switch (pcfg) {
      case 0: return tr ? ps_launch<8, 1, 1, 5, 3, true, 2>(a, splitk, ws, st) : ps_launch<8, 1, 1, 5, 3, false, 2>(a, splitk, ws, st);
      case 1: return tr ? ps_launch<4, 2, 2, 5, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 2, 2, 5, 3, false, 2>(a, splitk, ws, st);
      case 2: return tr ? ps_launch<4, 2, 2, 4, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 2, 2, 4, 3, false, 2>(a, splitk, ws, st);
      case 3: return tr ? ps_launch<4, 2, 1, 5, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 2, 1, 5, 3, false, 2>(a, splitk, ws, st);
      case 4: return tr ? ps_launch<4, 1, 1, 5, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 1, 1, 5, 3, false, 2>(a, splitk, ws, st);
      case 8: return tr ? ps_launch<4, 1, 2, 5, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 1, 2, 5, 3, false, 2>(a, splitk, ws, st);
      case 9: return tr ? ps_launch<4, 1, 2, 4, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 1, 2, 4, 3, false, 2>(a, splitk, ws, st);
      case 10: return tr ? ps_launch<2, 2, 2, 4, 3, true, 2>(a, splitk, ws, st) : ps_launch<2, 2, 2, 4, 3, false, 2>(a, splitk, ws, st);
      default: return tr ? ps_launch<4, 2, 1, 4, 3, true, 2>(a, splitk, ws, st) : ps_launch<4, 2, 1, 4, 3, false, 2>(a, splitk, ws, st);
    }
  }
  switch (pcfg) {
    case 0: return tr ? ps_launch<8, 1, 1, 5, 3, true>(a, splitk, ws, st) : ps_launch<8, 1, 1, 5, 3, false>(a, splitk, ws, st);
    case 1: return tr ? ps_launch<4, 2, 2, 5, 2, true>(a, splitk, ws, st) : ps_launch<4, 2, 2, 5, 2, false>(a, splitk, ws, st);
    case 2: return tr ? ps_launch<4, 2, 2, 4, 3, true>(a, splitk, ws, st) : ps_launch<4, 2, 2, 4, 3, false>(a, splitk, ws, st);
    case 3: return tr ? ps_launch<4, 2, 1, 5, 3, true>(a, splitk, ws, st) : ps_launch<4, 2, 1, 5, 3, false>(a, splitk, ws, st);
    case 4: return tr ? ps_launch<4, 1, 1, 5, 2, true>(a, splitk, ws, st) : ps_launch<4, 1, 1, 5, 2, false>(a, splitk, ws, st);
    case 6: return tr ? pw_launch<2, 5, 4, true>(a, splitk, ws, st) : pw_launch<2, 5, 4, false>(a, splitk, ws, st);
    case 7: return tr ? pw_launch<2, 4, 4, true>(a, splitk, ws, st) : pw_launch<2, 4, 4, false>(a, splitk, ws, st);
    case 8: return tr ? ps_launch<4, 1, 2, 5, 2, true>(a, splitk, ws, st) : ps_launch<4, 1, 2, 5, 2, false>(a, splitk, ws, st);
    case 9: return tr ? ps_launch<4, 1, 2, 4, 2, true>(a, splitk, ws, st) : ps_launch<4, 1, 2, 4, 2, false>(a, splitk, ws, st);
    case 10: return tr ? ps_launch<2, 2, 2, 4, 2, true>(a, splitk, ws, st) : ps_launch<2, 2, 2, 4, 2, false>(a, splitk, ws, st);
    default: return tr ? ps_launch<4, 2, 1, 4, 3, true>(a, splitk, ws, st) : ps_launch<4, 2, 1, 4, 3, false>(a, splitk, ws, st);
  }
}

}  // namespace ldmk

extern "C" long long ldmk_ps_bytes(int rows, int k) {
  if (rows <= 0 || k <= 0 || k % 16) return -1;
  return (long long)((rows + 31) / 32) * (k / 16) * 3072;
}

extern "C" long long ldmk_ps_bytes_h2(int rows, int k) {
  if (rows <= 0 || k <= 0 || k % 16) return -1;
  return (long long)((rows + 31) / 32) * (k / 16) * 2048;
}

extern "C" int ldmk_pack_ps(const float* src, int rows, int k, long long row_stride, long long k_stride, int batch, long long src_bstride,
                            void* dst, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(src && dst && rows > 0 && k > 0 && batch >= 1, "ldmk_pack_ps: bad args");
  LDMK_REQUIRE(k % 16 == 0, "ldmk_pack_ps: k=%d must be a multiple of 16", k);
  LDMK_REQUIRE(k_stride != 1 || (row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && src_bstride % 4 == 0),
               "ldmk_pack_ps: row-major sources are read as float4 (16-byte aligned rows)");
  const long long bytes = ldmk_ps_bytes(rows, k);
  hipLaunchKernelGGL(pack_ps_kernel<3>, dim3((rows + 31) / 32, (k + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, src, rows, k, row_stride,
                     k_stride, src_bstride, reinterpret_cast<unsigned char*>(dst), bytes, 1.0f, (int*)nullptr);
  return check_launch("ldmk_pack_ps");
}

extern "C" int ldmk_pack_ps_h2(const float* src, int rows, int k, long long row_stride, long long k_stride, int batch, long long src_bstride,
                               int scale_exp, int* range_flag, void* dst, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(src && dst && rows > 0 && k > 0 && batch >= 1, "ldmk_pack_ps_h2: bad args");
  LDMK_REQUIRE(k % 16 == 0, "ldmk_pack_ps_h2: k=%d must be a multiple of 16", k);
  LDMK_REQUIRE(k_stride != 1 || (row_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && src_bstride % 4 == 0),
               "ldmk_pack_ps_h2: row-major sources are read as float4 (16-byte aligned rows)");
  LDMK_REQUIRE(scale_exp >= -60 && scale_exp <= 60, "ldmk_pack_ps_h2: scale_exp=%d outside [-60, 60]", scale_exp);
  LDMK_REQUIRE(!range_flag || scale_exp == LDMK_F16X2_A_EXP, "ldmk_pack_ps_h2: range_flag is the activations' check: scale_exp must be %d", LDMK_F16X2_A_EXP);
  const long long bytes = ldmk_ps_bytes_h2(rows, k);
  hipLaunchKernelGGL(pack_ps_kernel<2>, dim3((rows + 31) / 32, (k + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, src, rows, k, row_stride,
                     k_stride, src_bstride, reinterpret_cast<unsigned char*>(dst), bytes, ldexpf(1.f, scale_exp), range_flag);
  return check_launch("ldmk_pack_ps_h2");
}

extern "C" int ldmk_ln_stats_ps(const float* x, int rows, int c, float eps, float* stats, void* dst, float guard, int* flag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && stats && dst && rows > 0 && c > 0, "ldmk_ln_stats_ps: bad args");
  LDMK_REQUIRE(c % 16 == 0 && c <= 1280, "ldmk_ln_stats_ps: C=%d must be a multiple of 16, at most 1280", c);
  LDMK_REQUIRE(!flag || guard > 0.f, "ldmk_ln_stats_ps: guard=%g must be positive", (double)guard);
  const dim3 grid((rows + 31) / 32);
  unsigned char* d = reinterpret_cast<unsigned char*>(dst);
  hipStream_t st = (hipStream_t)stream;
  int* nf = nullptr;
  if (c <= 192) hipLaunchKernelGGL(ln_stats_ps_kernel<3>, grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, nf);
  else if (c <= 320) hipLaunchKernelGGL(ln_stats_ps_kernel<5>, grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, nf);
  else if (c <= 640) hipLaunchKernelGGL(ln_stats_ps_kernel<10>, grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, nf);
  else hipLaunchKernelGGL(ln_stats_ps_kernel<20>, grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, nf);
  return check_launch("ldmk_ln_stats_ps");
}

extern "C" int ldmk_ln_stats_ps_h2(const float* x, int rows, int c, float eps, float* stats, void* dst, float guard, int* flag, int* range_flag,
                                   void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && stats && dst && range_flag && rows > 0 && c > 0, "ldmk_ln_stats_ps_h2: bad args");
  LDMK_REQUIRE(c % 16 == 0 && c <= 1280, "ldmk_ln_stats_ps_h2: C=%d must be a multiple of 16, at most 1280", c);
  LDMK_REQUIRE(!flag || guard > 0.f, "ldmk_ln_stats_ps_h2: guard=%g must be positive", (double)guard);
  const dim3 grid((rows + 31) / 32);
  unsigned char* d = reinterpret_cast<unsigned char*>(dst);
  hipStream_t st = (hipStream_t)stream;
  if (c <= 192) hipLaunchKernelGGL((ln_stats_ps_kernel<3, 2>), grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, range_flag);
  else if (c <= 320) hipLaunchKernelGGL((ln_stats_ps_kernel<5, 2>), grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, range_flag);
  else if (c <= 640) hipLaunchKernelGGL((ln_stats_ps_kernel<10, 2>), grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, range_flag);
  else hipLaunchKernelGGL((ln_stats_ps_kernel<20, 2>), grid, dim3(256), 0, st, x, rows, c, eps, stats, d, guard, flag, range_flag);
  return check_launch("ldmk_ln_stats_ps_h2");
}

extern "C" int ldmk_gn_apply_ps_h2(const float* x0, int c0, const float* x1, int c1, const float* coef, void* y_ps, int n, int hw, int silu,
                                   int* range_flag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x0 && coef && y_ps && range_flag && n > 0 && hw > 0 && c0 > 0, "ldmk_gn_apply_ps_h2: bad args");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr), "ldmk_gn_apply_ps_h2: x1/c1 mismatch");
  LDMK_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0 && (c0 + c1) % 16 == 0, "ldmk_gn_apply_ps_h2: c0 = %d, c1 = %d must be multiples of 8, their sum of 16", c0, c1);
  const long long rows = (long long)n * hw;
  LDMK_REQUIRE(ldmk_ps_bytes_h2((int)rows, c0 + c1) < (1LL << 31) && rows < (1LL << 31), "ldmk_gn_apply_ps_h2: output of 2 GiB or more");
  hipLaunchKernelGGL(gn_apply_ps_h2_kernel, dim3((unsigned)((rows + 31) / 32), (c0 + c1 + 63) / 64), dim3(256), 0, (hipStream_t)stream, x0, c0, x1, c1,
                     coef, reinterpret_cast<unsigned char*>(y_ps), rows, hw, silu, range_flag);
  return check_launch("ldmk_gn_apply_ps_h2");
}
