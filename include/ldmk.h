/* ldmk.h -- C ABI of libldmk.so: MI355X (gfx950) kernels for the latent-diffusion sampling path.
 *
 * The reference (GiannisPikoulis/dsml-thesis) has no FFI: its extension point is the YAML
 * `target:` factory (ldm/util.py:78-93) and everything below that is torch.nn ops.  This header
 * is therefore the boundary the *build* defines: one entry point per PyTorch op call-site family
 * on the hot path.  Each declaration cites the reference code it replaces
 * (paths relative to /root/reference/face_reenactment).
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers to fp32 unless stated
 *  - activations are NHWC ("token-major": [n][y][x][c] == [n*H*W][C]); latents at the sampler
 *    boundary stay NCHW like the reference's tensors
 *  - every call only enqueues work on `stream` (a hipStream_t passed as void*): no allocation,
 *    no synchronisation, no host callbacks -> legal inside hipStreamBeginCapture
 *  - return 0 on success, negative LDMK_E* otherwise; never throws. ldmk_last_error() returns
 *    a thread-local message for the last failure.
 */
#ifndef LDMK_H
#define LDMK_H
#ifdef __cplusplus
extern "C" {
#endif

#define LDMK_OK 0
#define LDMK_EINVAL (-1) /* bad shape / alignment / unsupported combination */
#define LDMK_EHIP (-2)   /* HIP launch error, see ldmk_last_error() */
#define LDMK_ENOMEM (-3) /* the caller's workspace is too small for the pinned plan; size it with ldmk_*_workspace_elems() */

int ldmk_version(void);
const char* ldmk_last_error(void);
/* Once per process and device, before the first launch: selects `device` (hipSetDevice), checks that it is a gfx950 part
 * and raises the dynamic-LDS limits of the GEMM kernels, so that no later call changes a function attribute (every later
 * call only enqueues work: capture-safe).  The library owns no device memory: every buffer, INCLUDING scratch, is the
 * caller's -- a non-PyTorch host sizes scratch with the *_workspace_elems queries below.  Calling a kernel entry point
 * without ldmk_init works too (attributes are then set lazily at the first launch of each tile shape).
 * Returns LDMK_EHIP when there is no such device / it is not gfx950. */
int ldmk_init(int device);

/* ------------------------------------------------------------------------------------------
 * Implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32):
 *     out[M][N] = epilogue( transform(A)[M][K] * W[K][N] )
 * replaces every nn.Conv2d 3x3 / 1x1 and nn.Linear on [n*H*W] rows:
 *   openaimodel.py:204,230 (ResBlock convs) :241 (skip 1x1) :150-153 (Downsample) :107,116-118
 *   (Upsample: nearest x2 folded into the gather) ; attention.py:161-168,173-177 (to_q/k/v,to_out)
 *   :37-64 (GEGLU FF) :232-236,244-248 (proj_in/out) ; model.py:95-129 (ResnetBlock convs,
 *   nin_shortcut) :157-176 (AttnBlock q/k/v/proj) :72-76 (asymmetric-pad stride-2 conv).
 * A-side prologues fuse GroupNorm(+SiLU) (util.py:214-216, openaimodel.py:201-203) and LayerNorm
 * (attention.py:203-205) into the operand staging; epilogues fuse bias, the timestep-embedding
 * add (openaimodel.py:264-273), residual adds (:275, attention.py:211-215,261) and GEGLU.
 */
enum { LDMK_A_ROWS = 0, LDMK_A_CONV3X3 = 1 };
enum { LDMK_TF_NONE = 0, LDMK_TF_AFFINE = 1, LDMK_TF_AFFINE_SILU = 2, LDMK_TF_LAYERNORM = 3,
       /* LayerNorm folded through the product (rows mode): with W' = diag(gamma) W,
        *     LN(x) W + b  ==  rstd * (x W' - mean * colsum(W')) + (beta^T W + b)
        * so A is staged raw (no per-element arithmetic next to the matrix cores) and the epilogue applies the two
        * per-row scalars.  The caller passes w = W' (and w_frag of W'), ln_colsum, bias = beta^T W + b -- all three from
        * ldmk_fold_layernorm -- and row_stats as for LDMK_TF_LAYERNORM; ln_gamma / ln_beta are not read. */
       LDMK_TF_LAYERNORM_FOLDED = 4 };
enum { LDMK_EPI_NONE = 0, LDMK_EPI_GEGLU = 1 };
/* arithmetic of the product.  F32: v_mfma_f32_32x32x2_f32, bit-identical to an fp32 fmaf chain -- the sampling path and
 * every parity test.  BF16: operands rounded to bf16 (RNE) while staged, v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * and fp32 prologue / epilogue -- the mixed-precision training step (BASELINE configs[4]); HBM tensors stay fp32. */
enum { LDMK_COMPUTE_F32 = 0, LDMK_COMPUTE_BF16 = 1, LDMK_COMPUTE_BF16X3 = 2, LDMK_COMPUTE_F16X2 = 3 };
/* F16X2 -- fp32-accurate products from THREE fp16 matrix instructions.  Measured on MI355X (tools/clock_probe.hip,
 * profiles/r04_clock_bf16.txt): a register-only loop of v_mfma_f32_32x32x16_{bf16,f16} on real operands sustains 1.5-1.7 of the
 * nominal 2.5 PFLOP/s -- the chip is power-limited there (2.3 PFLOP/s on all-zero operands) -- so past BF16X3 the lever is the
 * NUMBER of matrix instructions per product.  Each operand is scaled by a power of two into fp16's range and written
 *     x' = 2^e x = hi + lo,   hi = fp16(x'), lo = fp16(x' - hi)     (round-to-nearest-even; x' - hi is exact in fp32)
 * 2 x 11 significand bits: |x' - hi - lo| <= 2^-23 |x'| -- one fp32 ulp, the size of an fp32 rounding error -- for |x'| >= 2^-3
 * (below, lo is subnormal: an absolute 2^-25), where the BF16X3 split is exact.  fp16 x fp16 products are exact in fp32;
 *     a*b ~= 2^-(ea+eb) (lo*hi + hi*lo + hi*hi)                    (dropped: lo*lo <= 2^-22 |a*b|, typically 2^-24)
 * accumulate in one fp32 accumulator, the scales leave in the epilogue (exact).  Against float64 the result has 1.0-1.7 x the
 * RMS error of an fp32 dot product (1.7 at K = 160, 1.1 from K = 1440; tests/test_f16x2_gpu.py bounds it at 2 x) -- the fp32
 * accuracy class, a little behind BF16X3.  Scales: activations 2^LDMK_F16X2_A_EXP, fixed; weights 2^w_scale_exp, chosen per
 * matrix when it is packed (ldmk_pack_wsplit_h2) so that max |w'| lies in [2^13, 2^14).  An activation far below 2^-9 keeps
 * an absolute precision of 2^-31 rather than a relative one: invisible next to O(1) elements of the same row (an fp32
 * accumulation resolves 2^-24 of the sum), but a tensor that is uniformly ~1e-4 comes out with 8 x the fp32 form's error.  An activation with |x| >=
 * LDMK_F16X2_RANGE (or inf / NaN) would leave fp16: the kernels then write 1 to *range_flag (a device int the caller zeroes
 * once; never cleared here) and the caller repeats the work in BF16X3.  tile_cfg 0..6 (A split while it is staged, w_split from
 * ldmk_pack_wsplit_h2) and the pre-split tiles 23..28 / 31..33 (a_ps / w_ps / out_ps as two fp16 planes: ldmk_ps_bytes_h2 below). */
#define LDMK_F16X2_A_EXP 6
#define LDMK_F16X2_RANGE 1000.0f
/* BF16X3 -- fp32-accurate products on the bf16 matrix cores (the f32 MFMA of gfx950 peaks at 157 TFLOP/s, the bf16 one at
 * 2.5 PFLOP/s).  Every fp32 operand is written as the EXACT sum of three bf16 values
 *     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (3 x 8 significand bits = 24)
 * and the product is accumulated in fp32 from the six partial products that are not below fp32 resolution
 *     a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi        (dropped: mid*lo, lo*mid, lo*lo <= 2^-25 |a*b|)
 * with v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32).  Six 32-cycle instructions per 16 k instead of
 * eight 64-cycle ones: 2.67x the matrix rate of the F32 form at the same accuracy class (tests/test_split_gpu.py bounds both
 * against float64).  Activations are split while they are staged; the weights come pre-split (args.w_split, ldmk_pack_wsplit). */

typedef struct ldmk_igemm_args {
  int M, N, K;               /* K = 9*(c0+c1) for LDMK_A_CONV3X3 (weights packed by ldmk_pack_conv3x3), else c0+c1 */
  const float* a0;           /* first A source, NHWC rows of c0 channels                          */
  const float* a1;           /* optional second source (channel concat, openaimodel.py:736), c1   */
  int c0, c1;                /* both multiples of 32                                              */
  int a_mode;                /* LDMK_A_ROWS | LDMK_A_CONV3X3                                      */
  int in_h, in_w;            /* conv: stored input spatial size                                   */
  int out_h, out_w;          /* conv: output spatial size (M = n*out_h*out_w)                     */
  int stride, pad_lo;        /* conv: iy = oy*stride + dy - pad_lo                                */
  int upsample;              /* conv: 1 -> input is nearest-x2 upsampled on the fly; 2 -> zero-inserted x2
                                (data gradient of a stride-2 convolution: odd rows/columns read as 0) */
  int a_tf;                  /* LDMK_TF_*                                                         */
  const float* tf_coef;      /* AFFINE: [n][2][c0+c1] (scale plane, shift plane) from ldmk_gn_*   */
  const float* row_stats;    /* LAYERNORM: [M][2] (mean, rstd) from ldmk_ln_stats                 */
  const float* ln_gamma;     /* LAYERNORM: [K]                                                    */
  const float* ln_beta;      /* LAYERNORM: [K]                                                    */
  int rows_per_sample;       /* rows of A/out per batch item (H*W of the output)                  */
  const float* w;            /* b_trans=0: [K][ldb] row-major (packed) ; 1: [N][ldb] (torch [out][in]) */
  int b_trans, ldb;
  const float* bias;         /* [N] or NULL                                                       */
  const float* batch_vec;    /* [n][batch_vec_ld] per-sample vector added to every row, or NULL   */
  int batch_vec_ld;
  const float* residual;     /* [M][ldc] or NULL (may alias out)                                  */
  int epi;                   /* LDMK_EPI_*; GEGLU: packed (value,gate) 32-column pairs, out has N/2 cols */
  float* out;                /* [M][ldc]                                                          */
  int ldc;
  int batch;                 /* >1: batched GEMM over blockIdx.z with the strides below           */
  long long a_bstride, w_bstride, out_bstride;
  float alpha;               /* scale applied to the product before the epilogue (1.0 default)    */
  int tile_cfg;              /* 0 = choose from the problem size; 1..6 = pin an LDS-tiled workgroup shape, 7..12 = pin
                                a row-GEMM wave tile (32 TM x 32 TN: 1x5, 2x5, 1x4, 2x4, 1x2, 1x1), 13..20 = pin a
                                slab-GEMM shape for small row counts (csrc/sgemm.hip; wave tile x waves that split K
                                inside the workgroup: 2x1x4, 2x2x4, 1x1x4, 1x2x4, 1x1x8, 1x1x16, 2x1x8, 1x2x8; needs
                                w_frag; stride-1 3x3 convolutions and rows mode), 21..22 = pin a warp-specialised tile of
                                the LDMK_COMPUTE_BF16X3 arithmetic (csrc/igemm_ws.hip: 256x160 / 256x128, four consumer +
                                four producer waves; bitwise the results of tile_cfg 5 / 1 at equal splitk), 23..33 = pin a
                                pre-split tile (csrc/igemm_ps.hip: 23..28 = 256x160, 256x320, 256x256, 128x320, 128x160,
                                128x256; 29 / 30 = the warp-specialised 256x160 / 256x128 with four consumer + four LDS-DMA
                                waves; 31..33 = 256x160, 256x128, 128x256 on four waves with 2x2 / 2x4 wave tiles;
                                needs a_ps and w_ps, see the end of this struct).  The K-summation
                                order depends on (tile_cfg, splitk), so a caller that needs results that are
                                bitwise independent of the batch size pins both (ldmk_igemm_plan)          */
  int splitk;                /* 0 = choose; 1 = none; 2..64 = split K over that many workgroups            */
  float* splitk_ws;          /* scratch for split-K partial slabs (batch*splitk*M*N floats) or NULL        */
  long long splitk_ws_elems; /* capacity of splitk_ws in floats                                           */
  float* stats_out;          /* optional [M/32][N][3] GroupNorm partial records of the *output* (after the
                                epilogue), one per 32-row tile and column; needs M%32==0, rows_per_sample%32==0 */
  int compute;               /* LDMK_COMPUTE_*: F32 (default), BF16, BF16X3 or F16X2 matrix-core arithmetic (tile_cfg 1..6 only) */
  int* splitk_counters;      /* optional: >= ceil(M/64)*ceil(N/64)*batch ints, ZEROED once by the caller.  With it a split-K
                                GEMM is ONE launch: each tile's last-arriving workgroup sums the slabs (fixed order: bitwise
                                reproducible) and runs the epilogue; every launch leaves the counters zeroed again, so one
                                array serves all GEMMs of a stream.  NULL: the slabs are summed by a second (reduce) launch,
                                which is the FASTER form whenever the output has fewer tiles than the chip has CUs -- the
                                usual reason to split K -- because the combine then runs on those few workgroups only */
  int splitk_counters_len;
  const float* w_frag;       /* optional second copy of w in MFMA-fragment order (ldmk_pack_wfrag).  With it, rows-mode
                                problems may run on the wave-autonomous row GEMM (tile_cfg 7..12: no LDS, no barrier;
                                csrc/rgemm.hip), which is what the short-K Linear layers of the transformer blocks want */
  const float* ln_colsum;    /* LDMK_TF_LAYERNORM_FOLDED: [N] column sums of w (= diag(gamma) W), see ldmk_fold_layernorm */
  int raw_slabs;             /* != 0 (split-K plans only, splitk >= 2, batch <= 1): leave the raw partial slabs
                                [splitk][M][N] in splitk_ws and launch NO reduce kernel -- the consumer (ldmk_post,
                                ldmk_attn_self with qkv_slabs) sums them in slab order and applies bias / per-sample vector /
                                residual / GEGLU itself; `out`, `bias`, `batch_vec`, `residual` and `epi` are then not used */
  const float* skip_a0;      /* slab GEMM, LDMK_A_CONV3X3 only: K = 9*c0 + skip_c0 + skip_c1, the trailing K columns multiply the ROWS */
  const float* skip_a1;      /*   of the (two-source) tensor skip_a0 | skip_a1 at the output pixel -- the ResBlock's 1x1      */
  int skip_c0, skip_c1;      /*   skip_connection as extra K of its second 3x3 convolution (openaimodel.py:241,275): one GEMM  */
                             /*   instead of two.  Weights: the packed conv weight with the [skip_c0+skip_c1][N] matrix appended. */
  const void* w_split;       /* LDMK_COMPUTE_BF16X3 (b_trans = 0, tile_cfg 0..6): the weights as three bf16 images (hi, mid, lo)   */
  int w_split_ld;            /*   [3][N][w_split_ld], K-contiguous rows (w_split_ld >= K, a multiple of 8), made by               */
  long long w_split_bstride; /*   ldmk_pack_wsplit from `w`; batched GEMMs step w_split_bstride bf16 ELEMENTS per batch entry      */
  const void* a_split;       /* optional, LDMK_COMPUTE_BF16X3 on tile_cfg 0..6, rows mode, one source, a_tf NONE / LAYERNORM_FOLDED:  */
  int a_split_ld;            /*   the A operand pre-split as well, three bf16 images [3][M][a_split_ld] of a0 (ldmk_ln_stats_split   */
                             /*   writes them next to the row statistics): the kernel copies instead of splitting per N-tile.        */
                             /*   Results are bitwise those of the in-kernel split.                                                  */
  /* ---- tile_cfg 23..33 (csrc/igemm_ps.hip): LDMK_COMPUTE_BF16X3 on operands that are BOTH pre-split, in the PS layout:
   * for a matrix X[R][K] (K % 16 == 0) plane g (0 hi, 1 mid, 2 lo) of element (r, k) is the bf16 at index
   *     (((r / 32) (K / 16) + k / 16) 3 + g) 512 + ((k / 8 % 2) 32 + r % 32) 8 + k % 8
   * i.e. per (32-row block, 16-deep k-slab) three consecutive 1-KiB planes, each in the lane order of the MFMA operand; rows
   * R .. 32 ceil(R / 32) - 1 are padding (any finite or non-finite content: it only reaches output rows that are masked).
   * ldmk_ps_bytes(R, K) is the size.  The kernel moves these planes memory -> LDS with LDS-DMA loads and does no arithmetic
   * on them: rows mode, a_tf NONE or LAYERNORM_FOLDED, K % 32 == 0, N % 32 == 0.  Same products, product order and split-K
   * partition as tile_cfg 1 / 5: bitwise equal results at equal splitk. */
  const void* a_ps;          /* A[M][K] in the PS layout (ldmk_pack_ps, ldmk_ln_stats_ps, out_ps of a producer GEMM, ...); a0 is not read */
  long long a_ps_bstride;    /*   BYTES between batch entries                                                               */
  const void* w_ps;          /* the weights as X[N][K] (row = output column) in the PS layout: ldmk_pack_ps(w, N, K, 1, ldb)   */
  long long w_ps_bstride;    /*   BYTES between batch entries                                                               */
  void* out_ps;              /* optional: the result [M][ldc] written in the PS layout too (the A operand of the next GEMM; every  */
                             /*   element is split once, by its producer); `out` may then be NULL.  No split-K / batch / stats_out. */
  /* ---- LDMK_COMPUTE_F16X2 */
  int w_scale_exp;           /* the exponent ldmk_pack_wsplit_h2 scaled the weights by (w_split = the two fp16 images of 2^e W)    */
  int* range_flag;           /* device int: set to 1 when a staged activation leaves the scaled fp16 range (see above)              */
  /* ---- the fused QKV projection feeding ldmk_attn_self_h2_tiles (LDMK_COMPUTE_F16X2 on tile_cfg 23 / 27, no split-K):      */
  void* attn_kv_out;         /* optional: the K and V column thirds of the result ([q | k | v], N = 3 x 32 x attn_heads) are NOT   */
  int attn_tokens;           /*   stored in `out` but written as the attention's pre-split K / V tiles                               */
  int attn_heads;            /*   (ldmk_attn_kv_split_h2_bytes(M / attn_tokens, attn_tokens, attn_heads) bytes; attn_tokens % 64 == 0, */
                             /*   dividing M): bit for bit what ldmk_attn_self_h2's pre-pass writes from the fp32 K / V               */
} ldmk_igemm_args;

/* size in bytes of the PS layout of a [rows][k] matrix (-1: k not a multiple of 16) */
long long ldmk_ps_bytes(int rows, int k);
/* X[r][k] = src[r row_stride + k k_stride] (fp32; `batch` matrices src_bstride floats apart) -> PS layout, ldmk_ps_bytes(rows, k)
 * bytes per batch entry.  Activations [rows][ld]: (ld, 1); packed weights w[K][ldb] as X[N][K]: rows = N, (1, ldb). */
int ldmk_pack_ps(const float* src, int rows, int k, long long row_stride, long long k_stride, int batch, long long src_bstride,
                 void* dst, void* stream);
/* ldmk_ln_stats_guard and the rows themselves in the PS layout (the statistics pass reads every element anyway): what the
 * LDMK_TF_LAYERNORM_FOLDED GEMMs on tile_cfg 23..33 take as a_ps.  C % 16 == 0, C <= 1280; two-pass statistics in registers. */
int ldmk_ln_stats_ps(const float* x, int rows, int c, float eps, float* stats, void* dst, float guard, int* flag, void* stream);

/* The PS layout of the F16X2 arithmetic: the same index arithmetic with TWO fp16 planes (hi, lo of 2^e x) per unit -- 2-KiB units,
 * ldmk_ps_bytes_h2 bytes.  tile_cfg 23..28 / 31..33 with compute = LDMK_COMPUTE_F16X2 take a_ps / w_ps (and write out_ps) in this
 * form: activations scaled by 2^LDMK_F16X2_A_EXP by their producers, which also raise range_flag (ldmk_pack_ps_h2 with
 * scale_exp = LDMK_F16X2_A_EXP and a flag, ldmk_ln_stats_ps_h2, a GEMM's out_ps); weights by 2^w_scale_exp (ldmk_pack_ps_h2 with
 * range_flag = NULL).  Same products in the same order as tile_cfg 1 / 5 in that arithmetic: bitwise equal results. */
long long ldmk_ps_bytes_h2(int rows, int k);
int ldmk_pack_ps_h2(const float* src, int rows, int k, long long row_stride, long long k_stride, int batch, long long src_bstride,
                    int scale_exp, int* range_flag, void* dst, void* stream);
int ldmk_ln_stats_ps_h2(const float* x, int rows, int c, float eps, float* stats, void* dst, float guard, int* flag, int* range_flag,
                        void* stream);
/* ldmk_gn_apply (GroupNorm scale / shift [+ SiLU] of the channel concat x0 | x1, openaimodel.py:201-203,225-227) with its result
 * written ONCE as the [n hw][c0 + c1] matrix in the F16X2 form of the PS layout: the A operand of a 3x3 convolution on a conv-mode
 * pre-split tile -- ldmk_igemm with a_mode = LDMK_A_CONV3X3, a_ps = this buffer (c0 = the channel count, c1 = 0, a_tf = NONE),
 * w_ps = ldmk_pack_ps_h2 of the ldmk_pack_conv3x3 weights as X[N][K = 9 C], compute = LDMK_COMPUTE_F16X2 on tile_cfg 23 / 24 / 26 /
 * 27, splitk a divisor of C / 32.  The nine taps are per-lane LDS-DMA addresses into this one tensor (the halo reads zeros): bitwise
 * the result of tile_cfg 5 / 1 in LDMK_COMPUTE_F16X2 on the fp32 ldmk_gn_apply output at equal splitk.  c0, c1 multiples of 8, their
 * sum of 16. */
int ldmk_gn_apply_ps_h2(const float* x0, int c0, const float* x1, int c1, const float* coef, void* y_ps, int n, int hw, int silu,
                        int* range_flag, void* stream);

/* w[K][ldb] fp32 (row-major, as ldmk_igemm reads it with b_trans = 0; `batch` matrices w_bstride floats apart) -> the three
 * bf16 images [batch][3][N][ld_out] of its exact three-way split, transposed so that every output column's K run is
 * contiguous.  ld_out >= K, a multiple of 8; columns K..ld_out-1 are zero-filled. */
int ldmk_pack_wsplit(const float* w, int K, int N, int ldb, int batch, long long w_bstride, void* out, int ld_out, void* stream);
/* LDMK_COMPUTE_F16X2: the two fp16 images [batch][2][N][ld_out] (hi, lo) of 2^scale_exp w, same transposed layout.  The caller
 * picks scale_exp so that max |2^scale_exp w| lies in [2^13, 2^14) and passes it on as args.w_scale_exp. */
int ldmk_pack_wsplit_h2(const float* w, int K, int N, int ldb, int batch, long long w_bstride, int scale_exp, void* out, int ld_out,
                        void* stream);
/* the first image alone: w rounded to bf16 (nearest even), transposed, K-contiguous -- [N][ld_out].  Passed as args.w_split with
 * compute = LDMK_COMPUTE_BF16 (b_trans = 0) it replaces the in-kernel conversion of the fp32 weights: the training step packs
 * its forward weights once per optimiser step and every GEMM reads B fragments as single 16-byte LDS vectors. */
int ldmk_pack_wbf16t(const float* w, int K, int N, int ldb, void* out, int ld_out, void* stream);

int ldmk_igemm(const ldmk_igemm_args* args, void* stream);
/* Scratch (in floats) ldmk_igemm(args) needs in args->splitk_ws: batch * splitk * M * N for a split-K plan, 0 otherwise.
 * With args->splitk pinned (> 0) that is what the call will require (too little -> LDMK_ENOMEM); with splitk == 0 it is
 * what the plan chosen with unlimited scratch would use (ldmk_igemm itself never fails for lack of scratch in that case:
 * it plans within the scratch it is given).  Host flow without PyTorch:
 *     ldmk_igemm_plan(&a, &a.tile_cfg, &a.splitk);  n = ldmk_igemm_workspace_elems(&a);  a.splitk_ws = my_alloc(4 * n);
 *     a.splitk_ws_elems = n;  ldmk_igemm(&a, stream);
 * Returns a negative LDMK_E* code for invalid arguments. */
long long ldmk_igemm_workspace_elems(const ldmk_igemm_args* args);
/* every check ldmk_igemm(args) makes (shapes, alignment, the pinned tile_cfg / splitk against this problem, scratch
 * size), without launching: LDMK_OK or the code ldmk_igemm would return.  Lets a host validate a whole launch program
 * up front, and a planner ask "can this wave tile run this problem" before pinning it. */
int ldmk_igemm_check(const ldmk_igemm_args* args);
/* W[K][ldb] (row-major, as ldmk_igemm reads it with b_trans = 0) -> the fragment-order copy `w_frag` of K*N floats:
 * Wf[k/8][n/32][h][n%32][s] = W[8(k/8) + 4h + s][n], one contiguous 1-KiB wave load per four MFMAs.  K%8 == 0, N%32 == 0.
 * ldmk_wfrag_elems returns the size of that copy in floats (-1 when the shape cannot be packed). */
long long ldmk_wfrag_elems(int K, int N);
int ldmk_pack_wfrag(const float* w, int ldb, int K, int N, float* wfrag, void* stream);
/* Operands of LDMK_TF_LAYERNORM_FOLDED from a Linear's packed weight w[K][ldb] (N used columns), the LayerNorm's
 * gamma[K] / beta[K] and the Linear's bias[N] (or NULL):  w_out[k][n] = gamma[k] w[k][n] (row stride N),
 * colsum[n] = sum_k w_out[k][n], bias_out[n] = bias[n] + sum_k beta[k] w[k][n]  (sums in double). */
int ldmk_fold_layernorm(const float* w, int ldb, int K, int N, const float* gamma, const float* beta, const float* bias,
                        float* w_out, float* colsum, float* bias_out, void* stream);
/* the (tile_cfg, splitk) ldmk_igemm would choose for these sizes; reads M, N, K, epi, batch, splitk_ws* */
int ldmk_igemm_plan(const ldmk_igemm_args* args, int* tile_cfg, int* splitk);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) around a batched ldmk_igemm: a stride-1, pad-1 nn.Conv2d 3x3 (openaimodel.py:204,230) with 4
 * instead of 9 multiplications per output and input channel.  H and W even; tiles = n (H/2) (W/2).
 *   ldmk_winograd_input : V[16][tiles][c0+c1] = B^T d B of the 4x4 patches of (the channel concat of) x0 | x1 (NHWC), after
 *                         the optional GroupNorm scale/shift planes `coef` ([n][2][C], ldmk_gn_coef) and SiLU; padding
 *                         taps are zeros of the activated tensor.
 *   ldmk_igemm          : batch = 16, M = tiles, K = c0+c1, N = cout, a0 = V (a_bstride = tiles K), w = U (w_bstride = K N,
 *                         U[p] = (G g G^T)[p] as [K][N]: dsml_thesis_amd.ops.pack_winograd), out = M (out_bstride = tiles N)
 *   ldmk_winograd_output: out (NHWC) = A^T M A + bias + batch_vec[sample] + residual; optional stats_out = the GroupNorm
 *                         partial records of the result ([n H W / 32][cout][3], as ldmk_igemm's stats_out).
 */
long long ldmk_winograd_tiles(int n, int h, int w);
int ldmk_winograd_input(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h, int w,
                        float* v, void* stream);
/* ldmk_winograd_input writing V in the PS layout: 16 planes of ldmk_ps_bytes(tiles, c0 + c1) bytes each -- the a_ps operand
 * (a_ps_bstride = that size) of the batched GEMM on the pre-split tiles.  c0, c1 multiples of 16.  V is bit for bit the matrix
 * ldmk_winograd_input writes, split exactly. */
int ldmk_winograd_input_ps(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h, int w,
                           void* v_ps, void* stream);
/* the same in the F16X2 form of the layout (16 planes of ldmk_ps_bytes_h2(tiles, c0 + c1) bytes; V scaled by 2^6, range-checked) */
int ldmk_winograd_input_ps_h2(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h, int w,
                              void* v_ps, int* range_flag, void* stream);
int ldmk_winograd_output(const float* m, const float* bias, const float* batch_vec, int batch_vec_ld, const float* residual,
                         float* out, float* stats_out, int n, int h, int w, int cout, void* stream);

/* Nearest-x2 upsampling + Conv2d 3x3 (openaimodel.py:107-118) as four 2x2-tap convolutions on the low-resolution input
 * (one per output parity; exact: the collapsed taps carry the summed weights) -- 4/9 of the multiplications:
 *   ldmk_upconv_gather : A[4][n h w][4 c] (tap-major K), zeros outside the image
 *   ldmk_igemm         : batch = 4, M = n h w, K = 4 c, N = cout, w from dsml_thesis_amd.ops.pack_upconv (w_bstride = K N)
 *   ldmk_upconv_scatter: out (n, 2h, 2w, cout) NHWC = the 4 planes interleaved + bias; optional GroupNorm partial records */
int ldmk_upconv_gather(const float* x, int c, int n, int h, int w, float* a, void* stream);
/* ldmk_upconv_gather writing the four phase operands in the PS layout: 4 planes of ldmk_ps_bytes(n h w, 4 c) bytes. c % 16 == 0. */
int ldmk_upconv_gather_ps(const float* x, int c, int n, int h, int w, void* a_ps, void* stream);
int ldmk_upconv_gather_ps_h2(const float* x, int c, int n, int h, int w, void* a_ps, int* range_flag, void* stream);   /* F16X2 form */
int ldmk_upconv_scatter(const float* planes, const float* bias, float* out, float* stats_out, int n, int h, int w, int cout,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation statistics (HBM-bound, wave-shuffle reductions).
 * ldmk_gn_coef: GroupNorm(groups, C) statistics over an NHWC tensor that may be the channel
 *   concat of two tensors (groups may straddle the seam) -> per-(sample, channel)
 *   scale/shift planes coef[n][2][C] with  y = x*scale + shift == GroupNorm(x)*gamma + beta.
 *   util.py:214-216 (eps 1e-5), attention.py:76-77 & model.py:38-39 (eps 1e-6).
 *   `partial` is a caller-provided scratch of n*chunks*C*3 floats, chunks = ldmk_gn_chunks(hw).
 * ldmk_ln_stats: LayerNorm row statistics (mean, rstd), attention.py:203-205 (eps 1e-5).
 */
int ldmk_gn_chunks(int hw);
/* the two halves of ldmk_gn_coef, for callers that keep the per-tensor partial records
 * ([n][chunks][c][3] = shift, sum(x-shift), sum((x-shift)^2) per 32-pixel chunk) -- ldmk_igemm can emit the
 * same records from its epilogue (`stats_out`), which removes the statistics pass over the activation. */
int ldmk_gn_partial(const float* x, int c, int n, int hw, float* partial, void* stream);
int ldmk_gn_finalize(const float* partial0, int c0, const float* partial1, int c1, int n, int hw, int groups,
                     float eps, const float* gamma, const float* beta, float* coef, void* stream);
int ldmk_gn_coef(const float* x0, int c0, const float* x1, int c1, int n, int hw, int groups, float eps,
                 const float* gamma, const float* beta, float* partial, float* coef, void* stream);
int ldmk_ln_stats(const float* x, int rows, int c, float eps, float* stats, void* stream);
/* the same pass with a guard for LDMK_TF_LAYERNORM_FOLDED consumers: rows with |mean| * rstd > guard (and NaN statistics) set
 * *flag = 1 (device int, never cleared by the library).  The folded form subtracts mean * colsum(W') from x W' in fp32 and
 * loses accuracy ~ |mean| / std on such rows; the caller reads the flag on the host once per program and rebuilds it with
 * the unfolded prologue (LDMK_TF_LAYERNORM).  flag == NULL: plain ldmk_ln_stats. */
int ldmk_ln_stats_guard(const float* x, int rows, int c, float eps, float* stats, float guard, int* flag, void* stream);
/* the same statistics, and the rows themselves as the three bf16 images [3][rows][ld_split] of their exact split
 * (LDMK_COMPUTE_BF16X3's a_split operand): the pass reads every element anyway.  C % 4 == 0, C <= 1024, ld_split >= C, % 8 == 0. */
int ldmk_ln_stats_split(const float* x, int rows, int C, float eps, float* stats, void* split, int ld_split, void* stream);
/* ldmk_gn_apply: y[n][hw][c0+c1] = act(x*scale + shift) with the planes of ldmk_gn_coef, reading the
 *   (virtual) channel concat of x0|x1 and writing one contiguous NHWC tensor; silu != 0 applies SiLU
 *   (openaimodel.py:201-203, model.py:118-131).  One HBM-bound pass: each element is normalised once
 *   instead of 9 x (N / tile) times inside the following 3x3 convolution's operand staging. */
int ldmk_gn_apply(const float* x0, int c0, const float* x1, int c1, const float* coef, float* y, int n, int hw,
                  int silu, void* stream);
/* ldmk_resample2: the parameter-free resampling of a ResBlock(up=True / down=True) (resblock_updown, openaimodel.py:207-216,
 *   256-261), NHWC fp32, out of place.  (h, w) is the SMALLER of the two grids.  up != 0: y[n][2h][2w][c] = x[n][h][w][c]
 *   (Upsample(use_conv=False): F.interpolate(scale_factor=2, mode="nearest"), :110-118); up == 0: y[n][h][w][c] = mean of the
 *   2x2 block of x[n][2h][2w][c] (Downsample(use_conv=False): avg_pool2d(2, 2), :143-160).  C % 4 == 0. */
int ldmk_resample2(const float* x, float* y, int n, int h, int w, int c, int up, void* stream);
/* use_scale_shift_norm (openaimodel.py:267-271: h = out_norm(h) * (1 + scale) + shift, (scale, shift) = the halves of the
 * ResBlock's emb_layers output): folds a per-sample [n][ld] vector (scale at column 0, shift at column c) into the coefficient
 * planes coef[n][2][c] of ldmk_gn_finalize, in place. */
int ldmk_gn_coef_film(float* coef, const float* emb, int ld, int n, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * ldmk_post: everything that sits BETWEEN two GEMMs of a ResBlock / transformer block, as one launch -- the small-batch
 * (latency-bound) route of the UNet: batch 1 is the reference's shipped talking-face mode
 * (talking_face/progressive_sampling_difftalk.py:282-317, batch size 1 at :350), where a step is a chain of dependent
 * launches and every separate reduce / statistics / apply pass costs a launch ramp plus a memory round trip.
 *     v      = alpha * (slab_0 + slab_1 + ... in this order) + bias + batch_vec[sample] + residual
 *              -- the split-K reduce and the GEMM epilogue: openaimodel.py:264-275 (timestep vector, residual),
 *                 attention.py:211-215,261 (residual adds);  nslab = 1 with the other operands NULL reads a plain tensor
 *     raw    = v, or with geglu != 0 value * gelu(gate) of the packed (value | gate) 32-column pairs (attention.py:37-50)
 *     normed = LDMK_POST_GROUPNORM: GroupNorm(groups, eps)[+SiLU] over the channel concat (raw | x1), statistics per
 *              (sample, group) in two passes (openaimodel.py:201-203,225-227,683-684; attention.py:254; the concat of a
 *              skip connection is two base pointers, :736, and groups may straddle the seam);
 *              LDMK_POST_LAYERNORM: LayerNorm(raw) * gamma + beta per row (attention.py:203-205, eps 1e-5)
 * raw_out and norm_out are both optional (at least one).  Fixed summation orders: bitwise reproducible. */
enum { LDMK_POST_NONE = 0, LDMK_POST_GROUPNORM = 1, LDMK_POST_LAYERNORM = 2 };
typedef struct ldmk_post_args {
  const float* src;          /* [nslab][M][N] raw GEMM accumulators (ldmk_igemm with raw_slabs), or a plain [M][N] tensor */
  int nslab;                 /* >= 1 */
  long long slab_stride;     /* floats between slabs (>= M*N) */
  int M, N;                  /* rows (n * H * W) and columns of src; N % 4 == 0 */
  int rows_per_sample;       /* H * W */
  float alpha;               /* scale of the slab sum (1.0) */
  const float* bias;         /* [N] or NULL */
  const float* batch_vec;    /* [n][batch_vec_ld] per-sample vector or NULL */
  int batch_vec_ld;
  const float* residual;     /* [M][ldr] or NULL */
  int ldr;
  int geglu;                 /* != 0: src holds packed (value | gate) pairs, raw_out is [M][N/2]; no norm / residual / batch_vec */
  float* raw_out;            /* [M][ld_raw] or NULL */
  int ld_raw;
  int norm;                  /* LDMK_POST_* */
  const float* x1;           /* GroupNorm: second tensor of the channel concat, [M][c1] contiguous, or NULL */
  int c1;
  int groups;                /* GroupNorm: 32 in every reference layer */
  float eps;
  const float* gamma;        /* [N + c1] */
  const float* beta;         /* [N + c1] */
  int silu;                  /* GroupNorm: apply SiLU after the affine */
  float* norm_out;           /* [M][ld_norm], ld_norm >= N + c1 */
  int ld_norm;
  int gn_cache_floats;       /* (set by the library) */
  float* gn_scratch;         /* GroupNorm with rows_per_sample >= LDMK_POST_GN_TILED_ROWS: (mean, M2) pairs per (row tile, group), */
  long long gn_scratch_elems;/*   ldmk_post_scratch_elems(args) floats; the call is then two launches (row-tiled statistics, apply) */
} ldmk_post_args;
#define LDMK_POST_GN_TILED_ROWS 512
int ldmk_post(const ldmk_post_args* args, void* stream);
/* floats of gn_scratch ldmk_post(args) needs (0 for the single-launch forms); negative for invalid arguments */
long long ldmk_post_scratch_elems(const ldmk_post_args* args);

/* ------------------------------------------------------------------------------------------
 * Attention.
 * ldmk_attn_self: flash-style softmax(Q K^T * scale) V for d_head = 32 on the f32 matrix cores;
 *   qkv is the fused projection [n*tokens][3*C] (q | k | v), out [n*tokens][C].
 *   attention.py:178-192 (CrossAttention with context=None).
 * ldmk_attn_cross: same math for a short context (L <= 128 keys), K/V given as [n*L][C] rows;
 *   attention.py:170-193 with context != None.
 * ldmk_softmax_rows: in-place row softmax of x*scale; model.py:191-192 (VQGAN AttnBlock).
 */
int ldmk_attn_self(const float* qkv, float* out, int n, int tokens, int heads, float scale, void* stream);
/* ldmk_attn_self_x3: the same product with both matrix products in the fp32-accurate three-way bf16 split arithmetic
 *   (LDMK_COMPUTE_BF16X3 above: Q, K, V and the probabilities are exact sums of three bf16 values, six partial products each,
 *   fp32 accumulation, fp32 softmax): the accuracy class of ldmk_attn_self at the bf16 matrix rate. */
int ldmk_attn_self_x3(const float* qkv, float* out, int n, int tokens, int heads, float scale, void* stream);
/* ldmk_attn_self_x3p: ldmk_attn_self_x3 with K and V split ONCE, by a pre-pass, instead of once per 128-query workgroup: the
 *   pre-pass writes the three bf16 planes of every 64-key tile of every (sample, head) in MFMA-operand order into `kv_scratch`
 *   (ldmk_attn_kv_split_bytes(n, tokens, heads) bytes, caller-owned), the attention kernel moves tiles memory -> LDS with
 *   LDS-DMA loads and does no arithmetic on K / V.  Bitwise the results of ldmk_attn_self_x3. */
long long ldmk_attn_kv_split_bytes(int n, int tokens, int heads);
int ldmk_attn_self_x3p(const float* qkv, void* kv_scratch, float* out, int n, int tokens, int heads, float scale, void* stream);
/* the same, with the result ALSO (out != NULL) or ONLY (out == NULL) in the PS layout of the [n tokens][heads 32] matrix (out_ps,
 * ldmk_ps_bytes(n tokens, heads 32) bytes): the pre-split A operand of attn1.to_out on tile_cfg 23+.  tokens % 32 == 0. */
int ldmk_attn_self_x3p_ps(const float* qkv, void* kv_scratch, float* out, void* out_ps, int n, int tokens, int heads, float scale,
                          void* stream);
/* ldmk_attn_self_h2: the same product, fp32-accurate from THREE fp16 products per term (the F16X2 arithmetic: gfx950 sustains
 *   0.60-0.69 of its nominal 16-bit matrix rate on real operands -- power --, so what is left to cut is the number of matrix
 *   instructions).  An operand scaled by a power of two into fp16's range is x' = hi + lo, hi = fp16(x'), lo = fp16(x' - hi):
 *   2 x 11 significand bits, |x' - hi - lo| <= 2^-23 |x'|, the size of an fp32 rounding error (the three-way bf16 split is
 *   exact); hi hi + hi lo + lo hi accumulate in one fp32 accumulator, lo lo (<= 2^-22 of the product) is dropped.  Error against
 *   float64: 1.0-1.7 x that of an fp32 dot product (tests/test_f16x2_gpu.py).  K, V and the pre-scaled Q are scaled by 2^6, the
 *   probabilities by 2^14; |K|, |V|, |scale log2(e) Q| must stay below 1000: the kernels write 1 to *range_flag (device int,
 *   caller-zeroed, never cleared here) when an element is not, and the caller repeats the product with ldmk_attn_self_x3p.
 *   K / V pre-pass and LDS-DMA tiles as ldmk_attn_self_x3p; kv_scratch: ldmk_attn_kv_split_h2_bytes(n, tokens, heads) bytes. */
long long ldmk_attn_kv_split_h2_bytes(int n, int tokens, int heads);
int ldmk_attn_self_h2(const float* qkv, void* kv_scratch, float* out, int* range_flag, int n, int tokens, int heads, float scale,
                      void* stream);
/* the same, with the result ALSO (out != NULL) or ONLY (out == NULL) in the F16X2 form of the PS layout (ldmk_ps_bytes_h2(n tokens,
 * heads 32) bytes): the pre-split A operand of attn1.to_out.  tokens % 32 == 0. */
int ldmk_attn_self_h2_ps(const float* qkv, void* kv_scratch, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                         float scale, void* stream);
/* the same on K / V tiles that already exist (written by the QKV projection's epilogue, ldmk_igemm_args.attn_kv_out): no pre-pass;
 * only the q third of `qkv` is read. */
int ldmk_attn_self_h2_tiles(const float* qkv, const void* kv_tiles, float* out, void* out_ps, int* range_flag, int n, int tokens, int heads,
                            float scale, void* stream);
/* ldmk_attn_self_small: the same product for SMALL problems (batch 1-2: the reference's talking-face mode runs batch 1,
 *   talking_face/progressive_sampling_difftalk.py:350).  One workgroup per 32-query tile of a (sample, head), the keys split
 *   over its 4 / 8 waves and streamed from global memory without LDS staging, partial (max, sum, O) merged in wave order
 *   (bitwise reproducible).  `qkv` may be the nslab raw split-K slabs [nslab][n*tokens][3*C] of the fused QKV projection
 *   (ldmk_igemm with raw_slabs): operand loads sum them in slab order.  nslab = 1: a plain qkv tensor. */
int ldmk_attn_self_small(const float* qkv, int nslab, long long slab_stride, float* out, int n, int tokens, int heads,
                         float scale, void* stream);
int ldmk_attn_cross(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo,
                    int n, int tokens, int ctx_len, int heads, float scale, void* stream);
/* ... and for heads that are not 32 wide (reference kwargs num_heads / num_head_channels, openaimodel.py:443-469,542-549):
 * d_head in {32, 40, 64, 80}. */
int ldmk_attn_cross_d(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo, int n, int tokens,
                      int ctx_len, int heads, int d_head, float scale, void* stream);
/* Self attention for those head widths runs as batched GEMMs (ldmk_igemm, q k^T and p v per (sample, head)) + ldmk_softmax_rows on
 * head-major copies: ldmk_heads_gather copies the heads side by side in token rows src[n tokens][ld] from column col0 to
 * dst[n heads][tokens][dp], each head's d columns zero-padded to dp (a multiple of 32: exact for q k^T); ldmk_heads_scatter is the
 * inverse (first d columns).  The flash kernels (ldmk_attn_self*) stay d_head = 32. */
int ldmk_heads_gather(const float* src, int ld, int col0, float* dst, int n, int tokens, int heads, int d, int dp, void* stream);
int ldmk_heads_scatter(const float* src, float* dst, int ld, int n, int tokens, int heads, int d, int dp, void* stream);
int ldmk_softmax_rows(float* x, long long rows, int cols, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small dense layers on per-sample vectors (weight-bandwidth bound):
 *   out[b][N] = (silu_in ? silu(x[b]) : x[b]) [K] * W[K][N] + bias
 * time_embed MLP (openaimodel.py:506-511,723-724), all ResBlock emb_layers batched into one call
 * (:218-224,264), cross-attention to_k/to_v/to_out on a 1-token context (attention.py:175-176).
 */
int ldmk_dense_small(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                     int rows, int K, int N, int silu_in, void* stream);
/* silu_in: bit 0 = apply SiLU to x first; bit 1 (value 2) = the 16-byte-load form for rows <= 4 (N, ldo multiples of 4; w, bias,
 * out 16-byte aligned): half the time per call at batch 1.  The two forms sum K in different orders, so a caller that needs
 * results independent of how a job is sharded requests the form from the JOB's batch, never from `rows`. */
/* timestep_embedding, util.py:151-171: t int64 [n] -> emb [n][dim] = [cos(t*f) | sin(t*f)];
 * freqs[dim/2] = exp(-ln(max_period) * i / (dim/2)) is a per-model constant computed once on the host */
int ldmk_timestep_embedding(const long long* t, const float* freqs, float* emb, int n, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * Narrow-channel convolutions at the NCHW latent boundary.
 * ldmk_conv3x3_in : 3x3 pad 1 on an NCHW input that is the channel concat of up to two tensors
 *   (x and TF's 'motion_&_id' c_concat, ddpm2cond.py:1309) with <= 16 channels in total ->
 *   NHWC output.  openaimodel.py:519 (input_blocks.0), model.py:497-501 (Decoder.conv_in),
 *   model.py:382-386 (Encoder.conv_in).  w: [9][cin][cout].
 * ldmk_conv3x3_out: GroupNorm+SiLU prologue (coef planes) then 3x3 pad 1 to <= 4 channels,
 *   NHWC input -> NCHW output. openaimodel.py:683-685 (out), model.py:527-531,563-565 (conv_out).
 *   w: [9][cin][cout].
 */
int ldmk_conv3x3_in(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                    float* out, int n, int h, int w_, int cout, void* stream);
int ldmk_conv3x3_out(const float* x, const float* coef, const float* w, const float* bias, float* out,
                     int n, int h, int w_, int cin, int cout, void* stream);
/* ldmk_conv3x3_out_small: the same layer tiled for SMALL images (batch 1-2): 4x4-pixel workgroups, 16 lanes per pixel
 *   splitting the channel axis -- 64 workgroups for one 32x32 image where the 16x16-pixel tiles above give 4.  Same
 *   arguments; a different (fixed) summation order, so a caller that needs results independent of the batch split picks
 *   one of the two from the JOB's batch (engine.NetBuilder does). */
int ldmk_conv3x3_out_small(const float* x, const float* coef, const float* w, const float* bias, float* out,
                           int n, int h, int w_, int cin, int cout, void* stream);
/* 1x1 conv between narrow NCHW tensors (quant_conv / post_quant_conv, autoencoder.py:44-45) */
int ldmk_conv1x1_nchw(const float* x, const float* w, const float* bias, float* out, int n, int hw, int cin,
                      int cout, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sampler updates.  Coefficient tables live in device memory and are indexed by a device-side
 * step counter so that one captured hipGraph replays for every step.
 * ldmk_ddim_step: ddim.py:170-203 -- optional CFG combine (eps has 2n items: [uncond | cond]),
 *   pred_x0, dir_xt, x_prev.  table: [S][4] = (a_t, a_prev, sigma_t, sqrt_one_minus_at) float32;
 *   step_idx: device int32 holding the CURRENT index; advance = +1 walks it down (sampling, S-1 .. 0),
 *   advance = -1 walks it up (DDIM inversion, compute_latents.py:364-406, which is the same update with
 *   the table rows (a_prev, a_t, 0, sqrt(1-a_prev))), clamped to [0, n_steps-1]; ts (int64 [n_ts], the
 *   UNet's timestep input) is rewritten with timesteps[new index].  noise may be NULL (eta == 0).
 * ldmk_ddpm_step: ddpm.py:215-228,1049-1109 -- ancestral update with per-sample t (int64).
 *   tables: [T][4] = (sqrt_recip_ac, sqrt_recipm1_ac, post_mean_coef1, post_mean_coef2) and
 *   logvar[T] (posterior_log_variance_clipped).
 */
int ldmk_ddim_step(const float* x, const float* eps, const float* noise, const float* table, int* step_idx,
                   float cfg_scale, int cfg, float* x_prev, float* pred_x0, long long per_sample, int n,
                   const long long* timesteps, long long* ts, int n_ts, int advance, int n_steps, void* stream);
/* the counter/timestep advance of ldmk_ddim_step on its own (used by the ancestral loop): index -= advance,
 * clamped to [0, n_steps-1]; ts[0..n_ts) = timesteps[index] */
int ldmk_advance_timestep(int* step_idx, const long long* timesteps, long long* ts, int n_ts, int advance, int n_steps,
                          void* stream);
int ldmk_ddpm_step(const float* x, const float* eps, const float* noise, const float* tables, const float* logvar,
                   const long long* t, float* x_prev, long long per_sample, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * VQ nearest-codebook lookup, taming/modules/vqvae/quantize.py:276-285:
 *   idx = argmin_j ( |z|^2 + |e_j|^2 - 2 z.e_j ), z_q = e[idx]; z, z_q NCHW [n][dim][hw].
 */
int ldmk_vq_nearest(const float* z, const float* codebook, float* zq, int* idx, int n, int hw, int dim, int n_embed,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Talking-face per-frame conditioning front-end.
 * ldmk_audio_attention: Conv1DTemporalAttention.forward, talking_face/ldm/modules/encoders/modules.py:103-113:
 *   x [n][T][dim] audio window -> out [n][dim]; conv_w[l] is layer l's Conv1d weight packed [3][cin][cout]
 *   (cin/cout = dim,192,64,16,4,1), conv_b[l] its bias; conv_w / conv_b are DEVICE arrays of 5 device pointers;
 *   lin_w [T][T], lin_b [T] = attentionNet.0.  T <= 32.
 * ldmk_mask_rows: img[n][c][y >= y0[n]][:] = value, the lower-face mask of MEADBase3 (taming/data/custom.py:375-389).
 */
int ldmk_audio_attention(const float* x, int n, int T, int dim, const float* const* conv_w, const float* const* conv_b,
                         const float* lin_w, const float* lin_b, float* out, void* stream);
int ldmk_mask_rows(float* img, const int* y0, int n, int c, int h, int w, float value, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout helpers: weight repacking (once, after load_state_dict) and boundary transposes.
 * ldmk_permute3: dst[i2? ...] generic 3-D permutation of a contiguous [d0][d1][d2] tensor;
 *   perm gives, for each destination axis, the source axis (e.g. conv OIHW viewed as [O][I][9]
 *   -> [9][I][O] is perm = {2,1,0}).
 * ldmk_postprocess_frames: sample_affectnet.py:127,132: clamp((x+1)/2,0,1), NCHW -> NHWC.
 */
int ldmk_permute3(const float* src, float* dst, int d0, int d1, int d2, int p0, int p1, int p2, void* stream);
/* ldmk_pack_conv3x3: torch conv weight OIHW [cout][cin][3][3] -> the K order ldmk_igemm's LDMK_A_CONV3X3 mode walks:
 *   [cin/32][9][32][cout] (32-channel chunk major, tap minor, so that nine consecutive K slices re-read the same
 *   pixels and hit L1/L2).  cin % 32 == 0.  (The narrow boundary convs ldmk_conv3x3_in/out take [9][cin][cout].) */
int ldmk_pack_conv3x3(const float* src, float* dst, int cout, int cin, void* stream);
int ldmk_postprocess_frames(const float* x, float* out, int n, int c, int hw, void* stream);
/* out[m][c] += vec[m / rows_per_sample][c]  (cross-attention with a 1-token context collapses to
 * a per-sample vector; attention.py:170-193 with L_ctx == 1) */
int ldmk_add_rowvec(float* x, const float* vec, int vec_ld, long long rows, int c, int rows_per_sample, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (SURVEY §8f "next" row N1: LatentDiffusion.p_losses ddpm.py:1014-1047, backward of the UNet,
 * AdamW ddpm.py:1363-1385, EMA ema.py:25-44).  Data gradients of Conv2d/Linear reuse ldmk_igemm (b_trans for
 * Linear; ldmk_pack_dgrad3x3 weights for 3x3 convolutions; upsample=2 for stride-2 ones); weight gradients are
 * ldmk_wgrad; the rest are the HBM-bound kernels below.  Every reduction has a fixed order (reproducible grads).
 */
typedef struct ldmk_wgrad_args {
  int R;                     /* reduction rows: n*out_h*out_w (conv) or token rows                          */
  int Kw, N;                 /* dW is [Kw][N]; Kw = 9*c in the packed [c/32][9][32] row order for 3x3 convs   */
  const float* a;            /* what the forward GEMM consumed: NHWC rows of c channels (after norm/activation) */
  int c, lda;                /* channels (conv: multiple of 32); rows mode: row stride of a (>= Kw)          */
  int a_mode;                /* LDMK_A_ROWS | LDMK_A_CONV3X3                                                 */
  int in_h, in_w, out_h, out_w, stride, pad_lo, upsample;   /* conv geometry, as in ldmk_igemm_args          */
  const float* dy;           /* [R][ldy] gradient of the forward output                                     */
  int ldy;
  float* dw;                 /* [Kw][ldw]                                                                    */
  int ldw;
  int accumulate;            /* dw += alpha * product                                                        */
  float alpha;               /* 0 -> 1                                                                       */
  int batch;                 /* >1: batched over blockIdx.z (per-head attention gradients)                   */
  long long a_bstride, dy_bstride, dw_bstride;
  int splitr;                /* 0 = choose (ldmk_wgrad_plan); rows are split over that many workgroups       */
  float* ws;                 /* scratch for the partial slabs: batch*splitr*(Kw + (dbias?1:0))*N floats      */
  long long ws_elems;
  float* dbias;              /* optional [N]: column sums of dy (the bias gradient) from the same pass       */
  int compute;               /* LDMK_COMPUTE_*: BF16 rounds both operands to bf16 while staging (fp32 accumulate; the
                                bias column sums stay fp32)                                                    */
} ldmk_wgrad_args;
int ldmk_wgrad(const ldmk_wgrad_args* args, void* stream);
int ldmk_wgrad_plan(const ldmk_wgrad_args* args, int* splitr);
/* scratch floats ldmk_wgrad(args) needs in args->ws for args->splitr (0: the split ldmk_wgrad_plan would choose):
 * batch * splitr * (Kw + (dbias ? 1 : 0)) * N when splitr > 1, else 0; too little with a pinned splitr -> LDMK_ENOMEM */
long long ldmk_wgrad_workspace_elems(const ldmk_wgrad_args* args);
/* packed forward 3x3 weights [cin/32][9][32][cout] -> data-gradient weights [cout/32][9][32][cin], taps mirrored */
int ldmk_pack_dgrad3x3(const float* w_fwd, float* w_dgrad, int cin, int cout, void* stream);

/* GroupNorm(+SiLU) backward (util.py:214-216 + nn.SiLU).  mr = per-(sample, group) (mean, rstd) from the forward's
 * partial records; scratch = ldmk_gn_bwd_scratch_elems floats; dx0/dx1 follow the two sources of the channel concat. */
int ldmk_gn_group_stats(const float* partial0, int c0, const float* partial1, int c1, int n, int hw, int groups,
                        float eps, float* mr, void* stream);
int ldmk_gn_bwd_chunks(int hw);
long long ldmk_gn_bwd_scratch_elems(int n, int hw, int c, int groups);
int ldmk_gn_bwd(const float* x0, int c0, const float* x1, int c1, const float* dy, const float* coef, const float* mr,
                const float* gamma, int n, int hw, int groups, int silu, float* dx0, int acc0, float* dx1, int acc1,
                float* dgamma, float* dbeta, int acc_params, float* scratch, void* stream);
/* LayerNorm materialised forward / backward (attention.py:203-205); scratch = ldmk_ln_bwd_blocks(rows)*c*2 floats */
int ldmk_ln_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int rows, int c,
                  void* stream);
int ldmk_ln_bwd_blocks(int rows);
int ldmk_ln_bwd(const float* dy, const float* x, const float* stats, const float* gamma, float* dx, int acc_dx, int rows,
                int c, float* dgamma, float* dbeta, int acc_params, float* scratch, void* stream);
/* GEGLU unfused (attention.py:37-45): pre = [value | gate] of 2*inner columns */
int ldmk_geglu_fwd(const float* pre, float* f, long long rows, int inner, void* stream);
int ldmk_geglu_bwd(const float* pre, const float* df, float* dpre, long long rows, int inner, void* stream);
/* ds = p * (dp - sum_j dp_j p_j) * scale, in place over dp (softmax backward of the materialised attention) */
int ldmk_softmax_bwd_rows(const float* p, float* dp, long long rows, int cols, float scale, void* stream);
/* out[g][n] (+)= sum over the rows of group g of x[r][n]: bias gradients (groups = 1) and per-sample timestep-
 * embedding gradients (groups = batch); scratch = groups * ldmk_colsum_splits(rows_per_group) * n floats */
int ldmk_colsum_splits(int rows_per_group);
int ldmk_colsum(const float* x, int ldx, int rows_per_group, int groups, int n, float* out, int ldo, int accumulate,
                float* scratch, void* stream);
/* nearest-x2 upsample backward: dx[n][h][w][c] (+)= sum of the 2x2 block of du[n][2h][2w][c] */
int ldmk_sumpool2(const float* du, float* dx, int n, int h, int w, int c, int accumulate, void* stream);
int ldmk_silu(const float* x, float* y, long long n, void* stream);
int ldmk_silu_bwd(const float* x, const float* dy, float* dx, long long n, void* stream);
int ldmk_axpy(float* y, const float* x, float a, long long n, void* stream);
/* q_sample (ddpm.py:1009-1012), per = elements per sample; t int64 [n] */
int ldmk_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac, const float* sqrt_1mac,
                  float* xt, int n, int per, void* stream);
/* loss = mean((pred-target)^2), dpred = 2*(pred-target)/n (ddpm.py:324-334 'l2', :1034); scratch = 256 doubles */
int ldmk_mse_grad(const float* pred, const float* target, float* dpred, long long n, long long denom, float* loss,
                  double* scratch, void* stream);   /* denom: divisor of the mean (0 -> n); channel-padded tensors pass
                                                       the real element count */
/* Flash-style backward of ldmk_attn_self.  ldmk_attn_self_lse is the forward that also writes the per-row log-sum-exp
 * of the scaled scores, lse[n][heads][tokens]; ldmk_attn_self_bwd recomputes the probabilities tile by tile from it
 * (no [tokens][tokens] matrix in memory) and writes dqkv [n*tokens][3*C]; dsum = scratch of n*heads*tokens floats. */
int ldmk_attn_self_lse(const float* qkv, float* out, float* lse, int n, int tokens, int heads, float scale, void* stream);
int ldmk_attn_self_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* dsum,
                       int n, int tokens, int heads, float scale, void* stream);
/* The same forward / backward pair with bf16 matrix-core products (v_mfma_f32_32x32x16_bf16, operands rounded to nearest even,
 * fp32 accumulation; softmax, log-sum-exp, D and every stored tensor fp32): the attention of the bf16 training step
 * (BASELINE configs[4]) -- the split torch.autocast(bfloat16) makes for attention.py:178-192.  Same arguments. */
int ldmk_attn_self_lse_bf16(const float* qkv, float* out, float* lse, int n, int tokens, int heads, float scale, void* stream);
int ldmk_attn_self_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* dsum,
                            int n, int tokens, int heads, float scale, void* stream);
/* Backward of ldmk_attn_cross (context of ctx_len <= 128 tokens): dq [n*tokens][ldq], dk / dv [n*ctx_len][ldkv];
 * scratch = 2 * n*tokens*heads*ctx_len floats (probabilities and score gradients kept between the two passes). */
int ldmk_attn_cross_bwd(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* dout, int ldo,
                        float* dq, float* dk, float* dv, float* scratch, int n, int tokens, int ctx_len, int heads,
                        float scale, void* stream);
/* Backward of ldmk_audio_attention w.r.t. its parameters (the audio features are frozen wav2vec2 outputs): per-sample
 * gradients [n][ldmk_audio_attention_grad_elems(T, dim)] = conv weights (packed [3][cin][cout]) x5, conv biases x5,
 * Linear(T,T) weight, bias; the caller sums over the batch (ldmk_colsum). */
long long ldmk_audio_attention_grad_elems(int T, int dim);
int ldmk_audio_attention_bwd(const float* x, const float* dout, int n, int T, int dim, const float* const* conv_w,
                             const float* const* conv_b, const float* lin_w, const float* lin_b, float* grads_per_sample,
                             void* stream);
/* d_head = 32 layouts: token-major [n][tokens][parts][heads][32] (the fused qkv / attention output rows) <->
 * head-major [parts][n*heads][tokens][32] (contiguous per-head matrices for the batched backward GEMMs) */
int ldmk_head_permute(const float* src, float* dst, int n, int tokens, int parts, int heads, int to_heads, void* stream);
/* torch.optim.AdamW step over a flat buffer (step >= 1); LitEma update shadow -= omd * (shadow - p) */
int ldmk_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, void* stream);
int ldmk_ema(float* shadow, const float* p, long long n, float one_minus_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDMK_H */
