"""Every environment switch of the package, the library and bench.py, in ONE table.

The defaults are the shipped configuration: none of these has to be set to run the reference's workloads, and results do not depend
on the A/B switches beyond what each line says ("same bits" = bitwise identical output either way).  The package reads its switches
through `get()` (a registered name or KeyError); `libldmk.so` reads the "library" ones itself with getenv, once, at the first
launch that consults them; tests/test_host_logic.py checks that every `LDMK_*` read in the sources is in this table with the
default the call site uses, and that the table names nothing that is no longer read.  `python -m dsml_thesis_amd.switches` prints
the table (docs/SWITCHES.md is that output).

scope: "arithmetic" changes which products are formed (all forms meet the stated fp32 tolerance; the bits differ);
       "route"      picks between kernels / operand layouts for the same arithmetic (A/B measurements; same bits unless noted);
       "plan"       tile-plan lookup; "runtime" process set-up; "build" the hipcc build; "library" read inside libldmk.so;
       "bench"      bench.py only.
"""
import os

# name: (default as the reading site passes it -- None = unset, scope, what it does)
SWITCHES = {
    # ---- arithmetic
    "LDMK_SPLIT_BF16": ("1", "arithmetic", "0: every GEMM and the attention on the f32 matrix-core form (v_mfma_f32_32x32x2_f32); the "
                        "f32_mfma_form leg of bench.py"),
    "LDMK_F16X2": ("1", "arithmetic", "0: split products from the exact bf16x3 split (six bf16 MFMAs, operands bit-exact) instead of "
                   "F16X2 (three fp16 MFMAs, range-checked per site); the bf16x3_form leg of bench.py"),
    "LDMK_LN_UNFOLDED": (None, "arithmetic", "set: LayerNorm applied in the GEMM prologue instead of folded through the product (what "
                         "the folded-LayerNorm guard switches a model to by itself)"),
    "LDMK_LN_GUARD_RATIO": ("4.0", "arithmetic", "|mean| / std of a token row above which the folded LayerNorm raises its guard"),
    "LDMK_ATTN_H2_MIN_TOKENS": ("512", "arithmetic", "self attention runs in F16X2 from this many tokens per sample, bf16x3 below"),
    "LDMK_H2_CONV_RULE": ("1", "arithmetic", "0: untabled long-K convolutions (K >= 1440) do not take the F16X2 rule"),
    # ---- routes (A/B)
    "LDMK_PS": ("1", "route", "0: no pre-split (PS layout, LDS-DMA) GEMM tiles; same bits"),
    "LDMK_PSC": ("1", "route", "0: no conv-mode pre-split tile for 3x3 convolutions; same bits"),
    "LDMK_PSC_FORCE": (None, "route", "'cfg,splitk': force the conv-mode pre-split tile for every eligible convolution (A/B runs)"),
    "LDMK_POUT_PS": ("0", "route", "1: SpatialTransformer.proj_out on a pre-split tile instead of the row GEMM (measured slower)"),
    "LDMK_ATTN_PS": ("1", "route", "0: attn1.to_out does not read the attention result in the PS layout"),
    "LDMK_ATTN_PRESPLIT": ("1", "route", "0: K / V are split inside the attention kernel instead of once by their producer; same bits"),
    "LDMK_ATTN_PRESPLIT_MIN_TOKENS": ("2048", "route", "tokens per sample from which K / V are pre-split"),
    "LDMK_QKV_TILES": ("1", "route", "0: the fused QKV projection does not write the attention's K / V tiles from its epilogue"),
    "LDMK_NO_WINOGRAD": (None, "route", "set: no Winograd F(2x2,3x3) convolutions and no phase-split upsampling convolutions (direct "
                         "implicit GEMM everywhere; other bits, same tolerance)"),
    "LDMK_WINO_MIN_TILES": ("256", "route", "Winograd from this many 2x2 output tiles at the plan-policy batch"),
    "LDMK_WINO_MIN_CIN": ("320", "route", "Winograd / phase-split upsampling from this many input channels"),
    "LDMK_NO_SMALL_ROUTE": (None, "route", "set: batch 1-2 jobs run the batched launch program instead of the small-batch route"),
    "LDMK_SMALL_ROWS": ("4096", "route", "rows (batch x pixels at the policy batch) up to which the small-batch route is taken"),
    "LDMK_SPLITK_IN_LAUNCH": (None, "route", "set: split-K slabs combined by each tile's last workgroup instead of a reduce launch "
                              "(measured slower)"),
    "LDMK_TRAIN_NO_PACKED_W": (None, "route", "set: the training step's bf16 GEMMs read fp32 weights instead of the packed bf16 copies"),
    "LDMK_ATTN_QT": (None, "route", "1 | 2: query tiles per wave of the staged f32 attention kernel (A/B hook)"),
    # ---- plans
    "LDMK_NO_PLAN_TABLE": (None, "plan", "set: ignore igemm_plans.json (the library's built-in tile heuristic plans every GEMM)"),
    "LDMK_PLAN_MAX_RATIO": ("0", "plan", "> 0: a tuned plan is used only within this ratio of its tuned row count (0: nearest at any "
                            "distance)"),
    "LDMK_PLAN_TABLE": (None, "plan", "a file that replaces the `f32` section of igemm_plans.json (a flat {key: [cfg, splitk]} file as "
                        "the tuning tools write it, or another merged file whose section is taken)"),
    "LDMK_X3_TABLE": (None, "plan", "the same for the `bf16x3` section"),
    "LDMK_H2_TABLE": (None, "plan", "the same for the `f16x2` section"),
    "LDMK_PS_TABLE": (None, "plan", "the same for the `ps_bf16x3` section (pre-split tiles)"),
    "LDMK_PS_H2_TABLE": (None, "plan", "the same for the `ps_f16x2` section (pre-split tiles, F16X2)"),
    # ---- runtime / build
    "LDMK_LIBRARY": (None, "runtime", "path of the libldmk.so to load instead of the in-tree one"),
    "LDMK_FORCE_COLLECTIVE": (None, "runtime", "set: parallel.all_gather_items issues the real all-gather at world size 1 too (RCCL "
                              "rehearsal; bench.py sets it under LDMK_BENCH_FORCE_DIST)"),
    "LDMK_LIB_OUT": (None, "build", "output path of the build (an A/B build next to the shipped library)"),
    "LDMK_HIPCC_FLAGS": ("", "build", "extra hipcc flags, part of the build digest (-DLDMK_PS_PROBES, -DLDMK_IG_STAMPS: probe builds "
                         "for tools/; the shipped library carries neither)"),
    # ---- read by libldmk.so (getenv, once)
    "LDMK_ATTN_QB": (None, "library", "1 | 2: 32-query blocks per wave of the pre-split attention kernels (default by token count); "
                     "same bits"),
    "LDMK_ATTN_XCD": ("1", "library", "0: F16X2 attention workgroups in launch order instead of re-dealt over the XCDs; same bits"),
    "LDMK_ATTN_PIPE": ("2", "library", "F16X2 attention key loop: 2 pipelined + lazy running maximum, 1 pipelined with the exact "
                       "maximum, 0 phase-separated (1 and 0: same bits)"),
    "LDMK_IG_NFAST": ("1", "library", "0: LDS-tiled igemm workgroups in row-major tile order instead of column tiles adjacent; same bits"),
    "LDMK_IG_LEAN": ("1", "library", "0: the general igemm epilogue everywhere instead of the operand-set specialised copies; same bits"),
    "LDMK_PS_NFAST": ("1", "library", "as LDMK_IG_NFAST, pre-split tiles"),
    "LDMK_PS_LEAN": ("1", "library", "as LDMK_IG_LEAN, pre-split tiles"),
    "LDMK_WGRAD_TR": ("1", "library", "0: bf16 weight-gradient GEMM with the strided gather instead of transposed LDS reads (A/B)"),
    # ---- bench.py
    "LDMK_BENCH_BACKEND": ("nccl", "bench", "'gloo': rehearse the N > 1 path of bench.py on a one-GPU box"),
    "LDMK_BENCH_FORCE_DIST": (None, "bench", "set: create the process group and issue the clip leg's collective with a single rank"),
}

# read only by probe builds of the library (-DLDMK_PS_PROBES) and the tools that drive them; the shipped library ignores them
PROBE_ONLY = ("LDMK_PS_DEBUG", "LDMK_PS_STAGGER")


def get(name, default=None):
    """os.environ.get for a REGISTERED switch (KeyError otherwise: add the switch to the table, with its documentation)."""
    if name not in SWITCHES:
        raise KeyError(f"{name} is not a registered switch (dsml_thesis_amd/switches.py)")
    return os.environ.get(name, default)


def table_markdown():
    rows = ["| switch | default | scope | effect |", "|---|---|---|---|"]
    for name, (default, scope, doc) in SWITCHES.items():
        shown = "unset" if default is None else ("`" + (default or "''") + "`")
        rows.append(f"| `{name}` | {shown} | {scope} | {doc} |")
    return "\n".join(rows)


if __name__ == "__main__":
    print(table_markdown())
