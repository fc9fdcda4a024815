"""MFMA-pipe utilisation per kernel family from a rocprofv3 --pmc pass:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
  python tools/pmc_mfma.py DIR
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the 1024 SIMDs, with an MFMA in flight; rocprofv3 reports
GRBM_GUI_ACTIVE as the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so
utilisation = busy / (GRBM_GUI_ACTIVE / 8 x 1024), summed over the dispatches of a family.  GRBM_GUI_ACTIVE includes the
dispatch ramp-up/drain, which weighs on launches shorter than ~0.3 ms: the figure is a lower bound of the in-kernel one."""
import collections
import csv
import glob
import os
import sys

# (first match wins: the bf16x3 instantiations of igemm_kernel carry BF = 3 as their 9th template argument: '..., false, false, 3, true, false>')
FAM = (("igemm_psc_kernel", "igemm f16x2, conv-mode pre-split tile: taps gathered by LDS-DMA (igemm_psc)"),
       (", 2, false>(ldmk_igemm_args", "igemm f16x2, pre-split operands + LDS-DMA (igemm_ps, two fp16 planes)"),
       (", 2, true>(ldmk_igemm_args", "igemm f16x2, pre-split operands + LDS-DMA (igemm_ps, two fp16 planes)"),
       ("igemm_ps_kernel", "igemm bf16x3, pre-split operands + LDS-DMA (igemm_ps)"), ("igemm_pw_kernel", "igemm bf16x3, pre-split + warp-specialised (igemm_pw)"),
       ("false, 3, true", "igemm bf16x3 (six bf16 MFMAs per product)"), ("false, 4, true", "igemm f16x2 (three fp16 MFMAs per product)"), ("igemm_ws_kernel", "igemm bf16x3, warp-specialised tiles"),
       ("igemm_kernel", "igemm f32 MFMA (LDS-tiled, all tiles)"),
       ("rgemm_kernel", "row GEMM (all wave tiles)"), ("sgemm_kernel", "slab GEMM"), ("attn_h2_fwd", "attention forward, f16x2 (three fp16 MFMAs per product), K / V pre-split + LDS-DMA"), ("attn_x3p", "attention forward, bf16x3, K / V pre-split + LDS-DMA"), ("attn_x3", "attention forward, bf16x3"),
       ("attn_self", "attention forward, f32 MFMA"), ("wgrad_kernel", "wgrad"), ("attn_bwd", "attention backward"))


def main(d):
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(f, r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for (_, _, name), c in rows.items():
        fam = next((lab for key, lab in FAM if key in name), None)
        if fam is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        a = acc[fam]
        a[0] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
        a[1] += c["GRBM_GUI_ACTIVE"]
        a[2] += 1
    print(f"{'kernel family':44s} {'dispatches':>10s} {'MFMA busy cycles':>18s} {'GPU active cycles':>18s} {'MFMA pipe busy':>15s}")
    for fam, (busy, act, n) in acc.items():
        print(f"{fam:44s} {n:10d} {busy:18.4g} {act:18.4g} {100.0 * busy / (act / 8.0 * 1024.0):14.1f}%")


if __name__ == "__main__":
    main(sys.argv[1])
