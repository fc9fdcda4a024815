"""L2 <-> fabric traffic of one UNet step BY KERNEL FAMILY from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, no
tracing), corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 64 B per 128-B request on wide coalesced
reads: doubled; both counters are in KiB; Infinity-Cache hits are included).  Every dispatch of the profiled command between two
`timestep_embedding` launches belongs to one UNet evaluation, so bytes per step = the family's total / the number of evaluations.
   python tools/pmc_traffic.py FETCH_DIR WRITE_DIR KEY OUT.json [ALGORITHMIC_BYTES_PER_STEP]"""
import collections
import csv
import glob
import json
import os
import sys

# (first match wins)
FAMILIES = (
    (("igemm_kernel", "igemm_ws_kernel", "igemm_ps_kernel", "igemm_psc_kernel", "igemm_pw_kernel", "rgemm_kernel", "sgemm_kernel", "igemm_reduce"), "gemm"),
    (("attn_",), "attention"),
    (("wino_", "upconv_", "gn_apply_ps"), "conv_transforms"),          # Winograd input / output, upsampling gather / scatter, the GroupNorm-apply pass that writes the PS operand
    (("gn_", "ln_stats", "post_"), "norm_statistics"),
    (("conv3x3_in", "conv3x3_out", "dense_small", "timestep_embedding", "ddim_step", "ddim_advance"), "boundary_and_update"),
)
SKIP = ("pack_", "fold_layernorm", "permute", "Cijk_", "at::native", "rocclr")      # weight packing at load time, torch's own kernels


def family(name):
    if "ldmk::" not in name or any(s in name for s in SKIP):
        return None
    for keys, fam in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def collect(d, counter):
    tot, cnt, evals = collections.Counter(), collections.Counter(), 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r.get("Kernel_Name", "")
            if "timestep_embedding" in name:
                evals += 1
            fam = family(name)
            if fam is not None:
                tot[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
    return tot, cnt, evals


if __name__ == "__main__":
    fd, wd, key, out = sys.argv[1:5]
    algo = float(sys.argv[5]) if len(sys.argv) > 5 else None
    f_kib, f_cnt, f_ev = collect(fd, "FETCH_SIZE")
    w_kib, w_cnt, w_ev = collect(wd, "WRITE_SIZE")
    assert f_ev > 0 and w_ev > 0, "no UNet evaluation in the trace"
    fams = {}
    for fam in sorted(set(f_kib) | set(w_kib)):
        fetch = 2.0 * f_kib[fam] * 1024 / f_ev          # gfx950 correction x2, KiB -> bytes, per evaluation
        write = w_kib[fam] * 1024 / w_ev
        fams[fam] = dict(fetch_bytes_per_step=fetch, write_bytes_per_step=write, bytes_per_step=fetch + write,
                         dispatches_per_step=round(f_cnt[fam] / f_ev, 2))
    whole = sum(v["bytes_per_step"] for v in fams.values())
    res = dict(families=fams, whole_step_bytes=whole, hbm_bytes_per_step=fams.get("gemm", {}).get("bytes_per_step"),
               evaluations_counted=[f_ev, w_ev], correction="FETCH_SIZE x2 (gfx950), KiB -> bytes; L2 <-> fabric, Infinity-Cache hits included")
    if algo:
        res["algorithmic_bytes_per_step"] = algo
        res["ratio_to_algorithmic"] = whole / algo
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[key] = res
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
