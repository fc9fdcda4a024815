"""Build libldmk.so (the C-ABI HIP library) in-tree for gfx950.

`python -m dsml_thesis_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a
GPU; the .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys

from . import switches

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = switches.get("LDMK_LIB_OUT") or os.path.join(HERE, "libldmk.so")      # (LDMK_LIB_OUT: an A/B build next to the shipped one)
SOURCES = ["igemm.hip", "rgemm.hip", "norms.hip", "attention.hip", "small.hip", "wgrad.hip", "backward.hip", "attention_bwd.hip", "winograd.hip", "post.hip", "sgemm.hip", "attention_small.hip", "attention_bf16.hip", "igemm_ws.hip", "igemm_ps.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]
# probe builds only (tools/ps_probe.sh, tools/pw_stamps.py: LDMK_HIPCC_FLAGS=-DLDMK_PS_PROBES): extra flags are part of the digest
FLAGS += switches.get("LDMK_HIPCC_FLAGS", "").split()
# The attention kernels run their softmax on the MFMA results: keep the accumulators in VGPRs (MFMA VGPR form) instead of
# AGPRs, otherwise every score / output tile costs a v_accvgpr_read + v_accvgpr_write round trip per register
# (208 such moves per key tile in the forward kernel).  The GEMM kernels only touch their accumulators in the epilogue.
EXTRA_FLAGS = {"rgemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attention_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "attention_small.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "attention_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}   # (no NaNs: see x3_softmax)


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "ldmk.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build_lib(force=False, verbose=True):
    stamp = LIB + ".sha256"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    # per-object stamps: a source is recompiled only when it, a header or its flags changed (igemm.hip alone takes minutes)
    hdr = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            hdr.update(open(os.path.join(CSRC, f), "rb").read())
    hdr.update(open(os.path.join(HERE, "..", "include", "ldmk.h"), "rb").read())
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        flags = FLAGS + EXTRA_FLAGS.get(src, [])
        h = hashlib.sha256(hdr.digest())
        h.update(open(os.path.join(CSRC, src), "rb").read())
        h.update(" ".join(flags).encode())
        ostamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == h.hexdigest():
            continue
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd), ostamp, h.hexdigest()))
    failed = []
    for src, p, ostamp, dg in procs:
        if p.wait() != 0:
            failed.append(src)
        else:
            with open(ostamp, "w") as fh:
                fh.write(dg)
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
