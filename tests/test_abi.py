"""CPU: the C-ABI library builds, loads and exports every symbol include/ldmk.h declares
(no compute calls without a GPU), and the product refuses to run without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from dsml_thesis_amd.build import build_lib
    return build_lib(verbose=False)


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ldmk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ldmk_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ldmk.h but not exported"


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 (no C++-isms, no torch types)."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "ldmk.h")
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", hdr])
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)      # comments may mention PyTorch; signatures may not
    assert "torch" not in code.lower() and "tensor" not in code.lower()


def test_binding_matches_header(libpath):
    from dsml_thesis_amd import lib as L
    assert sorted(L.EXPORTED) == header_symbols()
    lib = L.load()
    assert lib.ldmk_version() >= 200
    assert lib.ldmk_gn_chunks(1024) == 32 and lib.ldmk_gn_chunks(33) == 2


def test_argument_validation_without_gpu(libpath):
    """Validation happens on the host before any launch, so it is testable without a GPU."""
    from dsml_thesis_amd import lib as L
    lib = L.load()
    a = L.IgemmArgs()
    assert lib.ldmk_igemm(ctypes.byref(a), None) == -1
    assert b"empty problem" in lib.ldmk_last_error()
    assert lib.ldmk_attn_self(1, 1, 1, 0, 5, 0.1, None) == -1
    assert b"must be positive" in lib.ldmk_last_error()
    assert lib.ldmk_vq_nearest(1, 1, 1, 1, 1, 4, 7, 16, None) == -1
    with pytest.raises(L.LdmkError, match="unsupported"):
        L.call("ldmk_vq_nearest", 1, 1, 1, 1, 1, 4, 7, 16, None)


def test_workspace_queries_and_enomem_without_gpu(libpath):
    """the caller owns scratch.  A host that is not PyTorch sizes it from the query alone; a pinned plan
    with too little scratch is LDMK_ENOMEM (-3), not a launch."""
    from dsml_thesis_amd import lib as L
    lib = L.load()
    a = L.IgemmArgs()
    a.M, a.N, a.K, a.c0, a.rows_per_sample, a.ldb, a.ldc = 1024, 640, 5760, 640, 64, 640, 640
    a.a0 = a.w = a.out = 4096                       # never dereferenced: validation precedes any launch
    a.a_mode, a.in_h, a.in_w, a.out_h, a.out_w, a.stride, a.pad_lo = L.A_CONV3X3, 8, 8, 8, 8, 1, 1
    a.tile_cfg, a.splitk = 4, 6
    assert lib.ldmk_igemm_workspace_elems(ctypes.byref(a)) == 6 * 1024 * 640
    a.splitk = 1
    assert lib.ldmk_igemm_workspace_elems(ctypes.byref(a)) == 0
    a.splitk = 0                                    # planner's wish with unlimited scratch: this shape splits K
    want = lib.ldmk_igemm_workspace_elems(ctypes.byref(a))
    assert want > 0 and want % (1024 * 640) == 0
    a.tile_cfg, a.splitk = 9, 0                     # row-GEMM tiles never split
    assert lib.ldmk_igemm_workspace_elems(ctypes.byref(a)) == 0
    a.tile_cfg, a.splitk, a.splitk_ws, a.splitk_ws_elems = 4, 6, 4096, 6 * 1024 * 640 - 1
    assert lib.ldmk_igemm(ctypes.byref(a), None) == -3
    assert b"needs a workspace" in lib.ldmk_last_error()
    with pytest.raises(L.LdmkError, match="rc=-3"):
        L.call("ldmk_igemm", ctypes.byref(a), None)
    b = L.IgemmArgs()
    assert lib.ldmk_igemm_workspace_elems(ctypes.byref(b)) == -1            # invalid arguments: LDMK_EINVAL
    w = L.WgradArgs()
    w.R, w.Kw, w.N, w.c, w.lda, w.ldy, w.ldw, w.a, w.dy, w.dw = 65536, 160, 160, 160, 160, 160, 160, 4096, 4096, 4096
    w.splitr = 8
    assert lib.ldmk_wgrad_workspace_elems(ctypes.byref(w)) == 8 * 160 * 160
    w.dbias = 4096
    assert lib.ldmk_wgrad_workspace_elems(ctypes.byref(w)) == 8 * 161 * 160
    w.ws, w.ws_elems = 4096, 100
    assert lib.ldmk_wgrad(ctypes.byref(w), None) == -3


def test_init_reports_a_missing_device(libpath):
    import torch
    from dsml_thesis_amd import lib as L
    lib = L.load()
    if torch.cuda.is_available():
        assert lib.ldmk_init(0) == 0
        assert lib.ldmk_init(99) == -1 and b"outside" in lib.ldmk_last_error()
    else:
        assert lib.ldmk_init(0) == -2 and b"no HIP device" in lib.ldmk_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dsml_thesis_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.LdmkError, match="no CPU fallback"):
        L.load()
