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
    assert lib.ldmk_version() >= 100
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


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dsml_thesis_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.LdmkError, match="no CPU fallback"):
        L.load()
