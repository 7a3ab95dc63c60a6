"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/zkhip.h declares.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from zk_cryptography_amd import _native
    _native.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.zkhip_version() >= 1


def test_no_oracle_or_cpu_fallback_in_product():
    """The product package must not import or link the oracle."""
    pkg = os.path.join(ROOT, "zk-cryptography_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "zkoracle" not in src and "oracle." not in src and "import oracle" not in src, f


def test_missing_gpu_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zk_cryptography_amd as z
    with pytest.raises(Exception):
        z.Multilinear(z.Fr.from_ints([1, 2]))
