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


def test_msm_batch_geometry_host_logic():
    """zkhip_msm_geometry_info (host only): for the problem sizes of a 2^20 opening, of tiny and of lopsided batches every problem's digit
    windows are 256 bits in all, of two widths one bit apart (no sparse window), between 4 and 16 bits, and the pass fits the sort's
    4096 partitions and the 2048-window table the sort kernels keep in LDS -- up to the 64 problems the entry point takes."""
    import ctypes as C
    from zk_cryptography_amd import _native
    _native.build()
    lib = C.CDLL(_native.LIB_PATH)
    lib.zkhip_msm_geometry_info.restype = C.c_int
    cases = [[1 << (19 - i) for i in range(20)], [1, 1, 1], [5], [1 << 19, 3], [4096] * 40, [1 << 17] * 8, [0, 7, 0], [1 << 19] * 2 + [1 << 10] * 30, [3] * 64, [1 << 14] * 64, [1 << 19] + [100] * 63, [1 << 20] * 64, [1 << 24] * 64]
    for sizes in cases:
        n = len(sizes)
        offs = (C.c_size_t * (n + 1))(*[sum(sizes[:j]) for j in range(n + 1)])
        first = (C.c_uint16 * (n + 1))()
        bits = (C.c_uint8 * 2048)()
        totals = (C.c_uint32 * 8)()
        rc = lib.zkhip_msm_geometry_info(offs, C.c_uint32(n), first, bits, totals)
        assert rc == 0, (sizes, rc)
        assert first[0] == 0 and first[n] == totals[0] <= 2048
        assert totals[3] <= 4096 and totals[2] % 8 == 0
        buckets = 0
        for j in range(n):
            w = [bits[v] for v in range(first[j], first[j + 1])]
            if n == 1:
                assert len(set(w)) == 1 and sum(w) >= 256, (sizes, w)       # one problem: uniform windows
            else:
                assert sum(w) == 256 and max(w) - min(w) <= 1 and 7 <= min(w) and max(w) <= 16, (sizes, j, w)
                assert w == sorted(w, reverse=True)                         # the wider windows first
            buckets += sum(1 << (b - 1) for b in w)
        assert buckets == totals[2], (sizes, buckets, totals[2])
    # wider windows for larger problems
    offs = (C.c_size_t * 3)(0, 1 << 18, (1 << 18) + (1 << 10))
    first = (C.c_uint16 * 3)(); bits = (C.c_uint8 * 2048)(); totals = (C.c_uint32 * 8)()
    assert lib.zkhip_msm_geometry_info(offs, C.c_uint32(2), first, bits, totals) == 0
    assert bits[first[0]] > bits[first[1]]
