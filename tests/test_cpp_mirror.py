"""The C++ host mirror (include/zkhip.hpp) runs the reference's own unit tests (same names, inputs and expected
values) plus oracle comparisons.  CPU: it must compile and link; GPU: it must pass."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")


def _build():
    from oracle import oracle
    from zk_cryptography_amd import _native
    oracle.build()
    _native.build()
    subprocess.check_call(["make", "-C", CPP, "-s", "test_mirror"])
    return os.path.join(CPP, "test_mirror")


def test_cpp_mirror_builds():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_mirror_passes_reference_tests():
    exe = _build()          # `make` is a no-op when the binary is newer than the headers it was built from (a stale one -- older than
                            # oracle/zkoracle.h or include/zkhip.h* -- must never run: the structs it allocates would be the old ones)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout[-4000:])
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert " 0 failed" in res.stdout
