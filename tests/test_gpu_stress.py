"""tools/stress_parity.py -- randomised shapes, every output compared with the oracle bit for bit -- under pytest with a
bounded budget, so that the driver's GPU run exercises it too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_stress_parity_60_seconds():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "60", "20261003"],
                         capture_output=True, text=True, timeout=600)
    tail = res.stdout[-3000:] + res.stderr[-2000:]
    assert res.returncode == 0, tail
    assert "MISMATCH" not in res.stdout, tail
