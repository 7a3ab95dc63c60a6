"""The library's host-side SHA-256 (csrc/host_fr.hpp) -- the hash behind MultiComposedSumcheckProver::prove's table-bytes pass
and the GKR outer transcript -- against hashlib, through both of its block functions (portable rounds and, where the CPU has
them, the SHA extensions)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = [0, 1, 3, 55, 56, 63, 64, 65, 119, 120, 127, 128, 129, 1000, 4096, 4097, 65536 + 61, (1 << 20) + 5]


def test_host_sha256_matches_hashlib(tmp_path):
    exe = str(tmp_path / "host_sha")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_sha.cpp")])
    out = subprocess.check_output([exe] + [str(n) for n in LENS]).decode().split("\n")
    assert out[0].startswith("sha_ext ")
    for n, got in zip(LENS, out[1:]):
        data = bytes(((i * 7 + 3 + (i >> 8)) & 0xFF) for i in range(n))
        assert got == hashlib.sha256(data).hexdigest(), n
