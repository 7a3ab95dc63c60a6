"""SURVEY 5's "ASan-enabled host build + determinism check": the HOST-only product code that does real arithmetic (csrc/host_g1.hpp,
csrc/host_fr.hpp, csrc/msm_geometry.hpp, csrc/host_util.hpp) built with g++ -fsanitize=address,undefined and run against the CPU oracle
(tests/cpp/host_sanitize.cpp).  Any sanitizer report, mismatch or difference between two runs fails.  (GPU sanitizers are not available on
this pool; the device code is covered by the parity suite.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_and_ubsan_matches_the_oracle(tmp_path):
    from oracle import oracle
    oracle.build()
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lzkoracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    outs = []
    for _ in range(2):
        p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        out, err = p.stdout.decode(), p.stderr.decode()
        assert p.returncode == 0, (out[-2000:], err[-4000:])
        assert "runtime error" not in err and "AddressSanitizer" not in err and "LeakSanitizer" not in err, err[-4000:]
        assert out.strip().endswith("ok") and "digest " in out, out
        outs.append(out)
    assert outs[0] == outs[1]          # deterministic: same digests over every produced value
