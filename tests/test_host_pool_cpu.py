"""The library's host thread pool (csrc/host_util.hpp, ZkHostPool: the per-problem host epilogues of a batched commit run on it):
every task exactly once, run() returns after the last one -- for 0 / 1 / 3 / 7 worker threads, hundreds of thousands of back-to-back
runs with tiny tasks (a late-waking worker of run N meeting the set-up of run N + 1 is what used to deadlock at ~1e5 runs), once as
a plain build and once under ThreadSanitizer.  A hang (timeout), a non-zero exit status or any TSan report fails the test."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_pool.cpp")


def _build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-pthread"] + extra + ["-o", exe, SRC])
    return exe


def _run(exe, workers, runs, timeout, env=None):
    p = subprocess.run([exe, str(workers), str(runs)], timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (p.returncode, out[-2000:], err[-4000:])
    assert out.startswith("ok %d workers %d runs" % (workers, runs)), out
    return err


@pytest.mark.parametrize("workers,runs", [(0, 300000), (1, 300000), (3, 1000000), (7, 300000)])
def test_host_pool_runs_every_task_once(tmp_path, workers, runs):
    _run(_build(tmp_path, "host_pool", ["-O2"]), workers, runs, timeout=600)


@pytest.mark.parametrize("workers", [7])
def test_host_pool_is_clean_under_thread_sanitizer(tmp_path, workers):
    exe = _build(tmp_path, "host_pool_tsan", ["-O1", "-g", "-fsanitize=thread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    err = _run(exe, workers, 300000, timeout=900, env=env)
    assert "ThreadSanitizer" not in err, err[-4000:]
