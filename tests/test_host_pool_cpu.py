"""The library's host thread pool (csrc/host_util.hpp, ZkHostPool: the per-problem host epilogues of a batched commit run on it):
every task exactly once, run() returns after the last one, for 0 / 1 / 7 worker threads and thousands of runs in a row."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workers", [0, 1, 7])
def test_host_pool_runs_every_task_once(tmp_path, workers):
    exe = str(tmp_path / "host_pool")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_pool.cpp")])
    out = subprocess.check_output([exe, str(workers), "3000"], timeout=120).decode()
    assert out.startswith("ok %d workers" % workers), out
