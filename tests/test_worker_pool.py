"""The persistent host worker pool (cerberus_amd/csrc/worker_pool.hpp, used by vilo_batch_create and the fleet's bookkeeping): exception
safety, fork safety and a clean process exit."""
import os
import subprocess

from conftest import ROOT


def test_worker_pool_survives_exceptions_fork_and_exit(tmp_path):
    exe = str(tmp_path / "wp_check")
    src = os.path.join(ROOT, "tests", "host_check", "worker_pool_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "worker pool OK" in out.stdout
