"""The C++ host mirror of the reference's factor classes (cerberus_amd/host/vilo_factors.h): builds everywhere,
runs on the GPU box (-m gpu) and agrees with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "tests", "host_check", "host_api_check")


def _build():
    lib = os.path.join(ROOT, "cerberus_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", EXE,
                           os.path.join(ROOT, "tests", "host_check", "host_api_check.cpp"), "-L", lib, "-lvilo_host", "-lvilo_gpu",
                           "-lvilo_synth", "-Wl,-rpath," + lib])


def test_host_mirror_compiles_and_links():
    _build()
    assert os.path.exists(EXE)
    hdr = open(os.path.join(ROOT, "cerberus_amd", "host", "vilo_factors.h")).read()
    for cls in ("IMULegFactor", "IMUFactor", "ProjectionTwoFrameOneCamFactor", "ProjectionTwoFrameTwoCamFactor",
                "ProjectionOneFrameTwoCamFactor", "MarginalizationFactor", "PoseLocalParameterization", "WindowSolver"):
        assert re.search(r"class %s|struct %s" % (cls, cls), hdr), cls


@pytest.mark.gpu
def test_host_mirror_runs_and_matches_oracle(cfg, ocfg):
    from cerberus_amd import synth
    from oracle import oracle_py as O
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vals = dict()
    for line in out.stdout.splitlines():
        vals[line.split()[0]] = line
    w = synth.make_window(cfg, n_landmarks=30, seed=5)
    O.fill_preint(ocfg, w)
    r_o, J_o = O.eval_imu_leg(ocfg, w.preint[0], [w.pose[0], w.speed_bias[0], w.leg_bias[0], w.pose[1], w.speed_bias[1], w.leg_bias[1]])
    got = [float(x) for x in vals["imu_leg_r0"].split()[1:4]]
    np.testing.assert_allclose(got, [r_o[0], r_o[1], r_o[30]], rtol=1e-7, atol=1e-7 * np.abs(r_o).max())
    before = w.clone_state()
    sm = O.solve_window(ocfg, w, O.default_opts(True, 4))
    O.gauge_fix(before, w)
    m = re.search(r"cost (\S+) -> (\S+)", vals["solve"])
    np.testing.assert_allclose(float(m.group(2)), sm.final_cost, rtol=1e-6)
    p0 = [float(x) for x in vals["pose0"].split()[1:4]]
    np.testing.assert_allclose(p0, w.pose[0, :3], atol=1e-9)
    assert "next_prior n 86" in vals["next_prior"] and "valid 1" in vals["next_prior"]


MD_EXE = os.path.join(ROOT, "tests", "host_check", "multi_device_check")


def _build_md():
    lib = os.path.join(ROOT, "cerberus_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", MD_EXE,
                           os.path.join(ROOT, "tests", "host_check", "multi_device_check.cpp"), "-L", lib, "-lvilo_gpu", "-lvilo_synth", "-Wl,-rpath," + lib])


def test_multi_device_check_compiles_and_links():
    _build_md()
    assert os.path.exists(MD_EXE)


@pytest.mark.gpu
def test_one_context_per_device_one_thread_each():
    """SURVEY 8(e)'s in-process form: one vilo_ctx per visible device, created and used from its own host thread, disjoint windows, all
    threads solving at once — and every window solved again alone on device 0: bitwise the same states and cost. On a one-GPU box the
    threads share device 0 (two contexts, two streams) and the program says that its multi-device leg was skipped."""
    _build_md()
    out = subprocess.run([MD_EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bitwise_equal 1" in out.stdout
    m = re.search(r"devices (\d+) threads (\d+)", out.stdout)
    ndev, nthr = int(m.group(1)), int(m.group(2))
    assert nthr == (min(ndev, 8) if ndev >= 2 else 2)
    assert ("multi_device_leg ran" in out.stdout) == (ndev >= 2)
    assert len(re.findall(r"^win \d+ \d+ cost ", out.stdout, flags=re.M)) == 24 * nthr
    print(out.stdout.splitlines()[0], "|", out.stdout.splitlines()[-1])
