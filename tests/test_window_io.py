"""Window dump format (include/vilo_window_io.h): Python writer/reader, byte-exact round trip through
the C header, and — on the GPU — replay of a dumped window (incl. a stored result) through the C-ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from cerberus_amd import synth, window_io
from oracle import oracle_py as O


def _window(cfg, ocfg, **kw):
    w = synth.make_window(cfg, **kw)
    O.fill_preint(ocfg, w)
    return w


def test_python_roundtrip(tmp_path, cfg, ocfg):
    w = _window(cfg, ocfg, n_landmarks=23, seed=9)
    after = [a + 1e-3 for a in w.clone_state()]
    p = str(tmp_path / "w.vwin")
    window_io.save(p, cfg, w, after=after, ref_summary=[12, 3.5, 1.25, 0], marginalization_flag=1)
    cfg2, w2, after2, ref2, flag = window_io.load(p)
    assert bytes(cfg2) == bytes(cfg) and flag == 1
    for a, b in zip(w.state_arrays(), w2.state_arrays()):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(after, after2):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ref2, [12, 3.5, 1.25, 0])
    np.testing.assert_array_equal(w.obs, w2.obs); np.testing.assert_array_equal(w.obs_is_stereo, w2.obs_is_stereo)
    np.testing.assert_array_equal(w.lm_obs_offset, w2.lm_obs_offset); np.testing.assert_array_equal(w.preint, w2.preint)
    assert w2.prior.struct.n == w.prior.struct.n and w2.prior.blocks() == w.prior.blocks()
    np.testing.assert_array_equal(w.prior.J0_matrix(), w2.prior.J0_matrix())
    # the oracle sees the same problem
    assert O.window_cost(ocfg, w) == O.window_cost(O.config_from(cfg2), w2)


@pytest.mark.parametrize("with_after,with_prior", [(True, True), (False, False)])
def test_c_header_roundtrip_is_byte_exact(tmp_path, cfg, ocfg, with_after, with_prior):
    exe = str(tmp_path / "window_io_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_check", "window_io_check.cpp")])
    w = _window(cfg, ocfg, n_landmarks=17, seed=4, with_prior=with_prior)
    a, b = str(tmp_path / "a.vwin"), str(tmp_path / "b.vwin")
    window_io.save(a, cfg, w, after=w.clone_state() if with_after else None, ref_summary=[5, 2.0, 1.0, 1] if with_after else None)
    out = subprocess.run([exe, a, b], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert open(a, "rb").read() == open(b, "rb").read()


def test_c_header_reader_survives_mutated_dumps(tmp_path):
    """A dump is untrusted input for whoever replays it: 2000 seeded mutations of the golden dump go through
    vilo_window_read built with the address and undefined-behaviour sanitizers — each comes back as a window or as an error code; the counts
    a header states are bounded by the file's size before anything is allocated for them."""
    exe = str(tmp_path / "window_io_fuzz")
    src = os.path.join(ROOT, "tests", "host_check", "window_io_check.cpp")
    base = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe, src]
    if subprocess.call(base[:5] + ["-fsanitize=address,undefined"] + base[5:], stderr=subprocess.DEVNULL) != 0:
        subprocess.check_call(base)   # (no sanitizer runtime on this host: the plain build still must not crash)
    golden = os.path.join(ROOT, "tests", "golden", "seq_window_rejected_steps.vwin")
    out = subprocess.run([exe, "--fuzz", golden, str(tmp_path / "scratch.vwin"), "2000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr[-3000:]
    assert "refused" in out.stdout


@pytest.mark.gpu
def test_replay_dumped_window_on_gpu(tmp_path, cfg, ocfg):
    """Dump a window together with the oracle's result as the stored 'reference result', reload it and replay it through
    the C-ABI: this is the harness a file dumped from the real Ceres stack goes through."""
    from cerberus_amd import api
    w = _window(cfg, ocfg, n_landmarks=40, seed=21)
    w_ref = _window(cfg, ocfg, n_landmarks=40, seed=21)
    opts = O.default_opts(fixed_iterations=True, max_num_iterations=6)
    sm = O.solve_window(ocfg, w_ref, opts)
    p = str(tmp_path / "w.vwin")
    window_io.save(p, cfg, w, after=w_ref.clone_state(), ref_summary=[sm.iterations, sm.initial_cost, sm.final_cost, sm.termination])
    cfg2, w2, after, ref, _ = window_io.load(p)
    ctx = api.Context(cfg2, 0)
    s = ctx.solve_windows([w2], api.default_solve_opts(True, int(ref[0])))[0]
    ctx.close()
    assert abs(s.final_cost - ref[2]) <= 1e-7 * abs(ref[2])
    for a, b in zip(w2.state_arrays(), after):
        assert np.abs(a - b).max() < 1e-6 * max(1.0, np.abs(b).max())
