import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "cerberus_amd", "lib", "libvilo_synth.so"),
            os.path.join(ROOT, "cerberus_amd", "lib", "libvilo_gpu.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__ as g
    g.build()


_ensure_built()


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def cfg():
    from cerberus_amd import synth
    return synth.default_config()


@pytest.fixture(scope="session")
def ocfg(cfg):
    from oracle import oracle_py as O
    return O.config_from(cfg)


@pytest.fixture(scope="session")
def small_window(cfg, ocfg):
    """40-landmark synthetic window with oracle-filled preintegration (CPU tests)."""
    from cerberus_amd import synth
    from oracle import oracle_py as O
    w = synth.make_window(cfg, n_landmarks=40, seed=7)
    O.fill_preint(ocfg, w)
    return w


def rand_unit_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def rand_pose(rng, scale=1.0):
    p = np.zeros(7)
    p[:3] = rng.normal(size=3) * scale
    p[3:] = rand_unit_quat(rng)
    return p
