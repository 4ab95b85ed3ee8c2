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


def rank_deficient_window(cfg, ocfg, seed=31, L=15):
    """No prior, no IMU factor on interval (0, 1) (sum_dt > 10 s) and ONE landmark anchored in frame 0 with two observations: the
    marginalised block [pose 0, lambda] (7 dims) is seen by 6 residual rows only, so Amm is rank deficient."""
    import numpy as np
    from cerberus_amd import synth
    from oracle import oracle_py as O
    prm = synth.default_params(n_landmarks=L, seed=seed, with_prior=False)
    w = synth.make_window(cfg, params=prm); O.fill_preint(ocfg, w)
    w.preint[0, 0] = 11.0
    # the first frame-0 landmark keeps two observations, the other frame-0 landmarks are removed: rebuild the observation arrays
    keep = []
    new_off = [0]
    first0 = True
    drop_lm = []
    for l in range(w.L):
        o0, o1 = w.lm_obs_offset[l], w.lm_obs_offset[l + 1]
        if w.lm_start_frame[l] == 0:
            if first0:
                o1 = o0 + 2; first0 = False
            else:
                drop_lm.append(l); continue
        keep.append((l, o0, o1)); new_off.append(new_off[-1] + (o1 - o0))
    idx = np.concatenate([np.arange(o0, o1) for _, o0, o1 in keep])
    lm = [l for l, _, _ in keep]
    w.obs = np.ascontiguousarray(w.obs[idx]); w.obs_is_stereo = np.ascontiguousarray(w.obs_is_stereo[idx])
    w.lm_start_frame = np.ascontiguousarray(w.lm_start_frame[lm]); w.inv_depth = np.ascontiguousarray(w.inv_depth[lm])
    w.truth_inv_depth = np.ascontiguousarray(w.truth_inv_depth[lm])
    w.lm_obs_offset = np.array(new_off, np.int32); w.L = len(lm); w.n_obs = len(idx)
    return w
