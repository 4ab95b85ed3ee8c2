"""Freezes the reference's IMULegIntegrationBase (oracle/_ref/libref.so) on the contact inputs no gait generator produces — the branches
of imu_leg_integration_base.cpp that the trot of cerberus_amd/host/synth.cpp (c in {0, 1}, two feet always down) never reaches:
  * :354-358  all four feet off the ground (sum of the flags < 1e-6): the leg-velocity uncertainties become 10e10, rho's RHO_NC_N
  * :183-194  the flag is `c >= 0.5`: non-binary c (0.3 / 0.5 / 0.7, values one ulp either side of 0.5, uniform noise)
  * :195-229 with :354-358  the same with contact_sensor_type 2: forces so low that every truncated logistic flag is 0
The ten intervals of the golden window (seed 5, 24 landmarks) keep their IMU / joint samples; only the contact columns (31:35) change —
`contact_edges(samples, offsets)` below is imported by the tests, so the inputs are regenerated, not stored.
Run where /root/reference exists:   python tests/golden/make_golden_edges.py"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BELOW, ABOVE = np.nextafter(0.5, 0.0), np.nextafter(0.5, 1.0)


def contact_edges(samples, offsets, seed=31):
    """Contact columns of the ten intervals, one edge case each (k = interval):
    0  five consecutive samples with all four feet in the air in the middle of a trot
    1  the whole interval in the air
    2  c cycling through 0.3 / 0.5 / 0.7, phase-shifted per leg (0.5 counts as contact)
    3  c one ulp below 0.5, exactly 0.5, one ulp above, 0.4999999, per sample and leg
    4  all legs at 0.3 (non-binary AND all in the air) for the first half, 0.7 after
    5  uniform noise in [0, 1]
    6  one foot down at 0.5 exactly, the others at 0.49
    7  in the air except for single samples of contact (the branch toggles from step to step)
    8  c = 1e-7 on all legs: the flags' sum is 0 < 1e-6 although the inputs' is not
    9  negative and > 1 values (a filter's over- and undershoot)"""
    rng = np.random.default_rng(seed)
    s = np.array(samples, copy=True)
    c = s[:, 31:35]

    def rows(k):
        return np.arange(offsets[k], offsets[k + 1])
    r = rows(0); c[r[5:10]] = 0.0
    r = rows(1); c[r] = 0.0
    r = rows(2); c[r] = np.array([0.3, 0.5, 0.7])[(np.arange(r.size)[:, None] + np.arange(4)[None, :]) % 3]
    r = rows(3); c[r] = np.array([BELOW, 0.5, ABOVE, 0.4999999])[(np.arange(r.size)[:, None] + 2 * np.arange(4)[None, :]) % 4]
    r = rows(4); c[r[:r.size // 2]] = 0.3; c[r[r.size // 2:]] = 0.7
    r = rows(5); c[r] = rng.uniform(size=(r.size, 4))
    r = rows(6); c[r] = 0.49; c[r, 2] = 0.5
    r = rows(7); c[r] = 0.0; c[r[::4], 1] = 1.0
    r = rows(8); c[r] = 1e-7
    r = rows(9); c[r] = rng.choice([-0.2, 0.2, 0.8, 1.3], size=(r.size, 4))
    return s


def force_edges(samples, offsets, seed=32):
    """contact_sensor_type 2: foot forces. Odd intervals: every leg unloaded (a few newtons of sensor noise) from the fifth sample on, so
    that the adaptive threshold (:201-212) ends above every reading and all four truncated flags are 0; even intervals: one leg loaded."""
    rng = np.random.default_rng(seed)
    s = np.array(samples, copy=True)
    f = 15.0 + 140.0 * s[:, 31:35] + 4.0 * rng.normal(size=(s.shape[0], 4))
    for k in range(len(offsets) - 1):
        r = np.arange(offsets[k], offsets[k + 1])
        if k % 2:
            f[r[4:]] = 3.0 + 1.0 * rng.normal(size=(r.size - 4, 4))
        else:
            f[r[4:]] = 3.0 + 1.0 * rng.normal(size=(r.size - 4, 4))
            f[r[4:], k % 4] = 150.0 + 4.0 * rng.normal(size=r.size - 4)
    s[:, 31:35] = f
    return s


def main():
    from cerberus_amd import synth
    from oracle import oracle_py as O
    from oracle import ref_py as R
    cfg = O.default_config()
    cfg2 = copy.copy(cfg)
    cfg2.contact_sensor_type = 2
    w = synth.make_window(synth.default_config(), n_landmarks=24, seed=5)
    ce, fe = contact_edges(w.samples, w.sample_offsets), force_edges(w.samples, w.sample_offsets)
    with R.as_oracle():
        flags = np.array([O.preintegrate_imu_leg(cfg, ce[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
        force = np.array([O.preintegrate_imu_leg(cfg2, fe[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
    np.savez_compressed(os.path.join(HERE, "preint_contact_edges.npz"), flags=flags, force=force, flag_seed=np.array(31), force_seed=np.array(32))
    print("wrote preint_contact_edges.npz", flags.shape, "largest covariance entries per interval:", np.abs(flags[:, 994:]).max(axis=1))


if __name__ == "__main__":
    main()
