"""The randomised sweeps of tools/ at a size the suites can afford, held to thresholds (fixed seeds: the same inputs every run).
  * oracle against the compiled reference (CPU; skipped where oracle/_ref/libref.so is absent): tools/oracle_vs_reference_sweep.py
  * HIP path against the oracle (-m gpu): tools/parity_sweep.py
Their full-size outputs are profiles/round6_oracle_vs_reference_sweep.txt and profiles/round6_parity_sweep.txt."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _held(worst, rules, value=lambda v: v):
    """every quantity of `worst` falls under exactly one rule (substring, bound) and respects it"""
    for k, v in worst.items():
        hits = [(sub, bound) for sub, bound in rules if sub in k]
        assert len(hits) == 1, (k, hits)
        assert value(v) < hits[0][1], (k, value(v), hits[0][1])


def test_oracle_against_the_compiled_reference_on_random_inputs():
    from oracle import ref_py as R
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    import oracle_vs_reference_sweep as S
    worst, n_int, n_marg = S.sweep(40, 424242)
    assert n_int >= 90 and n_marg >= 10
    _held(worst, [
        ("kinematics", 1e-14),
        ("state (33 scalars)", 1e-14),
        ("jacobian, per entry", 1e-6),                      # (entries of 1e-10 of a row's largest, to 1e-8 of themselves)
        ("covariance, per diagonal", 1e-13),
        ("IntegrationBase: state, jacobian, covariance", 1e-6),
        ("whitened residual, per entry", 1e-11),
        ("whitened Jacobians, per row", 1e-13),
        ("Projection factor", 1e-11),
        ("PoseLocalParameterization::Plus", 1e-15),
        ("MarginalizationFactor::Evaluate: residual", 1e-9),
        ("MarginalizationFactor::Evaluate: Jacobian", 1e-13),
        ("MARGIN_SECOND_NEW: information", 1e-11),
        ("MARGIN_SECOND_NEW: gradient", 1e-11),
        ("MARGIN_OLD: information", 5e-3),                  # (the compiled reference's eigen pseudo-inverse with the build's eigensolver: DESIGN 2)
        ("MARGIN_OLD: gradient", 5e-3),
        ("linearisation points", 1e-300),
    ])


@pytest.mark.gpu
def test_hip_path_against_the_oracle_on_random_windows_and_intervals():
    import parity_sweep as S
    r = S.sweep(60, 1234567)
    assert not r["differ"], r["differ"]                      # every window stops at the oracle's iteration for the oracle's reason
    assert sum(len(v) for v in r["split"].values()) <= 6     # (fixed iteration counts on stagnating windows: a noise-level decrease accepted on one side only)
    assert r["worst_cond"] < 50.0                            # equilibrated condition number of every covariance of two steps or more
    assert r["pd_count"][2] <= 2 and all("1 steps" in s for s in r["one_sided"])   # one-sided refusals: single-step intervals only
    _held(r["worst2"], [
        ("state (33 scalars)", 1e-14),
        ("jacobian, of its largest entry", 1e-13),
        ("covariance, of its largest entry", 1e-12),
        ("covariance, per diagonal", 1e-12),
        ("two steps or more: whitened residual, per entry", 1e-10),
        ("two steps or more: whitened Jacobians, per row", 1e-12),
        ("ONE step", 1e300),                                 # (rank-deficient covariance: no sqrt_info exists)
    ], value=lambda v: v[0])
    _held(r["worst"], [
        ("gauge fix", 1e-13),
        ("windows with a prior: states", 1e-8),
        ("windows with a prior: final cost", 1e-8),
        ("WITHOUT a prior (4 gauge directions free): states", 1e-6),
        ("WITHOUT a prior (4 gauge directions free): final cost", 1e-5),
        ("marginalise flag 1, eigen form: ", 1e-11),
        ("marginalise flag 1, factor form: ", 1e-11),
        ("marginalise flag 0, eigen form: ", 1e-4),          # (MARGIN_OLD: the floor FP64 inputs set is 6e-6 .. 2e-5, tests/test_golden.py)
        ("marginalise flag 0, factor form: ", 1e-4),
        ("NO prior (semi-definite): information", 1e-5),
        ("NO prior (semi-definite): gradient", 5e-3),
    ], value=lambda v: v[0])
