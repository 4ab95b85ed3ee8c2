"""Replay a dumped window (include/vilo_window_io.h) through the oracle and, if a GPU is present, the HIP path, and compare
with the result stored in the file (e.g. produced by the reference's real Ceres stack).
Usage: python tools/replay_window.py window.vwin [--iterations N] [--no-gpu]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerberus_amd import window_io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--iterations", type=int, default=12)
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    cfg, w, after, ref, flag = window_io.load(a.path)
    print("window: %d landmarks, %d observations, prior n = %d, marginalization_flag %d" % (w.L, w.n_obs, w.prior.struct.n if w.prior.struct.valid else 0, flag))
    names = ["pose", "speed_bias", "leg_bias", "ex_pose", "td", "inv_depth"]

    def report(tag, arrs, cost, its):
        print("%-8s iterations %3d  final cost %.10e" % (tag, its, cost))
        if after is not None:
            for n, x, r in zip(names, arrs, after):
                print("           max |%s - stored| = %.3e" % (n, float(np.abs(x - r).max()) if x.size else 0.0))
    if ref is not None:
        print("stored   iterations %3d  initial cost %.10e  final cost %.10e  termination %d" % (ref[0], ref[1], ref[2], ref[3]))
    before = w.clone_state()
    from oracle import oracle_py as O   # (test / analysis tooling: the oracle is never part of the product path)
    ocfg = O.config_from(cfg)
    sm = O.solve_window(ocfg, w, O.default_opts(fixed_iterations=False, max_num_iterations=a.iterations))
    report("oracle", w.state_arrays(), sm.final_cost, sm.iterations)
    if not a.no_gpu and os.path.exists("/dev/kfd"):
        from cerberus_amd import api
        w.set_state(before)
        ctx = api.Context(cfg, 0)
        s = ctx.solve_windows([w], api.default_solve_opts(False, a.iterations))[0]
        report("gpu", w.state_arrays(), s.final_cost, s.iterations)


if __name__ == "__main__":
    main()
