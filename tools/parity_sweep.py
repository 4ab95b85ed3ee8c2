"""A randomised GPU-against-oracle sweep of the whole path, beyond what the -m gpu suite holds at fixed seeds (run on the GPU box; the
oracle is the checker, as in tests/): N random windows — 1 .. 600 landmarks, with / without prior, IMU-leg or plain IMU factors, constant or
estimated extrinsics / leg biases / td — through

  solve      vilo_solve_windows (ONE call for all windows: the sub-batch pipeline when N allows) against oracle solve_window, fixed 8
             iterations and to convergence: final states relative to max(1, |state|), final cost relative
  gauge fix  vilo_gauge_fix against the oracle's on the solved states
  marginalise  both flags, eigen form and factor form, against the oracle's prior: J0^T J0 / J0^T r0 per kept block pair in units of the
             blocks' own diagonals (tests/marg_exact.py scaling, against the ORACLE's information here: its own distance from the exact
             complement is what tests/test_golden.py measures)

  factors    vilo_preintegrate on intervals of random length under the three contact models — the gait's flags mixed with fractional values
             around the 0.5 threshold and runs with every foot in the air, foot forces for model 2 — against the oracle's integration, and
             IMULegFactor::Evaluate on the resulting records at the windows' states

and prints the worst case per category with the window that produced it. python tools/parity_sweep.py [N] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cerberus_amd import api, synth  # noqa: E402
from cerberus_amd.synth import PriorData  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) if a.size else 0.0


def info_err(pg, po):
    """worst |dH_ij| / sqrt(H_ii H_jj), worst |db_i| / sqrt(H_ii) (relative to the largest whitened gradient entry) over the kept blocks"""
    from marg_exact import block_table
    n = po.struct.n
    Jo, ro = po.J0[: n * n].reshape(n, n), po.r0[:n]
    Jg, rg = pg.J0[: n * n].reshape(n, n), pg.r0[:n]
    Ho, bo, Hg, bg = Jo.T @ Jo, Jo.T @ ro, Jg.T @ Jg, Jg.T @ rg
    to, tg = block_table(po), block_table(pg)
    assert set(to) == set(tg) and pg.struct.n == n
    d = np.sqrt(np.maximum(np.diag(Ho), 1e-300))
    eh = eb = 0.0
    for a, (ia, la) in tg.items():
        ja, _ = to[a]
        eb = max(eb, float((np.abs(bg[ia:ia + la] - bo[ja:ja + la]) / d[ja:ja + la]).max()))
        for c, (ic, lc) in tg.items():
            jc, _ = to[c]
            eh = max(eh, float((np.abs(Hg[ia:ia + la, ic:ic + lc] - Ho[ja:ja + la, jc:jc + lc]) / np.outer(d[ja:ja + la], d[jc:jc + lc])).max()))
    return eh, eb / max(1.0, float(np.abs(bo / d).max()))


def sweep(N, seed):
    """-> dict(worst: windows' categories -> (value, window, spec), worst2: factor categories -> (value, contact model), split, differ,
    pd_count, cls_count, one_sided, worst_cond, n_int, seconds)"""
    rng = np.random.default_rng(seed)
    cfg, ocfg = synth.default_config(), O.default_config()
    ctx = api.Context(cfg, 0)
    specs = []
    for i in range(N):
        leg = int(rng.integers(0, 6) != 0)
        # (the generator's prior carries leg-bias blocks: windows with plain IMU factors go without one, as in tests/test_gpu_parity.py::_vins)
        specs.append(dict(n_landmarks=int(rng.choice([1, 3, 8, 20, 60, 130, 200, 333, 600])), seed=int(rng.integers(1, 1 << 30)),
                          with_prior=bool(rng.integers(0, 5) != 0) and leg == 1, use_leg=leg,
                          consts=[(0, 0, 1), (0, 1, 1), (1, 0, 1), (1, 1, 1), (0, 0, 0), (1, 0, 0)][int(rng.integers(0, 6))]))   # (leg bias, extrinsics, td constant)

    def fresh(sp):
        w = synth.make_window(cfg, n_landmarks=sp["n_landmarks"], seed=sp["seed"], with_prior=sp["with_prior"])
        w.use_leg = sp["use_leg"]
        w.leg_bias_const, w.ex_const, w.td_const = sp["consts"]
        O.fill_preint(ocfg, w)
        return w
    worst, differ, split, worst2 = {}, {}, {}, {}

    def note2(key, val, tag):
        if key not in worst2 or val > worst2[key][0]:
            worst2[key] = (val, tag)

    def note(key, val, i):
        if key not in worst or val > worst[key][0]:
            worst[key] = (val, i, specs[i])
    t0 = time.time()
    for name, opts_g, opts_o in (("solve, 8 fixed iterations", api.default_solve_opts(True, 8), O.default_opts(True, 8)),
                                 ("solve, to convergence (<= 12)", api.default_solve_opts(False, 12), O.default_opts(False, 12))):
        for leg, tdc in ((1, 1), (1, 0), (0, 1), (0, 0)):     # (one IMU factor kind per batch; a batch with a window that estimates td takes the 23-column visual rows, one without the compact ones)
            idx = [i for i in range(N) if specs[i]["use_leg"] == leg and specs[i]["consts"][2] == tdc]
            if not idx:
                continue
            wg = [fresh(specs[i]) for i in idx]
            sg = ctx.solve_windows(wg, opts_g)
            for k, i in enumerate(idx):
                wo = fresh(specs[i])
                before = wo.clone_state()
                so = O.solve_window(ocfg, wo, opts_o)
                kind = ", windows with a prior" if specs[i]["with_prior"] else ", windows WITHOUT a prior (4 gauge directions free)"
                note(name + kind + ": final cost", abs(sg[k].final_cost - so.final_cost) / max(1.0, abs(so.final_cost)), i)
                if sg[k].num_successful != so.num_successful:
                    # a step whose relative decrease is rounding noise (a window that stagnates with fixed iterations: tolerances are
                    # off) is accepted on one side and rejected on the other; from there the two are different, equally valid runs
                    split[name] = split.get(name, []) + [i]
                else:
                    note(name + kind + ": states", max(rel(a, b) for a, b in zip(wg[k].state_arrays(), wo.state_arrays())), i)
                if sg[k].iterations != so.iterations or sg[k].termination != so.termination:
                    differ[name] = differ.get(name, 0) + 1
                if name.startswith("solve, 8"):
                    # gauge fix and both marginalisations at the oracle's solved state (identical inputs for both sides)
                    wg2 = fresh(specs[i])
                    wg2.set_state(wo.clone_state())
                    ctx.gauge_fix(before, wg2)
                    O.gauge_fix(before, wo)
                    note("gauge fix", max(rel(a, b) for a, b in zip(wg2.state_arrays(), wo.state_arrays())), i)
                    wg2.set_state(wo.clone_state())
                    for mode in (0, 1):
                        po = PriorData()
                        if O.marginalize(ocfg, wo, mode, po)[0] != 0 or not po.struct.valid:
                            continue
                        for form in ("eigen", "factor"):
                            ctx.set_prior_form(form)
                            pg = PriorData()
                            ctx.marginalize(wg2, mode, pg)
                            assert pg.struct.valid == 1 and pg.blocks() == po.blocks()
                            if specs[i]["with_prior"]:
                                eh, eb = info_err(pg, po)
                                note("marginalise flag %d, %s form: information, per block diagonal" % (mode, form), eh, i)
                                note("marginalise flag %d, %s form: gradient, per block diagonal" % (mode, form), eb, i)
                            else:
                                # no prior: A' is semi-definite (the four gauge directions carry rounding noise that the 1e-8 eigenvalue
                                # threshold keeps or drops): compared in units of the largest entry, like the suite's no-prior test
                                n = po.struct.n
                                Jo, Jg = po.J0[: n * n].reshape(n, n), pg.J0[: n * n].reshape(n, n)
                                Ho, Hg = Jo.T @ Jo, Jg.T @ Jg
                                note("marginalise flag %d, %s form, NO prior (semi-definite): information, of the largest entry" % (mode, form), float(np.abs(Hg - Ho).max() / np.abs(Ho).max()), i)
                                bo, bg = Jo.T @ po.r0[:n], Jg.T @ pg.r0[:n]
                                note("marginalise flag %d, %s form, NO prior (semi-definite): gradient, of the largest entry" % (mode, form), float(np.abs(bg - bo).max() / np.abs(bo).max()), i)
                        ctx.set_prior_form("eigen")
    # ---- the factor level: preintegration (three contact models, contact inputs no gait produces) and IMULegFactor / IMUFactor::Evaluate ----
    import copy

    def relm(a, b):
        return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
    n_int = 0
    pd_count = [0, 0, 0]
    one_sided, cls_count, worst_cond = [], {}, [0.0]
    for ctype in (0, 1, 2):
        c2, o2 = copy.copy(cfg), copy.copy(ocfg)
        c2.contact_sensor_type = ctype; o2.contact_sensor_type = ctype
        cc = api.Context(c2, 0)
        for rep in range(max(4, N // 10)):
            w = synth.make_window(cfg, n_landmarks=3, seed=int(rng.integers(1, 1 << 30)))
            smp = np.array(w.samples, copy=True)
            nS = int(w.sample_offsets[-1])
            # contact inputs: the gait's flags, fractional values around the 0.5 threshold, runs with every foot in the air
            c = smp[:nS, 31:35]
            frac = rng.random(c.shape) < 0.25
            c[frac] = rng.choice([0.0, 0.3, 0.49999, 0.5, 0.7, 1.0], size=int(frac.sum()))
            for _ in range(3):
                a0 = int(rng.integers(0, max(1, nS - 6)))
                c[a0:a0 + int(rng.integers(1, 6))] = 0.0
            if ctype == 2:
                c[:] = 15.0 + 140.0 * c + 4.0 * rng.normal(size=c.shape)       # foot forces (tests/test_gpu_parity.py::_force_samples)
            # intervals of random length inside the window's ten (>= 2 samples), biases away from the generator's
            offs = [0]
            parts, lins = [], []
            for k in range(10):
                a0, a1 = int(w.sample_offsets[k]), int(w.sample_offsets[k + 1])
                n = int(rng.integers(2, a1 - a0 + 1))
                parts.append(smp[a0:a0 + n]); offs.append(offs[-1] + n)
                lins.append(w.lin[k] + np.concatenate([0.05 * rng.normal(size=3), 0.01 * rng.normal(size=3), 0.01 * rng.normal(size=4)]))
            S, L = np.ascontiguousarray(np.concatenate(parts)), np.ascontiguousarray(np.stack(lins))
            out = cc.preintegrate(S, np.array(offs, np.int32), L)
            P = [w.pose[:-1], w.speed_bias[:-1], w.leg_bias[:-1], w.pose[1:], w.speed_bias[1:], w.leg_bias[1:]]
            for k in range(10):
                b = O.preintegrate_imu_leg(o2, S[offs[k]:offs[k + 1]], L[k])
                # a covariance without sqrt_info (not positive definite in FP64: e.g. the 10e10 uncertainties of feet in the air next to
                # 1e-11 ones) must be refused by both sides
                try:
                    rg, Jg = cc.eval_imu_leg(out[k:k + 1], [p[k:k + 1] for p in P])
                    gpu_ok = True
                except api.ViloError:
                    gpu_ok = False
                try:
                    O.sqrt_info(b[33 + 961:].reshape(31, 31))
                    orc_ok = True
                except FloatingPointError:
                    orc_ok = False
                pd_count[0] += 1 if (gpu_ok and orc_ok) else 0
                pd_count[1] += 1 if (not gpu_ok and not orc_ok) else 0
                pd_count[2] += 1 if gpu_ok != orc_ok else 0
                n_int += 1
                tag = "preintegration, contact model %d: " % ctype
                note2(tag + "state (33 scalars)", float(np.abs(out[k][:33] - b[:33]).max() / max(1.0, np.abs(b[:33]).max())), ctype)
                note2(tag + "jacobian, of its largest entry", relm(out[k][33:33 + 961], b[33:33 + 961]), ctype)
                note2(tag + "covariance, of its largest entry", relm(out[k][33 + 961:], b[33 + 961:]), ctype)
                cg_, co_ = out[k][33 + 961:].reshape(31, 31), b[33 + 961:].reshape(31, 31)
                dd = np.sqrt(np.abs(np.diag(co_)))
                note2(tag + "covariance, per diagonal (|dC_ij| / sqrt(C_ii C_jj))", float((np.abs(cg_ - co_) / np.outer(dd, dd)).max()), ctype)
                # What the accuracy of LLT(cov^-1) depends on is the condition number after diagonal equilibration (the raw one — 1e11 for a
                # trot, 1e21 with feet in the air — is units: 1e-11 variances beside 10e10 ones; Cholesky does not see that scaling). It is
                # ~ 15 for every interval of two integration steps or more and ~ 1e16 for an interval of ONE step, whose covariance
                # V N V^T is rank-deficient: sqrt_info does not exist there, in any arithmetic.
                cov = b[33 + 961:].reshape(31, 31)
                dg = np.sqrt(np.abs(np.diag(cov)))
                ev = np.linalg.eigvalsh(0.5 * (cov + cov.T) / np.outer(dg, dg))
                cond = float(ev[-1] / ev[0]) if ev[0] > 0 else np.inf
                worst_cond[0] = max(worst_cond[0], cond if offs[k + 1] - offs[k] > 2 else 0.0)
                cls = "two steps or more" if offs[k + 1] - offs[k] > 2 else "ONE step (covariance rank-deficient, equilibrated cond ~ 1e16: no sqrt_info in any arithmetic)"
                if gpu_ok != orc_ok:
                    one_sided.append("%d steps, equilibrated cond %.1e" % (offs[k + 1] - offs[k] - 1, cond))
                if gpu_ok and orc_ok:
                    ro, Jo = O.eval_imu_leg(o2, b, [p[k] for p in P])
                    note2("IMULegFactor::Evaluate on those records, intervals of %s: whitened residual, per entry" % cls, float((np.abs(rg[0] - ro) / np.maximum(np.abs(ro), 1e-12 * np.abs(ro).max())).max()), ctype)
                    note2("IMULegFactor::Evaluate on those records, intervals of %s: whitened Jacobians, per row" % cls, float((np.linalg.norm(np.hstack([Jg[q][0] for q in range(6)]) - np.hstack(Jo), axis=1) / np.linalg.norm(np.hstack(Jo), axis=1)).max()), ctype)
                    cls_count[cls] = cls_count.get(cls, 0) + 1
        cc.close()
    ctx.close()
    return dict(worst=worst, worst2=worst2, split=split, differ=differ, pd_count=pd_count, cls_count=cls_count, one_sided=one_sided,
                worst_cond=worst_cond[0], n_int=n_int, seconds=time.time() - t0)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260930
    r = sweep(N, seed)
    worst, worst2, split, differ, pd_count, cls_count, one_sided = r["worst"], r["worst2"], r["split"], r["differ"], r["pd_count"], r["cls_count"], r["one_sided"]
    print("parity sweep: %d random windows (seed %d), %d random preintegration intervals, %.0f s" % (N, seed, r["n_int"], r["seconds"]))
    for k in sorted(worst2):
        print("  %-150s %.2e   (contact model %d)" % (k, worst2[k][0], worst2[k][1]))
    print("  covariances with sqrt_info on both sides: %d (%s), refused by both (not positive definite in FP64): %d, refused by one side only: %d (%s); largest equilibrated condition number of a covariance of two steps or more: %.1f"
          % (pd_count[0], ", ".join("%s: %d" % kv for kv in sorted(cls_count.items())), pd_count[1], pd_count[2], "; ".join(one_sided) or "-", r["worst_cond"]))
    for k in sorted(worst):
        v, i, sp = worst[k]
        print("  %-104s %.2e   (window %d: %d landmarks, prior %d, use_leg %d, consts %s)" % (k, v, i, sp["n_landmarks"], sp["with_prior"], sp["use_leg"], sp["consts"]))
    for k in sorted(split):
        print("  %s: %d window(s) where a step with a relative decrease of rounding noise is accepted on one side, rejected on the other (%s) — states not compared, costs are" % (k, len(split[k]), split[k]))
    for k in sorted(differ):
        print("  %s: %d windows stop at another iteration or for another reason than the oracle's" % (k, differ[k]))
    if not differ:
        print("  every window stops at the oracle's iteration, for the oracle's reason")


if __name__ == "__main__":
    main()
