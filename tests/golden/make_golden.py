"""Freezes outputs of the REFERENCE's own code (oracle/_ref/libref.so = the Cerberus factor sources compiled unmodified
from /root/reference, see oracle/ref_build/README.md) on seeded inputs into tests/golden/reference_vectors.npz.
Run in the build container, where /root/reference exists:   python tests/golden/make_golden.py
The fixtures let the parity tests check oracle AND HIP path against real reference outputs on a box without the
reference tree (the GPU box, a fresh clone)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import rand_pose  # noqa: E402
from cerberus_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from oracle import ref_py as R  # noqa: E402

RF = np.array([0.1805, -0.047, -0.0838, 0.21])
N_LANDMARKS, SEED = 24, 5


def main():
    cfg = O.default_config()
    out = {}
    rng = np.random.default_rng(2026)
    with R.as_oracle():   # every O.* call below executes the compiled reference
        # A1Kinematics
        q = np.array([0.2, 0.8, -1.6]) + 0.5 * rng.normal(size=(16, 3))
        lc = 0.21 + 0.02 * rng.normal(size=16)
        ks = [O.kin(q[i], lc[i], RF) for i in range(16)]
        out.update(kin_q=q, kin_lc=lc, kin_f=np.array([k["f"] for k in ks]), kin_J=np.array([k["J"] for k in ks]),
                   kin_df_drho=np.array([k["df_drho"] for k in ks]), kin_dJ_dq=np.array([k["dJ_dq"] for k in ks]),
                   kin_dJ_drho=np.array([k["dJ_drho"] for k in ks]))
        # preintegration of the 10 intervals of a synthetic window (inputs are regenerated from the seed by the tests)
        w = synth.make_window(synth.default_config(), n_landmarks=N_LANDMARKS, seed=SEED)
        O.fill_preint(cfg, w)
        out.update(win_landmarks=np.array(N_LANDMARKS), win_seed=np.array(SEED), preint=w.preint.copy(), preint_imu=w.preint_imu.copy())
        # IMULegFactor / IMUFactor at perturbed states
        P_all, r_all, J_all, ri_all, Ji_all = [], [], [], [], []
        for k in range(w.F - 1):
            P = [w.pose[k].copy(), w.speed_bias[k].copy(), w.leg_bias[k].copy(), w.pose[k + 1].copy(), w.speed_bias[k + 1].copy(), w.leg_bias[k + 1].copy()]
            for p in P:
                if p.size == 7:
                    p[:] = O.pose_plus(p, 1e-2 * rng.normal(size=6))
                else:
                    p += 1e-2 * rng.normal(size=p.size)
            r, J = O.eval_imu_leg(cfg, w.preint[k], P)
            ri, Ji = O.eval_imu(cfg, w.preint_imu[k], [P[0], P[1], P[3], P[4]])
            P_all.append(np.concatenate(P)); r_all.append(r); J_all.append(np.hstack(J)); ri_all.append(ri); Ji_all.append(np.hstack(Ji))
        out.update(imu_params=np.array(P_all), imuleg_r=np.array(r_all), imuleg_J=np.array(J_all), imu_r=np.array(ri_all), imu_J=np.array(Ji_all))
        # projection factors
        for kind, sizes in enumerate([[7, 7, 7, 1, 1], [7, 7, 7, 7, 1, 1], [7, 7, 1, 1]]):
            obs_l, par_l, r_l, J_l = [], [], [], []
            for _ in range(16):
                pi, pj = rand_pose(rng, 0.5), rand_pose(rng, 0.5)
                ex0, ex1 = rand_pose(rng, 0.05), rand_pose(rng, 0.05)
                ex1[0] += 0.1
                obs = np.concatenate([[0.3 * rng.normal(), 0.3 * rng.normal(), 1.0], [0.3 * rng.normal(), 0.3 * rng.normal(), 1.0],
                                      0.1 * rng.normal(size=2), 0.1 * rng.normal(size=2), [0.002, 0.004]])
                lam, td = np.array([abs(0.3 + 0.1 * rng.normal())]), np.array([0.01])
                P = [[pi, pj, ex0, lam, td], [pi, pj, ex0, ex1, lam, td], [ex0, ex1, lam, td]][kind]
                r, J = O.eval_proj(kind, cfg, obs, P)
                obs_l.append(obs); par_l.append(np.concatenate(P)); r_l.append(r); J_l.append(np.hstack(J))
            out.update({"proj%d_obs" % kind: np.array(obs_l), "proj%d_params" % kind: np.array(par_l), "proj%d_r" % kind: np.array(r_l),
                        "proj%d_J" % kind: np.array(J_l)})
        # PoseLocalParameterization::Plus
        x = np.array([rand_pose(rng, 1.0) for _ in range(16)])
        d = 0.1 * rng.normal(size=(16, 6))
        out.update(plus_x=x, plus_d=d, plus_out=np.array([O.pose_plus(x[i], d[i]) for i in range(16)]))
        # MarginalizationFactor::Evaluate on the synthetic prior of the window
        pr = w.prior
        params, off = [], 0
        for k in range(pr.struct.n_blocks):
            gs = pr.struct.block_size[k]
            x0 = pr.x0[off:off + gs].copy()
            off += gs
            params.append(O.pose_plus(x0, 1e-2 * rng.normal(size=6)) if gs == 7 else x0 + 1e-2 * rng.normal(size=gs))
        r, J = O.eval_prior(pr.struct, params)
        out.update(prior_params=np.concatenate(params), prior_r=r, prior_J=np.hstack(J))
    # MarginalizationInfo::marginalize (reference) -> order-independent information blocks, sorted by block id
    for mode in (0, 1):
        p = synth.PriorData()
        assert R.marginalize(cfg, w, mode, p) == 0
        Hb, bb, x0 = R.prior_information(p)
        ids = sorted(x0)
        out["marg%d_ids" % mode] = np.array(ids)
        out["marg%d_sizes" % mode] = np.array([x0[i].size for i in ids])
        out["marg%d_x0" % mode] = np.concatenate([x0[i] for i in ids])
        out["marg%d_H" % mode] = np.block([[Hb[(a, c)] for c in ids] for a in ids])
        out["marg%d_b" % mode] = np.concatenate([bb[i] for i in ids])
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
