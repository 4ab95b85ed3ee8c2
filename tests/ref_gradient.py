"""Test helper: cost and gradient of a window's least-squares problem assembled in Python from the factor classes' own Evaluate() — the
oracle's (default) or, inside `with ref_py.as_oracle():`, the COMPILED REFERENCE's (oracle/_ref/libref.so) — with the residual blocks
enumerated as Estimator::optimization adds them (estimator.cpp:1107-1216), ceres::HuberLoss(1.0) on the visual blocks (:1062) applied as
Ceres applies a loss (rho'(s) J^T r, cost rho(s) / 2) and PoseLocalParameterization's [I6; 0] Jacobian on the 7-dim blocks. Nothing of
the solvers under test is in it: it says what the cost function IS, so that a state can be tested for being a stationary point of it."""
import numpy as np

from oracle import oracle_py as O


def _blocks(w):
    """state arrays by block kind (VILO_BLK_*: 0 pose, 1 speed / bias, 2 leg bias, 3 extrinsic, 4 td)"""
    return {0: w.pose, 1: w.speed_bias, 2: w.leg_bias, 3: w.ex_pose, 4: w.td.reshape(1, 1)}


def cost_and_gradient(cfg, w, huber_delta=1.0):
    """Returns (cost, g, h): g[(kind, index)] / g[('lam', l)] = gradient in the block's LOCAL coordinates, h likewise = the diagonal of the
    Gauss-Newton Hessian sum_blocks rho' J^T J (what a gradient entry is measured against: g_i / sqrt(h_i) is dimensionless)."""
    F = w.F
    g, h = {}, {}
    cost = 0.0

    def add(key, J, r, wgt):
        J = J[:, :6] if J.shape[1] == 7 else J
        g[key] = g.get(key, 0.0) + wgt * (J.T @ r)
        h[key] = h.get(key, 0.0) + wgt * (J * J).sum(axis=0)

    def block(keys, r, Js, robust):
        nonlocal cost
        s = float(r @ r)
        if robust and s > huber_delta ** 2:
            cost += 0.5 * (2.0 * huber_delta * np.sqrt(s) - huber_delta ** 2)
            wgt = huber_delta / np.sqrt(s)
        else:
            cost += 0.5 * s
            wgt = 1.0
        for key, J in zip(keys, Js):
            add(key, J, r, wgt)
    st = _blocks(w)
    pr = w.prior
    if pr.struct.valid:
        keys = [(pr.struct.block_id[k] // 16, pr.struct.block_id[k] % 16) for k in range(pr.struct.n_blocks)]
        r, Js = O.eval_prior(pr.struct, [st[kind][idx] for kind, idx in keys])
        block(keys, r, Js, False)
    for k in range(F - 1):
        keys = [(0, k), (1, k), (2, k), (0, k + 1), (1, k + 1), (2, k + 1)]
        if w.use_leg:
            r, Js = O.eval_imu_leg(cfg, w.preint[k], [st[kd][i] for kd, i in keys])
        else:
            keys = [(0, k), (1, k), (0, k + 1), (1, k + 1)]
            r, Js = O.eval_imu(cfg, w.preint_imu[k], [st[kd][i] for kd, i in keys])
        block(keys, r, Js, False)
    td = w.td
    for l in range(w.L):
        s, o0, o1 = int(w.lm_start_frame[l]), int(w.lm_obs_offset[l]), int(w.lm_obs_offset[l + 1])
        f0 = w.obs[o0]
        lam = w.inv_depth[l:l + 1]
        for o in range(o0, o1):
            j = s + (o - o0)
            fj = w.obs[o]
            if j != s:
                obs = np.concatenate([f0[0:3], fj[0:3], f0[6:8], fj[6:8], [f0[10], fj[10]]])
                r, Js = O.eval_proj(0, cfg, obs, [w.pose[s], w.pose[j], w.ex_pose[0], lam, td])
                block([(0, s), (0, j), (3, 0), ("lam", l), (4, 0)], r, Js, True)
            if w.obs_is_stereo[o]:
                obs = np.concatenate([f0[0:3], fj[3:6], f0[6:8], fj[8:10], [f0[10], fj[10]]])
                if j != s:
                    r, Js = O.eval_proj(1, cfg, obs, [w.pose[s], w.pose[j], w.ex_pose[0], w.ex_pose[1], lam, td])
                    block([(0, s), (0, j), (3, 0), (3, 1), ("lam", l), (4, 0)], r, Js, True)
                else:
                    r, Js = O.eval_proj(2, cfg, obs, [w.ex_pose[0], w.ex_pose[1], lam, td])
                    block([(3, 0), (3, 1), ("lam", l), (4, 0)], r, Js, True)
    return cost, g, h


def free_gradient(w, g):
    """the entries of g (or h) that belong to blocks the solve may move (SetParameterBlockConstant, estimator.cpp:1074-1105), as one vector"""
    out = []
    for key, v in sorted(g.items(), key=lambda kv: (str(type(kv[0][0])), kv[0])):
        kind = key[0]
        if kind == 4 and w.td_const:
            continue
        if kind == 3 and w.ex_const:
            continue
        if kind == 2 and w.leg_bias_const:
            continue
        out.append(np.atleast_1d(v).ravel())
    return np.concatenate(out)


def dense_jacobian(cfg, w, huber_delta=1.0):
    """The whole problem as Ceres' evaluator hands it to the linear solver: r (all residual blocks stacked, loss-corrected) and the dense
    Jacobian in LOCAL coordinates over the free blocks — Corrector's first branch (rho'' <= 0 for Huber: residual and Jacobian of a block
    times sqrt(rho'), corrector.cc) — with the column layout {key: slice}. Inverse depths last."""
    rows_r, rows_J = [], []
    keys_seen = {}

    def free(key):
        k = key[0]
        return not ((k == 4 and w.td_const) or (k == 3 and w.ex_const) or (k == 2 and w.leg_bias_const))

    def block(keys, r, Js, robust):
        s = float(r @ r)
        sc = np.sqrt(huber_delta / np.sqrt(s)) if (robust and s > huber_delta ** 2) else 1.0
        rows_r.append(sc * r)
        ent = []
        for key, J in zip(keys, Js):
            J = J[:, :6] if J.shape[1] == 7 else J
            if free(key):
                keys_seen.setdefault(key, J.shape[1])
                ent.append((key, sc * J))
        rows_J.append(ent)
    # (same enumeration as cost_and_gradient)
    st = _blocks(w)
    pr = w.prior
    if pr.struct.valid:
        keys = [(pr.struct.block_id[k] // 16, pr.struct.block_id[k] % 16) for k in range(pr.struct.n_blocks)]
        r, Js = O.eval_prior(pr.struct, [st[kind][idx] for kind, idx in keys])
        block(keys, r, Js, False)
    for k in range(w.F - 1):
        if w.use_leg:
            keys = [(0, k), (1, k), (2, k), (0, k + 1), (1, k + 1), (2, k + 1)]
            r, Js = O.eval_imu_leg(cfg, w.preint[k], [st[kd][i] for kd, i in keys])
        else:   # USE_LEG 0: IMUFactor, no leg-bias blocks in the problem (estimator.cpp:1160-1171)
            keys = [(0, k), (1, k), (0, k + 1), (1, k + 1)]
            r, Js = O.eval_imu(cfg, w.preint_imu[k], [st[kd][i] for kd, i in keys])
        block(keys, r, Js, False)
    td = w.td
    for l in range(w.L):
        s, o0, o1 = int(w.lm_start_frame[l]), int(w.lm_obs_offset[l]), int(w.lm_obs_offset[l + 1])
        f0 = w.obs[o0]
        lam = w.inv_depth[l:l + 1]
        for o in range(o0, o1):
            j = s + (o - o0)
            fj = w.obs[o]
            if j != s:
                obs = np.concatenate([f0[0:3], fj[0:3], f0[6:8], fj[6:8], [f0[10], fj[10]]])
                r, Js = O.eval_proj(0, cfg, obs, [w.pose[s], w.pose[j], w.ex_pose[0], lam, td])
                block([(0, s), (0, j), (3, 0), (9, l), (4, 0)], r, Js, True)
            if w.obs_is_stereo[o]:
                obs = np.concatenate([f0[0:3], fj[3:6], f0[6:8], fj[8:10], [f0[10], fj[10]]])
                if j != s:
                    r, Js = O.eval_proj(1, cfg, obs, [w.pose[s], w.pose[j], w.ex_pose[0], w.ex_pose[1], lam, td])
                    block([(0, s), (0, j), (3, 0), (3, 1), (9, l), (4, 0)], r, Js, True)
                else:
                    r, Js = O.eval_proj(2, cfg, obs, [w.ex_pose[0], w.ex_pose[1], lam, td])
                    block([(3, 0), (3, 1), (9, l), (4, 0)], r, Js, True)
    cols, n = {}, 0
    for key in sorted(keys_seen):
        cols[key] = slice(n, n + keys_seen[key])
        n += keys_seen[key]
    m = sum(len(r) for r in rows_r)
    J = np.zeros((m, n))
    at = 0
    for r, ent in zip(rows_r, rows_J):
        for key, Jb in ent:
            J[at:at + len(r), cols[key]] += Jb
        at += len(r)
    return np.concatenate(rows_r), J, cols


def apply_step(w, cols, delta):
    """x (+) delta on w's state arrays: PoseLocalParameterization::Plus on the 7-dim blocks, plain addition elsewhere"""
    st = _blocks(w)
    for key, sl in cols.items():
        d = delta[sl]
        if key[0] == 9:
            w.inv_depth[key[1]] += d[0]
        elif key[0] in (0, 3):
            st[key[0]][key[1]][:] = O.pose_plus(st[key[0]][key[1]], d)
        else:
            st[key[0]][key[1]][:] += d
