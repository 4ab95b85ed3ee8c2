"""CPU: the product's inline device math (cerberus_amd/csrc/factors.hpp, compiled for the host by
tests/host_check) against the oracle. Catches transcription errors before any GPU minute is spent;
the GPU parity tests (-m gpu) then check the very same functions as executed by the HIP kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rand_pose
from oracle import oracle_py as O
from test_oracle_factors import _proj_setup

dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def hc():
    d = os.path.join(ROOT, "tests", "host_check")
    so = os.path.join(d, "libhostcheck.so")
    srcs = [os.path.join(d, "host_check.cpp"), os.path.join(ROOT, "cerberus_amd", "csrc", "factors.hpp"),
            os.path.join(ROOT, "cerberus_amd", "csrc", "vilo_math.hpp"), os.path.join(ROOT, "cerberus_amd", "csrc", "visual_lin.hpp"),
            os.path.join(ROOT, "cerberus_amd", "csrc", "assemble_compact.hpp"), os.path.join(ROOT, "cerberus_amd", "csrc", "preint_blocks.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, srcs[0]])
    lib = C.CDLL(so)
    lib.hc_correct.restype = C.c_double
    lib.hc_vis_lin.restype = C.c_double
    return lib


def P(a):
    return a.ctypes.data_as(dp)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_proj_matches_oracle(hc, ocfg, kind):
    rng = np.random.default_rng(20 + kind)
    for _ in range(20):
        obs, params = _proj_setup(rng, kind)
        r_o, J_o = O.eval_proj(kind, ocfg, obs, params)
        if kind == 0:
            pose_i, pose_j, ex0, lam, td = params; ex1 = ex0
        elif kind == 1:
            pose_i, pose_j, ex0, ex1, lam, td = params
        else:
            ex0, ex1, lam, td = params; pose_i = pose_j = ex0
        r = np.zeros(2); Ji = np.zeros((2, 6)); Jj = np.zeros((2, 6)); Je0 = np.zeros((2, 6)); Je1 = np.zeros((2, 6))
        Jl = np.zeros(2); Jt = np.zeros(2)
        hc.hc_proj(kind, P(obs), P(pose_i), P(pose_j), P(ex0), P(ex1), C.c_double(lam[0]), C.c_double(td[0]),
                   C.c_double(460.0 / 1.5), P(r), 1, P(Ji), P(Jj), P(Je0), P(Je1), P(Jl), P(Jt))
        np.testing.assert_allclose(r, r_o, rtol=1e-13, atol=1e-12)
        mine = {0: [Ji, Jj, Je0, Jl, Jt], 1: [Ji, Jj, Je0, Je1, Jl, Jt], 2: [Je0, Je1, Jl, Jt]}[kind]
        for a, b in zip(mine, J_o):
            b = b[:, :6] if b.shape[1] == 7 else b[:, 0]
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-11 * max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("outlier", [False, True])
def test_fused_visual_forms_match_the_literal_factor_bodies(hc, kind, outlier):
    """k_visual_linearize evaluates the projection factors from hoisted rotation products with the Huber weight folded into the
    projection Jacobian (csrc/visual_lin.hpp). Against the literal Evaluate bodies (proj_factor, pinned to the oracle and the compiled
    reference above) followed by the generic ceres Corrector: the corrected rows [J | r], d r / d lambda and rho(s)."""
    rng = np.random.default_rng(70 + kind)
    sq, a = 460.0 / 1.5, 1.0
    for it in range(40):
        obs, params = _proj_setup(rng, kind)
        if kind == 0:
            pose_i, pose_j, ex0, lam, td = params
            ex1 = ex0.copy(); ex1[1] = -0.025
        elif kind == 1:
            pose_i, pose_j, ex0, ex1, lam, td = params
        else:
            ex0, ex1, lam, td = params
            pose_i = rand_pose(rng, 0.3); pose_j = rand_pose(rng, 0.3)
        # make the residual consistent (inlier) or leave it as drawn (tens of pixels: outlier)
        r = np.zeros(2); Ji = np.zeros((2, 6)); Jj = np.zeros((2, 6)); Je0 = np.zeros((2, 6)); Je1 = np.zeros((2, 6))
        Jl = np.zeros(2); Jt = np.zeros(2)

        def literal():
            hc.hc_proj(kind, P(obs), P(pose_i), P(pose_j), P(ex0), P(ex1), C.c_double(lam[0]), C.c_double(td[0]), C.c_double(sq), P(r), 1,
                       P(Ji), P(Jj), P(Je0), P(Je1), P(Jl), P(Jt))
        literal()
        if not outlier:
            obs[3:5] += r / sq * (1.0 - 1e-3 * rng.uniform(0.1, 1.0, size=2))   # move the observation onto the projection: sub-pixel residual
            literal()
            assert r @ r < a * a
        else:
            assert r @ r > a * a
        # generic corrector on every column and on the residual
        cols = np.concatenate([Ji, Jj, Je0, Je1, Jt[:, None], Jl[:, None]], axis=1)
        want = np.zeros((2, 27)); rho0 = 0.0
        for c in range(26):
            r2 = r.copy(); j2 = np.ascontiguousarray(cols[:, c])
            rho0 = hc.hc_correct(C.c_double(a), P(r2), P(j2))
            want[:, c] = j2
            want[:, 26] = r2
        x0 = np.zeros(23); x1 = np.zeros(23); jl = np.zeros(2)
        got_rho = hc.hc_vis_lin(kind, P(obs), P(pose_i), P(pose_j), P(ex0), P(ex1), C.c_double(lam[0]), C.c_double(td[0]), C.c_double(sq),
                                C.c_double(a), P(x0), P(x1), P(jl))
        g23 = np.stack([x0, x1])
        # the Gram slot's 23 columns (trans 3 | rot_i 3 | rot_j 3 | ex0 6 | r | ex1 6 | td) back in the order [pose_i 6 | pose_j 6 | ex0 6 | ex1 6 | td | r]
        got = np.concatenate([g23[:, 0:6], -g23[:, 0:3], g23[:, 6:9], g23[:, 9:15], g23[:, 16:22], g23[:, 22:23], g23[:, 15:16]], axis=1)
        scale = max(1.0, np.abs(want).max())
        np.testing.assert_allclose(got[:, :25], want[:, :25], rtol=1e-11, atol=1e-12 * scale)
        np.testing.assert_allclose(jl, want[:, 25], rtol=1e-11, atol=1e-12 * scale)
        np.testing.assert_allclose(got[:, 25], want[:, 26], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(got_rho, rho0, rtol=1e-10, atol=1e-13)   # (rho = |r|^2 of a sub-pixel residual formed by cancellation)
        if kind != 1:
            assert np.all(got[:, 18:24] == 0.0) if kind == 0 else np.all(got[:, 0:12] == 0.0)


def test_imu_leg_raw_matches_oracle(hc, ocfg, small_window):
    w = small_window
    for k in range(10):
        params = [w.pose[k], w.speed_bias[k], w.leg_bias[k], w.pose[k + 1], w.speed_bias[k + 1], w.leg_bias[k + 1]]
        r_o, J_o = O.eval_imu_leg(ocfg, w.preint[k], params)
        r = np.zeros(31); J = np.zeros((31, 38))
        pre = np.ascontiguousarray(w.preint[k])
        hc.hc_imu_leg_raw(P(pre), C.c_double(9.805), *[P(np.ascontiguousarray(p)) for p in params], P(r), P(J))
        U = O.sqrt_info(pre[33 + 961:].reshape(31, 31))
        np.testing.assert_allclose(U @ r, r_o, rtol=1e-9, atol=1e-9 * np.abs(r_o).max())
        Jw = U @ J
        loc = np.concatenate([J_o[0][:, :6], J_o[1], J_o[2], J_o[3][:, :6], J_o[4], J_o[5]], axis=1)
        np.testing.assert_allclose(Jw, loc, rtol=1e-9, atol=1e-10 * np.abs(loc).max())


def test_imu_raw_matches_oracle(hc, ocfg, small_window):
    w = small_window
    for k in (0, 3, 9):
        params = [w.pose[k], w.speed_bias[k], w.pose[k + 1], w.speed_bias[k + 1]]
        r_o, J_o = O.eval_imu(ocfg, w.preint_imu[k], params)
        r = np.zeros(15); J = np.zeros((15, 30))
        pre = np.ascontiguousarray(w.preint_imu[k])
        hc.hc_imu_raw(P(pre), C.c_double(9.805), *[P(np.ascontiguousarray(p)) for p in params], P(r), P(J))
        U = O.sqrt_info(pre[17 + 225:].reshape(15, 15))
        np.testing.assert_allclose(U @ r, r_o, rtol=1e-9, atol=1e-9 * np.abs(r_o).max())
        loc = np.concatenate([J_o[0][:, :6], J_o[1], J_o[2][:, :6], J_o[3]], axis=1)
        np.testing.assert_allclose(U @ J, loc, rtol=1e-9, atol=1e-10 * np.abs(loc).max())


def test_imu_block_pool_and_gather_table_reproduce_the_raw_factors(hc, small_window):
    """Small batches linearise an IMU factor inside one wave (imu_fused_body): lane 0 evaluates seven 3 x 3 matrices and the residual
    (imu_blocks), every lane fetches its entries of the [J | r] operand image through a compile-time table (imu_gather_table). Emulated
    on the host entry by entry against what imu_leg_raw / imu_raw write: the same image, bit for bit — the expressions are shared, the
    table only says where an entry lives and which of 1 / -1 / T / -T multiplies it."""
    w = small_window
    for k in range(10):
        params = [np.ascontiguousarray(p) for p in (w.pose[k], w.speed_bias[k], w.leg_bias[k], w.pose[k + 1], w.speed_bias[k + 1], w.leg_bias[k + 1])]
        pre, pre_imu = np.ascontiguousarray(w.preint[k]), np.ascontiguousarray(w.preint_imu[k])
        # IMULegFactor
        r = np.zeros(31); J = np.zeros((31, 38))
        hc.hc_imu_leg_raw(P(pre), C.c_double(9.805), *[P(p) for p in params], P(r), P(J))
        img = np.zeros((32, 48))
        hc.hc_imu_blocks_gather(P(pre), P(pre_imu), 1, C.c_double(9.805), *[P(p) for p in params], P(img))
        np.testing.assert_array_equal(img[:31, :38], J)
        np.testing.assert_array_equal(img[:31, 38], r)
        assert not img[31].any() and not img[:, 39:].any()
        # IMUFactor, embedded: rows 0 .. 14, frame-j blocks from column 19
        r = np.zeros(15); J = np.zeros((15, 30))
        hc.hc_imu_raw(P(pre_imu), C.c_double(9.805), P(params[0]), P(params[1]), P(params[3]), P(params[4]), P(r), P(J))
        img = np.zeros((32, 48))
        hc.hc_imu_blocks_gather(P(pre), P(pre_imu), 0, C.c_double(9.805), *[P(p) for p in params], P(img))
        np.testing.assert_array_equal(img[:15, :15], J[:, :15])
        np.testing.assert_array_equal(img[:15, 19:34], J[:, 15:])
        np.testing.assert_array_equal(img[:15, 38], r)
        assert not img[15:].any() and not img[:15, 15:19].any() and not img[:15, 34:38].any()


def test_leg_kin_matches_oracle(hc):
    rng = np.random.default_rng(5)
    rf = np.array([0.1805, -0.047, -0.0838, 0.21])
    for _ in range(10):
        q = np.array([0.1, 0.8, -1.5]) + 0.4 * rng.normal(size=3)
        lc = 0.21 + 0.01 * rng.normal()
        k = O.kin(q, lc, rf)
        f = np.zeros(3); J = np.zeros(9); dfr = np.zeros(3); dJ = np.zeros(27); dJr = np.zeros(9)
        hc.hc_leg_kin(P(q), C.c_double(lc), P(rf), P(f), P(J), P(dfr), P(dJ), P(dJr))
        np.testing.assert_allclose(f, k["f"], atol=1e-15)
        np.testing.assert_allclose(J.reshape(3, 3), k["J"], atol=1e-15)
        np.testing.assert_allclose(dfr, k["df_drho"], atol=1e-15)
        for n in range(3):  # oracle dJ_dq: 9x3, column n = vec_colmajor(dJ/dq_n)
            np.testing.assert_allclose(dJ[9 * n:9 * n + 9].reshape(3, 3), k["dJ_dq"][:, n].reshape(3, 3).T, atol=1e-15)
        np.testing.assert_allclose(dJr.reshape(3, 3), k["dJ_drho"].reshape(3, 3).T, atol=1e-15)


def test_pose_plus_and_prior_dx(hc):
    rng = np.random.default_rng(6)
    x = rand_pose(rng); d = 0.05 * rng.normal(size=6)
    out = np.zeros(7)
    hc.hc_pose_plus(P(x), P(d), P(out))
    np.testing.assert_allclose(out, O.pose_plus(x, d), atol=1e-15)
    dx = np.zeros(6)
    hc.hc_prior_dx(P(out), P(x), 7, P(dx))
    np.testing.assert_allclose(dx[:3], d[:3], atol=1e-15)
    np.testing.assert_allclose(dx[3:], d[3:], rtol=2e-3)  # 2 vec(dq) is first order in theta


def test_corrector(hc):
    # ceres Corrector semantics restated at marginalization_factor.cpp:46-77
    for r in ([0.3, -0.2], [2.0, 1.5], [0.0, 0.0]):
        r = np.array(r); j = np.array([0.7, -1.1])
        s = r @ r
        rho = O.huber(1.0, s)
        sr1 = np.sqrt(rho[1])
        jr = sr1 * j  # rho[2] <= 0 for Huber -> alpha = 0
        rr = sr1 * r
        r2 = r.copy(); j2 = j.copy()
        rho0 = hc.hc_correct(C.c_double(1.0), P(r2), P(j2))
        np.testing.assert_allclose([rho0], [rho[0]])
        np.testing.assert_allclose(r2, rr, atol=1e-15)
        np.testing.assert_allclose(j2, jr, atol=1e-15)


def _gram26_cols(g23):
    """23-column rows -> the 26-column view [pose_s 6 | pose_j 6 | ex0 6 | ex1 6 | td | r]."""
    return np.concatenate([g23[:, 0:6], -g23[:, 0:3], g23[:, 6:9], g23[:, 9:15], g23[:, 16:22], g23[:, 22:23], g23[:, 15:16]], axis=1)


def test_compact_rows_and_their_assembly_match_the_23_column_form(hc):
    """The compact 16-column rows of the solve passes (visual_lin.hpp: one FP64-MFMA tile per camera) and k_assemble's expansion of
    their per-slot Grams (assemble_compact.hpp: the extrinsic-translation blocks as 3 x 3 transforms of the B blocks, the 256 owner
    threads emulated on the host) against a dense accumulation of the 23-column rows: pose system (td left out: a constant block in
    this mode) and gradient of a window of several chunks (start frames), stereo and mono observations, inliers and outliers."""
    rng = np.random.default_rng(2026)
    sq, a = 460.0 / 1.5, 1.0
    F = 11
    poses = np.stack([rand_pose(rng, 0.3) for _ in range(F)])
    q0 = poses[0, 3:].copy()
    for f in range(1, F):
        poses[f, 3:] = q0 + 0.08 * rng.normal(size=4); poses[f, 3:] /= np.linalg.norm(poses[f, 3:])
    ex0 = np.array([0.1, 0.025, 0.11, 0.5, -0.5, 0.5, -0.5]); ex1 = ex0.copy(); ex1[1] = -0.025
    ex0[3:] += 0.01 * rng.normal(size=4); ex0[3:] /= np.linalg.norm(ex0[3:])
    ex1[3:] += 0.01 * rng.normal(size=4); ex1[3:] /= np.linalg.norm(ex1[3:])
    td = 0.003
    chunks = [(0, 5, 11), (0, 3, 4), (2, 6, 9), (5, 4, 6), (9, 3, 2), (10, 2, 1)]   # (start frame, landmarks, frames seen)
    H = np.zeros((80, 80)); g = np.zeros(80)
    slots = []; tab = []
    cd_of = lambda s, j: np.concatenate([6 * s + np.arange(6), 6 * j + np.arange(6), 66 + np.arange(12), [78]])
    for (s, n, km) in chunks:
        tab.append(s | (km << 8) | (len(slots) << 16))
        lams = 1.0 / rng.uniform(2, 10, size=n)
        oi = [np.array([rng.uniform(-.5, .5), rng.uniform(-.5, .5), 1.0]) for _ in range(n)]
        vi = [rng.normal(size=2) * 0.2 for _ in range(n)]
        for t in range(km):
            j = s + t
            Cg = [np.zeros((16, 16)), np.zeros((16, 16))]
            for l in range(n):
                if rng.uniform() < 0.15 and t > 0:
                    continue   # not observed in this frame
                for cam in ((1,) if t == 0 else (0, 1)):
                    if cam == 1 and rng.uniform() < 0.3:
                        continue   # mono observation
                    kind = 2 if t == 0 else cam
                    obs = np.concatenate([oi[l], [rng.uniform(-.5, .5), rng.uniform(-.5, .5), 1.0], vi[l], rng.normal(size=2) * 0.2, [0.003, 0.001]])
                    x0 = np.zeros(23); x1 = np.zeros(23); jl = np.zeros(2)
                    args = (kind, P(obs), P(poses[s]), P(poses[j]), P(ex0), P(ex1), C.c_double(lams[l]), C.c_double(td), C.c_double(sq), C.c_double(a))
                    rho = hc.hc_vis_lin(*args, P(x0), P(x1), P(jl))
                    k0 = np.zeros(16); k1 = np.zeros(16); jlc = np.zeros(2); tc = np.zeros(12)
                    hc.hc_vis_lin_c.restype = C.c_double
                    rho_c = hc.hc_vis_lin_c(*args, P(k0), P(k1), P(jlc), P(tc))
                    assert rho_c == rho
                    np.testing.assert_array_equal(jlc, jl)
                    g23 = np.stack([x0, x1]); k16 = np.stack([k0, k1]); tc = tc.reshape(4, 3)
                    # the columns both forms have are the same numbers; the translation columns of the extrinsics come back as tc
                    if kind == 2:   # no pose columns: the B columns carry reduce ric2^T = the tic column (k_assemble: identity transforms at t = 0)
                        np.testing.assert_array_equal(k16[:, 0:3], g23[:, 9:12])
                        assert np.all(k16[:, 3:9] == 0.0) and np.all(g23[:, 0:9] == 0.0)
                    else:
                        np.testing.assert_array_equal(k16[:, 0:9], g23[:, 0:9])
                    np.testing.assert_array_equal(k16[:, 9:12], g23[:, 12:15])
                    np.testing.assert_array_equal(k16[:, 12:15], g23[:, 19:22])
                    np.testing.assert_array_equal(k16[:, 15], g23[:, 15])
                    np.testing.assert_array_equal(tc[0:2], g23[:, 9:12])
                    np.testing.assert_array_equal(tc[2:4], g23[:, 16:19])
                    J26 = _gram26_cols(g23)
                    if t == 0:
                        assert np.all(J26[:, 0:12] == 0.0)
                    cd = cd_of(s, j)
                    for rrow in range(2):
                        np.add.at(H, (cd[:, None], cd[None, :]), np.outer(J26[rrow, :25], J26[rrow, :25]))
                        np.add.at(g, cd, J26[rrow, :25] * J26[rrow, 25])
                    Cg[cam] += k16.T @ k16
            G = Cg[0] + Cg[1]
            slot = np.zeros(184)
            for x in range(16):
                for y in range(x, 16):
                    slot[x * 16 - (x * (x - 1)) // 2 + (y - x)] = G[x, y]
            slot[136:] = Cg[1][0:3, :].reshape(-1)
            slots.append(slot)
    slots = np.ascontiguousarray(np.stack(slots))
    tab = np.array(tab, dtype=np.uint32)
    Hc = np.zeros((80, 80)); gc = np.zeros(80)
    hc.hc_assemble_compact(len(chunks), tab.ctypes.data_as(C.POINTER(C.c_uint32)), P(slots), P(np.ascontiguousarray(poses)), P(Hc), P(gc))
    keep = [i for i in range(80) if i not in (78, 79)]
    Hl = np.tril(H)[np.ix_(keep, keep)]
    scale = np.abs(Hl).max()
    np.testing.assert_allclose(Hc[np.ix_(keep, keep)], Hl, rtol=0, atol=2e-13 * scale)
    assert np.all(np.triu(Hc, 1) == 0.0)
    np.testing.assert_allclose(gc[keep], g[keep], rtol=0, atol=2e-13 * np.abs(g).max())
    assert np.count_nonzero(Hl) > 1500
    # the full batch's pose assembly runs the same classes pass by pass (<= 6 frames through a six-slot stage, straight-line bodies with
    # clamped reads and +0.0 for the frames beyond a pass: kernels_asm_full.hip): same system, sums over a chunk's frames reach their target
    # once per pass
    Hp = np.zeros((80, 80)); gp = np.zeros(80)
    hc.hc_assemble_compact_passes(len(chunks), tab.ctypes.data_as(C.POINTER(C.c_uint32)), P(slots), P(np.ascontiguousarray(poses)), P(Hp), P(gp))
    np.testing.assert_allclose(Hp[np.ix_(keep, keep)], Hl, rtol=0, atol=2e-13 * scale)
    assert np.all(np.triu(Hp, 1) == 0.0)
    np.testing.assert_allclose(gp[keep], g[keep], rtol=0, atol=2e-13 * np.abs(g).max())
    np.testing.assert_allclose(Hp, Hc, rtol=0, atol=1e-14 * scale)


def test_lane_parallel_preintegration_blocks_match_the_blocks_written_out(hc):
    """cerberus_amd/csrc/preint_blocks.hpp (table-driven, one lane per 3 x 3 entry: what k_preint_imu_leg / k_repropagate run per sample)
    against dF = F - I and V of IMULegIntegrationBase::midPointIntegration written block by block with 3 x 3 temporaries
    (imu_leg_integration_base.cpp:376-465). Same products, sums of scaled terms instead of scaled sums: equal to a few ulp."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        inp = np.zeros(37 + 8 * 27)
        for o in (0, 9):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            inp[o:o + 9] = O.quat_to_rot(q).ravel() if hasattr(O, "quat_to_rot") else np.linalg.qr(rng.normal(size=(3, 3)))[0].ravel()
        inp[18:27] = rng.normal(size=9) * [1, 1, 1, 9, 9, 9, 9, 9, 9]
        inp[27:36] = np.linalg.qr(rng.normal(size=(3, 3)))[0].ravel()
        inp[36] = 0.0025 * (1 + 0.1 * rng.random())
        inp[37:] = rng.normal(size=8 * 27)
        out = []
        for mode in (0, 1):
            dF, V = np.zeros(32 * 31), np.zeros(32 * 48)
            hc.hc_preint_blocks(mode, P(inp), P(dF), P(V))
            out.append((dF.reshape(32, 31), V.reshape(32, 48)))
        for a, b, name in ((out[1][0], out[0][0], "dF"), (out[1][1], out[0][1], "V")):
            assert (a != 0).sum() == (b != 0).sum() and ((a != 0) == (b != 0)).all(), name   # the same sparsity pattern
            scale = np.abs(b).max()
            assert np.abs(a - b).max() <= 4e-16 * max(1.0, scale) + 1e-15 * np.abs(b).max(), (name, np.abs(a - b).max())
        assert np.count_nonzero(out[0][0]) >= 140 and np.count_nonzero(out[0][1]) >= 300, (np.count_nonzero(out[0][0]), np.count_nonzero(out[0][1]))
        # k-steps (4 noise columns) of V N V^T whose products with one half of the rows kernels_preint.hip does not issue: that half of V
        # must be structurally zero there — in the blocks as the reference writes them and in the lane-parallel ones — and every other
        # k-step must have something in both halves (else the masks leave work on the table without saying so)
        lo, hi = C.c_uint(0), C.c_uint(0)
        hc.hc_v_kstep_masks(C.byref(lo), C.byref(hi))
        assert lo.value & hi.value == 0
        for _, V in out:
            for kk in range(12):
                top, bot = V[:16, 4 * kk:4 * kk + 4], V[16:, 4 * kk:4 * kk + 4]
                if (lo.value >> kk) & 1:
                    assert not bot.any() and top.any(), kk
                elif (hi.value >> kk) & 1:
                    assert not top.any() and bot.any(), kk
                else:
                    assert top.any() and bot.any(), kk
        dlo, cols = C.c_uint(0), (C.c_int * 16)()
        hc.hc_df_kstep_mask(C.byref(dlo), cols)
        assert sorted(cols) == [3, 4, 5, 6, 7, 8] + list(range(21, 31))
        for dF, _ in out:
            for kk in range(4):
                bot = dF[16:, [cols[4 * kk + u] for u in range(4)]]
                assert bot.any() != bool((dlo.value >> kk) & 1), kk   # masked k-steps: nothing in rows 16 .. 31; the others: something
