"""Freezes the gauge fix of Estimator::double2vector (estimator.cpp:905-957) at and around its Euler-singularity branch (:925-934) into
gauge_fix_edges.npz. estimator.cpp itself cannot be compiled here (ROS, OpenCV); oracle/ref_build/ref_driver.cpp::ref_gauge_fix writes its
statements down around the reference's OWN Utility::R2ypr / ypr2R (utils/utility.h:83-125, compiled from /root/reference) and the
quaternion <-> matrix conversions of the build's Eigen stand-in. Cases = pitch of frame 0 before / after the solve, in degrees:
the branch is taken if either is within one degree of +-90. Inputs and outputs are both stored (small).
Run where /root/reference exists:   python tests/golden/make_golden_gauge.py"""
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation as Rot

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# (pitch before, pitch after): both singular, one of them, neither (just outside), exactly 90, regular
CASES = [(89.5, 89.5), (-89.5, -89.5), (90.0, 90.0), (-90.0, -90.0), (88.9, 89.2), (89.2, 88.9), (88.9, 88.9), (-88.9, -89.2), (89.01, 89.01), (88.99, 88.99),
         (0.0, 0.0), (35.0, -20.0), (89.999, 90.0)]


def rotate_window(arrs, Rg):
    """every frame's pose / velocity by the same world rotation"""
    pose, sb = arrs[0].copy(), arrs[1].copy()
    for i in range(pose.shape[0]):
        pose[i, :3] = Rg @ pose[i, :3]
        pose[i, 3:] = (Rot.from_matrix(Rg) * Rot.from_quat(pose[i, 3:])).as_quat()
        sb[i, :3] = Rg @ sb[i, :3]
    return pose, sb


def make_case(O, w, p_before, p_after, seed):
    rng = np.random.default_rng(seed)
    arrs = w.clone_state()
    R0 = Rot.from_quat(arrs[0][0, 3:]).as_matrix()
    yaw, roll = rng.uniform(-170, 170), rng.uniform(-40, 40)
    pb, sbb = rotate_window(arrs, O.ypr2R([yaw, p_before, roll]) @ R0.T)
    # "after": what a solve leaves — the same window drifted along the gauge (yaw + translation) and moved a little everywhere,
    # its frame 0 at the stated pitch
    pa, sba = rotate_window(arrs, O.ypr2R([yaw + rng.uniform(-3, 3), p_after, roll + rng.uniform(-0.5, 0.5)]) @ R0.T)
    pa[:, :3] += rng.normal(size=3) * 0.05
    for i in range(1, pa.shape[0]):
        pa[i] = O.pose_plus(pa[i], 1e-3 * rng.normal(size=6))
    pa[:, 3:] *= 1.0 + 1e-9 * rng.normal(size=(pa.shape[0], 1))     # para_Pose quaternions come back from the solver unnormalised
    sba = sba + 1e-3 * rng.normal(size=sba.shape)
    ex = arrs[3].copy()
    ex[:, 3:] *= 1.0 + 1e-9 * rng.normal(size=(2, 1))
    return pb, sbb, pa, sba, ex


def run_fix(fn, w, before_pose, before_sb, pa, sba, ex):
    """fn(before_arrays, w) fixes w's states in place (oracle_py.gauge_fix / Context.gauge_fix signature)"""
    st = w.clone_state()
    before = [before_pose, before_sb] + [a.copy() for a in st[2:]]
    w.set_state([pa.copy(), sba.copy(), st[2].copy(), ex.copy(), st[4].copy(), st[5].copy()])
    fn(before, w)
    out = w.clone_state()
    w.set_state(st)
    return out[0], out[1], out[3]


def main():
    import ctypes as C
    from cerberus_amd import synth
    from oracle import oracle_py as O
    from oracle import ref_py as R
    w = synth.make_window(synth.default_config(), n_landmarks=24, seed=5)

    def ref_fix(before_arrays, win):
        sb = O.WindowState()
        keep = [np.ascontiguousarray(a) for a in before_arrays]
        sb.pose, sb.speed_bias, sb.leg_bias, sb.ex_pose, sb.td, sb.inv_depth = [k.ctypes.data_as(O.dp) for k in keep]
        _, sa = win.desc(O)
        R.ref_lib().ref_gauge_fix(C.byref(sb), C.byref(sa), C.c_int(win.F))
    out = dict(cases=np.array(CASES))
    taken = []
    for n, (p0, p1) in enumerate(CASES):
        pb, sbb, pa, sba, ex = make_case(O, w, p0, p1, 100 + n)
        fp, fs, fe = run_fix(ref_fix, w, pb, sbb, pa, sba, ex)
        with R.as_oracle():
            y0, y1 = O.R2ypr(Rot.from_quat(pb[0, 3:]).as_matrix()), O.R2ypr(Rot.from_quat(pa[0, 3:]).as_matrix())
        taken.append(abs(abs(y0[1]) - 90) < 1.0 or abs(abs(y1[1]) - 90) < 1.0)
        for name, a in (("before_pose", pb), ("before_sb", sbb), ("after_pose", pa), ("after_sb", sba), ("after_ex", ex), ("fixed_pose", fp),
                        ("fixed_sb", fs), ("fixed_ex", fe)):
            out["%s_%d" % (name, n)] = a
    out["branch_taken"] = np.array(taken)
    np.savez_compressed(os.path.join(HERE, "gauge_fix_edges.npz"), **out)
    print("wrote gauge_fix_edges.npz; singular branch taken:", list(zip(CASES, taken)))


if __name__ == "__main__":
    main()
