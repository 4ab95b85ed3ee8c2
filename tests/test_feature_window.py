"""Host-side feature window (cerberus_amd/host/vilo_feature_window.*) against the reference's own
FeatureManager (src/featureTracker/feature_manager.cpp compiled unmodified into oracle/_ref/libref.so): the same random
track history goes through both, every operation is followed by a full dump comparison."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "libref.so")
HOST = os.path.join(ROOT, "cerberus_amd", "lib", "libvilo_host.so")

dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)


class FW:
    def __init__(self, lib, prefix):
        self.L, self.p = lib, prefix
        self.f("create").restype = C.c_void_p
        self.h = C.c_void_p(self.f("create")())

    def f(self, name):
        return getattr(self.L, self.p + name)

    def call(self, name, *args):
        fn = self.f(name)
        fn.restype = C.c_int
        return fn(self.h, *args)

    def add_frame(self, fc, ids, obs, stereo, td):
        c3 = (C.c_int * 3)()
        ids = np.ascontiguousarray(ids, np.int32); obs = np.ascontiguousarray(obs, np.float64); stereo = np.ascontiguousarray(stereo, np.uint8)
        kf = self.call("add_frame", C.c_int(fc), C.c_int(len(ids)), ids.ctypes.data_as(ip), obs.ctypes.data_as(dp), stereo.ctypes.data_as(up), C.c_double(td), c3)
        return kf, list(c3)

    def dump(self):
        tot = C.c_int()
        n = self.call("dump", None, None, None, None, C.byref(tot))
        info = np.zeros((max(n, 1), 4), np.int32); depth = np.zeros(max(n, 1)); obs = np.zeros((max(tot.value, 1), 11)); st = np.zeros(max(tot.value, 1), np.uint8)
        self.call("dump", info.ctypes.data_as(ip), depth.ctypes.data_as(dp), obs.ctypes.data_as(dp), st.ctypes.data_as(up), C.byref(tot))
        return info[:n], depth[:n], obs[:tot.value], st[:tot.value]


def _same(a, b, depth_tol=1e-9):
    ia, da, oa, sa = a.dump(); ib, db, ob, sb = b.dump()
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(oa, ob)
    np.testing.assert_array_equal(sa, sb)
    np.testing.assert_allclose(da, db, rtol=depth_tol, atol=0)


def obs_checksum(obs, stereo):
    """order-sensitive checksum of a dump's observation rows (exact: plain float64 arithmetic on identical inputs)"""
    w = np.arange(1, len(obs) + 1, dtype=np.float64)
    return np.array([float(len(obs)), float((obs * w[:, None]).sum()), float((stereo * w).sum())])


def _rot(rng, s):
    w = s * rng.normal(size=3); th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def drive(fws, seed, check, steps=40):
    """The same random track history through every feature window of `fws`; check(kind) after each operation. Returns how often
    each slide branch / the depth path was taken."""
    rng = np.random.default_rng(seed)
    W = 10
    alive, next_id, frame_count = {}, 0, 0
    branches = {"shift": 0, "back": 0, "front": 0, "triangulated": 0}
    tic = np.array([[0.2, 0.03, 0.05], [0.2, -0.02, 0.05]]); ric = np.stack([_rot(rng, 0.02), _rot(rng, 0.02)])

    def all_call(name, *args):
        return [f.call(name, *args) for f in fws]
    for step in range(steps):
        # tracks: each alive feature survives with p = 0.85, new ones are born; points drift slowly
        alive = {i: p + 0.01 * rng.normal(size=3) * [1, 1, 0] for i, p in alive.items() if rng.random() < 0.85}
        n_new = int(rng.integers(5, 25)) if len(alive) < 90 else 0
        for _ in range(n_new):
            alive[next_id] = np.array([0.5 * rng.normal(), 0.4 * rng.normal(), 1.0]); next_id += 1
        ids = np.array(sorted(alive), np.int32)
        rng.shuffle(ids)   # arrival order must not matter (the reference walks a map sorted by id)
        obs = np.zeros((len(ids), 11)); stereo = (rng.random(len(ids)) < 0.8).astype(np.uint8)
        for r, i in enumerate(ids):
            obs[r, :3] = alive[i]; obs[r, 3:6] = alive[i] + [-0.02, 0.001, 0.0]; obs[r, 6:8] = 0.05 * rng.normal(size=2); obs[r, 8:10] = 0.05 * rng.normal(size=2)
        td = 0.002
        res = [f.add_frame(frame_count, ids, obs, stereo, td) for f in fws]
        assert all(r == res[0] for r in res)
        check("add_frame")
        # poses of the window
        Ps = np.cumsum(0.05 * rng.normal(size=(W + 1, 3)), axis=0); Rs = np.stack([_rot(rng, 0.1) for _ in range(W + 1)])
        args = [np.ascontiguousarray(x).ctypes.data_as(dp) for x in (Ps, Rs, tic, ric)]
        all_call("triangulate", *args)
        check("triangulate")
        counts = all_call("feature_count")
        assert all(c == counts[0] for c in counts)
        n = counts[0]
        if frame_count < W:
            frame_count += 1
            continue
        # what Estimator::optimization does around the solve: depth vector out, (perturbed) depths back, failures removed
        if n:
            d0 = np.zeros(n)
            fws[0].call("depth_vector", d0.ctypes.data_as(dp))
            for f in fws[1:]:
                d1 = np.zeros(n)
                f.call("depth_vector", d1.ctypes.data_as(dp))
                np.testing.assert_allclose(d1, d0, rtol=1e-8)
            branches["triangulated"] += int((d0 > 0).sum())
            x = d0 * (1 + 0.05 * rng.normal(size=n)); x[rng.random(n) < 0.05] *= -1.0
            all_call("set_depth", x.ctypes.data_as(dp))
            check("set_depth")
            all_call("remove_failures")
            check("remove_failures")
        if step % 7 == 3:
            out = np.ascontiguousarray(rng.choice(ids, size=min(3, len(ids)), replace=False), np.int32)
            all_call("remove_outlier", out.ctypes.data_as(ip), C.c_int(len(out)))
            check("remove_outlier")
        # slide: the keyframe decision of add_frame picks the branch (estimator.cpp:1460 / feature_manager.cpp:433-509)
        if res[0][0]:
            if step % 2:
                m = [np.ascontiguousarray(v).ctypes.data_as(dp) for v in (Rs[0] @ ric[0], Ps[0] + Rs[0] @ tic[0], Rs[1] @ ric[0], Ps[1] + Rs[1] @ tic[0])]
                all_call("remove_back_shift_depth", *m); branches["shift"] += 1
            else:
                all_call("remove_back"); branches["back"] += 1
        else:
            all_call("remove_front", C.c_int(frame_count)); branches["front"] += 1
        check("slide")
        if step == 25:
            all_call("clear_depth")
            check("clear_depth")
    return branches


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HOST)), reason="needs libref.so (reference tree) and libvilo_host.so")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_feature_window_matches_reference_feature_manager(seed):
    ours, ref = FW(C.CDLL(HOST), "vilo_fw_"), FW(C.CDLL(REF), "ref_fm_")
    branches = drive([ours, ref], seed, lambda kind: _same(ours, ref, depth_tol=1e-8 if kind in ("triangulate", "slide") else 1e-9))
    assert min(branches.values()) > 0, branches   # every slide branch and the depth path were exercised


GOLDEN = os.path.join(ROOT, "tests", "golden", "feature_manager_history.npz")


@pytest.mark.skipif(not os.path.exists(HOST), reason="needs libvilo_host.so")
def test_feature_window_against_the_frozen_reference_history():
    """tests/golden/feature_manager_history.npz: the dumps of the reference's FeatureManager after every operation of the seed-0
    history (tests/golden/make_golden_rank3.py), so the pin also holds where /root/reference is absent."""
    g = np.load(GOLDEN)
    ours = FW(C.CDLL(HOST), "vilo_fw_")
    k = [0]

    def check(kind):
        i = k[0]
        info, depth, obs, st = ours.dump()
        a, b = int(g["track_off"][i]), int(g["track_off"][i + 1])
        assert str(g["kinds"][i]) == kind
        np.testing.assert_array_equal(info, g["info"][a:b])
        np.testing.assert_array_equal(obs_checksum(obs, st), g["obs_sum"][i])
        np.testing.assert_allclose(depth, g["depth"][a:b], rtol=1e-8)
        k[0] += 1
    drive([ours], 0, check)
    assert k[0] == len(g["kinds"])
