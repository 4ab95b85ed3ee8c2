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
pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HOST)), reason="needs libref.so (reference tree) and libvilo_host.so")

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


def _rot(rng, s):
    w = s * rng.normal(size=3); th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_feature_window_matches_reference_feature_manager(seed):
    rng = np.random.default_rng(seed)
    ours, ref = FW(C.CDLL(HOST), "vilo_fw_"), FW(C.CDLL(REF), "ref_fm_")
    W = 10
    alive, next_id, frame_count = {}, 0, 0
    branches = {"shift": 0, "back": 0, "front": 0, "triangulated": 0}
    tic = np.array([[0.2, 0.03, 0.05], [0.2, -0.02, 0.05]]); ric = np.stack([_rot(rng, 0.02), _rot(rng, 0.02)])
    for step in range(40):
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
        ra, rb = ours.add_frame(frame_count, ids, obs, stereo, td), ref.add_frame(frame_count, ids, obs, stereo, td)
        assert ra == rb
        _same(ours, ref)
        # poses of the window
        Ps = np.cumsum(0.05 * rng.normal(size=(W + 1, 3)), axis=0); Rs = np.stack([_rot(rng, 0.1) for _ in range(W + 1)])
        args = [np.ascontiguousarray(x).ctypes.data_as(dp) for x in (Ps, Rs, tic, ric)]
        keep = [Ps, Rs]   # noqa: F841 (buffers must outlive the calls)
        ours.call("triangulate", *args); ref.call("triangulate", *args)
        _same(ours, ref, depth_tol=1e-8)
        assert ours.call("feature_count") == ref.call("feature_count")
        n = ours.call("feature_count")
        if frame_count < W:
            frame_count += 1
            continue
        # what Estimator::optimization does around the solve: depth vector out, (perturbed) depths back, failures removed
        if n:
            da, db = np.zeros(n), np.zeros(n)
            ours.call("depth_vector", da.ctypes.data_as(dp)); ref.call("depth_vector", db.ctypes.data_as(dp))
            np.testing.assert_allclose(da, db, rtol=1e-8)
            branches["triangulated"] += int((db > 0).sum())
            x = db * (1 + 0.05 * rng.normal(size=n)); x[rng.random(n) < 0.05] *= -1.0
            ours.call("set_depth", x.ctypes.data_as(dp)); ref.call("set_depth", x.ctypes.data_as(dp))
            _same(ours, ref)
            ours.call("remove_failures"); ref.call("remove_failures")
            _same(ours, ref)
        if step % 7 == 3:
            out = np.ascontiguousarray(rng.choice(ids, size=min(3, len(ids)), replace=False), np.int32)
            ours.call("remove_outlier", out.ctypes.data_as(ip), C.c_int(len(out))); ref.call("remove_outlier", out.ctypes.data_as(ip), C.c_int(len(out)))
            _same(ours, ref)
        # slide: the keyframe decision of add_frame picks the branch (estimator.cpp:1460 / feature_manager.cpp:433-509)
        if ra[0]:
            if step % 2:
                m = [np.ascontiguousarray(v).ctypes.data_as(dp) for v in (Rs[0] @ ric[0], Ps[0] + Rs[0] @ tic[0], Rs[1] @ ric[0], Ps[1] + Rs[1] @ tic[0])]
                ours.call("remove_back_shift_depth", *m); ref.call("remove_back_shift_depth", *m); branches["shift"] += 1
            else:
                ours.call("remove_back"); ref.call("remove_back"); branches["back"] += 1
        else:
            ours.call("remove_front", C.c_int(frame_count)); ref.call("remove_front", C.c_int(frame_count)); branches["front"] += 1
        _same(ours, ref, depth_tol=1e-8)
        if step == 25:
            ours.call("clear_depth"); ref.call("clear_depth")
            _same(ours, ref)
    assert min(branches.values()) > 0, branches   # every slide branch and the depth path were exercised
