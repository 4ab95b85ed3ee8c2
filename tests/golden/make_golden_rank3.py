"""Freezes reference outputs for the SURVEY 8(f) rows added after reference_vectors.npz was made (that file is left alone so its
random stream stays what the existing tests expect):
  * feature_manager_history.npz: the reference's FeatureManager (oracle/_ref/libref.so, ref_fm_*) dumped after every operation of
    the seed-0 history of tests/test_feature_window.py::drive
  * preint_force_model.npz: IMULegIntegrationBase with contact_sensor_type 2 on the ten intervals of the golden window with
    force-valued contact inputs (tests/test_oracle_vs_reference.py::force_samples)
Run where /root/reference exists:   python tests/golden/make_golden_rank3.py"""
import copy
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_feature_window as tfw  # noqa: E402
from test_oracle_vs_reference import force_samples  # noqa: E402
from cerberus_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from oracle import ref_py as R  # noqa: E402


def main():
    ref = tfw.FW(C.CDLL(tfw.REF), "ref_fm_")
    rec = dict(kinds=[], info=[], depth=[], obs_sum=[], track_off=[0])

    def check(kind):
        info, depth, obs, st = ref.dump()
        # observations pass through the manager unchanged (only regrouped): a position-weighted checksum per dump stands for them
        rec["kinds"].append(kind); rec["info"].append(info.copy()); rec["depth"].append(depth.copy()); rec["obs_sum"].append(tfw.obs_checksum(obs, st))
        rec["track_off"].append(rec["track_off"][-1] + len(info))
    tfw.drive([ref], 0, check)
    np.savez_compressed(os.path.join(HERE, "feature_manager_history.npz"), kinds=np.array(rec["kinds"]), info=np.concatenate(rec["info"]),
                        depth=np.concatenate(rec["depth"]), obs_sum=np.array(rec["obs_sum"]), track_off=np.array(rec["track_off"]))
    cfg2 = copy.copy(O.default_config())
    cfg2.contact_sensor_type = 2
    w = synth.make_window(synth.default_config(), n_landmarks=24, seed=5)
    smp = force_samples(w.samples, seed=99)
    with R.as_oracle():
        pre = np.array([O.preintegrate_imu_leg(cfg2, smp[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
    np.savez_compressed(os.path.join(HERE, "preint_force_model.npz"), preint=pre, force_seed=np.array(99))
    print("wrote feature_manager_history.npz (%d dumps), preint_force_model.npz" % len(rec["kinds"]))


if __name__ == "__main__":
    main()
