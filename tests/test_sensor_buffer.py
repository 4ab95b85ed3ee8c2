"""Sensor-interval extraction (cerberus_amd/host/vilo_sensor_buffer.*): which messages an image interval
holds, which stay queued and the dt each one is integrated with — getIMUAndLegInterval / processMeasurements
(estimator.cpp:349-397, 400-521) restated — and the whole path stamps -> processIMULeg / processImage on the device."""
import numpy as np
import pytest

H = 1.0 / 500.0


def _stamps(frame, t_prev):
    """time stamps of the samples of one Stream frame: t_prev + i*h, the last one exactly at the image"""
    n = len(frame["samples"])
    return [t_prev + (i + 1) * H for i in range(n - 1)] + [frame["header"]]


def test_interval_extraction_and_dt(cfg):
    from cerberus_amd import sequence
    stream = sequence.Stream(cfg, seed=5)
    sw = sequence.SlidingWindow(None, cfg)          # the first ten images need no device
    tic, ric, _ = stream.extrinsics()
    sw.set_extrinsics(tic, ric, 0.0)
    mp = sequence.MeasurementProcessor(sw)
    f0 = stream.next()
    assert mp.input_feature(f0["header"], f0["ids"], f0["obs"], f0["stereo"]) == 0      # "wait for imu and leg": no message yet
    mp.input_sample(f0["header"], f0["samples"][0])
    assert mp.process() == 1
    # first image: initFirstIMUPose from the averaged accelerometer (estimator.cpp:524-544), frame_count -> 1
    st = sw.state()
    assert st["frame_count"] == 1
    g = f0["samples"][0][1:4]
    np.testing.assert_allclose(st["Rs"][0] @ g / np.linalg.norm(g), [0, 0, 1], atol=1e-12)     # measured gravity along +z
    assert abs(np.arctan2(st["Rs"][0][1, 0], st["Rs"][0][0, 0])) < 1e-12                       # yaw removed (Utility::g2R)
    assert mp.queue_size() == 1                                                                # the message at the image stays queued
    t_prev, late_prev = f0["header"], 0.0
    for k in range(1, 7):
        f = stream.next()
        ts = _stamps(f, t_prev)
        for t, s in zip(ts[:-1], f["samples"][:-1]):
            mp.input_sample(t, s)
        # the image arrives before the message that closes its interval: nothing happens
        assert mp.input_feature(f["header"], f["ids"], f["obs"], f["stereo"]) == 0 and sw.state()["frame_count"] == k
        # the first message stamped at or after the image closes the interval, whatever its own stamp
        late = 0.0 if k % 2 else 0.4 * H
        mp.input_sample(f["header"] + late, f["samples"][-1])
        assert mp.process() == 1 and sw.state()["frame_count"] == k + 1
        got = mp.last_interval()
        want = f["samples"]
        if late_prev:
            # the closing message of the previous image was stamped after that image: it is not dropped (stamp > t0), so it
            # opens this interval too, integrated over stamp - t0, and shortens the next message's dt
            assert len(got) == len(want) + 1
            np.testing.assert_allclose(got[0, 0], late_prev, rtol=1e-9)
            np.testing.assert_allclose(got[1, 0], H - late_prev, rtol=1e-9)
            got = got[1:]
            np.testing.assert_allclose(got[1:, 0], want[1:, 0], rtol=1e-9, atol=1e-15)
        else:
            assert len(got) == len(want)
            np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=1e-9, atol=1e-15)              # dt rule of estimator.cpp:456-462
        np.testing.assert_array_equal(got[:, 1:], want[:, 1:])                                    # measurements pass through
        assert mp.queue_size() == 1
        t_prev, late_prev = f["header"], late
    # two images queued behind missing messages are both processed once the messages arrive
    fa, fb = stream.next(), stream.next()
    assert mp.input_feature(fa["header"], fa["ids"], fa["obs"], fa["stereo"]) == 0
    assert mp.input_feature(fb["header"], fb["ids"], fb["obs"], fb["stereo"]) == 0
    for f, tp in ((fa, t_prev), (fb, fa["header"])):
        for t, s in zip(_stamps(f, tp), f["samples"]):
            mp.input_sample(t, s)
    b0 = mp.busy_ms()
    assert mp.process() == 2 and sw.state()["frame_count"] == 9
    # the processor keeps the account of the wall time spent inside its entry points (bench.py's replay figure)
    assert 0.0 < b0 < mp.busy_ms() < 1e4


@pytest.mark.gpu
def test_stamped_messages_give_the_same_estimate_as_direct_feeding(cfg):
    from cerberus_amd import api, sequence
    ctx = api.Context(cfg, 0)
    N = 18

    def run(stamped):
        stream = sequence.Stream(cfg, seed=21)
        sw = sequence.SlidingWindow(ctx, cfg)
        sw.set_extrinsics(*stream.extrinsics())
        mp = sequence.MeasurementProcessor(sw) if stamped else None
        t_prev = 0.0
        for k in range(N):
            f = stream.next()
            if k == 0:
                t = f["truth"]
                sw.init_first_pose(t[0:3], sequence.quat_to_R(t[3:7]).ravel(), t[7:10])
            if stamped:
                for ts, s in zip([f["header"]] if k == 0 else _stamps(f, t_prev), f["samples"]):
                    mp.input_sample(ts, s)
                assert mp.input_feature(f["header"], f["ids"], f["obs"], f["stereo"]) == 1
            else:
                sw.process_samples(f["samples"])
                sw.process_image(f["header"], f["ids"], f["obs"], f["stereo"])
            t_prev = f["header"]
        return sw.state()
    a, b = run(True), run(False)
    assert a["n_optimizations"] == b["n_optimizations"] == N - 10
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "Rho"):
        np.testing.assert_allclose(a[key], b[key], rtol=0, atol=1e-7)    # dt from stamp differences vs the generator's dt: 1e-17 apart
    ctx.close()
