"""Bitwise A/B of sqrt_info between two builds of libvilo_gpu (run on the GPU box): residuals and Jacobians of vilo_eval_imu_leg
(= sqrt_info times the raw factor) over an odd number of preintegration records must be equal bit for bit.
usage: python tools/micro/prepare_ab.py cerberus_amd/lib/libvilo_gpu_r4.so [n_windows]
(a subprocess per library: the library is chosen when cerberus_amd.api is first used)"""
import os, subprocess, sys, hashlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def run(nw):
    from cerberus_amd import api, synth
    from oracle import oracle_py as O
    cfg = synth.default_config()
    ocfg = O.config_from(cfg)
    ctx = api.Context(cfg)
    pre, P = [], [[] for _ in range(6)]
    for s in range(nw):
        w = synth.make_window(cfg, n_landmarks=8, seed=100 + s)
        O.fill_preint(ocfg, w)
        pre.append(w.preint)
        for k, a in enumerate([w.pose[:-1], w.speed_bias[:-1], w.leg_bias[:-1], w.pose[1:], w.speed_bias[1:], w.leg_bias[1:]]):
            P[k].append(a)
    pre = np.concatenate(pre)[:-1]   # odd count: the last wave has one record
    P = [np.concatenate(p)[:-1] for p in P]
    r, Js = ctx.eval_imu_leg(pre, P)
    h = hashlib.sha256(r.tobytes())
    for J in Js:
        h.update(J.tobytes())
    print("RESULT", pre.shape[0], h.hexdigest(), float(np.abs(r).max()))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run(int(sys.argv[2]))
        sys.exit(0)
    other = sys.argv[1]
    nw = sys.argv[2] if len(sys.argv) > 2 else "13"
    outs = []
    for libpath in (None, other):
        env = dict(os.environ)
        if libpath:
            env["VILO_GPU_LIB"] = os.path.join(ROOT, libpath)
        else:
            env.pop("VILO_GPU_LIB", None)
        o = subprocess.run([sys.executable, __file__, "--child", nw], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
        print(libpath or "default", line[-1] if line else ("FAILED\n" + o.stderr[-2000:]))
        outs.append(line[-1] if line else None)
    print("BITWISE EQUAL" if outs[0] and outs[0] == outs[1] else "DIFFERENT")
