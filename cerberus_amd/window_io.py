"""Reader / writer of the one-window dump format of include/vilo_window_io.h: what
Estimator::optimization() reads (and optionally its result), so windows dumped from the reference's real Ceres stack can
be replayed here (tools/replay_window.py) and synthetic windows can be shipped to such a machine."""
import ctypes as C
import struct

import numpy as np

from . import _ctypes as T
from . import synth

MAGIC = b"VILOWIN1"
NB = T.MAX_PRIOR_BLOCKS


def save(path, cfg, w, after=None, ref_summary=None, marginalization_flag=0):
    """w: synth.Window (states = before). after: optional list of the six state arrays (the reference's result)."""
    F, L = w.F, w.L
    has_prior = 1 if (w.prior is not None and w.prior.struct.valid) else 0
    hdr = np.array([1, F, L, w.n_obs, w.use_leg, w.leg_bias_const, w.ex_const, w.td_const, has_prior, 1 if after is not None else 0,
                    marginalization_flag, C.sizeof(cfg), 0, 0, 0, 0], np.int32)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(hdr.tobytes())
        f.write(bytes(cfg))
        for a in w.state_arrays():
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
        f.write(np.ascontiguousarray(w.lm_start_frame, np.int32).tobytes())
        f.write(np.ascontiguousarray(w.lm_obs_offset, np.int32).tobytes())
        f.write(np.ascontiguousarray(w.obs, np.float64).tobytes())
        f.write(np.ascontiguousarray(w.obs_is_stereo, np.uint8).tobytes())
        f.write(b"\0" * ((8 - (w.n_obs & 7)) & 7))
        f.write(np.ascontiguousarray(w.preint if w.use_leg else w.preint_imu, np.float64).tobytes())
        if has_prior:
            p = w.prior.struct
            ph = np.zeros(2 + 3 * NB + 3, np.int32)
            ph[0], ph[1] = p.n, p.n_blocks
            ph[2:2 + NB] = list(p.block_id); ph[2 + NB:2 + 2 * NB] = list(p.block_size); ph[2 + 2 * NB:2 + 3 * NB] = list(p.block_idx)
            ph[2 + 3 * NB] = p.valid
            xs = sum(p.block_size[k] for k in range(p.n_blocks))
            f.write(ph.tobytes())
            f.write(w.prior.x0[:xs].tobytes()); f.write(w.prior.J0[:p.n * p.n].tobytes()); f.write(w.prior.r0[:p.n].tobytes())
        if after is not None:
            for a in after:
                f.write(np.ascontiguousarray(a, np.float64).tobytes())
            f.write(np.asarray(ref_summary if ref_summary is not None else [0, 0, 0, 0], np.float64).tobytes())


def load(path):
    """Returns (cfg, window, after_or_None, ref_summary_or_None, marginalization_flag)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise ValueError("not a VILOWIN1 file")
    hdr = np.frombuffer(buf, np.int32, 16, 8)
    if hdr[0] != 1:
        raise ValueError("unsupported version %d" % hdr[0])
    F, L, n_obs, use_leg = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
    cfg = T.Config()
    if hdr[11] != C.sizeof(cfg):
        raise ValueError("vilo_config size mismatch")
    if F != T.F:
        raise ValueError("only full windows (n_frames = %d) are supported by the Python container" % T.F)
    off = 8 + 64
    C.memmove(C.byref(cfg), buf[off:off + C.sizeof(cfg)], C.sizeof(cfg)); off += C.sizeof(cfg)
    w = synth.Window(L, n_obs, 0)

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype, n, off).copy()
        off += a.nbytes
        return a

    def take_state():
        return [take(np.float64, 7 * F).reshape(F, 7), take(np.float64, 9 * F).reshape(F, 9), take(np.float64, 4 * F).reshape(F, 4),
                take(np.float64, 14).reshape(2, 7), take(np.float64, 1), take(np.float64, L)]

    w.set_state(take_state())
    w.lm_start_frame[:] = take(np.int32, L); w.lm_obs_offset[:] = take(np.int32, L + 1)
    w.obs[:] = take(np.float64, 11 * n_obs).reshape(n_obs, 11); w.obs_is_stereo[:] = take(np.uint8, n_obs)
    off += (8 - (n_obs & 7)) & 7
    w.use_leg = use_leg
    w.leg_bias_const, w.ex_const, w.td_const = int(hdr[5]), int(hdr[6]), int(hdr[7])
    if use_leg:
        w.preint[:] = take(np.float64, (F - 1) * T.PREINT_DOUBLES).reshape(F - 1, -1)
    else:
        w.preint_imu[:] = take(np.float64, (F - 1) * T.PREINT_IMU_DOUBLES).reshape(F - 1, -1)
    if hdr[8]:
        ph = take(np.int32, 2 + 3 * NB + 3)
        p = w.prior.struct
        p.n, p.n_blocks = int(ph[0]), int(ph[1])
        for k in range(NB):
            p.block_id[k], p.block_size[k], p.block_idx[k] = int(ph[2 + k]), int(ph[2 + NB + k]), int(ph[2 + 2 * NB + k])
        p.valid = int(ph[2 + 3 * NB])
        xs = sum(p.block_size[k] for k in range(p.n_blocks))
        w.prior.x0[:xs] = take(np.float64, xs); w.prior.J0[:p.n * p.n] = take(np.float64, p.n * p.n); w.prior.r0[:p.n] = take(np.float64, p.n)
        w.prior.rebind()
    else:
        w.prior.struct.valid = 0
    after = ref = None
    if hdr[9]:
        after = take_state()
        ref = take(np.float64, 4)
    return cfg, w, after, ref, int(hdr[10])
