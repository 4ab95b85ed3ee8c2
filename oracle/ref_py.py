"""TEST INFRASTRUCTURE. ctypes access to oracle/_ref/libref.so — the reference's OWN factor sources compiled from
/root/reference against the header shim in oracle/ref_build/shim (see oracle/ref_build/README.md). The entry points
take the oracle's PODs, so `with as_oracle():` lets every helper of oracle_py run against the compiled reference."""
import contextlib
import ctypes as C
import os

import numpy as np

from . import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "libref.so")
_ref = None


def available():
    return os.path.exists(PATH)


def ref_lib():
    global _ref
    if _ref is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref.so missing: run `make -C oracle/ref_build` where /root/reference exists")
        L = C.CDLL(PATH)
        for name in ("fk", "jac", "dfk_drho", "dJ_dq", "dJ_drho"):
            getattr(L, "ref_" + name).argtypes = [O.dp, C.c_double, O.dp, O.dp]
        L.ref_marginalize.restype = C.c_int
        _ref = L
    return _ref


class _Alias:
    """orc_<name> -> ref_<name>"""

    def __init__(self, L):
        self._L = L

    def __getattr__(self, name):
        return getattr(self._L, "ref_" + name[4:]) if name.startswith("orc_") else getattr(self._L, name)


@contextlib.contextmanager
def as_oracle():
    O.lib()
    saved = O._lib
    O._lib = _Alias(ref_lib())
    try:
        yield
    finally:
        O._lib = saved


def marginalize(cfg, w, mode, prior_out):
    """Reference MarginalizationInfo on window w; the prior comes back in the reference's own block order."""
    d, s = w.desc(O)
    pr = C.cast(C.pointer(prior_out.struct), C.POINTER(O.Prior))
    return ref_lib().ref_marginalize(C.byref(cfg), C.byref(d), C.byref(s), C.c_int(mode), pr)


def prior_information(prior):
    """Order-independent content of a prior: {(id_a, id_b): H block}, {id: b block} with H = J0^T J0, b = J0^T r0,
    and the linearisation points {id: x0}."""
    n = prior.struct.n
    J = prior.J0[: n * n].reshape(n, n)
    r = prior.r0[:n]
    H, b = J.T @ J, J.T @ r
    blocks, off = {}, 0
    for k in range(prior.struct.n_blocks):
        bid, gs, idx = prior.struct.block_id[k], prior.struct.block_size[k], prior.struct.block_idx[k]
        ls = 6 if gs == 7 else gs
        blocks[bid] = (idx, ls, prior.x0[off:off + gs].copy())
        off += gs
    Hb = {(a, c): H[ia:ia + la, ic:ic + lc] for a, (ia, la, _) in blocks.items() for c, (ic, lc, _) in blocks.items()}
    bb = {a: b[ia:ia + la] for a, (ia, la, _) in blocks.items()}
    x0 = {a: x for a, (_, _, x) in blocks.items()}
    return Hb, bb, x0


def repropagate_imu_leg(cfg, samples, lin0, lins):
    """The reference's own IMULegIntegrationBase: constructor + push_back at lin0, then repropagate() once per row of lins on the SAME
    object; its public state after each (oracle_py.repropagate_imu_leg is the restatement)."""
    s = np.ascontiguousarray(samples, dtype=np.float64)
    lin0 = np.ascontiguousarray(lin0, dtype=np.float64)
    lins = np.atleast_2d(np.ascontiguousarray(lins, dtype=np.float64))
    out = np.zeros((1 + lins.shape[0], O.PREINT_DOUBLES))
    ref_lib().ref_repropagate_imu_leg(C.byref(cfg), C.cast(s.ctypes.data, C.POINTER(O.Sample)), C.cast(s[1:].ctypes.data, C.POINTER(O.Sample)),
                                      C.c_int(s.shape[0] - 1), lin0.ctypes.data_as(O.dp), C.c_int(lins.shape[0]), lins.ctypes.data_as(O.dp),
                                      C.cast(out.ctypes.data, C.POINTER(O.Preint)))
    return out


def time_evaluate(cfg, w, reps=20):
    """Seconds per pass over ALL cost functions of window w (prior, IMULegFactors, projection factors as Estimator::optimization adds them),
    each Evaluate()d with Jacobians by the compiled reference, timed inside the library; and the number of residual blocks per pass."""
    L = ref_lib()
    L.ref_time_evaluate.restype = C.c_double
    d, s = w.desc(O)
    n = C.c_int(0)
    t = L.ref_time_evaluate(C.byref(cfg), C.byref(d), C.byref(s), C.c_int(reps), C.byref(n))
    return t, n.value
