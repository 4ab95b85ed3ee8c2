"""Test helper: the Schur complement of MarginalizationInfo::marginalize (marginalization_factor.cpp:262-296) evaluated in 60-digit
arithmetic (mpmath) from the FP64 normal equations the oracle assembled, and the comparison of a prior with it PER KEPT BLOCK PAIR, every
entry scaled by the square roots of the exact matrix's own diagonal entries (D^-1/2 H D^-1/2): a block whose entries are 1e-6 of the
largest one in the matrix is held to the same relative accuracy as the largest."""
import numpy as np


def exact_schur(A, b, m, digits=60):
    """A = [[Amm, Amr], [Arm, Arr]] (dropped dimensions first), b likewise -> (Arr - Arm Amm^-1 Amr, br - Arm Amm^-1 bm) as float64 arrays
    rounded from the 60-digit result. Amm is symmetrised as the reference does (:281)."""
    import mpmath as mp
    with mp.workdps(digits):
        Am, bm = mp.matrix(A.tolist()), mp.matrix(b.tolist())
        Amm = (Am[:m, :m] + Am[:m, :m].T) / 2
        Ai = mp.inverse(Amm)
        H = Am[m:, m:] - Am[m:, :m] * (Ai * Am[:m, m:])
        g = bm[m:, 0] - Am[m:, :m] * (Ai * bm[:m, 0])
        return np.array(H.tolist(), dtype=np.float64), np.array(g.tolist(), dtype=np.float64).ravel()


def block_table(prior):
    """{block id: (offset in the prior's own ordering, local size)}"""
    s = prior.struct
    return {s.block_id[k]: (s.block_idx[k], 6 if s.block_size[k] == 7 else s.block_size[k]) for k in range(s.n_blocks)}


def scaled_errors(prior, H_exact, b_exact, exact_blocks):
    """prior: a synth.PriorData (any block order); H_exact / b_exact in the order exact_blocks = block_table(the oracle's prior) describes.
    Returns (worst |dH_ij| / sqrt(H_ii H_jj) over all kept block pairs, worst |db_i| / sqrt(H_ii) relative to max(1, max_i |b_i| / sqrt(H_ii))
    — the gradient in whitened units, held to the accuracy of its largest component —, the same two normalised by the largest entry of H / b
    instead: what the tests used before)."""
    n = prior.struct.n
    J, r = prior.J0[: n * n].reshape(n, n), prior.r0[:n]
    H, g = J.T @ J, J.T @ r
    mine = block_table(prior)
    assert set(mine) == set(exact_blocks)
    d = np.sqrt(np.diag(H_exact))
    eh = eb = gh = gb = 0.0
    for a, (ia, la) in mine.items():
        ja, _ = exact_blocks[a]
        eb = max(eb, float((np.abs(g[ia:ia + la] - b_exact[ja:ja + la]) / d[ja:ja + la]).max()))
        gb = max(gb, float(np.abs(g[ia:ia + la] - b_exact[ja:ja + la]).max() / np.abs(b_exact).max()))
        for c, (ic, lc) in mine.items():
            jc, _ = exact_blocks[c]
            diff = np.abs(H[ia:ia + la, ic:ic + lc] - H_exact[ja:ja + la, jc:jc + lc])
            eh = max(eh, float((diff / np.outer(d[ja:ja + la], d[jc:jc + lc])).max()))
            gh = max(gh, float(diff.max() / np.abs(H_exact).max()))
    return eh, eb / max(1.0, float(np.abs(b_exact / d).max())), gh, gb
