"""Host logic of `0 < max_vector_count < k` (a10, online_update_bfgs.py:233-245, 309-310) on the CPU: the product's m x k
factor bookkeeping (`covariance._woodbury_factors`, `_real_inner`) driven by a NumPy stand-in for the d-sized device
kernels (row rescale + weighted Gram, pair append), against the dense matrices the reference produced
(tests/golden/covariance_trunc.npz, identity prior, d = 15).  The GPU test of the same fixture runs the real kernels."""
import numpy as np
import pytest

import inputs
from free_hunch_amd.covariance import CovarianceHessianBFGS, _woodbury_factors, _woodbury_inner


class NumpyRep:
    """diag(D) + diag(r) B M B^T diag(r) over a shared base - the HBM layout of covariance.py in NumPy"""

    def __init__(self, D):
        self.D, self.r, self.M = D.copy(), np.ones_like(D), np.zeros((0, 0))

    def dense(self, B):
        W = self.r[:, None] * B
        return np.diag(self.D) + W @ self.M @ W.T


def invert(src, dst, B, shift=0.0):
    """fh_rep_invert + the inner matrix: returns the Gram matrix the kernels would leave in G"""
    Dn = src.D + shift
    src.D = Dn
    G = B.T @ ((src.r ** 2 / Dn)[:, None] * B)
    dst.D, dst.r = 1.0 / Dn, src.r / Dn
    return G


@pytest.mark.parametrize("tag", ["id_d15_max1", "id_d15_max3"])
def test_truncation_bookkeeping_matches_reference_dense_matrices(gold, tag):
    g = gold("covariance_trunc")
    meta = eval(str(g[f"{tag}__meta"]))
    d, n = 15, meta["kw"]["max_vector_count"]
    s0 = meta["sigma0"]
    C, Ci, H, Hi = (NumpyRep(np.ones(d)) for _ in range(4))
    Bc, Bh = np.zeros((d, 0)), np.zeros((d, 0))
    Ci.D = 1.0 / C.D
    H.D = (C.D / s0 ** 2 - 1) / s0 ** 2
    Hi.D = 1.0 / H.D
    Q = {"C": (np.zeros((0, 0), complex),) * 2, "Ci": (np.zeros((0, 0), complex),) * 2}
    real = CovarianceHessianBFGS._real_inner
    steps = inputs.script(meta["script_seed"], meta["shape"], meta["n_steps"], s0, None)
    truncated = 0
    for si, (what, a) in enumerate(steps):
        pre = f"{tag}__{si}_"
        if what == "time":
            sc = float(np.float32(a["sigma_next"] ** -2 - a["sigma"] ** -2))
            sh = -float(np.float32(a["sigma_next"] ** 2 - a["sigma"] ** 2))
            G = invert(Ci, C, Bc, sc)
            Q["C"] = _woodbury_factors(*Q["Ci"], G)
            C.M = real(Q["C"][0] @ Q["C"][0].T - Q["C"][1] @ Q["C"][1].T)
            G = invert(Hi, H, Bh, sh)
            H.M = _woodbury_inner(Hi.M, G)
        else:
            x, xn = a["x"].numpy().reshape(-1), a["xn"].numpy().reshape(-1)
            m0, m1 = a["m0"].numpy().reshape(-1), a["m1"].numpy().reshape(-1)
            s2 = a["sigma"] ** 2
            dx, de = xn - x, s2 * (m1 - m0)
            cdx = C.dense(Bc) @ dx
            gamma, q = 1.0 / (dx @ de), dx @ cdx
            mc = Bc.shape[1]
            Bc = np.concatenate([Bc, (de / C.r)[:, None], (cdx / C.r)[:, None]], 1)
            Bh = np.concatenate([Bh, (de / H.r)[:, None], (cdx / H.r)[:, None]], 1)
            Mc = np.zeros((mc + 2, mc + 2)); Mc[:mc, :mc] = C.M; Mc[mc, mc], Mc[mc + 1, mc + 1] = gamma, -1 / q
            C.M = Mc
            mh = H.M.shape[0]
            Mh = np.zeros((mh + 2, mh + 2)); Mh[:mh, :mh] = H.M; Mh[mh, mh], Mh[mh + 1, mh + 1] = gamma / s2 ** 2, -1 / (q * s2 ** 2)
            H.M = Mh
            H.D = (C.D / s2 - 1) / s2
            QU, QV = Q["C"]
            k = QU.shape[1]
            QU2, QV2 = np.zeros((mc + 2, k + 1), complex), np.zeros((mc + 2, k + 1), complex)
            QU2[:mc, :k], QV2[:mc, :k] = QU, QV
            QU2[mc, k], QV2[mc + 1, k] = np.sqrt(complex(gamma)), 1 / np.sqrt(complex(q))
            Q["C"] = (QU2, QV2)
            G = invert(C, Ci, Bc)
            Q["Ci"] = _woodbury_factors(*Q["C"], G)
            Ci.M = real(Q["Ci"][0] @ Q["Ci"][0].T - Q["Ci"][1] @ Q["Ci"][1].T)
            G = invert(H, Hi, Bh)
            Hi.M = _woodbury_inner(H.M, G)
            if Q["C"][0].shape[1] > n:  # drop_vectors + _rederive_after_truncation
                truncated += 1
                QU, QV = Q["C"][0][:, -n:], Q["C"][1][:, -n:]
                Q["C"] = (QU, QV)
                C.M = real(QU @ QU.T - QV @ QV.T)
                G = invert(C, Ci, Bc)
                Q["Ci"] = _woodbury_factors(*Q["C"], G)
                Ci.M = real(Q["Ci"][0] @ Q["Ci"][0].T - Q["Ci"][1] @ Q["Ci"][1].T)
                Bh, H.r, H.D, H.M = Bc.copy(), C.r.copy(), (C.D / s2 - 1) / s2, C.M / s2 ** 2
                G = invert(H, Hi, Bh)
                Hi.M = _woodbury_inner(H.M, G)
        assert Q["C"][0].shape[1] == int(g[pre + "k"])
        for nm, rep, B in (("C", C, Bc), ("Ci", Ci, Bc), ("H", H, Bh), ("Hi", Hi, Bh)):
            ref = np.asarray(g[pre + nm])
            assert np.abs(ref.imag).max() < 1e-9 * max(1.0, np.abs(ref.real).max())
            assert np.abs(rep.dense(B) - ref.real).max() < 1e-8 * max(1.0, np.abs(ref.real).max()), (tag, si, nm)
    assert truncated >= 2


def test_folded_dct_blur_bases_equal_blur_then_dct():
    """Host side of the folded CG operator (free_hunch_amd.measurements.folded_dct_blur_bases): with the circular blur
    A(X) = F_col X F_row^T given by 1-D tap lists, P_col u P_row^T = dct2(A^T u) and P_col^T v P_row = A(idct2(v)), against
    SciPy's orthonormal DCT and explicit rolls - asymmetric taps, so that a transposed or mirrored basis cannot pass."""
    import scipy.fft
    from free_hunch_amd.measurements import dct_basis_longdouble, folded_dct_blur_bases
    S = 32
    g = np.random.default_rng(5)
    x = g.standard_normal((S, S))
    C = dct_basis_longdouble(S).astype(np.float64)
    assert np.abs(C @ x @ C.T - scipy.fft.dctn(x, type=2, norm="ortho")).max() < 1e-13
    col_dy, row_dx = np.arange(-4, 5), np.arange(-6, 7)
    cw, rw = g.uniform(0.1, 1.0, 9), g.uniform(0.1, 1.0, 13)

    def blur(X, sign):  # sign = +1: A (out[i] = sum w in[i - d]);  -1: A^T
        out = sum(w * np.roll(X, sign * int(dd), axis=0) for dd, w in zip(col_dy, cw))
        return sum(w * np.roll(out, sign * int(dd), axis=1) for dd, w in zip(row_dx, rw))

    P_row, P_col, P_row_t, P_col_t = folded_dct_blur_bases(col_dy, cw, row_dx, rw, S)
    assert np.abs(P_col @ x @ P_row.T - scipy.fft.dctn(blur(x, -1), type=2, norm="ortho")).max() < 1e-13
    assert np.abs(P_col_t @ x @ P_row - blur(scipy.fft.idctn(x, type=2, norm="ortho"), +1)).max() < 1e-13
    assert np.array_equal(P_row_t, P_row.T) and np.array_equal(P_col_t, P_col.T)


def test_symmetric_half_bases_reproduce_the_dct_passes():
    """`pack_symmetric_halves` + the two transposed passes of k_dct_sym in NumPy: forward = butterflies x_n +- x_{S-1-n} against
    the even / odd half bases, inverse = even / odd samples and E +- O outputs; both equal SciPy's orthonormal DCT-II / III."""
    import scipy.fft
    from free_hunch_amd.measurements import dct_basis_longdouble, pack_symmetric_halves
    S, H = 128, 64
    C = dct_basis_longdouble(S).astype(np.float64)
    fwd, inv = pack_symmetric_halves(C)
    assert fwd.shape == (2, H, H) and inv.shape == (2, H, H)
    assert pack_symmetric_halves(C + np.triu(np.ones((S, S)), 1) * 1e-6) is None  # not mirror-symmetric: dense passes
    x = np.random.default_rng(3).standard_normal((S, S))

    def sym_pass(Ah, X, INV):  # out[k][r] = sum_n P[k][n] X[r][n]
        out = np.zeros((S, S))
        if not INV:
            s, d = X[:, :H] + X[:, ::-1][:, :H], X[:, :H] - X[:, ::-1][:, :H]
            out[0::2], out[1::2] = Ah[0] @ s.T, Ah[1] @ d.T
        else:
            E, O = Ah[0] @ X[:, 0::2].T, Ah[1] @ X[:, 1::2].T
            out[:H], out[::-1][:H] = E + O, E - O
        return out

    assert np.abs(sym_pass(fwd, sym_pass(fwd, x, False), False) - scipy.fft.dctn(x, type=2, norm="ortho")).max() < 1e-13
    assert np.abs(sym_pass(inv, sym_pass(inv, x, True), True) - scipy.fft.idctn(x, type=2, norm="ortho")).max() < 1e-13
