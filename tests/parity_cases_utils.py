"""The reference's `utils` test matrix (test/test_utils.py:100-621) restated as property tests on the
product, shared by the CPU host-logic tests (NumPy test double) and the GPU tests (HIP library).

Same parametrisations - plus the complex matrices the reference had to comment out for its CI
(test_utils.py:358-364, 403-409, 548-552) - and the same inequalities (Drkosova, Greenbaum,
Rozloznik, Strakos: Numerical stability of GMRES, BIT 1995, (2.3)-(2.5)), each recomputed here with
plain NumPy from the arrays the product returns.
"""
import itertools

import numpy as np
import scipy.linalg

from krypy_amd import utils
from oracle.inputs import zoo_matrices

EPS = np.finfo(float).eps
_B = np.diag(np.linspace(1, 5, 10))
_FACTORS = [0.0, 1.0, 1.0j, 1.0 + 1.0j, 1e8, 1e-8]


def _ip_Bs(matrix_form=False):
    out = [None, utils.MatrixLinearOperator(_B), lambda x, y: x.T.conj().dot(_B.dot(y))]
    return out + [_B] if matrix_form else out


def _ip(X, Y, weighted):
    return X.conj().T.dot(_B.dot(Y) if weighted else Y)


def _nrm(X, weighted):
    """utils.norm of a block: sqrt of the 2-norm of its Gram matrix"""
    return np.sqrt(np.linalg.norm(_ip(X, X, weighted), 2))


def case_house_givens():
    n = 0
    for a, b, length in itertools.product(_FACTORS, _FACTORS, (10, 1)):
        x = np.ones((length, 1), dtype=np.array([a]).dtype) * b
        x[0] = a
        H = utils.House(x)
        y = H.apply(x)
        I = np.eye(length)
        Hm = H.matrix()
        xn = np.linalg.norm(x, 2)
        assert np.linalg.norm(H.apply(I) - Hm, 2) <= 1e-14
        assert np.linalg.norm(Hm - Hm.T.conj(), 2) <= 1e-14
        assert np.linalg.norm(I - Hm.T.conj().dot(Hm), 2) <= 1e-14
        assert abs(xn - abs(y[0, 0])) <= 1e-14 * xn
        assert abs(1 - abs(H.alpha)) <= 1e-14
        assert abs(y[0, 0] - H.alpha * H.xnorm) <= 1e-14 * xn
        if length > 1:
            assert np.linalg.norm(y[1:], 2) <= 1e-14 * xn
        n += 1
    for a, b in itertools.product(_FACTORS, _FACTORS):
        x = np.array([[a], [b]])
        G = utils.Givens(x)
        y = G.apply(x)
        xn = np.linalg.norm(x, 2)
        assert np.linalg.norm(np.eye(2) - G.G.T.conj().dot(G.G), 2) <= 1e-14
        assert abs(xn - abs(y[0, 0])) <= 1e-14 * xn
        assert abs(y[1, 0]) <= 1e-14 * xn
        n += 1
    return n


def case_projection():
    n = 0
    Xs = [np.eye(10, 1), np.eye(10, 5), np.eye(10, 5) + 1e-1 * np.ones((10, 5)), np.eye(10),
          np.zeros((10, 0)), np.eye(10, 3) * (1 + 0.5j) + 0.1j * np.ones((10, 3))]
    for X, Ys, ipi, iterations in itertools.product(Xs, (None, 0, 1), range(3), (1, 2, 3)):
        Y = None if Ys is None else X + Ys
        P = utils.Projection(X, Y, ip_B=_ip_Bs()[ipi], iterations=iterations)
        N, k = X.shape
        I = np.eye(N)
        z = np.ones((10, 1)) / np.sqrt(10.0)
        PI = P.apply(I)
        assert np.linalg.norm(P.apply(I - PI), 2) < 1e-13
        if k > 0:
            assert np.linalg.norm(X - P.apply(X), 2) < 1e-13
            assert np.linalg.norm(_ip(X if Y is None else Y, I - PI, ipi > 0), 2) < 1e-12
        else:
            assert np.linalg.norm(PI) == 0
        assert np.linalg.norm(I - PI - P.apply_complement(I), 2) < 1e-13
        assert np.linalg.norm(P.operator() * z - P.apply(z)) == 0
        assert np.linalg.norm(P.operator_complement() * z - P.apply_complement(z)) == 0
        assert np.linalg.norm(P.matrix() - PI, 2) < 1e-13
        # adjoint (utils.py:554-638): <P x, y> = <x, P^* y> in the inner product used, same for the
        # complement, and the operators expose it
        if k > 0:
            rngp = np.random.default_rng(n)
            xa, ya = rngp.standard_normal((N, 2)), rngp.standard_normal((N, 2))
            lhs = _ip(P.apply(xa), ya, ipi > 0)
            rhs = _ip(xa, P.apply_adj(ya), ipi > 0)
            assert np.linalg.norm(lhs - rhs) < 1e-10 * max(1.0, np.linalg.norm(lhs))
            lhs = _ip(P.apply_complement(xa), ya, ipi > 0)
            rhs = _ip(xa, P.apply_complement_adj(ya), ipi > 0)
            assert np.linalg.norm(lhs - rhs) < 1e-10 * max(1.0, np.linalg.norm(lhs))
            assert np.linalg.norm(P.operator().adj * ya - P.apply_adj(ya)) == 0
            assert np.linalg.norm(P.operator_complement().adj * ya - P.apply_complement_adj(ya)) == 0
        a = np.ones((N, 1))
        want = _ip(X if Y is None else Y, a, ipi > 0)
        _, Ya = P.apply(a, return_Ya=True)
        assert np.allclose(Ya, want, atol=1e-6)
        _, Ya = P.apply_complement(a, return_Ya=True)
        assert np.allclose(Ya, want, atol=1e-6)
        n += 1
    return n


def case_qr():
    n = 0
    Xs = [np.eye(10, 5), scipy.linalg.hilbert(10)[:, :5], np.eye(10, 4) + 1j * scipy.linalg.hilbert(10)[:, :4]]
    for X, ipi, reorthos in itertools.product(Xs, range(3), (0, 1, 2)):
        N, k = X.shape
        smax = scipy.linalg.svd(X, compute_uv=False)[0]
        Q, R = utils.qr(X, ip_B=_ip_Bs()[ipi], reorthos=reorthos)
        assert Q.shape == (N, k) and R.shape == (k, k)
        assert np.linalg.norm(Q.dot(R) - X, 2) <= 1e-14 * smax
        assert np.linalg.norm(_ip(Q, Q, ipi > 0) - np.eye(k), 2) <= (1e-8 if reorthos < 1 else 1e-14)
        assert np.linalg.norm(np.tril(R, -1)) == 0
        n += 1
    return n


def case_angles():
    n = 0
    FGs = [np.eye(10, 1), 1j * np.eye(10, 1), np.eye(10, 4), np.eye(10)[:, -4:],
           np.eye(10, 4).dot(np.diag([1, 1e1, 1e2, 1e3])), np.eye(10, 4)]
    for F, G, ipi, vec in itertools.product(FGs, FGs, range(3), (False, True)):
        res = utils.angles(F, G, ip_B=_ip_Bs()[ipi], compute_vectors=vec)
        theta = res[0] if vec else res
        k, l = F.shape[1], G.shape[1]
        m = min(k, l)
        assert theta.shape == (max(k, l),)
        assert np.all(theta >= 0) and np.all(theta <= np.pi / 2) and np.all(np.diff(theta) >= -1e-15)
        assert np.all(np.abs(theta[m:] - np.pi / 2) <= 1e-14)
        if vec:
            _, U, V = res
            w = ipi > 0
            assert np.linalg.norm(_ip(U, U, w) - np.eye(k), 2) <= 1e-14
            assert np.linalg.norm(_ip(V, V, w) - np.eye(l), 2) <= 1e-14
            UV = _ip(U, V, w)
            want = np.zeros((k, l))
            want[:m, :m] = np.diag(np.cos(theta[:m]))
            assert np.linalg.norm(UV - want, 2) <= 1e-14
        n += 1
    return n


def case_hegedus():
    n = 0
    m = np.arange(1, 11.0)
    m[-1] = 1.0
    xs = [np.ones((10, 1)), np.full((10, 1), 1.0j + 1)]
    x0s = [np.zeros((10, 1)), np.linspace(1, 5, 10).reshape((10, 1))] + xs
    op = lambda P, v: v if P is None else P.dot(v)      # noqa: E731
    for (name, A, _), as_op, x, x0, M, Ml, ipi in itertools.product(
            zoo_matrices(), (False, True), xs, x0s, (None, np.diag(m)), (None, np.diag(m)), range(3)):
        b = A.dot(x)
        x0n = utils.hegedus(utils.MatrixLinearOperator(A) if as_op else A, b, x0, M, Ml, _ip_Bs()[ipi])
        w = ipi > 0

        def rnorm(xx):
            r = op(Ml, b - A.dot(xx))
            return np.sqrt(abs(_ip(r, op(M, r), w)[0, 0]))
        assert x0n.shape == (10, 1)
        assert rnorm(x0n) <= rnorm(x0) + 1e-13, (name, n)
        n += 1
    return n


def _assert_arnoldi(A, v, V, H, P, maxiter, ortho, M, weighted, An):
    """assert_arnoldi of the reference (test_utils.py:440-545), recomputed with NumPy."""
    N = v.shape[0]
    k = H.shape[1]
    assert k <= maxiter
    invariant = H.shape[0] == k
    assert V.shape[1] == H.shape[0]
    Mv = v if M is None else M.dot(v)
    v1n = np.sqrt(abs(_ip(v, Mv, weighted)[0, 0]))
    first = P if P is not None else V
    assert np.linalg.norm(first[:, [0]] - v / v1n) <= 1e-14
    assert np.linalg.norm(np.tril(H, -2)) == 0
    if ortho == "lanczos":
        assert np.linalg.norm(H[:k, :k] - H[:k, :k].T.conj()) == 0 and np.isreal(H).all()
    d = np.diag(H[1:, :])
    assert np.isreal(d).all() and (d.real >= 0).all()
    AV = A.dot(V if invariant else V[:, :-1])
    MAV = AV if M is None else M.dot(AV)
    res = MAV - V.dot(H)
    resn = _nrm(res, weighted)
    assert resn <= k * (N ** 1.5) * EPS * An * (5 if weighted else 1) * 4, (resn, ortho)   # (2.3)
    ortho_res = np.eye(V.shape[1]) - _ip(V, P if P is not None else V, weighted)
    ortho_resn = np.linalg.norm(ortho_res, 2)
    if ortho == "house":
        ortho_tol = (k ** 1.5) * N * EPS * 4                                               # (2.4)
    else:
        sv = scipy.linalg.svd(np.column_stack([V[:, [0]], (MAV[:, :-1] if invariant else MAV)]),
                              compute_uv=False)
        ortho_tol = np.inf if sv[-1] == 0 else (k ** 2) * N * EPS * sv[0] / sv[-1] * 4   # (2.5)
    if (ortho not in ("mgs", "cgs") or N != k) and ortho != "lanczos":
        assert ortho_resn <= ortho_tol, (ortho, ortho_resn, ortho_tol)
    proj_res = _ip(P if P is not None else V, MAV, weighted) - H
    assert np.linalg.norm(proj_res, 2) <= 10 * (ortho_resn * An + resn * _nrm(V, weighted)) + 1e-13


def case_arnoldi(orthos=("mgs", "dmgs", "house", "cgs2", "lanczos")):
    n = 0
    vs = [np.ones((10, 1)), np.eye(10, 1), (1 + 1j) * np.ones((10, 1))]
    for (name, A, flags), as_op, v, maxiter, ortho, M, ipi in itertools.product(
            zoo_matrices(), (False, True), vs, (1, 5, 9, 10), orthos, (None, _B), range(4)):
        if ortho == "house" and (ipi > 0 or M is not None):
            continue
        if ortho == "lanczos" and not flags.get("self_adjoint"):
            continue
        if ortho == "lanczos" and (ipi > 0 or M is not None):
            # (A must be self-adjoint in the inner product used: make it so, like test_linsys.py:64-71)
            A_use = np.linalg.inv(_B).dot(A) if ipi > 0 else A
            if M is not None:
                continue            # M A is not self-adjoint for these diagonal pairs in general
        else:
            A_use = A
        An = np.linalg.norm(A_use, 2)
        res = utils.arnoldi(utils.MatrixLinearOperator(A_use) if as_op else A_use, v, maxiter=maxiter,
                            ortho=ortho, M=M, ip_B=_ip_Bs(True)[ipi])
        V, H = res[0], res[1]
        P = res[2] if M is not None else None
        _assert_arnoldi(A_use, v, V, H, P, maxiter, ortho, M, ipi > 0, An)
        n += 1
    return n


def case_ritz_matrix():
    n = 0
    mats = zoo_matrices()
    herm = {"spd", "hpd"}
    for (name, A, flags), as_op, v, maxiter, ipi, with_V, kind in itertools.product(
            [mm for mm in mats if mm[0] in ("spd", "hpd", "nonsymm", "comp_nonsymm")], (False, True),
            [np.ones((10, 1)), np.eye(10, 1)], (1, 5, 9, 10), range(3), (True, False),
            ("ritz", "harmonic", "harmonic_improved")):
        is_h = name in herm
        A_use = np.linalg.inv(_B).dot(A) if (is_h and ipi > 0) else A     # self-adjoint w.r.t. ip_B
        An = np.linalg.norm(A_use, 2)
        ortho = "house" if ipi == 0 else "dmgs"
        V, H = utils.arnoldi(utils.MatrixLinearOperator(A_use) if as_op else A_use, v, maxiter=maxiter,
                             ortho=ortho, ip_B=_ip_Bs()[ipi])
        N, nn = 10, H.shape[1]
        if with_V:
            theta, U, resnorm, Z = utils.ritz(H, V=V, hermitian=is_h, type=kind)
            assert np.linalg.norm(V[:, :nn].dot(U) - Z, 2) <= 1e-14
        else:
            theta, U, resnorm = utils.ritz(H, hermitian=is_h, type=kind)
            Z = V[:, :nn].dot(U)
        assert theta.shape == (nn,) and U.shape == (nn, nn) and resnorm.shape == (nn,) and Z.shape == (N, nn)
        for i in range(nn):
            assert abs(np.linalg.norm(U[:, i], 2) - 1) <= 1e-14
        R = A_use.dot(Z) - Z.dot(np.diag(theta))
        w = ipi > 0
        for i in range(nn):
            assert abs(_nrm(R[:, [i]], w) - resnorm[i]) <= 1e-13 * An * 5, (name, kind, i)
        if kind == "ritz":
            assert np.linalg.norm(_ip(V[:, :nn], R, w), 2) <= 1e-13 * An * 5
        if nn == N:
            ev = (scipy.linalg.eigh if is_h and ipi == 0 else scipy.linalg.eig)(A_use)[0]
            assert np.all(np.abs(np.sort(np.abs(ev)) - np.sort(np.abs(theta))) <= 1e-12 * An)
        n += 1
    return n


CASES = [case_house_givens, case_projection, case_qr, case_angles, case_hegedus, case_arnoldi,
         case_ritz_matrix]
