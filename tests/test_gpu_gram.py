"""Inner products with a block on both sides on the FP64 matrix cores (krypy_amd/csrc/kernels.h: k_gram_mfma behind
kh_gemm_tn): utils.inner of the reference (/root/reference/krypy/utils.py:160-193) where both arguments are blocks - <W, V> of
Projection.__init__ (utils.py:478-520), <U, AU> and <V, AU> of the deflated solvers' and the Ritz set-up.

* against NumPy at every tile shape (full, partial, one column on the left, several tiles on both sides), row counts around
  the 64-row chunk (0 ... 63 rows of tail, fewer rows than one chunk), sub-blocks at column offsets, a block with itself;
* against the per-column kernels (the switch off) at 1e-13, and the same bits from run to run;
* a deflated solve through it against the CPU oracle;
* the other block product of the set-up, a block times a small matrix (kh_gemm_nn with 2 ... 16 output columns: k_panel_gemm_mfma),
  against NumPy and against the per-column kernel."""
import numpy as np
import pytest

from oracle import krylov_ref as ref
from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


def _blocks(hip, n, kx, ky, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, kx))
    Y = rng.standard_normal((n, ky)) + 0.25 * X[:, :1]
    return X, Y, hip.upload(X), hip.upload(Y)


@pytest.mark.parametrize("n", [1, 37, 64, 65, 1000, 4096 + 63, 100_003, 3_000_000])
@pytest.mark.parametrize("nx,ny", [(16, 16), (5, 3), (1, 2), (17, 16), (16, 17), (40, 33)])
def test_block_inner_products_against_numpy(hip, n, nx, ny):
    """(nx, ny) from one partial tile to 3 x 3 tiles; the bound is the forward error of a sum of n products."""
    if n >= 1_000_000 and nx * ny > 300:
        pytest.skip("the large case runs the single-tile and the 2-tile shapes")
    X, Y, Xd, Yd = _blocks(hip, n, nx, ny, n + nx)
    t0 = hip.get("n_gram_mfma")
    G = hip.gemm_tn(Xd, 0, nx, Yd, 0, ny)
    tiles = hip.get("n_gram_mfma") - t0
    want = X.T.dot(Y)
    scale = np.sqrt(np.sum(X * X, axis=0))[:, None] * np.sqrt(np.sum(Y * Y, axis=0))[None, :]
    assert G.shape == (nx, ny)
    assert np.max(np.abs(G - want) / scale) < 1e-14 * max(1.0, np.sqrt(n) / 8)
    G2 = hip.gemm_tn(Xd, 0, nx, Yd, 0, ny)
    assert np.array_equal(G, G2), "the same bits from run to run"
    was = hip.get("gram_mfma")
    hip.set("gram_mfma", 0)
    try:
        Gc = hip.gemm_tn(Xd, 0, nx, Yd, 0, ny)
    finally:
        hip.set("gram_mfma", was)
    assert np.max(np.abs(G - Gc) / scale) < 1e-13
    expect_kernel(tiles == ((nx + 15) // 16) * ((ny + 15) // 16), "16 x 16 tiles through k_gram_mfma: %d" % tiles)
    if tiles:
        assert np.array_equal(G, G2)


def test_sub_blocks_and_a_block_with_itself(hip):
    n = 200_001
    X, Y, Xd, Yd = _blocks(hip, n, 24, 20, 5)
    G = hip.gemm_tn(Xd, 3, 18, Yd, 2, 16)
    want = X[:, 3:21].T.dot(Y[:, 2:18])
    assert np.max(np.abs(G - want)) < 1e-10 * np.max(np.abs(want))
    S = hip.gemm_tn(Xd, 0, 24, Xd, 0, 24)
    assert np.max(np.abs(S - X.T.dot(X))) < 1e-10 * np.max(np.abs(S))
    assert np.max(np.abs(S - S.T)) < 1e-12 * np.max(np.abs(S))
    # one column on the right keeps the per-column kernel (the bits every dot product elsewhere has)
    t0 = hip.get("n_gram_mfma")
    g1 = hip.gemm_tn(Xd, 0, 24, Yd, 4, 1)
    assert hip.get("n_gram_mfma") == t0
    assert np.array_equal(g1[:, 0], hip.dot_panel(Xd, 0, 24, Yd, 4))


def test_deflated_solve_through_it_against_the_oracle(hip):
    """Projection.__init__'s <W, V> (utils.py:478-520) and E = <U, AU> through the kernel: a deflated GMRES solve with eight
    vectors against the CPU oracle."""
    from krypy_amd import deflation, linsys

    A = ref.laplace2d(300, 200)
    N = A.shape[0]
    b = np.random.default_rng(3).standard_normal(N)
    U = np.random.default_rng(4).standard_normal((N, 8))
    t0 = hip.get("n_gram_mfma")
    try:
        d = deflation.DeflatedGmres(linsys.LinearSystem(A, b, self_adjoint=True), U=U, maxiter=40, tol=1e-30)
    except Exception as e:      # ConvergenceError carries the solver
        d = e.solver
    used = hip.get("n_gram_mfma") - t0
    o = ref.deflated_gmres(A, b, U, tol=1e-30, maxiter=40)
    got, want = np.array(d.resnorms), np.array(o.resnorms)
    assert len(got) == len(want) and np.max(np.abs(got - want) / want) < 1e-9
    expect_kernel(used >= 1, "the set-up's block inner products took k_gram_mfma: %d tiles" % used)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 4096 + 63, 100_003, 3_000_001])
@pytest.mark.parametrize("k,nc", [(116, 16), (5, 2), (1, 3), (64, 16), (65, 7), (130, 16)])
def test_block_times_a_small_matrix_against_numpy(hip, n, k, nc):
    """kh_gemm_nn with 2 ... 16 output columns (k_panel_gemm_mfma: the Ritz vectors [V_n, U] @ coeffs,
    /root/reference/krypy/deflation.py:840-847): Y = beta Y + alpha X C against NumPy for beta = 0, 1 and 0.5, one to three
    passes of 64 columns, tails of 0 ... 31 rows, and against the per-column kernel (the switch off) at 1e-13."""
    if n >= 1_000_000 and k > 70:
        pytest.skip("the large case runs the one-pass shapes")
    rng = np.random.default_rng(n + k)
    X = rng.standard_normal((n, k + 2))
    C = rng.standard_normal((k, nc))
    Y0 = rng.standard_normal((n, nc + 1))
    Xd = hip.upload(X)
    scale = np.abs(X[:, 1:k + 1]).dot(np.abs(C)) + np.abs(Y0[:, 1:]) + 1e-300
    for alpha, beta in ((1.0, 0.0), (-0.75, 1.0), (2.0, 0.5)):
        Yd = hip.upload(Y0)
        p0 = hip.get("n_panel_gemm")
        hip.gemm_nn(Xd, 1, k, C, alpha, beta, Yd, 1)
        passes = hip.get("n_panel_gemm") - p0
        got = Yd.download()
        want = beta * Y0[:, 1:] + alpha * X[:, 1:k + 1].dot(C)
        assert np.array_equal(got[:, 0], Y0[:, 0]), "the column in front of the output is untouched"
        assert np.max(np.abs(got[:, 1:] - want) / scale) < 1e-14 * max(4.0, np.sqrt(k))
        was = hip.get("gram_mfma")
        hip.set("gram_mfma", 0)
        try:
            Yc = hip.upload(Y0)
            hip.gemm_nn(Xd, 1, k, C, alpha, beta, Yc, 1)
        finally:
            hip.set("gram_mfma", was)
        assert np.max(np.abs(got[:, 1:] - Yc.download()[:, 1:]) / scale) < 1e-13
        expect_kernel(passes == (k + 63) // 64, "passes of k_panel_gemm_mfma: %d" % passes)


def test_one_output_column_keeps_the_per_column_kernel(hip):
    """The x update of every solver (V[:, :k] @ y, linsys.py:941-949) has one output column: k_multiaxpy and its bits."""
    n, k = 50_001, 40
    rng = np.random.default_rng(9)
    X, c = rng.standard_normal((n, k)), rng.standard_normal(k)
    Xd, Yd, Zd = hip.upload(X), hip.alloc(n, 1), hip.alloc(n, 1)
    p0 = hip.get("n_panel_gemm")
    hip.gemm_nn(Xd, 0, k, c, 1.0, 0.0, Yd, 0)
    assert hip.get("n_panel_gemm") == p0
    was = hip.get("gram_mfma")
    hip.set("gram_mfma", 0)
    try:
        hip.gemm_nn(Xd, 0, k, c, 1.0, 0.0, Zd, 0)
    finally:
        hip.set("gram_mfma", was)
    assert np.array_equal(Yd.download(), Zd.download())


def test_qr_of_a_device_block_with_a_dependent_column_goes_back_to_its_input(hip):
    """utils.qr (/root/reference/krypy/utils.py:680-707) on a DEVICE block keeps no spare copy: when the fused factorisation meets a
    dependent column (R[i, i] < 1e-15) it starts again from the input block, which is still there - the result is the host
    path's (the reference's guard: the column is left unnormalised), and the input is untouched."""
    from krypy_amd import utils

    n = 5000
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, 5))
    X[:, 3] = X[:, 1]                    # column 3 is column 1: R[3, 3] = 0
    ipI = utils.IdentityLinearOperator((n, n))
    Xd = hip.upload(X)
    Qd, Rd = utils.qr(Xd, ip_B=ipI, reorthos=1)
    Qh, Rh = utils.qr(X, ip_B=ipI, reorthos=1)
    assert np.array_equal(Xd.download(), X), "the input block is not written"
    assert abs(Rd[3, 3]) < 1e-12 and abs(Rh[3, 3]) < 1e-12
    assert np.max(np.abs(Rd - Rh)) < 1e-12 * np.max(np.abs(Rh))
    Q = Qd.download()
    assert np.max(np.abs(Q - Qh)) < 1e-10
    keep = [0, 1, 2, 4]
    assert np.linalg.norm(Q[:, keep].T.dot(Q[:, keep]) - np.eye(4)) < 1e-12
    # and an independent block through the fused path: Q R = X, orthonormal columns
    X2 = rng.standard_normal((n, 6))
    Q2, R2 = utils.qr(hip.upload(X2), ip_B=ipI, reorthos=1)
    Q2 = Q2.download()
    assert np.linalg.norm(Q2.dot(R2) - X2) < 1e-12 * np.linalg.norm(X2)
    assert np.linalg.norm(Q2.T.dot(Q2) - np.eye(6)) < 1e-13
