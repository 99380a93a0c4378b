"""GPU parity tests (-m gpu): the shared parity cases on the REAL HIP library through the
C ABI, plus kernel-level checks against SciPy/NumPy on the same seeded inputs, and
size-independent properties at BASELINE.json's full size (N = 10^7).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from krypy_amd import linsys, utils
from oracle import krylov_ref as ref
from oracle.inputs import lap2d_system
from tests import parity_cases as pc
from tests import parity_cases_complex as pcc
from tests import parity_cases_utils as pcu

from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu

SIMPLE = [
    pc.case_toy_known_answers, pc.case_toy_custom_inner_product, pc.case_toy_deflated,
    pc.case_toy_solver_attributes, pc.case_api_errors, pc.case_inner_norm_panels,
    pc.case_arnoldi_steps, pc.case_arnoldi_invariant, pc.case_qr_projection,
    pc.case_operator_algebra, pc.case_restart_failure, pc.case_minres_jacobi,
    pc.case_minres_cg_sparse, pc.case_cg_dense, pc.case_deflated_gmres_recycling,
    pc.case_recycling_gmres_lap3d, pc.case_recycling_factories_toy, pc.case_inner_product_matrix_B,
    pc.case_solver_zoo, pc.case_ritz, pc.case_arnoldi_house, pc.case_basis_growth,
    pc.case_lanczos_window, pc.case_api_surface, pc.case_arnoldi_interleaved, pc.case_estimate_time,
    pc.case_input_kinds, pc.case_edge_cases, pc.case_matrix_preconditioner, pc.case_callable_preconditioner,
] + pcc.CASES


def test_native_library_is_the_one_running(hip):
    """The context is the ctypes binding of libkrylov_hip.so, not a test double."""
    from krypy_amd import _hip

    assert isinstance(hip, _hip.Context)
    info = hip.info()
    assert info["compute_units"] >= 64 and info["mem_total"] > 2 ** 34
    with open("/proc/self/maps") as f:
        assert "libkrylov_hip.so" in f.read()


@pytest.mark.parametrize("case", SIMPLE, ids=lambda f: f.__name__)
def test_case(hip, case):
    case()


@pytest.mark.parametrize("nx,rhs,ortho", [(64, "ones", "mgs"), (64, "rng1", "mgs"),
                                           (64, "rng1", "cgs2"), (128, "ones", "mgs"),
                                           (128, "rng1", "mgs"), (128, "rng1", "cgs")])
def test_restarted_gmres(hip, nx, rhs, ortho):
    pc.case_restarted_gmres(nx, rhs, ortho)


@pytest.mark.parametrize("ortho", ["mgs", "dmgs", "cgs", "cgs2"])
def test_one_cycle_nx200(hip, ortho):
    pc.case_one_cycle_nx200(ortho)


# ---- kernel level ----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4097, 100003])
def test_vector_kernels_edge_sizes(hip, n):
    rng = np.random.default_rng(n)
    V = rng.standard_normal((n, 5))
    w = rng.standard_normal((n, 1))
    Vd, Wd = hip.upload(V), hip.upload(w)
    scale = np.linalg.norm(V, axis=0) * np.linalg.norm(w) + 1e-300
    got = hip.dot_panel(Vd, 0, 5, Wd, 0)
    assert np.max(np.abs(got - V.T.dot(w[:, 0])) / scale) < 1e-14
    assert abs(hip.nrm2(Wd, 0) - np.linalg.norm(w)) <= 1e-14 * np.linalg.norm(w)
    h = rng.standard_normal(5)
    hip.axpy_panel(Vd, 0, 5, h, Wd, 0)
    want = w[:, 0].copy()
    for j in range(5):
        want = want - h[j] * V[:, j]
    assert np.array_equal(Wd.download()[:, 0], want)          # multiply-then-subtract, in order
    C = rng.standard_normal((5, 2))
    Y = hip.alloc(n, 2)
    hip.gemm_nn(Vd, 0, 5, C, 1.0, 0.0, Y, 0)
    assert np.allclose(Y.download(), V.dot(C), rtol=1e-13, atol=1e-13)
    Z = hip.alloc(n, 1)
    hip.waxpby(Z, 0, 2.0, Vd, 1, -0.5, Vd, 2)
    assert np.array_equal(Z.download()[:, 0], 2.0 * V[:, 1] + (-0.5) * V[:, 2])
    hip.vdiv(Z, 0, Vd, 3, 3.7)
    assert np.array_equal(Z.download()[:, 0], V[:, 3] / 3.7)


def test_reductions_are_deterministic(hip):
    rng = np.random.default_rng(0)
    V = hip.upload(rng.standard_normal((1 << 20, 3)))
    W = hip.upload(rng.standard_normal((1 << 20, 1)))
    first = hip.dot_panel(V, 0, 3, W, 0)
    for _ in range(5):
        assert np.array_equal(hip.dot_panel(V, 0, 3, W, 0), first)


@pytest.mark.parametrize("kind", ["lap2d", "random", "empty_rows", "long_row", "rect"])
def test_csr_spmv_bit_identical_to_scipy(hip, kind):
    rng = np.random.default_rng(11)
    if kind == "lap2d":
        A = ref.laplace2d(97, 53)
    elif kind == "random":
        A = _random_csr(20000, 20000, 40, 3)
    elif kind == "empty_rows":
        A = sp.random(5000, 5000, density=1e-3, random_state=4, format="csr")
        A = sp.vstack([A[:100], sp.csr_matrix((300, 5000)), A[400:]]).tocsr()
    elif kind == "long_row":
        A = sp.random(300, 50000, density=1e-3, random_state=5, format="lil")
        A[7, :] = rng.standard_normal(50000)             # 50000 nnz > LDS tile
        A = A.tocsr()
    else:
        A = sp.random(1000, 3000, density=5e-3, random_state=6, format="csr")
    A.sort_indices()
    x = rng.standard_normal((A.shape[1], 2))
    Ad = hip.csr(A)
    X, Y = hip.upload(x), hip.alloc(A.shape[0], 2)
    hip.apply(Ad, X, 0, Y, 0, 2)
    got, want = Y.download(), A.dot(x)
    if kind == "long_row":
        mask = np.ones(A.shape[0], bool)
        mask[7] = False
        assert np.array_equal(got[mask], want[mask])
        assert np.allclose(got[7], want[7], rtol=1e-12)    # tree-reduced row: not bit-ordered
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("solver", ["gmres", "minres", "gmres_complex"])
def test_chain_timeout_is_recovered(hip, solver):
    """A timeout of the chain kernel's grid-wide reduction (workgroups not co-resident on a shared GPU) must
    not fail the solve: kh_arnoldi_step_end switches the chain off and runs the step again on the per-column
    kernels from the intact columns 0..k - also when look-ahead steps were already in flight on the garbage.
    kh_ctx_set("chain_fault", 1) makes the next chain launch set the error word and halve every coefficient."""
    A = ref.laplace2d(300, 220)
    rng = np.random.default_rng(5)
    b = rng.standard_normal(A.shape[0])
    kw = dict(maxiter=40, tol=1e-30)
    if solver == "gmres_complex":
        A = (A + 0.3j * sp.diags(rng.standard_normal(A.shape[0]))).tocsr()
        b = b + 1j * rng.standard_normal(A.shape[0])

    def run(fault_at, reset=True):
        if reset:
            hip.set("chain", 1)
        ls = linsys.LinearSystem(A, b, self_adjoint=(solver == "minres"))
        cls = linsys.Minres if solver == "minres" else linsys.Gmres

        class Faulty(cls):
            def _finalize_iteration(self, yk, resnorm):
                if self.iter == fault_at:
                    hip.set("chain_fault", 1)
                return super(Faulty, self)._finalize_iteration(yk, resnorm)

        before = hip.get("n_chain_recovered")
        try:
            sol = Faulty(ls, **kw)
        except utils.ConvergenceError as e:
            sol = e.solver
        return sol, hip.get("n_chain_recovered") - before

    try:
        good, n0 = run(-1)
        bad, n1 = run(7)
        # the switch-off is not for life (round 4): 33 clean steps are fewer than the 100 that re-arm inside a solve, so
        # the chain family is still off here - and comes back with the next basis (step k = 0), whose solve runs chain
        # launches again and reproduces the undisturbed history
        off_after = hip.get("chain")
        recov = hip.get("chain_recoveries")
        rearmed0, chain0 = hip.get("n_chain_rearmed"), hip.counters()["chain"]
        again, n2 = run(-1, reset=False)
        rearmed1, chain1 = hip.get("n_chain_rearmed"), hip.counters()["chain"]
    finally:
        hip.set("chain", 1)
    assert np.array_equal(np.asarray(again.resnorms), np.asarray(good.resnorms))
    assert len(bad.resnorms) == len(good.resnorms) == 41
    # what ran (judged after the comparisons below: tests/support/kernel_expect.py)
    expect_kernel(n0 == 0 and n1 >= 1, "recoveries clean / faulted run 0 / >= 1: %r" % ((n0, n1),))
    expect_kernel(off_after == 0 and recov == 1, "chain off after the timeout, one recovery: %r" % ((off_after, recov),))
    expect_kernel(n2 == 0 and rearmed1 - rearmed0 == 1 and chain1 - chain0 >= 40,
                  "re-armed with the next basis: %r" % ((n2, rearmed0, rearmed1, chain0, chain1),))
    expect_kernel(hip.get("chain") == 1 and hip.get("chain_recoveries") == 0, "chain on again, recoveries reset")
    # the per-column kernels and the chain kernel differ in the order of their partial sums only
    # (test_mgs_chain_every_register_shape): the recovered solve is the undisturbed one to rounding
    assert np.max(np.abs(np.asarray(bad.resnorms) - np.asarray(good.resnorms)) / np.asarray(good.resnorms)) < 1e-9
    assert np.linalg.norm(bad.xk - good.xk) < 1e-9 * np.linalg.norm(good.xk)


@pytest.mark.parametrize("kind", ["lap2d_banded", "lap2d_csr", "random", "long_row"])
@pytest.mark.parametrize("d", [1, 2, 5, 16, 32])
def test_csr_panel_apply_streams_the_matrix_once(hip, kind, d):
    """kh_apply of a CSR operator to a block of d vectors (A U of the deflation set-up, deflation.py:47; Ritz
    residuals, deflation.py:849-855): ONE pass over the matrix (k_spmm_stream / k_spmm_dia), every column
    bit-identical to scipy's A.dot(X) (csr_matvecs adds a row's products left to right, per column)."""
    rng = np.random.default_rng(17)
    if kind.startswith("lap2d"):
        A = ref.laplace2d(301, 97)
    elif kind == "random":
        A = _random_csr(30011, 30011, 30, 8)
    else:
        A = sp.random(300, 40000, density=1e-3, random_state=5, format="lil")
        A[11, :] = rng.standard_normal(40000)
        A = A.tocsr()
    A.sort_indices()
    hip.set("spmv_dia", 0 if kind == "lap2d_csr" else 1)
    try:
        Ad = hip.csr(A)
        expect_kernel((Ad.diagonals > 0) == kind.startswith("lap2d"), "a banded copy exists for the stencil only: %r" % ((kind, Ad.diagonals),))
        x = rng.standard_normal((A.shape[1], d))
        X, Y = hip.upload(x), hip.alloc(A.shape[0], d + 1)
        before = hip.get("n_spmm")
        hip.apply(Ad, X, 0, Y, 1, d)                       # (an offset in the output block, too)
        got, want = Y.download(1, d), A.dot(x)
        expect_kernel(hip.get("n_spmm") - before == (1 if d >= 2 else 0), "hip.get(\"n_spmm\") - before == (1 if d >= 2 else 0)")
    finally:
        hip.set("spmv_dia", 1)
    if kind == "long_row":
        mask = np.ones(A.shape[0], bool)
        mask[11] = False
        assert np.array_equal(got[mask], want[mask])
        assert np.allclose(got[11], want[11], rtol=1e-12)   # tree-reduced row: not bit-ordered
    else:
        assert np.array_equal(got, want)


def _random_csr(n_rows, n_cols, per_row, seed):
    """About `per_row` random entries per row, sorted, no duplicates (scipy.sparse.random draws without replacement
    from n_rows * n_cols positions: half a minute at 30,000^2)."""
    r = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n_rows), per_row)
    cols = r.integers(0, n_cols, rows.size)
    A = sp.coo_matrix((r.standard_normal(rows.size), (rows, cols)), shape=(n_rows, n_cols)).tocsr()
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    return A


def _banded(n, offsets, seed, holes=0.0):
    rng = np.random.default_rng(seed)
    diags = []
    for o in offsets:
        d = rng.standard_normal(n - abs(o))
        d[d == 0.0] = 1.0
        if holes:
            d[rng.random(d.size) < holes] = 0.0
        diags.append(d)
    A = sp.diags(diags, offsets, shape=(n, n)).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    return A


@pytest.mark.parametrize("kind,nd", [
    ("tridiag", 3), ("lap2d", 5), ("lap3d", 7), ("nine", 9), ("holes", 5), ("generic11", 11), ("two", 2),
    ("edge1023", 5), ("edge1025", 5), ("tiny", 3), ("wide", 4), ("corners", 0), ("explicit_zero", 0), ("sparse_diags", 0),
    ("too_many", 0)])
def test_banded_spmv_bit_identical_to_scipy(hip, kind, nd):
    """k_spmv_dia (banded copy of a CSR operator): detection, and the same bits as csr_matvec."""
    if kind == "tridiag":
        A = _banded(30011, (-1, 0, 1), 1)
    elif kind == "lap2d":
        A = ref.laplace2d(97, 53)
    elif kind == "lap3d":
        A = ref.laplace3d(23).tocsr()
    elif kind == "nine":
        A = _banded(40000, (-201, -200, -199, -1, 0, 1, 199, 200, 201), 2)
    elif kind == "holes":
        A = _banded(25000, (-150, -1, 0, 1, 150), 3, holes=0.2)
    elif kind == "generic11":
        A = _banded(9000, (-900, -30, -3, -2, -1, 0, 1, 2, 3, 30, 900), 4)
    elif kind == "two":
        A = _banded(5000, (0, 17), 5)
    elif kind == "edge1023":
        A = _banded(1023, (-31, -1, 0, 1, 31), 6)
    elif kind == "edge1025":
        A = _banded(1025, (-31, -1, 0, 1, 31), 7)
    elif kind == "tiny":
        A = _banded(3, (-1, 0, 1), 8)
    elif kind == "wide":
        A = _banded(4096, (-1000, 0, 1, 1000), 9)
    elif kind == "corners":
        A = _banded(4096, (-4095, -1, 0, 1, 4095), 13)    # periodic tridiagonal: 60 % full, CSR serves
    elif kind == "explicit_zero":
        A = _banded(3000, (-1, 0, 1), 10)
        A.data[7] = 0.0                                   # a stored zero: the CSR kernel must serve
    elif kind == "sparse_diags":
        A = _banded(3000, (-7, 0, 7), 11, holes=0.5)      # less than 70 % full
    else:
        A = _banded(3000, tuple(range(-20, 21)), 12)      # 41 diagonals > 32
    n = A.shape[0]
    rng = np.random.default_rng(99)
    x = rng.standard_normal((n, 3))
    Ad = hip.csr(A)
    if os.environ.get("KRYPY_AMD_SPMV_DIA", "") != "0":
        assert Ad.diagonals == nd
    X, Y = hip.upload(x), hip.alloc(n, 3)
    hip.apply(Ad, X, 0, Y, 0, 3)
    assert np.array_equal(Y.download(), A.dot(x))
    # fused epilogues of the banded kernel: explicit residual + norm, first MGS coefficient
    b = rng.standard_normal((n, 1))
    R = hip.alloc(n, 1)
    nrm = hip.residual(Ad, hip.upload(b), 0, X, 1, R, 0)
    want = b[:, 0] - A.dot(x[:, 1])
    assert np.array_equal(R.download()[:, 0], want)
    assert abs(nrm - np.linalg.norm(want)) <= 1e-14 * np.linalg.norm(want)
    if n >= 8:
        V, W = hip.alloc(n, 4), hip.alloc(n, 2)
        v0 = x[:, [0]] / np.linalg.norm(x[:, 0])
        V.upload(0, v0)
        h = hip.arnoldi_step(Ad, None, V, None, W, 0, 0, 0, 1, 0)
        w = A.dot(v0[:, 0])
        a0 = float(np.dot(v0[:, 0], w))
        assert abs(h[0] - a0) <= 1e-13 * np.linalg.norm(w)
        w = w - a0 * v0[:, 0]
        assert abs(h[1] - np.linalg.norm(w)) <= 1e-13 * np.linalg.norm(w)


@pytest.mark.parametrize("kind", ["lap2d", "lap3d", "nonsym", "random"])
def test_shard_spmv_with_ghost_rows_on_one_gpu(hip, kind):
    """A block-row shard's SpMV (ghost columns, krypy_amd/dist.py) on the one GPU: every slab of a
    3-way split, ghost entries written with kh_mat_set_ghost instead of the halo exchange, must
    reproduce its rows of the global product bit for bit - through the banded kernel's ghost-row
    variant for the stencil matrices, through the CSR kernel for the random one."""
    from krypy_amd import dist

    if kind == "lap2d":
        A, align = ref.laplace2d(64, 45), 64
    elif kind == "lap3d":
        A, align = ref.laplace3d(18).tocsr(), 18 * 18
    elif kind == "nonsym":
        A, align = _banded(30000, (-300, -2, -1, 0, 1, 300), 21), 1      # even and odd offsets, one-sided -2
    else:
        A = (sp.random(6000, 6000, density=2e-3, random_state=8, format="csr") + sp.eye(6000)).tocsr()
        A.sort_indices()
        align = 1
    n = A.shape[0]
    x = np.random.default_rng(5).standard_normal(n)
    want = A.dot(x)
    b = np.random.default_rng(6).standard_normal(n)
    cuts = dist.slab_cuts(n, 3, align)
    for p in range(3):
        r0, r1 = cuts[p], cuts[p + 1]
        A_local, nrp, nrn = dist.localize_columns(A[r0:r1], r0, n)
        Ad = hip.csr(A_local, n_cols=A_local.shape[1])
        hip.set_halo(Ad, 0, 0, nrp, nrn)                      # no communicator: nothing is sent
        hip.set_ghost(Ad, np.concatenate([x[r0 - nrp:r0], x[r1:r1 + nrn]]))
        if os.environ.get("KRYPY_AMD_SPMV_DIA", "") != "0":
            assert (Ad.diagonals > 0) == (kind != "random"), (kind, p, Ad.diagonals)
        X, Y = hip.upload(x[r0:r1]), hip.alloc(r1 - r0, 1)
        hip.apply(Ad, X, 0, Y, 0, 1)
        assert np.array_equal(Y.download()[:, 0], want[r0:r1]), (kind, p)
        R = hip.alloc(r1 - r0, 1)
        hip.residual(Ad, hip.upload(b[r0:r1]), 0, X, 0, R, 0)
        assert np.array_equal(R.download()[:, 0], b[r0:r1] - want[r0:r1]), (kind, p)


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_banded_spmv_measurement_modes(hip, mode):
    """KRYPY_AMD_SPMV_DIA is read once per process: the non-default settings (CSR kernel, 1 and 2
    row pairs per lane) are checked in a child process - same bits as SciPy in every mode."""
    import subprocess
    import sys
    code = (
        "import numpy as np, scipy.sparse as sp\n"
        "from krypy_amd import _hip\n"
        "from oracle import krylov_ref as ref\n"
        "ctx = _hip.get_context()\n"
        "for A in (ref.laplace2d(97, 53), ref.laplace3d(21).tocsr(), ref.laplace2d(1500, 700)):\n"
        "    x = np.random.default_rng(1).standard_normal((A.shape[0], 2))\n"
        "    Ad = ctx.csr(A)\n"
        "    assert (Ad.diagonals > 0) == (%r != '0')\n"
        "    Y = ctx.alloc(A.shape[0], 2)\n"
        "    ctx.apply(Ad, ctx.upload(x), 0, Y, 0, 2)\n"
        "    assert np.array_equal(Y.download(), A.dot(x))\n"
        "print('ok')\n" % mode)
    env = dict(os.environ, KRYPY_AMD_SPMV_DIA=mode)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_dense_gemv_and_diag(hip):
    rng = np.random.default_rng(2)
    for n, m in ((1, 1), (37, 41), (512, 512), (1000, 999)):
        A = rng.standard_normal((n, m))
        x = rng.standard_normal((m, 1))
        Y = hip.alloc(n, 1)
        hip.apply(hip.dense(A), hip.upload(x), 0, Y, 0, 1)
        assert np.allclose(Y.download(), A.dot(x), rtol=1e-13, atol=1e-12)
    d = rng.standard_normal(777)
    x = rng.standard_normal((777, 1))
    Y = hip.alloc(777, 1)
    hip.apply(hip.diag(d), hip.upload(x), 0, Y, 0, 1)
    assert np.array_equal(Y.download()[:, 0], d * x[:, 0])


def test_fused_residual(hip):
    A, b = lap2d_system(150, rhs="rng1")
    x = np.random.default_rng(9).standard_normal((A.shape[0], 1))
    R = hip.alloc(A.shape[0], 1)
    nrm = hip.residual(hip.csr(A), hip.upload(b), 0, hip.upload(x), 0, R, 0)
    want = b - A.dot(x[:, 0])
    assert np.array_equal(R.download()[:, 0], want)
    assert abs(nrm - np.linalg.norm(want)) < 1e-14 * nrm


def test_arnoldi_step_matches_oracle_step_by_step(hip):
    """kh_arnoldi_step against the oracle's arnoldi_step on the same inputs, every step."""
    A, b = lap2d_system(120, rhs="rng1")
    m = 40
    for ortho, gs, sweeps in (("mgs", 0, 1), ("dmgs", 0, 2), ("mgs", 1, 2)):
        st = ref.arnoldi_init(A, b, m, ortho=ortho)
        V = hip.alloc(A.shape[0], m + 1)
        W = hip.alloc(A.shape[0], 2)
        V.upload(0, st.V[:, :1])
        Ad = hip.csr(A)
        for k in range(m):
            ref.arnoldi_step(st)
            hcol = hip.arnoldi_step(Ad, None, V, None, W, 0, k, 0, sweeps, gs)
            assert np.linalg.norm(hcol - st.H[: k + 2, k]) < 1e-12 * np.linalg.norm(hcol)
        assert np.linalg.norm(V.download() - st.V) < 1e-10 * np.sqrt(m)


# ---- full size (config 2: N = 10^7): size-independent properties -------------------------------
def test_full_size_properties(hip):
    """nx=4000, ny=2500 5-point Laplacian, N = 10^7, nnz = 49,987,000 (BASELINE.json config 2).

    SpMV is checked bit-for-bit against SciPy; one 30-step Arnoldi relation A V_k = V_{k+1} H
    is verified on the device (residual and orthogonality), for both Gram-Schmidt variants."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(4000, 2500)
    N = A.shape[0]
    assert N == 10_000_000 and A.nnz == 49_987_000 and A.indices.dtype == np.int32
    b = np.random.default_rng(0).standard_normal(N)
    ls = linsys.LinearSystem(A, b)
    x = np.random.default_rng(1).standard_normal((N, 1))
    assert np.array_equal(ls.A * x, A.dot(x))
    # first Arnoldi steps at full size against the CPU oracle, iterate for iterate (1e-10)
    st = ref.arnoldi_init(A, b, 6)
    ar = utils.Arnoldi(A, b.reshape(-1, 1), maxiter=6, ortho="mgs")
    for _ in range(6):
        ref.arnoldi_step(st)
        ar.advance()
    assert np.linalg.norm(ar.H - st.H) < 1e-10 * np.linalg.norm(st.H)
    assert np.linalg.norm(ar._V.download(6, 1)[:, 0] - st.V[:, 6]) < 1e-10
    del st
    for ortho in ("mgs", "cgs2"):
        ar = utils.Arnoldi(A, b.reshape(-1, 1), maxiter=30, ortho=ortho)
        for _ in range(30):
            ar.advance()
        H = ar.H
        assert np.all(np.diag(H, -1) > 0)
        G = hip.gemm_tn(ar._V, 0, 31, ar._V, 0, 31)
        assert np.linalg.norm(G - np.eye(31)) < 1e-12
        # || A V_30 - V_31 H ||_F through the device: column by column
        T = hip.alloc(N, 2)
        Ad = ls.A._device_matrix()
        worst = 0.0
        for j in (0, 7, 29):
            hip.apply(Ad, ar._V, j, T, 0, 1)
            hip.gemm_nn(ar._V, 0, 31, H[:, j], -1.0, 1.0, T, 0)
            worst = max(worst, hip.nrm2(T, 0))
        assert worst < 1e-12 * np.linalg.norm(H, 2)


def test_rccl_single_rank_communicator(hip):
    """The RCCL plumbing (dlopen, unique id, ncclCommInitRank, ncclAllReduce on the library's
    stream) on the one GPU this box has; the 2-rank logic is covered by tests/test_dist_gloo.py."""
    from krypy_amd import _hip

    ctx = _hip.Context(0)
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(0, 1, uid)
    vals = np.array([1.5, -2.0, 3.25])
    assert np.array_equal(ctx.allreduce_host(vals.copy()), vals)
    V = ctx.upload(np.arange(12.0).reshape(6, 2, order="F"))
    W = ctx.upload(np.ones(6))
    assert np.allclose(ctx.dot_panel(V, 0, 2, W, 0), [15.0, 51.0])
    ctx.close()


def _second_context(chain, fused_spmv=True):
    """A private context with the register-resident MGS chain (and the operator fused into its
    prologue) switched on/off (both read at creation)."""
    import os
    from krypy_amd import _hip

    want = {"KRYPY_AMD_MGS_CHAIN": "1" if chain else "0", "KRYPY_AMD_CHAIN_SPMV": "1" if fused_spmv else "0"}
    old = {k: os.environ.get(k) for k in want}
    os.environ.update(want)
    try:
        return _hip.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("shape", [("lap2d", 2000, 1600), ("lap2d", 2400, 2400), ("lap2d", 3000, 2200),
                                   ("lap2d", 4000, 2500), ("lap3d", 150, 0), ("lap3d", 190, 0), ("lap3d", 215, 0),
                                   ("holes", 6_600_001, 0), ("lap3d", 225, 0)])
def test_operator_fused_into_the_chain_prologue(hip, shape):
    """Banded operator + long vector: the chain kernel computes w = A v_k in its prologue instead of
    reading what a separate SpMV launch wrote.  Same arithmetic in the same order, so H and the basis
    must come out bit for bit as with the SpMV launch (16 ... 40 rows per lane, 5 and 7 diagonals,
    single and double sweeps; round 4: 48 rows per lane - 225^3 = 11.4 M rows, the last 8 rows of w in LDS, where the
    plain chain kernel has the prologue and there is neither a three-pass Lanczos kernel nor LDS parking)."""
    kind, a, b_ = shape
    if kind == "holes":      # nonsymmetric pattern, odd and even offsets, 10 % of the slots empty, odd n
        A = _banded(a, (-3000, -1, 0, 2, 2999), 17, holes=0.1)
    else:
        A = ref.laplace2d(a, b_) if kind == "lap2d" else ref.laplace3d(a).tocsr()
    n = A.shape[0]
    v = np.random.default_rng(3).standard_normal(n)
    dj = np.linspace(0.5, 1.5, n)
    m = 6
    out = []
    for fused in (True, False):
        ctx = _second_context(True, fused)
        Ad, Md = ctx.csr(A), ctx.diag(dj)
        expect_kernel(Ad.diagonals in (5, 7), "the operator has its diagonal-major copy: %r" % (Ad.diagonals,))
        res = {}
        # plain / double-sweep MGS (LDS-parking kernel), Lanczos with its pre-subtraction, Jacobi (plain kernel),
        # Lanczos with Jacobi (MINRES + M: config 3).  Steps with ONE Gram-Schmidt link - every Lanczos step, the
        # first step of the others - run the three-pass kernel of lanczos.h when the operator is fused
        for name, use_m, lanczos in (("mgs", False, False), ("lanczos", False, True), ("jacobi", True, False),
                                     ("lanczos_jacobi", True, True)):
            before = ctx.counters()
            lz0 = ctx.get("n_lanczos_fused")
            V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
            P = ctx.alloc(n, m + 1) if use_m else None
            if use_m:
                nrm = np.sqrt(np.dot(v, dj * v))
                P.upload(0, v / nrm)
                V.upload(0, dj * v / nrm)
            else:
                V.upload(0, v / np.linalg.norm(v))
            H = np.zeros((m + 1, m))
            for k in range(m):
                start = k if lanczos else 0
                hk = float(H[k, k - 1]) if (lanczos and k > 0) else 0.0
                hcol = ctx.arnoldi_step(Ad, Md if use_m else None, V, P, W, 0, k, start,
                                        2 if (k == 4 and not lanczos) else 1, 0, hk)
                H[start: k + 2, k] = hcol[start: k + 2]
            res[name] = (H, V.download(), P.download() if use_m else np.zeros(1))
            c = ctx.counters()
            expect_kernel(c["chain"] - before["chain"] == m, "c[\"chain\"] - before[\"chain\"] == m: %r" % ((name, c),))
            lds_on = os.environ.get("KRYPY_AMD_CHAIN_LDS", "1") != "0"
            lz = ctx.get("n_lanczos_fused") - lz0
            long_shape = n > 10_480_000
            lz_on = fused and os.environ.get("KRYPY_AMD_LANCZOS_FUSED", "1") != "0" and not long_shape
            expect_kernel(lz == ((m if lanczos else 1) if lz_on else 0), "lz == ((m if lanczos else 1) if lz_on else 0): %r" % ((name, lz),))
            # (48 rows per lane: k_mgs_chain_long keeps a third of the column on the chip and is counted with the LDS family)
            long_on = os.environ.get("KRYPY_AMD_CHAIN_LONG", "1") != "0" and n <= 12_580_000
            on_chip = lds_on and not use_m and (not long_shape or long_on)
            expect_kernel(c["chain_lds"] - before["chain_lds"] == ((m - lz) if on_chip else 0), "c[\"chain_lds\"] - before[\"chain_lds\"] == ((m - lz) if on_chip else 0): %r" % ((name, c, on_chip),))
            expect_kernel(c["chain_fused"] - before["chain_fused"] == (m if fused else 0), "c[\"chain_fused\"] - before[\"chain_fused\"] == (m if fused else 0): %r" % ((name, c),))
            del V, W, P
        out.append(res)
        ctx.close()
    for name in out[0]:
        (Hf, Vf, Pf), (Hs, Vs, Ps) = out[0][name], out[1][name]
        assert np.array_equal(Hf, Hs), name
        assert np.array_equal(Vf, Vs), name
        assert np.array_equal(Pf, Ps), name
    Hf, Vf, _ = out[0]["mgs"]
    assert np.linalg.norm(A.dot(Vf[:, :m]) - Vf.dot(Hf)) < 1e-12 * np.linalg.norm(Hf)


@pytest.mark.parametrize("shape", [(300, 300), (500, 500), (1000, 700), (1500, 800)])
def test_lanczos_kernel_on_short_vectors(hip, shape):
    """The three-pass Lanczos kernel (lanczos.h) also serves the 4 / 8-rows-per-lane shapes (N from 65,536 up: padded
    blocks): a MINRES iteration at N = 10^5 is ONE launch instead of SpMV + chain + update.  Bit for bit the SpMV +
    chain pair, with and without the Jacobi diagonal, a deferred MINRES update riding along."""
    nx, ny = shape
    A = ref.laplace2d(nx, ny)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    v = rng.standard_normal(n)
    dj = np.linspace(0.5, 1.5, n)
    Vm, Wm0, y0 = rng.standard_normal((n, 1)), rng.standard_normal((n, 2)), rng.standard_normal((n, 1))
    m = 6
    out = []
    for fused in (True, False):
        ctx = _second_context(True, fused)
        Ad, Md = ctx.csr(A), ctx.diag(dj)
        res = {}
        for name, use_m in (("lanczos", False), ("lanczos_jacobi", True)):
            lz0, rides0 = ctx.get("n_lanczos_fused"), ctx.get("n_minres_rides")
            V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
            P = ctx.alloc(n, m + 1) if use_m else None
            Wm, yk = ctx.upload(Wm0), ctx.upload(y0)
            if use_m:
                nrm = np.sqrt(np.dot(v, dj * v))
                P.upload(0, v / nrm)
                V.upload(0, dj * v / nrm)
            else:
                V.upload(0, v / np.linalg.norm(v))
            H = np.zeros((m + 1, m))
            for k in range(m):
                if k >= 2:          # the recurrences of iteration k - 2, as Minres defers them
                    ctx.minres_update(V, k - 2, Wm, k & 1, 0.3, -0.2, 1.7, 0.4, yk, 0, defer=True)
                hk = float(H[k, k - 1]) if k > 0 else 0.0
                hcol = ctx.arnoldi_step(Ad, Md if use_m else None, V, P, W, 0, k, k, 1, 0, hk)
                H[k: k + 2, k] = hcol[k: k + 2]
            ctx.minres_flush()
            res[name] = (H, V.download(), P.download() if use_m else np.zeros(1), Wm.download(), yk.download())
            expect_kernel(ctx.get("n_lanczos_fused") - lz0 == (m if fused else 0), "ctx.get(\"n_lanczos_fused\") - lz0 == (m if fused else 0): %r" % ((name, fused),))
            expect_kernel(ctx.get("n_minres_rides") - rides0 == (m - 2 if fused else 0), "ctx.get(\"n_minres_rides\") - rides0 == (m - 2 if fused else 0): %r" % ((name, fused),))
        out.append(res)
        ctx.close()
    for name in out[0]:
        for a_, b_ in zip(out[0][name], out[1][name]):
            assert np.array_equal(a_, b_), name


@pytest.mark.parametrize("shape", [(120, 120), (39, 41), (700, 300)])
def test_mgs_chain_kernel_equals_link_kernels(hip, shape):
    """The one-launch register-resident chain (chain.h) and the per-column link kernels are two
    implementations of the same reference-order MGS: identical up to the order of the partial sums
    (1e-13), for even n (chain) and odd n (chain not eligible -> link kernels on both sides),
    plain, double (dmgs), Jacobi-preconditioned and Lanczos steps."""
    nx, ny = shape
    A = ref.laplace2d(nx, ny)
    n = A.shape[0]
    b = np.random.default_rng(5).standard_normal(n)
    d = np.linspace(0.5, 1.5, n)
    m = 24
    results = []
    for chain in (True, False):
        ctx = _second_context(chain)
        Ad, Md = ctx.csr(A), ctx.diag(d)
        out = {}
        for name, sweeps, use_m, lanczos in (("mgs", 1, False, False), ("dmgs", 2, False, False),
                                              ("mgsM", 1, True, False), ("lanczos", 1, False, True)):
            V = ctx.alloc(n, m + 1)
            P = ctx.alloc(n, m + 1) if use_m else None
            W = ctx.alloc(n, 2)
            v0 = b / np.linalg.norm(b)
            if use_m:
                nrm = np.sqrt(np.dot(b, d * b))
                P.upload(0, b / nrm)
                V.upload(0, d * b / nrm)
            else:
                V.upload(0, v0)
            H = np.zeros((m + 1, m))
            for k in range(m):
                start = k if lanczos else 0
                hk = float(H[k, k - 1]) if (lanczos and k > 0) else 0.0
                hcol = ctx.arnoldi_step(Ad, Md if use_m else None, V, P, W, 0, k, start, sweeps, 0, hk)
                H[start: k + 2, k] = hcol[start: k + 2]
            out[name] = (H, V.download())
        results.append(out)
        ctx.close()
    for name in results[0]:
        Hc, Vc = results[0][name]
        Hl, Vl = results[1][name]
        assert np.linalg.norm(Hc - Hl) < 1e-12 * np.linalg.norm(Hl), name
        assert np.linalg.norm(Vc - Vl) < 1e-10, name
    # and both agree with the oracle
    st = ref.arnoldi_init(A, b, m)
    for _ in range(m):
        ref.arnoldi_step(st)
    assert np.linalg.norm(results[0]["mgs"][0] - st.H) < 1e-11 * np.linalg.norm(st.H)


@pytest.mark.parametrize("n", [3000, 4100, 20000, 99856, 130000, 250000, 700000])
def test_column_ring_kernel_for_short_vectors(hip, n):
    """k_mgs_chain_small (chain.h): vectors of 4 / 8 double2 rows per lane without a preconditioner keep a ring of whole
    basis columns in registers, requested several links ahead; on one XCD a ninth wave per workgroup does the sums.  Same
    partials, same order of additions as the general chain kernels - the same bits wherever both run the same geometry
    (everything but 1.3e5 < N <= 2.6e5, where the general kernel runs 8 rows per lane on one XCD and this one 4 rows on
    the whole chip) - for the operator in the prologue (padded blocks, N >= 4096) and a separate SpMV (masked blocks,
    a CSR operator that is not banded, with two double sweeps); a faked timeout inside it is recovered; the Arnoldi
    relation holds and the oracle's H is reproduced."""
    from krypy_amd import _hip

    nx = 100
    A = ref.laplace2d(nx, n // nx) if n % nx == 0 else _banded(n, (-700, -1, 0, 1, 700), 3)
    Ar = (A + sp.diags(np.random.default_rng(2).standard_normal(n - 37) * 0.01, 37, shape=(n, n))).tocsr()   # 6 diagonals: no banded copy
    v = np.random.default_rng(4).standard_normal(n)
    m = 14
    ctx = _hip.Context(0)
    ctx.set("chain_blk", 0)      # (steps with eight or more links would go to the blocked kernel, which is not a bit-for-bit one: tests/test_gpu_blocked.py)
    chain_configured = ctx.get("chain") == 1      # (KRYPY_AMD_MGS_CHAIN=0: the link kernels everywhere, except behind this test's own re-arming)
    res = {}
    for small in (1, 0):
        ctx.set("chain_small", small)
        c0 = ctx.get("n_chain_small")
        for name, mat in (("prologue", A), ("spmv", Ar)):
            Ad = ctx.csr(mat)
            V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
            V.upload(0, v / np.linalg.norm(v))
            H = np.zeros((m + 1, m))
            for k in range(m):
                if small and name == "spmv" and k == 9:
                    ctx.set("chain_fault", 1)          # this launch reports a timeout: re-run on the link kernels
                hcol = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if (name == "spmv" and k in (5, 6)) else 1, 0, 0.0)
                H[: k + 2, k] = hcol[: k + 2]
                if small and name == "spmv" and k == 9:
                    expect_kernel(ctx.get("chain") == 0, "the recovery switches the chain off")       # (back on for the rest)
                    ctx.set("chain", 1)
            res[small, name] = (H, V.download())
        used = ctx.get("n_chain_small") - c0
        expect_kernel((used >= 2 * m - 1) == bool(small), "k_mgs_chain_small launches: %r" % ((small, used),))
    ctx.close()
    same_geometry = not (131072 < n <= 262144)
    for name, mat in (("prologue", A), ("spmv", Ar)):
        Hs, Vs = res[1, name]
        Hg, Vg = res[0, name]
        if same_geometry and name == "prologue" and chain_configured:
            assert np.array_equal(Hs, Hg), name
            assert np.array_equal(Vs, Vg), name
        else:            # (other partial sums, or - "spmv" - one step of the ring run on the link kernels)
            assert np.linalg.norm(Hs - Hg) < 1e-12 * np.linalg.norm(Hg), name
            assert np.linalg.norm(Vs - Vg) < 1e-10, name
        assert np.linalg.norm(mat.dot(Vs[:, :m]) - Vs.dot(Hs)) < 1e-12 * np.linalg.norm(Hs), name
    if n <= 100000:
        st = ref.arnoldi_init(A, v, m)
        for _ in range(m):
            ref.arnoldi_step(st)
        assert np.linalg.norm(res[1, "prologue"][0] - st.H) < 1e-11 * np.linalg.norm(st.H)


def test_completion_tags_replace_the_per_step_event(hip):
    """The last kernel of a chained Arnoldi step writes a completion tag behind the H column in pinned memory and
    kh_arnoldi_step_end polls it; steps whose last kernel does not write one (link kernels, panel Gram-Schmidt) keep
    their event.  Same H columns either way (bit for bit), with look-ahead (several steps in flight), through a faked
    timeout (the re-run step waits for its event) and for a whole solve."""
    from krypy_amd import _hip, linsys

    A = ref.laplace2d(300, 200)
    n = A.shape[0]
    v = np.random.default_rng(8).standard_normal(n)
    m = 12
    res = {}
    for tag in (1, 0):
        ctx = _hip.Context(0)
        ctx.set("tag_wait", tag)
        Ad = ctx.csr(A)
        V, W = ctx.alloc(n, m + 2), ctx.alloc(n, 2)
        V.upload(0, v / np.linalg.norm(v))
        H = np.zeros((m + 2, m + 1))
        w0 = ctx.get("n_tag_waits")
        ctx.arnoldi_step_begin(Ad, None, V, None, W, 0, 0, 0, 1, 0, 0.0, 0)
        for k in range(m):                     # step k + 1 is in flight while step k is collected
            if k == 6:
                ctx.set("chain_fault", 1)
            ctx.arnoldi_step_begin(Ad, None, V, None, W, 0, k + 1, 0, 1, 1 if k == 9 else 0, 0.0, (k + 1) % 4)
            H[: k + 2, k] = ctx.arnoldi_step_end(k % 4, k + 2)
            if ctx.get("chain") == 0:
                ctx.set("chain", 1)
        H[: m + 2, m] = ctx.arnoldi_step_end(m % 4, m + 2)
        waits = ctx.get("n_tag_waits") - w0
        expect_kernel((waits >= m - 3) if tag else (waits == 0), "tag waits: %r" % ((tag, waits),))
        res[tag] = (H, V.download())
        ctx.close()
    # (the step behind the faked timeout is re-run on the link kernels, the one after it consumed its garbage and is
    # re-run too: same launches in both runs)
    assert np.array_equal(res[1][0], res[0][0])
    assert np.array_equal(res[1][1], res[0][1])
    Hm, Vm = res[1]
    assert np.linalg.norm(A.dot(Vm[:, : m + 1]) - Vm.dot(Hm)) < 1e-12 * np.linalg.norm(Hm)


def test_gmres_cycle_in_c_equals_the_per_step_loop(hip, monkeypatch):
    """kh_gmres_cycle (Arnoldi steps with look-ahead on the device, Givens QR and the residual recurrence on the host
    in C: linsys.py:951-997 in one call) against the per-step Python loop (KRYPY_AMD_GMRES_CYCLE=0): same iteration
    counts, same residual histories, same H / R / iterate - converging solves (the call stops at the tolerance), an
    exhausted maxiter (the last iteration is the Python loop's), restarts, Jacobi, the panel Gram-Schmidt, and a solve
    that finds an invariant subspace (the C loop hands that step back)."""
    from oracle.inputs import toy_system

    A, b = lap2d_system(70, rhs="rng1")
    d = A.diagonal()
    M = sp.diags(1.0 / d).tocsr()
    At, bt = toy_system()[:2]
    cases = [("converge", lambda: linsys.Gmres(linsys.LinearSystem(A, b), tol=1e-9, maxiter=400, store_arnoldi=True)),
             ("jacobi cgs2", lambda: linsys.Gmres(linsys.LinearSystem(A, b, M=M), tol=1e-9, maxiter=400, ortho="cgs2")),
             ("maxiter", lambda: linsys.Gmres(linsys.LinearSystem(A, b), tol=1e-12, maxiter=30)),
             ("restarted", lambda: linsys.RestartedGmres(linsys.LinearSystem(A, b), tol=1e-8, maxiter=25, max_restarts=40)),
             ("invariant", lambda: linsys.Gmres(linsys.LinearSystem(sp.diags(np.r_[np.ones(50), 2 * np.ones(50)]).tocsr(),
                                                                    np.ones(100)), tol=1e-12, maxiter=50)),
             ("toy", lambda: linsys.Gmres(linsys.LinearSystem(At, bt), tol=1e-5)),
             # default maxiter = N: the basis starts small and doubles on demand - the C call runs up to the last
             # column there is, Arnoldi.advance grows the blocks, the next call takes over
             ("growing", lambda: linsys.Gmres(linsys.LinearSystem(A, b), tol=1e-9))]

    def run(make):
        try:
            return make(), False
        except utils.ConvergenceError as e:
            return e.solver, True

    for name, make in cases:
        if name == "growing":
            monkeypatch.setattr(utils.Arnoldi, "_max_initial_cols", 8)
        c0 = hip.get("n_cycle_steps")
        s1, f1 = run(make)
        used = hip.get("n_cycle_steps") - c0
        monkeypatch.setenv("KRYPY_AMD_GMRES_CYCLE", "0")
        c1 = hip.get("n_cycle_steps")
        s0, f0 = run(make)
        expect_kernel(hip.get("n_cycle_steps") == c1, "hip.get(\"n_cycle_steps\") == c1")
        monkeypatch.delenv("KRYPY_AMD_GMRES_CYCLE")
        expect_kernel(used > 0, "used > 0: %r" % (name,))
        assert f1 == f0 and len(s1.resnorms) == len(s0.resnorms), (name, len(s1.resnorms), len(s0.resnorms))
        r1, r0 = np.array(s1.resnorms), np.array(s0.resnorms)
        if name == "restarted":      # open loop over restarts: last-bit differences of the two Givens generators grow
            assert np.max(np.abs(r1[:26] - r0[:26]) / r0[:26]) < 1e-11, name       # (SURVEY 0); the first cycle is tight
            assert np.max(np.abs(r1[:-1] - r0[:-1]) / r0[:-1]) < 1e-6, name
            assert np.linalg.norm(s1.xk - s0.xk) <= 1e-7 * np.linalg.norm(s0.xk), name
            continue
        assert np.max(np.abs(r1[:-1] - r0[:-1]) / r0[:-1]) < 1e-11, name
        assert np.linalg.norm(s1.xk - s0.xk) <= 1e-10 * np.linalg.norm(s0.xk), name
        if name in ("converge", "maxiter", "toy"):
            # round 4: the C loop calls the host layer's own BLAS drotg (kh_ctx_set_rotg) - the two loops produce the
            # same bits, not just the same numbers
            assert np.array_equal(r1[:-1], r0[:-1]), name
            assert np.array_equal(s1.R, s0.R), name
        if hasattr(s1, "R") and hasattr(s0, "R"):
            k = len(r1) - 1
            assert np.linalg.norm(s1.R[:k, :k] - s0.R[:k, :k]) <= 1e-12 * np.linalg.norm(s0.R[:k, :k]), name
        if name == "converge":
            assert np.linalg.norm(s1.H - s0.H) <= 1e-12 * np.linalg.norm(s0.H)
            assert np.linalg.norm(s1.V - s0.V) <= 1e-10
        if name == "invariant":
            assert s1.arnoldi.invariant and s0.arnoldi.invariant and len(r1) == 3
    # a timeout of the chain kernel's reduction inside the C loop: the step is re-run on the per-column kernels by
    # kh_arnoldi_step_end, the look-ahead step behind it too - same solve
    try:
        good, _ = run(cases[0][1])
        before = hip.get("n_chain_recovered")
        hip.set("chain_fault", 1)
        bad, _ = run(cases[0][1])
        expect_kernel(hip.get("n_chain_recovered") > before, "hip.get(\"n_chain_recovered\") > before")
    finally:
        hip.set("chain", 1)
    assert len(bad.resnorms) == len(good.resnorms)
    assert np.max(np.abs(np.array(bad.resnorms[:-1]) - np.array(good.resnorms[:-1])) / np.array(good.resnorms[:-1])) < 1e-9


def test_cg_cycle_in_c_equals_the_per_step_loop(hip, monkeypatch):
    """kh_cg_cycle (iterations of the fused CG step with omega, rho and the convergence test formed in C: linsys.py:622-690
    in one call) against the per-step Python loop (KRYPY_AMD_CG_CYCLE=0): the same residual history, rhos and trace BIT FOR
    BIT, the same iterate - sparse and dense operators, Jacobi, an exhausted maxiter, short and long vectors."""
    from oracle.inputs import dense_spd_system

    A, b = lap2d_system(70, rhs="rng1")
    M = sp.diags(1.0 / A.diagonal()).tocsr()
    A2, b2 = lap2d_system(700, rhs="rng1")
    Ad, bd = dense_spd_system(512)
    hpd = dict(self_adjoint=True, positive_definite=True)
    cases = [("sparse", lambda: linsys.Cg(linsys.LinearSystem(A, b, **hpd), tol=1e-10, maxiter=600)),
             ("jacobi", lambda: linsys.Cg(linsys.LinearSystem(A, b, M=M, **hpd), tol=1e-10, maxiter=600)),
             ("long", lambda: linsys.Cg(linsys.LinearSystem(A2, b2, **hpd), tol=1e-30, maxiter=120)),
             ("dense", lambda: linsys.Cg(linsys.LinearSystem(Ad, bd, **hpd), tol=1e-9, maxiter=200)),
             ("maxiter", lambda: linsys.Cg(linsys.LinearSystem(A, b, **hpd), tol=1e-13, maxiter=30))]

    def run(make):
        try:
            return make()
        except utils.ConvergenceError as e:
            return e.solver

    for name, make in cases:
        c0 = hip.get("n_cg_cycle_steps")
        s1 = run(make)
        used = hip.get("n_cg_cycle_steps") - c0
        monkeypatch.setenv("KRYPY_AMD_CG_CYCLE", "0")
        c1 = hip.get("n_cg_cycle_steps")
        s0 = run(make)
        expect_kernel(hip.get("n_cg_cycle_steps") == c1, "hip.get(\"n_cg_cycle_steps\") == c1")
        monkeypatch.delenv("KRYPY_AMD_CG_CYCLE")
        expect_kernel(used > 0, "used > 0: %r" % (name,))
        assert s1.resnorms == s0.resnorms and s1.rhos == s0.rhos and s1.iter == s0.iter, name
        assert list(s1.cg_trace) == list(s0.cg_trace), name
        assert np.array_equal(s1.xk, s0.xk), name


def test_minres_cycle_in_c_equals_the_per_step_loop(hip, monkeypatch):
    """kh_minres_cycle (Lanczos steps with look-ahead on the device; the symmetric fill, the QR update with the two
    remembered rotations, the rotated right-hand side and the deferred recurrence updates on the host in C:
    linsys.py:791-853 in one call) against the per-step Python loop (KRYPY_AMD_MINRES_CYCLE=0): same iteration counts,
    the same residual history and Lanczos matrix BIT FOR BIT (both loops call the same BLAS drotg), the same iterate -
    converging solves with and without Jacobi, a run through several slides of the basis window, an exhausted maxiter,
    a stored basis that grows on demand, an invariant subspace (the C loop hands that step back), short (one-launch
    three-pass kernel) and padded long vectors."""
    A, b = lap2d_system(70, rhs="rng1")
    M = sp.diags(1.0 / A.diagonal()).tocsr()
    A2, b2 = lap2d_system(300, rhs="rng1")
    M2 = sp.diags(1.0 / A2.diagonal()).tocsr()
    D = sp.diags(np.r_[np.ones(50), 2 * np.ones(50)]).tocsr()
    cases = [("converge", lambda: linsys.Minres(linsys.LinearSystem(A, b, self_adjoint=True), tol=1e-9, maxiter=600)),
             ("jacobi", lambda: linsys.Minres(linsys.LinearSystem(A, b, M=M, self_adjoint=True), tol=1e-9, maxiter=600)),
             ("window", lambda: linsys.Minres(linsys.LinearSystem(A2, b2, M=M2, self_adjoint=True), tol=1e-30, maxiter=200)),
             ("maxiter", lambda: linsys.Minres(linsys.LinearSystem(A, b, self_adjoint=True), tol=1e-13, maxiter=30)),
             ("stored", lambda: linsys.Minres(linsys.LinearSystem(A, b, self_adjoint=True), tol=1e-9, maxiter=600,
                                              store_arnoldi=True)),
             ("invariant", lambda: linsys.Minres(linsys.LinearSystem(D, np.ones(100), self_adjoint=True), tol=1e-12,
                                                 maxiter=50))]

    def run(make):
        try:
            return make(), False
        except utils.ConvergenceError as e:
            return e.solver, True

    for name, make in cases:
        if name == "stored":
            monkeypatch.setattr(utils.Arnoldi, "_max_initial_cols", 8)
        c0 = hip.get("n_minres_cycle_steps")
        s1, f1 = run(make)
        used = hip.get("n_minres_cycle_steps") - c0
        monkeypatch.setenv("KRYPY_AMD_MINRES_CYCLE", "0")
        c1 = hip.get("n_minres_cycle_steps")
        s0, f0 = run(make)
        expect_kernel(hip.get("n_minres_cycle_steps") == c1, "hip.get(\"n_minres_cycle_steps\") == c1")
        monkeypatch.delenv("KRYPY_AMD_MINRES_CYCLE")
        if name == "stored":
            monkeypatch.undo()
        expect_kernel(used > 0, "used > 0: %r" % (name,))
        assert f1 == f0 and len(s1.resnorms) == len(s0.resnorms), (name, len(s1.resnorms), len(s0.resnorms))
        if name == "window":
            assert len(s1.resnorms) == 201
            expect_kernel(used >= 190, "kh_minres_cycle steps >= 190: %r" % (used,))
        r1, r0 = np.array(s1.resnorms), np.array(s0.resnorms)
        assert np.array_equal(r1[:-1], r0[:-1]), (name, np.max(np.abs(r1[:-1] - r0[:-1]) / r0[:-1]))
        assert abs(r1[-1] - r0[-1]) <= 1e-9 * max(r0[-1], 1e-300) or r0[-1] < 1e-8, name       # (the explicit residual a solve ends with)
        k = s1.lanczos.iter
        assert k == s0.lanczos.iter
        assert np.array_equal(s1.lanczos.H[: k + 1, :k], s0.lanczos.H[: k + 1, :k]), name
        assert np.linalg.norm(s1.xk - s0.xk) <= 1e-13 * np.linalg.norm(s0.xk), name
        if name == "stored":
            assert np.array_equal(s1.V, s0.V)
        if name == "invariant":
            assert s1.lanczos.invariant and s0.lanczos.invariant


@pytest.mark.parametrize("n", [3000, 14400, 65538, 100000, 210000, 262144])
def test_short_vectors_run_on_one_xcd(hip, n):
    """4 ... 32 workgroups: a Gram-Schmidt link is its grid-wide sum.  The ONEX instantiations of the chain kernels put
    all working workgroups on one XCD, where the sum is an L2 round trip (chain.h).  Same partials, same order of the
    additions: H and the basis bit for bit as with the workgroups spread over the eight XCDs - plain, double sweep,
    Jacobi (plain kernel), Lanczos, complex; masked (n < 65536) and padded blocks."""
    from krypy_amd import _hip

    nx = 100
    A = ref.laplace2d(nx, n // nx) if n % nx == 0 else _banded(n, (-700, -1, 0, 1, 700), 3)
    rng = np.random.default_rng(4)
    v = rng.standard_normal(n)
    dj = np.linspace(0.5, 1.5, n)
    Az = (A + 0.3j * sp.diags(rng.standard_normal(n))).tocsr()
    vz = v + 1j * rng.standard_normal(n)
    m = 12
    ctx = _hip.Context(0)
    out = []
    for onex in (1, 0):
        ctx.set("chain_onex", onex)
        c0 = ctx.get("n_chain_onex")
        res = {}
        Ad, Md, Azd = ctx.csr(A), ctx.diag(dj), ctx.csr(Az)
        for name, use_m, lanczos, cplx in (("mgs", False, False, False), ("jacobi", True, False, False),
                                           ("lanczos", False, True, False), ("complex", False, False, True)):
            dt = complex if cplx else float
            V, W = ctx.alloc(n, m + 1, dtype=dt), ctx.alloc(n, 2, dtype=dt)
            P = ctx.alloc(n, m + 1) if use_m else None
            v0 = vz if cplx else v
            if use_m:
                nrm = np.sqrt(np.dot(v, dj * v))
                P.upload(0, v / nrm)
                V.upload(0, dj * v / nrm)
            else:
                V.upload(0, v0 / np.linalg.norm(v0))
            H = np.zeros((m + 1, m), dtype=dt)
            for k in range(m):
                start = k if lanczos else 0
                hk = float(H[k, k - 1].real) if (lanczos and k > 0) else 0.0
                hcol = ctx.arnoldi_step(Azd if cplx else Ad, Md if use_m else None, V, P, W, 0, k, start,
                                        2 if (k == 5 and not lanczos and not cplx) else 1, 0, hk)
                H[start: k + 2, k] = hcol[start: k + 2]
            res[name] = (H, V.download())
        used = ctx.get("n_chain_onex") - c0
        expect_kernel((used > 0) == bool(onex), "(used > 0) == bool(onex): %r" % ((onex, used),))
        out.append(res)
    ctx.close()
    for name in out[0]:
        if n * (2 if name == "complex" else 1) <= 131072 and name != "lanczos":      # the same 4 rows per lane either way: the same bits
            assert np.array_equal(out[0][name][0], out[1][name][0]), name
            assert np.array_equal(out[0][name][1], out[1][name][1]), name
        elif name == "lanczos":       # (one link per step: never on one XCD - the same kernel both times)
            assert np.array_equal(out[0][name][0], out[1][name][0]), name
            assert np.array_equal(out[0][name][1], out[1][name][1]), name
        else:                    # 8 rows per lane on <= 32 workgroups against 4 rows on twice as many: other partial sums
            assert np.linalg.norm(out[0][name][0] - out[1][name][0]) < 1e-12 * np.linalg.norm(out[1][name][0]), name
            assert np.linalg.norm(out[0][name][1] - out[1][name][1]) < 1e-10, name
    Hf, Vf = out[0]["mgs"]
    assert np.linalg.norm(A.dot(Vf[:, :m]) - Vf.dot(Hf)) < 1e-12 * np.linalg.norm(Hf)


@pytest.mark.parametrize("rows", [8, 16, 24, 32, 40, 48, 56])
@pytest.mark.parametrize("cplx", [False, True])
def test_mgs_chain_every_register_shape(hip, rows, cplx):
    """The chain kernel is instantiated for 4 ... 40 `double2` rows of w per lane; the vector length
    picks the instantiation.  One size per shape (real and complex, i.e. the LDS-parking kernel in both
    flavours): chain == per-column link kernels up to the order of the partial sums, and the
    Arnoldi relation A V_m = V_{m+1} H holds."""
    if cplx and rows > 40:
        pytest.skip("48 / 56 rows per lane (part of w in LDS) exist for real vectors")
    # 48 / 56: vectors beyond what the register file holds (N = 12 M / 14 M: config 5's 12.5 M-row shards)
    n2 = {8: 700_000, 16: 1_500_000, 24: 2_800_000, 32: 3_900_000, 40: 5_000_000, 48: 6_000_000,
          56: 7_000_000}[rows]   # double2 per vector
    n = n2 if cplx else 2 * n2                     # complex entries are one double2 each
    rng = np.random.default_rng(rows)
    A = sp.diags([np.full(n - 1, -1.0), np.linspace(2.0, 3.0, n), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr()
    if cplx:
        A = (A + sp.diags(1j * np.linspace(0.1, 0.5, n))).tocsr()
    b = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0)
    m = 5
    dt = complex if cplx else float
    res = []
    for chain in (True, False):
        ctx = _second_context(chain)
        Ad = ctx.csr(A)
        V, W = ctx.alloc(n, m + 1, dtype=dt), ctx.alloc(n, 2, dtype=dt)
        V.upload(0, b / np.linalg.norm(b))
        H = np.zeros((m + 1, m), dtype=dt)
        for k in range(m):
            H[: k + 2, k] = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if k == 3 else 1, 0)
        expect_kernel(ctx.counters()["chain"] == (m if chain else 0), "ctx.counters()[\"chain\"] == (m if chain else 0)")      # the one-launch kernel really ran
        res.append((H, V.download()))
        ctx.close()
    (Hc, Vc), (Hl, Vl) = res
    assert np.linalg.norm(Hc - Hl) < 1e-12 * np.linalg.norm(Hl)
    assert np.linalg.norm(Vc - Vl) < 1e-11 * np.sqrt(m)
    assert np.linalg.norm(A.dot(Vc[:, :m]) - Vc.dot(Hc)) < 1e-12 * np.linalg.norm(Hc)
    G = Vc.conj().T.dot(Vc)
    assert np.linalg.norm(G - np.eye(m + 1)) < 1e-12
    # the register-resident panel kernels (k_cgs_dots / k_cgs_update, real and - since round 2 - complex) have their
    # own batch shape per instantiation: same check against the chunked panel kernels, odd and even column counts
    res = []
    for chain in (True, False):
        ctx = _second_context(chain)
        Ad = ctx.csr(A)
        V, W = ctx.alloc(n, m + 1, dtype=dt), ctx.alloc(n, 2, dtype=dt)
        V.upload(0, b / np.linalg.norm(b))
        H = np.zeros((m + 1, m), dtype=dt)
        for k in range(m):
            H[: k + 2, k] = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if k == 3 else 1, 1)
        expect_kernel((ctx.counters()["cgs_register"] > 0) == chain, "(ctx.counters()[\"cgs_register\"] > 0) == chain")
        res.append((H, V.download()))
        ctx.close()
    (Hc, Vc), (Hl, Vl) = res
    assert np.linalg.norm(Hc - Hl) < 1e-12 * np.linalg.norm(Hl)
    assert np.linalg.norm(Vc - Vl) < 1e-11 * np.sqrt(m)
    assert np.linalg.norm(A.dot(Vc[:, :m]) - Vc.dot(Hc)) < 1e-12 * np.linalg.norm(Hc)


def test_multi_rank_code_path_on_one_gpu(hip):
    """The code path libkrylov_hip takes on N > 1 GPUs (partial sums -> k_reduce_partials -> device
    scalar -> ncclAllReduce -> consumer kernels reading the scalar; halo exchange hook; panel
    all-reduces) exercised with a 1-rank RCCL communicator in forced mode: results must equal the
    single-GPU path.  The 2-rank sharding logic itself is covered by tests/test_dist_gloo.py."""
    import os
    from krypy_amd import _hip, dist as kdist

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    A, b = lap2d_system(90, rhs="rng1")
    n = A.shape[0]
    op = kdist.ShardedCSROperator(A, 0, n, ctx)       # one slab = the whole matrix, empty halos
    assert op.halo == (0, 0, 0, 0)
    Ad = op._device_matrix()
    m = 20
    for ortho, gs, sweeps in (("mgs", 0, 1), ("dmgs", 0, 2), ("mgs", 1, 1), ("mgs", 1, 2)):
        st = ref.arnoldi_init(A, b, m, ortho=ortho)
        V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
        V.upload(0, st.V[:, :1])
        for k in range(m):
            ref.arnoldi_step(st)
            hcol = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, sweeps, gs)
            assert np.linalg.norm(hcol - st.H[: k + 2, k]) < 1e-12 * np.linalg.norm(hcol), (ortho, gs, k)
        assert np.linalg.norm(V.download() - st.V) < 1e-10
    x = np.random.default_rng(3).standard_normal((n, 1))
    X, R, B = ctx.upload(x), ctx.alloc(n, 1), ctx.upload(b)
    nrm = ctx.residual(Ad, B, 0, X, 0, R, 0)
    want = b - A.dot(x[:, 0])
    assert np.array_equal(R.download()[:, 0], want) and abs(nrm - np.linalg.norm(want)) < 1e-13 * nrm
    assert abs(ctx.nrm2(X, 0) - np.linalg.norm(x)) < 1e-14 * np.linalg.norm(x)
    Z, YK = ctx.alloc(n, 1), ctx.alloc(n, 1)
    rho = ctx.cg_update(0.5, X, 0, B, 0, YK, 0, R, 0, None, None, 0)
    r2 = want - 0.5 * b
    assert abs(rho - np.dot(r2, r2)) < 1e-12 * rho
    # the sharded SpMV runs as two launches - interior row blocks while the halo exchange is on the
    # communication stream, boundary blocks after it (forced mode on one rank: a one-block boundary at both ends)
    # - for the banded and for the CSR-stream kernel: the same bits as one launch
    expect_kernel(ctx.get("n_spmv_split") > 0, "ctx.get(\"n_spmv_split\") > 0")
    A2 = ref.laplace2d(300, 211)
    op2 = kdist.ShardedCSROperator(A2, 0, A2.shape[0], ctx)
    x2 = np.random.default_rng(8).standard_normal((A2.shape[0], 1))
    X2, Y2 = ctx.upload(x2), ctx.alloc(A2.shape[0], 1)
    for dia in (1, 0):
        ctx.set("spmv_dia", dia)
        for split in (1, 0):
            ctx.set("spmv_split", split)
            before = ctx.get("n_spmv_split")
            ctx.apply(op2._device_matrix(), X2, 0, Y2, 0, 1)
            expect_kernel((ctx.get("n_spmv_split") - before) == split, "(ctx.get(\"n_spmv_split\") - before) == split")
            assert np.array_equal(Y2.download(), A2.dot(x2)), (dia, split)
    ctx.set("spmv_dia", 1)
    ctx.set("spmv_split", 1)
    ctx.close()


def test_sharded_deflated_gmres_through_rccl_path_on_one_gpu(hip):
    """DeflatedGmres / Minres / Cg on a ShardedCSROperator with a 1-rank communicator in forced
    multi-rank mode (every reduction goes through k_reduce_partials + ncclAllReduce, the projector's
    W^T z panels included) equal the plain single-GPU solves."""
    import os
    from krypy_amd import _hip, deflation, dist as kdist, linsys

    A, b = lap2d_system(60, rhs="rng1")
    n = A.shape[0]
    U = np.random.default_rng(4).standard_normal((n, 6))
    ls0 = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)
    want = (deflation.DeflatedGmres(ls0, U=U, tol=1e-9, maxiter=300, store_arnoldi=True),
            linsys.Minres(ls0, tol=1e-9, maxiter=600), linsys.Cg(ls0, tol=1e-9, maxiter=600))
    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    old = _hip._install_context_for_testing(ctx)
    try:
        op = kdist.ShardedCSROperator(A, 0, n, ctx)
        ls = linsys.LinearSystem(op, b, self_adjoint=True, positive_definite=True)
        got = (deflation.DeflatedGmres(ls, U=U, tol=1e-9, maxiter=300, store_arnoldi=True),
               linsys.Minres(ls, tol=1e-9, maxiter=600), linsys.Cg(ls, tol=1e-9, maxiter=600))
        for g, w in zip(got, want):
            assert len(g.resnorms) == len(w.resnorms)
            assert np.allclose(g.resnorms[:40], w.resnorms[:40], rtol=1e-9)
            assert np.linalg.norm(g.xk - w.xk) < 1e-8 * np.linalg.norm(w.xk)
        assert np.allclose(got[0].E, want[0].E, rtol=1e-10, atol=1e-12)
        # (an unrestarted 211-step MGS drifts between two summation orders: compare the early part)
        assert np.allclose(got[0].C[:, :40], want[0].C[:, :40], rtol=1e-8, atol=1e-10)
    finally:
        _hip._install_context_for_testing(old)
        ctx.close()


def test_reference_solver_matrix(hip):
    """All 13,216 solves of the reference's solver test matrix (6 matrices, 3 of them complex, x inner
    products x right-hand sides x preconditioners x solvers x parameters) against the reference's own
    outcomes (tests/golden/solver_matrix.npz)."""
    stats = pcc.case_reference_solver_matrix()
    assert stats["n"] == 13216


def test_reference_deflation_matrix(hip):
    """The reference's deflated-solver test matrix (576 solves, real + complex) against its recorded
    outcomes and the E / C / B_ / Ritz identities of test/test_deflation.py."""
    assert pcc.case_reference_deflation_matrix()["n"] == 576


@pytest.mark.parametrize("case", pcu.CASES, ids=lambda f: f.__name__)
def test_reference_utils_matrix(hip, case):
    """The reference's utils test matrix (test/test_utils.py: House, Givens, Projection, qr, angles,
    hegedus, Arnoldi in every ortho mode, Ritz pairs) on real AND complex matrices."""
    assert case() > 20


@pytest.mark.parametrize("shape", [(1024, 1024), (1500, 1111), (4100, 777), (2048, 64), (1030, 63)])
def test_dense_panel_apply_on_matrix_cores(hip, shape):
    """kh_apply of a dense operator to a panel (k_gemm_dense_mfma, v_mfma_f64_16x16x4_f64): ragged row
    and column counts, 2..33 columns, an asymmetric panel (a transposed tile write would show)."""
    n, m = shape
    rng = np.random.default_rng(n + m)
    A = rng.standard_normal((n, m))
    Ad = hip.dense(A)
    for nc in (2, 5, 16, 17, 33):
        X = rng.standard_normal((m, nc)) * np.arange(1, nc + 1)
        Xd = hip.upload(X)
        Yd = hip.alloc(n, nc + 1)
        Yd.upload(0, np.full((n, nc + 1), 7.0))
        hip.apply(Ad, Xd, 0, Yd, 0, nc)
        got = Yd.download()
        want = A.dot(X)
        assert np.allclose(got[:, :nc], want, rtol=1e-12, atol=1e-11 * np.abs(want).max())
        assert np.all(got[:, nc] == 7.0)          # the column behind the panel is untouched
    # the single-column GEMV and the panel kernel agree
    x = rng.standard_normal((m, 2))
    Y1, Y2 = hip.alloc(n, 2), hip.alloc(n, 2)
    Xd = hip.upload(x)
    hip.apply(Ad, Xd, 0, Y1, 0, 2)
    hip.apply(Ad, Xd, 0, Y2, 0, 1)
    hip.apply(Ad, Xd, 1, Y2, 1, 1)
    assert np.allclose(Y1.download(), Y2.download(), rtol=1e-12, atol=1e-11)


def test_abi_fuzz_random_calls_against_numpy(hip):
    """tools/abi_fuzz.py, 40 fixed seeds: every vector / operator entry on random windows of random blocks (bit-level
    for the CSR / banded SpMV and SpMM), and every fused entry (Arnoldi / Lanczos step with each option, residual,
    MINRES / CG recurrences, projector) against the NumPy restatement of its semantics, real and complex."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import abi_fuzz
    from tests.support.numpy_context import NumpyContext
    dbl = NumpyContext()
    total = 0
    for seed in range(40):
        total += abi_fuzz.one_round(hip, seed, 200_000)
        total += abi_fuzz.step_round(hip, dbl, seed, 200_000)
        total += abi_fuzz.shard_round(hip, seed)      # block-row shards with hand-written ghost entries, bit for bit
        total += abi_fuzz.cycle_round(hip, dbl, seed)  # kh_gmres_cycle against step-by-step + NumPy Givens, deferred MINRES update
        total += abi_fuzz.minres_cycle_round(hip, dbl, seed)   # kh_minres_cycle against the NumPy restatement of its contract
    assert total > 1000


def test_solve_fuzz_random_solves_against_the_oracle(hip):
    """tools/solve_fuzz.py, 16 fixed seeds: CG / MINRES / GMRES on random banded systems of random size with random
    preconditioners, initial guesses, tolerances and Gram-Schmidt modes - same iteration count and residual history
    as oracle/krylov_ref.py, up to the oracle's own movement under a last-bit perturbation of b."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import solve_fuzz
    for seed in range(16):
        solve_fuzz.one_solve(seed)
        solve_fuzz.one_solve_extra(seed)        # complex CG / MINRES / GMRES (oracle/krylov_ref_c.py), deflated GMRES


@pytest.mark.parametrize("kind,n", [("band", 400_000), ("ragged", 150_000)])
def test_general_csr_operators_of_the_bench_tool(hip, kind, n):
    """The two non-stencil operators tools/bench_configs.py times at N = 5e6 / 2e6 (about nine entries per row at random
    places in a band; 3 ... 40 entries per row), here at a size the oracle handles in seconds: no banded copy, the
    CSR-stream kernel bit-identical to scipy's csr_matvec (utils.py:1593-1594), and a GMRES(40) cycle - SpMV launch +
    chain kernel, w through HBM - against the CPU oracle at 1e-10."""
    import importlib.util
    import os
    from krypy_amd import linsys, utils

    spec = importlib.util.spec_from_file_location(
        "bench_configs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_configs.py"))
    bc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bc)
    A = bc.general_csr(kind, n)
    b = np.random.default_rng(0).standard_normal(n)
    Ad = hip.csr(A)
    assert Ad.diagonals == 0
    X, Y = hip.upload(b), hip.alloc(n, 1)
    # the row blocks of "band" touch ~2,100 neighbouring columns each: x comes from an LDS window loaded once per block
    # (k_spmv_stream<.., WIN>); "ragged" spans +-20,000 columns: gathers from global memory.  Both ways, the fused residual
    # and dot epilogues included: scipy's bits.
    w0 = hip.get("n_spmv_win")
    hip.apply(Ad, X, 0, Y, 0, 1)
    assert np.array_equal(Y.download()[:, 0], A.dot(b))
    used = hip.get("n_spmv_win") - w0
    R, B = hip.alloc(n, 1), hip.upload(np.arange(n) % 7 - 3.0)
    nrm = hip.residual(Ad, B, 0, X, 0, R, 0)
    assert np.array_equal(R.download()[:, 0], (np.arange(n) % 7 - 3.0) - A.dot(b)) and abs(nrm - np.linalg.norm(R.download())) <= 1e-13 * nrm
    hip.set("spmv_win", 0)
    try:
        hip.apply(Ad, X, 0, Y, 0, 1)
        assert np.array_equal(Y.download()[:, 0], A.dot(b))
    finally:
        hip.set("spmv_win", 1)
    expect_kernel(used == (1 if kind == "band" else 0), "products through the LDS window: %r" % (used,))
    f0 = hip.counters()["chain_fused"]
    try:
        sol = linsys.Gmres(linsys.LinearSystem(A, b), maxiter=40, tol=1e-30)
    except utils.ConvergenceError as e:
        sol = e.solver
    expect_kernel(hip.counters()["chain_fused"] == f0, "hip.counters()[\"chain_fused\"] == f0")          # (nothing to fuse: the operator is not banded)
    want = ref.gmres(A, b, tol=1e-30, maxiter=40)
    res, wres = np.array(sol.resnorms), np.array(want.resnorms)
    assert len(res) == len(wres) == 41
    # (these systems are diagonally dominant: the residual reaches rounding level within the 40 steps, where its
    # digits are noise in the oracle too - the history is compared down to 1e-6 of the right-hand side)
    live = wres > 1e-6
    assert live.sum() >= 5
    assert np.max(np.abs(res[live] - wres[live]) / wres[live]) < 1e-10
    assert np.linalg.norm(sol.xk[:, 0] - want.xk) < 1e-10 * np.linalg.norm(want.xk)


@pytest.mark.parametrize("n_rows,n_cols", [(1, 7), (5, 64), (257, 511), (1030, 1030), (4099, 2048), (16390, 1536)])
def test_dense_gemv_rows_per_wave_same_bits(hip, n_rows, n_cols):
    """k_gemv_dense<ROWS> (kernels.h): two or four rows of the operator per wave share every load of x; each row keeps its
    own accumulators and their order, so the product is bit for bit the one-row-per-wave kernel's - ragged row counts
    (a wave with fewer rows than ROWS, a workgroup with idle waves), odd column counts (the unaligned path), and the
    default choice (four rows from 16 K rows on)."""
    rng = np.random.default_rng(n_rows)
    A = rng.standard_normal((n_rows, n_cols))
    x = rng.standard_normal(n_cols)
    Ad, X, Y = hip.dense(A), hip.upload(x), hip.alloc(n_rows, 1)
    got = {}
    try:
        for rows in (1, 2, 4, 0):
            hip.set("gemv_rows", rows)
            Y.upload(0, np.full((n_rows, 1), np.nan))
            hip.apply(Ad, X, 0, Y, 0, 1)
            got[rows] = Y.download()[:, 0]
    finally:
        hip.set("gemv_rows", 0)
    assert np.allclose(got[1], A.dot(x), rtol=1e-12, atol=1e-11 * np.abs(A).max() * np.abs(x).max() * n_cols)
    for rows in (2, 4, 0):
        assert np.array_equal(got[rows], got[1]), rows


def test_dense_operator_formed_on_the_device(hip, golden):
    """bench.py --config 4 forms A = G G^T / n + I on the device (kh_apply's panel path on the FP64 matrix cores, then
    kh_dense_from_block) and hands it to the solvers as a utils.DeviceOperator - no host image.  At n = 512 (1,100 at the
    second size: not a multiple of the panel width): the operator's entries against NumPy's to rounding, its products
    against A x, and CG through it against the reference's fixture (krypy/linsys.py:593-689; cg_dense_n512: the same
    iteration count, residual history and iterate)."""
    from krypy_amd import linsys, utils

    for n in (512, 1100):
        rng = np.random.default_rng(0)
        G = rng.standard_normal((n, n))
        b = rng.standard_normal(n)
        A = G.dot(G.T) / n + np.eye(n)
        Gop = hip.dense(G)
        Gt = hip.upload(np.ascontiguousarray(G.T))
        Y = hip.alloc(n, n)
        hip.apply(Gop, Gt, 0, Y, 0, n)
        Aop = hip.dense_from_block(Y, 0, n, 1.0 / n, 1.0)
        op = utils.DeviceOperator(Aop)
        # the entries: apply to the identity, block by block
        E = hip.upload(np.eye(n))
        Z = hip.alloc(n, n)
        hip.apply(Aop, E, 0, Z, 0, n)
        got = Z.download()
        assert np.max(np.abs(got - A)) < 1e-13 * np.max(np.abs(A)) * 10
        x = rng.standard_normal((n, 3))
        assert np.allclose(op.dot(x), A.dot(x), rtol=1e-12, atol=1e-12)
        assert np.allclose(op.adj.dot(x), A.dot(x), rtol=1e-12, atol=1e-12)
        ls = linsys.LinearSystem(op, b, self_adjoint=True, positive_definite=True)
        s = linsys.Cg(ls, tol=1e-8, maxiter=200)
        assert np.linalg.norm(A.dot(s.xk[:, 0]) - b) <= 1.0001e-8 * np.linalg.norm(b)
        if n == 512:
            g = golden("cg_dense_n512")
            assert s.iter == int(g["iter"])
            res = np.array(s.resnorms)
            assert np.max(np.abs(res[:-1] - g["resnorms"][:-1]) / g["resnorms"][:-1]) < 1e-9
            # (the last entry is the EXPLICIT residual b - A x at the 4e-9 level: cancellation)
            assert abs(res[-1] - g["resnorms"][-1]) < 1e-6 * g["resnorms"][-1]
            assert np.linalg.norm(s.xk[:, 0] - g["xk"]) < 1e-10 * np.linalg.norm(g["xk"])
