"""The REAL halo exchange of a block-row shard on one MI355X (sharded operator = /root/reference/krypy/utils.py:1593-1594).

The sharded SpMV exchanges its boundary entries with grouped ``ncclSend`` / ``ncclRecv`` calls on a second stream while
the interior rows are multiplied (``krylov_hip.hip: apply_one``, ``comm.hip: comm_halo_exchange``).  With one GPU per
box the exchange itself never ran in rounds 1 and 2 (a 1-rank communicator has no neighbours).  ``halo_loopback``
makes the one rank its own previous and next neighbour: the slab of an operator that is PERIODIC across the slab
boundary sends its first / last rows to itself.  Everything between the host call and the boundary rows' result is
then the code that runs on N > 1 GPUs: the grouped point-to-point calls, the communication stream, the two events.

Expected values come from SciPy on the tripled (non-periodic) operator applied to (x, x, x): the same rows with the
same in-row order, so the comparison is bit for bit."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref

from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


@pytest.fixture
def loop_ctx(hip):
    """A context with a 1-rank RCCL communicator in forced multi-rank mode and the loopback switch on."""
    from krypy_amd import _hip

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    ctx.set("halo_loopback", 1)
    yield ctx
    ctx.close()


def _stencil(kind, rng):
    """(operator on three slabs, slab length)"""
    if kind == "lap2d":            # 300 x (3 * 70) grid, slabs of 70 grid lines: ghost width 300
        return ref.laplace2d(300, 210), 300 * 70
    if kind == "lap3d":            # 40 x 40 x (3 * 30): ghost width one plane
        T = lambda m: sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], [-1, 0, 1])     # noqa: E731
        I = sp.identity                                                                               # noqa: E731,E741
        A = (sp.kron(I(90), sp.kron(I(40), T(40))) + sp.kron(I(90), sp.kron(T(40), I(40))) +
             sp.kron(T(90), sp.kron(I(40), I(40)))).tocsr()
        A.sort_indices()
        return A, 40 * 40 * 30
    n = 20000
    if kind == "nonsym":           # nonsymmetric banded, one-sided far diagonal: different widths up and down
        offs = (-700, -2, -1, 0, 1, 300)
        A = sp.diags([rng.uniform(0.5, 1.5, 3 * n - abs(o)) for o in offs], offs, format="csr")
        A.sort_indices()
        return A, n
    # random pattern inside a band (CSR-stream kernel: no diagonal structure); about nine entries per row
    r = np.random.default_rng(11)
    rows = np.repeat(np.arange(3 * n), 8)
    cols = np.clip(rows + r.integers(-899, 900, rows.size), 0, 3 * n - 1)
    A = (sp.coo_matrix((r.standard_normal(rows.size), (rows, cols)), shape=(3 * n, 3 * n)) + sp.identity(3 * n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A, n


@pytest.mark.parametrize("kind", ["lap2d", "lap3d", "nonsym", "random"])
def test_periodic_slab_exchanges_its_halo_with_itself(loop_ctx, kind):
    """Ghost buffer after kh_apply == the planes a periodic neighbour would have sent, product == the periodic
    operator, bit for bit - banded (k_spmv_dia<HALO>) and CSR-stream kernel, split launches and serial order."""
    from krypy_amd import dist

    ctx = loop_ctx
    rng = np.random.default_rng(3)
    Abig, n = _stencil(kind, rng)
    A_local, nrp, nrn = dist.localize_columns(Abig[n:2 * n], n, 3 * n)
    assert nrp > 0 and nrn > 0
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)          # a ring of identical slabs: what goes to the previous one is its "next" ghost region
    banded = kind != "random"
    x = rng.standard_normal(n)
    want = Abig[n:2 * n].dot(np.tile(x, 3))
    b = rng.standard_normal(n)
    X, Y, R, B = ctx.upload(x), ctx.alloc(n, 1), ctx.alloc(n, 1), ctx.upload(b)
    for dia in ((1, 0) if banded else (0,)):
        ctx.set("spmv_dia", dia)
        for split in (1, 0):
            ctx.set("spmv_split", split)
            ctx.set_ghost(Ad, np.full(nrp + nrn, np.nan))     # whatever is there now is not what the test reads later
            e0, s0 = ctx.get("n_halo_exchange"), ctx.get("n_spmv_split")
            ctx.apply(Ad, X, 0, Y, 0, 1)
            expect_kernel(ctx.get("n_halo_exchange") == e0 + 1 and ctx.get("n_spmv_split") == s0 + split, "ctx.get(\"n_halo_exchange\") == e0 + 1 and ctx.get(\"n_spmv_split\") == s0 + split")
            g = ctx.get_ghost(Ad, nrp + nrn)
            assert np.array_equal(g, np.concatenate([x[n - nrp:], x[:nrn]])), (kind, dia, split)
            assert np.array_equal(Y.download()[:, 0], want), (kind, dia, split)
            nrm = ctx.residual(Ad, B, 0, X, 0, R, 0)           # fused epilogue across the two launches
            assert np.array_equal(R.download()[:, 0], b - want), (kind, dia, split)
            assert abs(nrm - np.linalg.norm(b - want)) <= 1e-13 * nrm
    if banded and os.environ.get("KRYPY_AMD_SPMV_DIA", "") != "0":
        assert Ad.diagonals > 0
    ctx.set("spmv_dia", 1)
    ctx.set("spmv_split", 1)


@pytest.mark.parametrize("kind", ["lap2d", "lap3d", "nonsym"])
def test_halo_inside_the_spmv_launch(loop_ctx, kind):
    """xh (csrc/xr_dev.h, kernels.h: k_spmv_dia<..., XH>): the banded SpMV of a shard stores its boundary rows into the
    neighbours' IPC-mapped ghost granules and polls its own INSIDE its one launch - no ncclSend / ncclRecv kernel, no second
    stream, no split.  Here the rank is its own neighbour (the periodic slab of the tests above): product, fused residual
    and fused dot bit for bit the RCCL exchange's / SciPy's on the tripled operator, a hundred applications back to back
    (epochs, both parities), and the counters say which path ran."""
    from krypy_amd import dist

    ctx = loop_ctx
    rng = np.random.default_rng(3)
    Abig, n = _stencil(kind, rng)
    A_local, nrp, nrn = dist.localize_columns(Abig[n:2 * n], n, 3 * n)
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)
    if Ad.diagonals == 0:
        pytest.skip("no diagonal-major copy of the shard (KRYPY_AMD_SPMV_DIA=0): the banded kernel carries the in-launch halo")
    handle = ctx.xh_export(Ad)
    assert len(handle) == 64
    ctx.xh_attach(Ad, None, 0, 0, None, 0, self_loop=True)
    ctx.xh_enable(Ad, True)
    b = rng.standard_normal(n)
    R, B, Y = ctx.alloc(n, 1), ctx.upload(b), ctx.alloc(n, 1)
    e0, x0 = ctx.get("n_halo_exchange"), ctx.get("n_halo_xh")
    for rep in range(100):
        x = rng.standard_normal(n)
        want = Abig[n:2 * n].dot(np.tile(x, 3))
        X = ctx.upload(x)
        ctx.apply(Ad, X, 0, Y, 0, 1)
        assert np.array_equal(Y.download()[:, 0], want), (kind, rep)
        if rep % 10 == 0:
            nrm = ctx.residual(Ad, B, 0, X, 0, R, 0)
            assert np.array_equal(R.download()[:, 0], b - want), (kind, rep)
            assert abs(nrm - np.linalg.norm(b - want)) <= 1e-13 * nrm
    expect_kernel(ctx.get("n_halo_exchange") == e0 and ctx.get("n_halo_xh") - x0 == 110,
                  "no RCCL exchange, 110 in-launch ones: %r" % ((ctx.get("n_halo_exchange") - e0, ctx.get("n_halo_xh") - x0),))
    ctx.xh_enable(Ad, False)                    # ... and the RCCL exchange again
    ctx.apply(Ad, X, 0, Y, 0, 1)
    assert np.array_equal(Y.download()[:, 0], want)
    expect_kernel(ctx.get("n_halo_exchange") == e0 + 1, "back on the RCCL exchange")
    # detached (what every rank does when the collective decision comes out "off" after some had attached) and attached again:
    # the exchange is back inside the launch, the same bits
    ctx.xh_detach(Ad)
    with pytest.raises(Exception):
        ctx.xh_enable(Ad, True)                 # (nothing attached: refused)
    ctx.xh_attach(Ad, None, 0, 0, None, 0, self_loop=True)
    ctx.xh_enable(Ad, True)
    x1 = ctx.get("n_halo_xh")
    ctx.apply(Ad, X, 0, Y, 0, 1)
    assert np.array_equal(Y.download()[:, 0], want)
    expect_kernel(ctx.get("n_halo_xh") == x1 + 1, "in the launch again after detach + attach")


@pytest.mark.parametrize("kind", ["lap2d", "random"])
def test_sharded_panel_apply_exchanges_once_and_streams_the_matrix_once(loop_ctx, kind):
    """kh_apply of a shard to a block of d vectors (A U of the deflation set-up, deflation.py:47; Ritz residuals,
    deflation.py:849-855): ONE grouped exchange for all d columns' halos and ONE pass over the matrix
    (k_spmm_stream with ghost columns) instead of d exchanges + d passes - bit-identical to the column loop and to
    SciPy on the periodic operator."""
    from krypy_amd import dist

    ctx = loop_ctx
    rng = np.random.default_rng(7)
    Abig, n = _stencil(kind, rng)
    A_local, nrp, nrn = dist.localize_columns(Abig[n:2 * n], n, 3 * n)
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)
    for d in (2, 5, 16):
        X = rng.standard_normal((n, d))
        want = Abig[n:2 * n].dot(np.vstack([X, X, X]))
        Xd, Y, Y1 = ctx.upload(X), ctx.alloc(n, d + 1), ctx.alloc(n, d)
        e0, m0 = ctx.get("n_halo_exchange"), ctx.get("n_spmm")
        ctx.apply(Ad, Xd, 0, Y, 1, d)                       # (an offset in the output block)
        expect_kernel(ctx.get("n_halo_exchange") == e0 + 1 and ctx.get("n_spmm") == m0 + 1, "ctx.get(\"n_halo_exchange\") == e0 + 1 and ctx.get(\"n_spmm\") == m0 + 1")
        assert np.array_equal(Y.download()[:, 1:], want), (kind, d)
        for c in range(d):                                  # the column loop: d exchanges, d passes
            ctx.apply(Ad, Xd, c, Y1, c, 1)
        expect_kernel(ctx.get("n_halo_exchange") == e0 + 1 + d, "ctx.get(\"n_halo_exchange\") == e0 + 1 + d")
        assert np.array_equal(Y1.download(), want), (kind, d)


def test_complex_periodic_slab(loop_ctx):
    """(re, im) pairs through the same exchange: complex CSR shard (k_zspmv_stream with ghost columns)."""
    from krypy_amd import dist

    ctx = loop_ctx
    rng = np.random.default_rng(5)
    n = 15000
    offs = (-400, -1, 0, 1, 400)
    Abig = sp.diags([rng.standard_normal(3 * n - abs(o)) + 1j * rng.standard_normal(3 * n - abs(o)) for o in offs], offs,
                    format="csr")
    Abig.sort_indices()
    A_local, nrp, nrn = dist.localize_columns(Abig[n:2 * n], n, 3 * n)
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    X, Y = ctx.upload(x), ctx.alloc(n, 1, dtype=complex)
    e0 = ctx.get("n_halo_exchange")
    ctx.apply(Ad, X, 0, Y, 0, 1)
    expect_kernel(ctx.get("n_halo_exchange") == e0 + 1, "ctx.get(\"n_halo_exchange\") == e0 + 1")
    assert np.array_equal(ctx.get_ghost(Ad, nrp + nrn), np.concatenate([x[n - nrp:], x[:nrn]]))
    assert np.array_equal(Y.download()[:, 0], Abig[n:2 * n].dot(np.tile(x, 3)))


def test_thousand_exchanges_with_an_all_reduce_right_behind(loop_ctx):
    """Two streams issue RCCL work on one communicator: the exchange on the communication stream, the inner
    products' all-reduce on the compute stream right behind the boundary rows.  1000 rounds of SpMV + norm with
    changing data: no hang, and every round's numbers are the periodic operator's."""
    from krypy_amd import dist

    ctx = loop_ctx
    Abig, n = _stencil("lap2d", None)
    A_local, nrp, nrn = dist.localize_columns(Abig[n:2 * n], n, 3 * n)
    Ad = ctx.csr(A_local, n_cols=A_local.shape[1])
    ctx.set_halo(Ad, nrn, nrp, nrp, nrn)
    P = (Abig[n:2 * n, :n] + Abig[n:2 * n, n:2 * n] + Abig[n:2 * n, 2 * n:]).tocsr()      # the periodic operator itself
    rng = np.random.default_rng(9)
    x = rng.standard_normal(n)
    x /= np.linalg.norm(x)
    X, Y = ctx.upload(x), ctx.alloc(n, 1)
    e0 = ctx.get("n_halo_exchange")
    for it in range(1000):                       # power iteration: y = P x, x = y / ||y||
        check = it % 100 == 99                   # host replay of this step only (the iterate itself is downloaded)
        if check:
            xprev = X.download()[:, 0]
        ctx.apply(Ad, X, 0, Y, 0, 1)
        nrm = ctx.nrm2(Y, 0)                     # partial sums -> ncclAllReduce on the compute stream -> host
        ctx.vdiv(X, 0, Y, 0, nrm)
        if check:
            y = P.dot(xprev)
            assert abs(nrm - np.linalg.norm(y)) <= 1e-13 * nrm, it
            assert np.allclose(X.download()[:, 0], y / np.linalg.norm(y), rtol=0, atol=1e-15), it
    expect_kernel(ctx.get("n_halo_exchange") - e0 == 1000, "ctx.get(\"n_halo_exchange\") - e0 == 1000")


def test_gmres_on_a_periodic_slab_through_the_exchange(loop_ctx):
    """A whole solve whose every operator application exchanges a halo: RestartedGmres on the periodic slab against
    the CPU oracle on the periodic matrix (iterate-for-iterate, 1e-10)."""
    from krypy_amd import _hip, dist, linsys, utils

    ctx = loop_ctx
    Abig, n = _stencil("lap2d", None)
    A_slab = Abig[n:2 * n]
    P = (A_slab[:, :n] + A_slab[:, n:2 * n] + A_slab[:, 2 * n:]).tocsr() + sp.identity(n) * 0.05     # shifted: nonsingular
    # the slab of the shifted operator
    Abig2 = (Abig + sp.identity(3 * n) * 0.05).tocsr()
    Abig2.sort_indices()
    b = np.random.default_rng(2).standard_normal(n)

    old = _hip._install_context_for_testing(ctx)
    try:
        op = dist.ShardedCSROperator(Abig2[n:2 * n], n, 3 * n, ctx)
        nsp, nsn, nrp, nrn = op.halo
        assert (nsp, nsn) == (0, 0) and nrp > 0 and nrn > 0          # one rank: the table knows no neighbours ...
        op.halo = (nrn, nrp, nrp, nrn)                               # ... the ring does
        ctx.set_halo(op._device_matrix(), *op.halo)
        e0 = ctx.get("n_halo_exchange")
        ls = linsys.LinearSystem(op, b)
        try:
            s = linsys.Gmres(ls, maxiter=40, tol=1e-12, ortho="mgs")
        except utils.ConvergenceError as e:
            s = e.solver
        o = ref.gmres(P, b, maxiter=40, tol=1e-12)
        got, want = np.array(s.resnorms), np.array(o.resnorms)
        assert got.shape == want.shape
        assert np.max(np.abs(got[:-1] - want[:-1]) / want[:-1]) < 1e-10
        assert np.linalg.norm(s.xk[:, 0] - o.xk) <= 1e-10 * np.linalg.norm(o.xk)
        expect_kernel(ctx.get("n_halo_exchange") - e0 >= 40, "ctx.get(\"n_halo_exchange\") - e0 >= 40")
    finally:
        _hip._install_context_for_testing(old)


def test_all_reduces_per_arnoldi_step_are_counted(loop_ctx):
    """What the ranks of a node would issue per Arnoldi step (kh_ctx_get "n_allreduce", counted on the forced multi-rank
    path of a 1-rank communicator): the panel form two ncclAllReduce calls - the coefficient panel, the norm -, the
    reference order two as well since round 4 (all coefficients from one pass, corrected with the Gram table:
    tests/test_gpu_blocked.py), k + 2 (one per Gram-Schmidt link and the norm) with that form switched off."""
    from krypy_amd import _hip, utils

    old = _hip._install_context_for_testing(loop_ctx)
    try:
        loop_ctx.set("halo_loopback", 0)
        A = ref.laplace2d(200, 150)
        v = np.random.default_rng(2).standard_normal((A.shape[0], 1))
        per_step = {}
        for ortho in ("cgs", "mgs", "mgs per column"):
            loop_ctx.set("mgs_lowsync", 0 if ortho == "mgs per column" else 1)
            ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=20, ortho=ortho.split()[0])
            n0 = loop_ctx.get("n_allreduce")
            for _ in range(20):
                ar.advance()
            ar._settle()
            per_step[ortho] = (loop_ctx.get("n_allreduce") - n0) / 20.0
        loop_ctx.set("mgs_lowsync", 1)
        expect_kernel(abs(per_step["cgs"] - 2.0) < 0.2, "cgs: 2 all-reduces per step: %r" % (per_step,))
        expect_kernel(abs(per_step["mgs"] - 2.0) < 0.2, "mgs (one-reduction form): 2 all-reduces per step: %r" % (per_step,))
        expect_kernel(abs(per_step["mgs per column"] - (sum(k + 2 for k in range(20)) / 20.0)) < 1.5, "mgs per column: k + 2: %r" % (per_step,))
    finally:
        _hip._install_context_for_testing(old)
