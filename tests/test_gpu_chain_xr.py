"""The register-resident Gram-Schmidt chain with the cross-rank stage inside every grid-wide sum (krypy_amd/csrc/chain_xr.hip,
chain.h: grid_sum<true>): on N ranks a reference-order Arnoldi step (/root/reference/krypy/utils.py:1012-1034) on a slab of
2.5 M ... 14.68 M rows is the sharded SpMV + ONE launch - no all-reduce call, the local basis read once.

* one rank in loopback (a forced 1-rank communicator, every sum through the rank's own mailbox) at the full shapes of
  16 ... 56 rows per lane: the CPU oracle at 1e-10, and THE SAME BITS as the one-GPU chain kernel (with one rank the rank-ordered
  sum adds nothing: any difference would be a defect of the stage, not rounding);
* small slabs through the masked instantiations (the shape chosen for a few compute units: kh_ctx_set "chain_xr_cus"), whole
  restarted and deflated solves against the oracle;
* two PROCESSES on one GPU (tests/support/xr_worker.py): see tests/test_gpu_xr.py, which compares their `chainxr_*` solves."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref
from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


@pytest.fixture
def xr_ctx(hip):
    """A context in forced multi-rank mode (1-rank RCCL communicator) with the xr transport on in loopback."""
    from krypy_amd import _hip, dist as kdist

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    if os.environ.get("KRYPY_AMD_XR", "1") == "0" or os.environ.get("KRYPY_AMD_CHAIN_XR", "1") == "0":
        ctx.close()
        pytest.skip("KRYPY_AMD_XR=0 / KRYPY_AMD_CHAIN_XR=0: the in-launch cross-rank sums are switched off")
    assert kdist.enable_xr(ctx, kdist.TcpRendezvous(0, 1)) is True
    old = _hip._install_context_for_testing(ctx)
    yield ctx
    _hip._install_context_for_testing(old)
    ctx.close()


def _tridiag(n):
    return sp.diags([np.full(n - 1, -1.0), np.linspace(2.0, 3.0, n), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr()


def _steps(ctx, A, v, m):
    Ad = ctx.csr(A)
    n = A.shape[0]
    V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
    V.upload(0, v)
    H = np.zeros((m + 1, m))
    for k in range(m):
        H[: k + 2, k] = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 1, 0)
    return H, V.download()


@pytest.mark.parametrize("rows,n", [(16, 3_000_000), (24, 5_000_000), (32, 8_000_002), (40, 10_000_000), (48, 12_500_000), (56, 14_400_000)])
def test_full_shapes_in_loopback_equal_the_one_gpu_chain_kernel(hip, xr_ctx, rows, n):
    """Six Arnoldi steps at every register shape of the cross-rank chain (5 M rows: one of TWO ranks' slab of the benchmark
    problem; 12.5 M: one of eight ranks' slab of config 5): no all-reduce call, k + 2 exchanges per step, the one-GPU chain
    kernel's bits (one rank: the rank-ordered sum is the device's own total), the Arnoldi relation."""
    m = 6
    A = _tridiag(n)
    v = np.random.default_rng(rows).standard_normal(n)
    v /= np.linalg.norm(v)
    a0, c0, x0 = xr_ctx.get("n_allreduce"), xr_ctx.get("n_chain_xr"), xr_ctx.get("n_xr")
    Hx, Vx = _steps(xr_ctx, A, v, m)
    used = (xr_ctx.get("n_allreduce") - a0, xr_ctx.get("n_chain_xr") - c0, xr_ctx.get("n_xr") - x0)
    g0 = hip.counters()["chain"]
    H1, V1 = _steps(hip, A, v, m)
    one_gpu_chain = hip.counters()["chain"] - g0
    assert np.linalg.norm(A.dot(Vx[:, :m]) - Vx.dot(Hx)) < 1e-12 * np.linalg.norm(Hx)
    assert np.linalg.norm(Vx.T.dot(Vx) - np.eye(m + 1)) < 1e-12
    assert np.linalg.norm(Hx - H1) < 1e-12 * np.linalg.norm(H1)
    # (the one-GPU run may compute w in the kernel's prologue: the same bits as the SpMV launch, tested elsewhere.  Bits are
    # compared when BOTH contexts took the chain kernel: under KRYPY_AMD_MGS_CHAIN=0 the fresh loopback context takes one
    # all-reduce per link while the session's context may have had the chain switched back on by an earlier test.)
    if one_gpu_chain == m and used[1] == m:
        assert np.array_equal(Hx, H1) and np.array_equal(Vx, V1), "one rank in loopback must reproduce the one-GPU chain kernel's bits"
    expect_kernel(used == (0, m, sum(k + 2 for k in range(m))), "(all-reduce calls, chain launches with the stage, exchanges) = %r" % (used,))


@pytest.mark.parametrize("n,cus", [(90_000, 4), (184_901, 4), (229_000, 4), (61_003, 4), (1_000, 4), (700_000, 32)])
def test_small_slabs_through_the_masked_shapes_against_the_oracle(xr_ctx, n, cus):
    """The shape chosen for `cus` compute units (what lets two processes share one device in tests/test_gpu_xr.py): 24, 48, 56
    and 16 rows per lane on four workgroups, one workgroup, 24 rows on 32 - unpadded vectors, i.e. the masked instantiations;
    20 steps against the CPU oracle's MGS at 1e-10."""
    from krypy_amd import utils

    ctx = xr_ctx
    ctx.set("chain_xr_cus", cus)
    ctx.set("chain_blk2", 0)
    try:
        m = 20
        A = _tridiag(n) if n % 1000 else ref.laplace2d(n // 100, 100) if n >= 100_000 else _tridiag(n)
        v = np.random.default_rng(n).standard_normal((A.shape[0], 1))
        st = ref.arnoldi_init(A, v[:, 0], m, ortho="mgs")
        for _ in range(m):
            ref.arnoldi_step(st)
        ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=m, ortho="mgs")
        a0, c0 = ctx.get("n_allreduce"), ctx.get("n_chain_xr")
        for _ in range(m):
            ar.advance()
        ar._settle()
        used = (ctx.get("n_allreduce") - a0, ctx.get("n_chain_xr") - c0)
        hn = np.linalg.norm(st.H)
        assert np.linalg.norm(np.array(ar.H) - st.H) < 1e-10 * hn
        assert np.max(np.abs(ar.V - st.V)) < 1e-9
        expect_kernel(used == (0, m), "(all-reduce calls, chain launches with the stage) = %r" % (used,))
    finally:
        ctx.set("chain_xr_cus", 0)
        ctx.set("chain_blk2", 1)


def test_whole_solves_restarted_and_deflated(xr_ctx):
    """Restarted GMRES and a deflated solve (the projector's sums through the mailboxes, then the chain with the stage) on the
    multi-rank path against the CPU oracle."""
    from krypy_amd import deflation, linsys

    ctx = xr_ctx
    ctx.set("chain_xr_cus", 4)
    ctx.set("chain_blk2", 0)
    try:
        A = ref.laplace2d(300, 300)
        b = np.random.default_rng(1).standard_normal(A.shape[0])
        c0 = ctx.get("n_chain_xr")
        sol = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=40, max_restarts=3, tol=1e-30, ortho="mgs") \
            if False else None
        try:
            sol = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=40, max_restarts=2, tol=1e-30, ortho="mgs")
        except Exception as e:      # ConvergenceError carries the solver
            sol = e.solver
        o = ref.restarted_gmres(A, b, tol=1e-30, maxiter=40, max_restarts=2)
        got, want = np.array(sol.resnorms), np.array(o.resnorms)
        assert len(got) == len(want)
        assert np.max(np.abs(got[:41] - want[:41]) / want[:41]) < 1e-10
        expect_kernel(ctx.get("n_chain_xr") - c0 >= 120, "every step took the chain with the cross-rank stage: %r" % (ctx.get("n_chain_xr") - c0,))
        # deflated: two exact eigenvectors of the 2-D Laplacian as U
        nx = 300
        i = np.arange(1, nx + 1)
        s1, s2 = np.sin(np.pi * i / (nx + 1)), np.sin(2 * np.pi * i / (nx + 1))
        U = np.stack([np.kron(s1, s1), np.kron(s1, s2)], axis=1)
        c1 = ctx.get("n_chain_xr")
        try:
            d = deflation.DeflatedGmres(linsys.LinearSystem(A, b, self_adjoint=True), U=U, maxiter=30, tol=1e-30, ortho="mgs")
        except Exception as e:
            d = e.solver
        od = ref.deflated_gmres(A, b, U, tol=1e-30, maxiter=30)
        gd, wd = np.array(d.resnorms), np.array(od.resnorms)
        assert len(gd) == len(wd) and np.max(np.abs(gd - wd) / wd) < 1e-9
        expect_kernel(ctx.get("n_chain_xr") - c1 >= 30, "the deflated steps took it too: %r" % (ctx.get("n_chain_xr") - c1,))
    finally:
        ctx.set("chain_xr_cus", 0)
        ctx.set("chain_blk2", 1)


def test_switched_off_the_one_reduction_form_takes_over(xr_ctx):
    """KRYPY_AMD_CHAIN_XR=0 / kh_ctx_set("chain_xr", 0): the same steps through the one-reduction form (two sums across the
    ranks per step) - the same H at 1e-10."""
    ctx = xr_ctx
    n = 3_000_000
    A = _tridiag(n)
    v = np.random.default_rng(2).standard_normal(n)
    v /= np.linalg.norm(v)
    Hx, _ = _steps(ctx, A, v, 5)
    ctx.set("chain_xr", 0)
    try:
        c0, l0 = ctx.get("n_chain_xr"), ctx.get("n_lowsync")
        Hl, _ = _steps(ctx, A, v, 5)
        used = (ctx.get("n_chain_xr") - c0, ctx.get("n_lowsync") - l0)
    finally:
        ctx.set("chain_xr", 1)
    assert np.linalg.norm(Hx - Hl) < 1e-10 * np.linalg.norm(Hl)
    expect_kernel(used == (0, 5), "(chain launches with the stage, one-reduction steps) = %r" % (used,))
