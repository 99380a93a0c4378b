"""The eight-wave blocked Gram-Schmidt kernel (krypy_amd/csrc/chain_blk2.h): one grid-wide sum per block of four basis
columns for vectors of 4 ... 6 rows of 16 B per lane - and, on N ranks, the sums crossing the ranks INSIDE the launch through
the xr mailboxes: the local basis is read once per Arnoldi step and the step issues no all-reduce call.  Same recurrence as
/root/reference/krypy/utils.py:1012-1029 in exact arithmetic, another rounding: compared at north_star's 1e-10 with the
per-column kernels and with the CPU oracle, never bit for bit."""
import os

import numpy as np
import pytest

from oracle import krylov_ref as ref
from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


def _cycle(linsys, utils, ls, m=100, **kw):
    try:
        return linsys.Gmres(ls, maxiter=m, tol=1e-8, store_arnoldi=True, **kw)
    except utils.ConvergenceError as e:
        return e.solver


@pytest.mark.parametrize("nx,ny,cw", [(4000, 313, 1), (1500, 1000, 1), (1201, 907, 1), (1250, 1270, 1), (1500, 1000, 2), (1201, 907, 0),
                                      (2001, 901, 12), (4000, 625, 12)])
def test_one_gpu_long_short_vectors_against_the_per_column_kernels(hip, nx, ny, cw):
    """One GPU, 1.09 ... 1.59 M rows (5, 6 and 7 rows per lane; the first shape is the slab one of eight ranks holds of the
    benchmark problem): a whole GMRES(100) cycle through the blocked kernel - steps of eight links and more - against the
    same cycle on the per-column kernels: residual history, Hessenberg matrix and iterate at 1e-10, the basis as orthogonal
    (within a factor two).  cw = 1 is what runs by default (wave 0 without rows; 7 rows per lane from 1.38 M rows on), cw = 2
    stops that form at 6 rows (1.5 M rows then take the 512-lane form with 6 rows), cw = 0 is the 512-lane form throughout."""
    from krypy_amd import linsys, utils

    # cw = 12: the default shapes at sizes where they are the ones with ONE block in registers (8 ... 11 rows per lane, 1.6 ... 2.5 M
    # rows)
    hip.set("chain_blk2_cw", 1 if cw == 12 else cw)
    try:
        _one_gpu_case(hip, linsys, utils, nx, ny)
    finally:
        hip.set("chain_blk2_cw", 1)


def _one_gpu_case(hip, linsys, utils, nx, ny):

    A = ref.laplace2d(nx, ny)
    b = np.random.default_rng(3).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b)
    out = {}
    for blk in (1, 0):
        hip.set("chain_blk2", blk)
        try:
            n0, r0 = hip.get("n_chain_blk2"), hip.get("n_blk_rebuild")
            s = _cycle(linsys, utils, ls)
            Vb = s.arnoldi._V
            G = hip.gemm_tn(Vb, 0, 101, Vb, 0, 101)
            out[blk] = dict(res=np.array(s.resnorms), H=np.array(s.H), x=np.array(s.xk), orth=float(np.linalg.norm(G - np.eye(101))),
                            launches=hip.get("n_chain_blk2") - n0, rebuilds=hip.get("n_blk_rebuild") - r0)
            del s, Vb
        finally:
            hip.set("chain_blk2", 1)
    a, c = out[1], out[0]
    assert np.max(np.abs(a["res"] - c["res"]) / c["res"]) < 1e-10
    assert np.linalg.norm(a["H"] - c["H"]) < 1e-10 * np.linalg.norm(c["H"])
    assert np.linalg.norm(a["x"] - c["x"]) < 1e-10 * np.linalg.norm(c["x"])
    assert a["orth"] <= 2.0 * c["orth"] + 1e-13, (a["orth"], c["orth"])
    expect_kernel(a["launches"] == 93 and c["launches"] == 0, "blocked launches 93 / 0 (steps k = 7 .. 99): %r" % ((a["launches"], c["launches"]),))
    expect_kernel(a["rebuilds"] == 1, "one Gram-table rebuild per sequence: %r" % (a["rebuilds"],))


def test_one_gpu_against_the_oracle_at_the_shard_size(hip):
    """N = 1,252,000 (4000 x 313): one whole GMRES(100) cycle through the blocked kernel against the CPU oracle."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(4000, 313)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    n0 = hip.get("n_chain_blk2")
    sol = _cycle(linsys, utils, linsys.LinearSystem(A, b))
    used = hip.get("n_chain_blk2") - n0
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=1)
    except ImportError:
        lim = None
    want = ref.gmres(A, b, tol=1e-8, maxiter=100)
    if lim is not None:
        lim.restore_original_limits()
    res, wres = np.array(sol.resnorms), np.array(want.resnorms)
    assert len(res) == len(wres) == 101
    assert np.max(np.abs(res - wres) / wres) < 1e-10
    assert np.linalg.norm(np.array(sol.H) - want.H) < 1e-10 * np.linalg.norm(want.H)
    assert np.linalg.norm(sol.xk[:, 0] - want.xk) < 1e-10 * np.linalg.norm(want.xk)
    expect_kernel(used == 93, "93 blocked launches: %r" % (used,))


@pytest.fixture
def xr_ctx(hip):
    """A context in forced multi-rank mode (1-rank RCCL communicator) with the xr transport on in loopback: every sum of
    the blocked kernel goes through the rank's own mailbox - the code a rank of N runs, on one GPU."""
    from krypy_amd import _hip, dist as kdist

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    if os.environ.get("KRYPY_AMD_XR", "1") == "0":
        ctx.close()
        pytest.skip("KRYPY_AMD_XR=0: the in-launch cross-rank sums need the transport")
    assert kdist.enable_xr(ctx, kdist.TcpRendezvous(0, 1)) is True
    old = _hip._install_context_for_testing(ctx)
    yield ctx
    _hip._install_context_for_testing(old)
    ctx.close()


@pytest.mark.parametrize("nx,ny", [(30, 30), (90, 90), (301, 211), (1000, 700), (4000, 313), (1500, 1000), (2001, 901), (4000, 625)])
def test_cross_rank_sums_inside_the_launch_loopback(xr_ctx, nx, ny):
    """ortho='mgs' on the multi-rank path with the xr transport on: every Arnoldi step is SpMV + ONE launch whose
    grid-wide sums include the cross-rank stage (publish to every rank's mailbox, poll the own one, add in rank order) -
    no all-reduce call.  30 steps against the CPU oracle's MGS and against the one-reduction form of the same context
    (chain_blk2 = 0) at 1e-10 (H) / 1e-9 (basis); 900 rows (masked, one workgroup), 8,100, 63,511 (odd), 700,000 (4 rows per
    lane), 1.25 M (5 rows: one of eight ranks' slab of the benchmark problem), 1.5 M (7 rows), and - ONE block in registers - 1.8 M
    (8 rows, odd and masked) and 2.5 M rows (11 rows: one of FOUR ranks' slab)."""
    from krypy_amd import utils

    ctx = xr_ctx
    A = ref.laplace2d(nx, ny)
    n = A.shape[0]
    v = np.random.default_rng(5).standard_normal((n, 1))
    m = 30
    st = ref.arnoldi_init(A, v[:, 0], m, ortho="mgs")
    for _ in range(m):
        ref.arnoldi_step(st)
    out = {}
    for blk in (1, 0):
        ctx.set("chain_blk2", blk)
        try:
            ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=m, ortho="mgs")
            a0, b0, x0 = ctx.get("n_allreduce"), ctx.get("n_chain_blk2"), ctx.get("n_xr")
            for _ in range(m):
                ar.advance()
            ar._settle()
            out[blk] = dict(H=np.array(ar.H), V=ar.V, allred=ctx.get("n_allreduce") - a0, steps=ctx.get("n_chain_blk2") - b0,
                            xr=ctx.get("n_xr") - x0)
            del ar
        finally:
            ctx.set("chain_blk2", 1)
    a, c = out[1], out[0]
    hn = np.linalg.norm(st.H)
    assert np.linalg.norm(a["H"] - st.H) < 1e-10 * hn and np.linalg.norm(c["H"] - st.H) < 1e-10 * hn
    assert np.linalg.norm(a["H"] - c["H"]) < 1e-11 * hn
    assert np.max(np.abs(a["V"] - st.V)) < 1e-9 and np.max(np.abs(a["V"] - c["V"])) < 1e-9
    G = a["V"].T.dot(a["V"]) - np.eye(m + 1)
    Gc = c["V"].T.dot(c["V"]) - np.eye(m + 1)
    assert np.linalg.norm(G) <= 2.0 * np.linalg.norm(Gc) + 1e-13, (np.linalg.norm(G), np.linalg.norm(Gc))
    expect_kernel(a["steps"] == m and c["steps"] == 0, "blocked steps m / 0: %r" % ((a["steps"], c["steps"]),))
    expect_kernel(a["allred"] == 0, "no all-reduce call in a step: %r" % (a["allred"],))
    expect_kernel(a["xr"] == sum((k + 1 + 3) // 4 + 1 for k in range(m)), "one exchange per block of four links + the norm: %r" % (a["xr"],))


def test_cross_rank_sums_inside_the_launch_whole_solves(xr_ctx):
    """Restarted GMRES (every cycle a new basis: the table starts over with step k = 0), and a GMRES whose basis grows on
    demand, through the blocked kernel with the in-launch exchange, against the CPU oracle."""
    from krypy_amd import linsys, utils

    ctx = xr_ctx
    A = ref.laplace2d(96, 96)
    b = np.random.default_rng(1).standard_normal(A.shape[0])
    b0 = ctx.get("n_chain_blk2")
    sol = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=40, max_restarts=30, tol=1e-8, ortho="mgs")
    o = ref.restarted_gmres(A, b, tol=1e-8, maxiter=40, max_restarts=30)
    got, want = np.array(sol.resnorms), np.array(o.resnorms)
    assert len(got) == len(want)
    assert np.max(np.abs(got[:40] - want[:40]) / want[:40]) < 1e-10
    assert np.linalg.norm(sol.xk[:, 0] - o.xk) < 1e-7 * np.linalg.norm(o.xk)
    assert np.linalg.norm(A.dot(sol.xk[:, 0]) - b) <= 1.0001e-8 * np.linalg.norm(b)
    expect_kernel(ctx.get("n_chain_blk2") - b0 >= len(got) - 1, "every step took the blocked kernel")
    x, s2 = __import__("krypy_amd").gmres(A, b, tol=1e-8, maxiter=400)          # (basis grown on demand)
    o2 = ref.gmres(A, b, tol=1e-8, maxiter=400)
    r2, w2 = np.array(s2.resnorms), np.array(o2.resnorms)
    assert len(r2) == len(w2) and np.max(np.abs(r2[:60] - w2[:60]) / w2[:60]) < 1e-10
