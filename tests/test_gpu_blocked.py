"""The blocked reference-order Gram-Schmidt kernel (krypy_amd/csrc/chain_blk.h: one grid-wide sum per block of four
basis columns, Gram entries of the block from a table the sequence carries along).  Same recurrence as
/root/reference/krypy/utils.py:1012-1029 in exact arithmetic, another rounding: compared at north_star's 1e-10 with
the per-column kernels and with the CPU oracle, never bit for bit."""
import numpy as np
import pytest

from oracle import krylov_ref as ref

pytestmark = pytest.mark.gpu


def _cycle(linsys, utils, ls, **kw):
    try:
        return linsys.Gmres(ls, maxiter=100, tol=1e-14, store_arnoldi=True, **kw)
    except utils.ConvergenceError as e:
        return e.solver


@pytest.mark.parametrize("nx,ny", [(260, 250), (640, 500), (1000, 1000)])
def test_blocked_kernel_equals_the_per_column_kernels(hip, nx, ny):
    """One GMRES(100) cycle with the blocked kernel and with the per-column kernel of the same shape: residual history,
    Hessenberg matrix and iterate at 1e-10, the basis as orthogonal (||V^T V - I||_F within a factor two).  65,000 rows:
    all workgroups on one XCD (every communication wave gathers from the L2); 320,000: spread over the chip, the XCD
    leaders gather all records; 10^6: the two-level exchange.  The Gram table is rebuilt once per sequence (the first
    blocked step) and carried by the launches from then on."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(nx, ny)
    b = np.random.default_rng(3).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b)
    out = {}
    for blk in (1, 0):
        hip.set("chain_blk", blk)
        try:
            n0, r0 = hip.get("n_chain_blk"), hip.get("n_blk_rebuild")
            s = _cycle(linsys, utils, ls)
            Vb = s.arnoldi._V
            G = hip.gemm_tn(Vb, 0, 101, Vb, 0, 101)
            out[blk] = dict(res=np.array(s.resnorms), H=np.array(s.H), x=np.array(s.xk),
                            orth=float(np.linalg.norm(G - np.eye(101))), launches=hip.get("n_chain_blk") - n0,
                            rebuilds=hip.get("n_blk_rebuild") - r0)
            del s, Vb
        finally:
            hip.set("chain_blk", 1)
    a, c = out[1], out[0]
    assert a["launches"] == 93 and c["launches"] == 0, (a["launches"], c["launches"])      # steps k = 7 .. 99
    assert a["rebuilds"] == 1, a["rebuilds"]
    assert np.max(np.abs(a["res"] - c["res"]) / c["res"]) < 1e-10
    assert np.linalg.norm(a["H"] - c["H"]) < 1e-10 * np.linalg.norm(c["H"])
    assert np.linalg.norm(a["x"] - c["x"]) < 1e-10 * np.linalg.norm(c["x"])
    assert a["orth"] <= 2.0 * c["orth"] + 1e-13, (a["orth"], c["orth"])


def test_blocked_kernel_against_the_oracle_at_a_million_rows(hip):
    """N = 10^6 (1000 x 1000 grid), one whole GMRES(100) cycle through the blocked kernel against the CPU oracle: the 101
    residual norms, the Hessenberg matrix and the iterate at 1e-10."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(1000, 1000)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    n0 = hip.get("n_chain_blk")
    try:
        sol = linsys.Gmres(linsys.LinearSystem(A, b), maxiter=100, tol=1e-8, store_arnoldi=True)
    except utils.ConvergenceError as e:
        sol = e.solver
    assert hip.get("n_chain_blk") - n0 == 93
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


def test_gram_table_is_rebuilt_when_the_basis_is_touched(hip):
    """The table belongs to one Arnoldi sequence: consecutive steps of one basis block.  A write to the block through
    any other entry point (here: a column is downloaded and uploaded again, unchanged), a second Arnoldi object that
    interleaves its steps, a basis that grows (another block): the next blocked step rebuilds the rows from the basis
    instead of trusting them, and the result is the one of the undisturbed run."""
    from krypy_amd import linsys, utils

    A = ref.laplace2d(300, 300)
    rng = np.random.default_rng(11)
    v = rng.standard_normal((A.shape[0], 1))

    def run(disturb):
        ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=40, ortho="mgs")
        other = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v[::-1].copy(), maxiter=40, ortho="mgs") if disturb == "interleave" else None
        r0 = hip.get("n_blk_rebuild")
        for k in range(40):
            ar.advance()
            if disturb == "touch" and k in (12, 25):
                ar._settle()
                col = ar._V.download(3, 1)
                ar._V.upload(3, col[:, 0])
            if other is not None and k % 5 == 4:
                other.advance()
        V, H = ar.get()[:2]
        return np.array(V), np.array(H), hip.get("n_blk_rebuild") - r0

    V0, H0, r_plain = run(None)
    V1, H1, r_touch = run("touch")
    V2, H2, r_inter = run("interleave")
    assert r_plain == 1 and r_touch == 3, (r_plain, r_touch)
    assert r_inter >= 2, r_inter
    for V, H in ((V1, H1), (V2, H2)):
        assert np.linalg.norm(H - H0) < 1e-10 * np.linalg.norm(H0)
        assert np.linalg.norm(V - V0) < 1e-9
