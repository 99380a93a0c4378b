"""The blocked reference-order Gram-Schmidt kernel (krypy_amd/csrc/chain_blk.h: one grid-wide sum per block of four
basis columns, Gram entries of the block from a table the sequence carries along).  Same recurrence as
/root/reference/krypy/utils.py:1012-1029 in exact arithmetic, another rounding: compared at north_star's 1e-10 with
the per-column kernels and with the CPU oracle, never bit for bit."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref

from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


def _cycle(linsys, utils, ls, **kw):
    try:
        return linsys.Gmres(ls, maxiter=100, tol=1e-14, store_arnoldi=True, **kw)
    except utils.ConvergenceError as e:
        return e.solver


@pytest.mark.parametrize("nx,ny,rowless", [(260, 250, 1), (640, 500, 1), (640, 500, 0), (1000, 1000, 1), (1000, 1000, 0),
                                           (1024, 1024, 1)])
def test_blocked_kernel_equals_the_per_column_kernels(hip, nx, ny, rowless):
    """One GMRES(100) cycle with the blocked kernel and with the per-column kernel of the same shape: residual history,
    Hessenberg matrix and iterate at 1e-10, the basis as orthogonal (||V^T V - I||_F within a factor two).  65,000 rows:
    all workgroups on one XCD (every communication wave gathers from the L2); 320,000: spread over the chip, the XCD
    leaders gather all records; 10^6: the two-level exchange.  Spread over the chip, eight workgroups WITHOUT rows run in
    front of the others and gather the sums (rowless = 0: switched off; 1024 x 1024: 256 workgroups with rows, no
    room for them).  The Gram table is rebuilt once per sequence (the first blocked step) and carried by the launches
    from then on."""
    from krypy_amd import linsys, utils

    hip.set("blk_nx", 8 if rowless else 0)
    A = ref.laplace2d(nx, ny)
    b = np.random.default_rng(3).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b)
    out = {}
    for blk in (1, 0):
        hip.set("chain_blk", blk)
        try:
            n0, r0, x0 = hip.get("n_chain_blk"), hip.get("n_blk_rebuild"), hip.get("n_blk_rowless")
            s = _cycle(linsys, utils, ls)
            Vb = s.arnoldi._V
            G = hip.gemm_tn(Vb, 0, 101, Vb, 0, 101)
            out[blk] = dict(res=np.array(s.resnorms), H=np.array(s.H), x=np.array(s.xk),
                            orth=float(np.linalg.norm(G - np.eye(101))), launches=hip.get("n_chain_blk") - n0,
                            rebuilds=hip.get("n_blk_rebuild") - r0, rowless=hip.get("n_blk_rowless") - x0)
            del s, Vb
        finally:
            hip.set("chain_blk", 1)
    hip.set("blk_nx", 8)
    a, c = out[1], out[0]
    assert np.max(np.abs(a["res"] - c["res"]) / c["res"]) < 1e-10
    assert np.linalg.norm(a["H"] - c["H"]) < 1e-10 * np.linalg.norm(c["H"])
    assert np.linalg.norm(a["x"] - c["x"]) < 1e-10 * np.linalg.norm(c["x"])
    assert a["orth"] <= 2.0 * c["orth"] + 1e-13, (a["orth"], c["orth"])
    # which kernel ran: judged after the comparisons (tests/support/kernel_expect.py)
    expect_kernel(a["launches"] == 93 and c["launches"] == 0,
                  "blocked launches 93 / 0 (steps k = 7 .. 99): %r" % ((a["launches"], c["launches"]),))
    expect_kernel(a["rebuilds"] == 1, "one Gram-table rebuild per sequence: %r" % (a["rebuilds"],))
    # (one XCD: nobody without rows; 1024 x 1024: 256 workgroups with rows fill the chip)
    expect_kernel(a["rowless"] == (93 if rowless and 70000 < nx * ny <= 248 * 4096 else 0),
                  "launches with rowless workgroups: %r" % (a["rowless"],))


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
    expect_kernel(hip.get("n_chain_blk") - n0 == 93, "hip.get(\"n_chain_blk\") - n0 == 93")
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
    for V, H in ((V1, H1), (V2, H2)):
        assert np.linalg.norm(H - H0) < 1e-10 * np.linalg.norm(H0)
        assert np.linalg.norm(V - V0) < 1e-9
    expect_kernel(r_plain == 1 and r_touch == 3, "rebuilds plain / touched 1 / 3: %r" % ((r_plain, r_touch),))
    expect_kernel(r_inter >= 2, "rebuilds with interleaved sequences >= 2: %r" % (r_inter,))


def test_gram_table_does_not_survive_a_freed_and_reallocated_basis(hip):
    """The table is keyed on (basis block handle, next step).  A basis that is dropped in the middle of a sequence and
    a new block of the same shape taken right away lands on the SAME handle when the block comes back from the pool
    (krypy_amd/_hip.py: _pool_take) and, freed for real (`_pool_flush` -> kh_vec_free -> forget_steps), possibly on the
    same device address behind a new handle.  Either way the key could still match while the block holds another
    basis.  Checked by continuing a DIFFERENT sequence, at the very step number the dead one had reached, on the block
    that took the dead one's place: the result must be the undisturbed second sequence (the columns arrive through
    kh_vec_upload, which withdraws the table: chain_blk_touch)."""
    from krypy_amd import utils

    A = ref.laplace2d(300, 300)
    n = A.shape[0]
    rng = np.random.default_rng(12)
    v1, v2 = rng.standard_normal((n, 1)), rng.standard_normal((n, 1))
    op = utils.get_linearoperator(A.shape, A)
    m, kcut = 30, 14

    def steps(ar, count):
        for _ in range(count):
            ar.advance()
        ar._settle()

    ref2 = utils.Arnoldi(op, v2, maxiter=m, ortho="mgs")          # the undisturbed second sequence
    steps(ref2, m)
    V_want, H_want = np.array(ref2.get()[0]), np.array(ref2.get()[1])
    del ref2
    same_handle = 0
    for flush in (False, True):
        a2 = utils.Arnoldi(op, v2, maxiter=m, ortho="mgs")
        steps(a2, kcut)
        cols = a2._V.download(0, kcut + 1)
        a1 = utils.Arnoldi(op, v1, maxiter=m, ortho="mgs")
        steps(a1, kcut)                                  # the table now describes a1's basis, next step: kcut
        shape = (a1._V.n, a1._V.ncols)
        h1 = a1._V.handle.value
        del a1                                           # its block goes back to the pool ...
        if flush:
            hip._pool_flush()                            # ... or is freed for real
        fresh = hip.alloc(shape[0], shape[1], zero=False)
        same_handle += int(fresh.handle.value == h1)
        fresh.upload(0, cols)
        a2._V = fresh
        r0 = hip.get("n_blk_rebuild")
        steps(a2, m - kcut)
        V, H = np.array(a2.get()[0]), np.array(a2.get()[1])
        assert np.linalg.norm(H - H_want) < 1e-10 * np.linalg.norm(H_want), flush
        assert np.linalg.norm(V - V_want) < 1e-9, flush
        expect_kernel(hip.get("n_blk_rebuild") - r0 >= 1, "the block that replaced a dropped basis is rebuilt from (flush=%r)" % flush)
        del a2, fresh
    expect_kernel(same_handle >= 1, "the pool handed the dropped block's handle out again: %d" % same_handle)


@pytest.fixture
def forced_ctx(hip):
    """A context with a 1-rank RCCL communicator in forced multi-rank mode: every inner product goes through the partial
    sums -> device scalar -> ncclAllReduce path the ranks of a node take, the chain kernels are off."""
    import os
    from krypy_amd import _hip

    os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    try:
        ctx = _hip.Context(0)
        ctx.comm_init(0, 1, ctx.comm_unique_id())
    finally:
        del os.environ["KRYPY_AMD_FORCE_MULTI"]
    old = _hip._install_context_for_testing(ctx)
    yield ctx
    _hip._install_context_for_testing(old)
    ctx.close()


@pytest.mark.parametrize("nx,ny", [(90, 90), (301, 211), (1200, 1000), (2000, 1500), (2500, 2001)])
def test_one_reduction_reference_order_gram_schmidt(forced_ctx, nx, ny):
    """ortho='mgs' on the multi-rank path (round 4): all k + 1 coefficients of a step from ONE pass over the local basis
    and ONE all-reduce - alpha = (I + U^T)^{-1} V^T w with the strict upper Gram table U carried by the sequence, the
    reference's loop (utils.py:1012-1029) in exact arithmetic - instead of k + 1 dependent all-reduces.  30 Arnoldi
    steps against the CPU oracle's MGS and against the per-column path of the same context at 1e-10 (H) / 1e-9 (basis);
    two all-reduces per step whatever k; 8,100 rows (4 rows per lane, padded), 63,511 (odd), 1.2 M (8 rows per lane),
    3 M (16 rows: the second right-hand side fills LDS), 5 M (24 rows: eight of its rows in registers - a rank's share of the
    benchmark problem on two GPUs); the table is rebuilt from the basis when the
    block is not the one the sequence has been writing (a basis grown on demand)."""
    from krypy_amd import utils

    ctx = forced_ctx
    A = ref.laplace2d(nx, ny)
    n = A.shape[0]
    v = np.random.default_rng(5).standard_normal((n, 1))
    m = 30
    st = ref.arnoldi_init(A, v[:, 0], m, ortho="mgs")
    for _ in range(m):
        ref.arnoldi_step(st)
    out = {}
    for low in (1, 0):
        ctx.set("mgs_lowsync", low)
        try:
            ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=m, ortho="mgs")
            a0, l0 = ctx.get("n_allreduce"), ctx.get("n_lowsync")
            for _ in range(m):
                ar.advance()
            ar._settle()
            out[low] = dict(H=np.array(ar.H), V=ar.V, allred=ctx.get("n_allreduce") - a0, steps=ctx.get("n_lowsync") - l0)
            del ar
        finally:
            ctx.set("mgs_lowsync", 1)
    a, c = out[1], out[0]
    expect_kernel(a["steps"] == m and c["steps"] == 0, "one-reduction steps m / 0: %r" % ((a["steps"], c["steps"]),))
    expect_kernel(a["allred"] == 2 * m, "2 all-reduces per step (the coefficients, the norm): %r" % (a["allred"],))
    expect_kernel(c["allred"] == sum(k + 2 for k in range(m)), "per-link path: one all-reduce per link + the norm: %r" % (c["allred"],))
    hn = np.linalg.norm(st.H)
    assert np.linalg.norm(a["H"] - st.H) < 1e-10 * hn and np.linalg.norm(c["H"] - st.H) < 1e-10 * hn
    assert np.linalg.norm(a["H"] - c["H"]) < 1e-11 * hn
    assert np.max(np.abs(a["V"] - st.V)) < 1e-9 and np.max(np.abs(a["V"] - c["V"])) < 1e-9
    G = a["V"].T.dot(a["V"]) - np.eye(m + 1)
    Gc = c["V"].T.dot(c["V"]) - np.eye(m + 1)
    assert np.linalg.norm(G) <= 2.0 * np.linalg.norm(Gc) + 1e-13, (np.linalg.norm(G), np.linalg.norm(Gc))


def test_one_reduction_gram_schmidt_whole_solves(forced_ctx, monkeypatch):
    """Whole solves through the one-reduction path against the CPU oracle: a restarted GMRES (every cycle a new basis:
    the table starts over), a GMRES whose basis grows on demand (a new block mid-sequence: the table is rebuilt from the
    basis), GMRES(100) cycles at 1.25 M rows - the shard one of eight ranks holds of the benchmark problem - with the
    orthogonality of the basis next to the per-column path's."""
    from krypy_amd import linsys, utils
    from oracle.inputs import lap2d_system

    ctx = forced_ctx
    A, b = lap2d_system(64, rhs="rng1")
    l0 = ctx.get("n_lowsync")
    try:
        sol = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=30, max_restarts=60, tol=1e-9)
    except utils.ConvergenceError as e:
        sol = e.solver
    want = ref.restarted_gmres(A, b, maxiter=30, max_restarts=60, tol=1e-9)
    expect_kernel(ctx.get("n_lowsync") - l0 >= 30, "ctx.get(\"n_lowsync\") - l0 >= 30")
    assert len(sol.resnorms) == len(want.resnorms)
    r, wr = np.array(sol.resnorms), np.array(want.resnorms)
    assert np.max(np.abs(r[:31] - wr[:31]) / wr[:31]) < 1e-10          # the first cycle closed-loop; open loop beyond
    assert np.linalg.norm(sol.xk[:, 0] - want.xk) < 1e-8 * np.linalg.norm(want.xk)
    # a basis that grows on demand: the new block is not the one the table belongs to
    monkeypatch.setattr(utils.Arnoldi, "_max_initial_cols", 8)
    r0 = ctx.get("n_ls_rebuild")
    try:
        sol = linsys.Gmres(linsys.LinearSystem(A, b), maxiter=60, tol=1e-9)
    except utils.ConvergenceError as e:
        sol = e.solver
    monkeypatch.undo()
    want = ref.gmres(A, b, maxiter=60, tol=1e-9)
    expect_kernel(ctx.get("n_ls_rebuild") - r0 >= 3, "ctx.get(\"n_ls_rebuild\") - r0 >= 3")
    r, wr = np.array(sol.resnorms), np.array(want.resnorms)
    assert len(r) == len(wr) and np.max(np.abs(r[:-1] - wr[:-1]) / wr[:-1]) < 1e-10
    assert np.linalg.norm(sol.xk[:, 0] - want.xk) < 1e-10 * np.linalg.norm(want.xk)
    # the 1/8 shard of the benchmark problem: one cycle both ways
    A2 = ref.laplace2d(4000, 313)
    b2 = np.random.default_rng(0).standard_normal(A2.shape[0])
    out = {}
    for low in (1, 0):
        ctx.set("mgs_lowsync", low)
        try:
            try:
                s = linsys.Gmres(linsys.LinearSystem(A2, b2), maxiter=100, tol=1e-14, store_arnoldi=True)
            except utils.ConvergenceError as e:
                s = e.solver
            Vb = s.arnoldi._V
            G = ctx.gemm_tn(Vb, 0, 101, Vb, 0, 101)
            out[low] = (np.array(s.resnorms), float(np.linalg.norm(G - np.eye(101))))
            del s, Vb
        finally:
            ctx.set("mgs_lowsync", 1)
    assert np.max(np.abs(out[1][0] - out[0][0]) / out[0][0]) < 1e-10
    assert out[1][1] <= 2.0 * out[0][1] + 1e-13, (out[1][1], out[0][1])


@pytest.mark.parametrize("n,d", [(3_000_000, 16), (5_000_001, 7), (8_000_000, 16), (10_200_000, 16), (12_500_000, 16),
                                 (14_000_000, 3)])
def test_projector_with_the_vector_in_registers(hip, n, d):
    """Projection.apply_complement (utils.py:604-627) as ONE launch with z register-resident (proj_reg.h: both sweeps, a
    grid-wide sum of d values each, T c on every workgroup) against the four-launch form and against NumPy: z and
    <Y, z_in> at 1e-13 of their norms.  16 ... 56 rows per lane (the last two shapes keep rows of z in LDS), d = 16 and
    ragged d, T and WRH given or identity."""
    rng = np.random.default_rng(n % 1000 + d)
    Wh = np.empty((n, d), order="F")
    Vh = np.empty((n, d), order="F")
    for j in range(d):
        Wh[:, j] = rng.standard_normal(n) / np.sqrt(n)
        Vh[:, j] = rng.standard_normal(n) / np.sqrt(n)
    a = rng.standard_normal(n)
    ident = d == 7
    T = None if ident else rng.standard_normal((d, d))
    WRH = None if ident else rng.standard_normal((d, d))
    Wd, Vd = hip.upload(Wh), hip.upload(Vh)
    pj = hip.proj_create(Wd, Vd, d, T, WRH, 2)
    A = hip.upload(a)
    out = {}
    # 1: one launch, z in registers; 0: the passes over W and V through the register-resident panel kernels (what N ranks
    # run, with the all-reduce in between); 2: the chunked kernels (k_multidot<16> / k_multiaxpy<16>)
    for reg in (1, 0, 2):
        hip.set("proj_reg", 1 if reg == 1 else 0)
        hip.set("proj_panel", 0 if reg == 2 else 1)
        try:
            Z = hip.alloc(n, 1)
            c0, p0 = hip.get("n_proj_reg"), hip.get("n_proj_panel")
            ya = hip.proj_apply_complement(pj, A, 0, Z, 0, want_ya=True)
            out[reg] = (Z.download()[:, 0], np.array(ya), hip.get("n_proj_reg") - c0, hip.get("n_proj_panel") - p0)
            del Z
        finally:
            hip.set("proj_reg", 1)
            hip.set("proj_panel", 1)
    expect_kernel(out[1][2] == 1 and out[0][2] == 0 and out[2][2] == 0, "one-launch projector 1 / 0 / 0: %r" % ((out[1][2], out[0][2], out[2][2]),))
    expect_kernel(out[1][3] == 0 and out[0][3] == (2 if n > 8_400_000 else 0) and out[2][3] == 0,
                  "panel passes (40 rows per lane and more): %r" % ((out[1][3], out[0][3], out[2][3]),))
    z = a.copy()
    ya_want = None
    for it in range(2):
        c = Wh.T.dot(z)
        if it == 0:
            ya_want = c if WRH is None else WRH.dot(c)
        z = z - Vh.dot(c if T is None else T.dot(c))
    zn = np.linalg.norm(z)
    for reg in (1, 0, 2):
        assert np.linalg.norm(out[reg][0] - z) < 1e-13 * zn * max(1.0, np.linalg.norm(T) if T is not None else 1.0), reg
        assert np.linalg.norm(out[reg][1] - ya_want) < 1e-13 * max(np.linalg.norm(ya_want), 1.0), reg
    assert np.linalg.norm(out[1][0] - out[0][0]) < 1e-13 * zn * max(1.0, np.linalg.norm(T) if T is not None else 1.0)


def test_epoch_wrap_in_the_middle_of_a_blocked_sequence(hip):
    """The epoch counter of the grid-wide sums is brought back to 1 after ~4e9 sums: granules zeroed, election stamps
    cleared.  The blocked kernel's Gram table lives behind the same allocation - ADVICE r04: the reset used to zero it
    while (blk_V, blk_next) still vouched for it, and the rest of the running sequence silently lost its corrections.
    Forced here (kh_ctx_set "chain_epoch") between two blocked steps of one Arnoldi sequence whose basis is NOT
    orthonormal to working precision by then (a non-normal operator makes the defect, hence the corrections, large
    enough to see): the H columns and the basis must be the undisturbed run's."""
    from krypy_amd import utils

    n = 90000
    rng = np.random.default_rng(21)
    A = (ref.laplace2d(300, 300) + sp.diags(rng.standard_normal(n - 1) * 0.8, 1, shape=(n, n))).tocsr()
    v = rng.standard_normal((n, 1))
    m = 36

    def run(wrap_at):
        ar = utils.Arnoldi(utils.get_linearoperator(A.shape, A), v, maxiter=m, ortho="mgs")
        w0, r0, b0 = hip.get("n_epoch_wraps"), hip.get("n_blk_rebuild"), hip.get("n_chain_blk")
        for k in range(m):
            if k == wrap_at:
                ar._settle()
                hip.set("chain_epoch", 0xfff00000 + 5)
            ar.advance()
        ar._settle()
        V, H = ar.get()[:2]
        return np.array(V), np.array(H), hip.get("n_epoch_wraps") - w0, hip.get("n_blk_rebuild") - r0, hip.get("n_chain_blk") - b0

    V0, H0, w_plain, r_plain, b_plain = run(-1)
    V1, H1, w_wrap, r_wrap, b_wrap = run(20)
    assert np.linalg.norm(H1 - H0) < 1e-11 * np.linalg.norm(H0)
    assert np.linalg.norm(V1 - V0) < 1e-10
    expect_kernel(w_plain == 0 and w_wrap == 1, "epoch wraps 0 / 1: %r" % ((w_plain, w_wrap),))
    # (the settle in front of the forced wrap begins the step in flight again: one launch more)
    expect_kernel(b_wrap in (b_plain, b_plain + 1) and b_plain >= m - 8, "the same blocked launches: %r" % ((b_plain, b_wrap),))
    expect_kernel(r_wrap == r_plain + 1, "the table is rebuilt once more after the wrap: %r" % ((r_plain, r_wrap),))


def test_projector_timeout_is_recovered(hip):
    """The one-launch deflation projector (proj_reg.h) reports a timed-out grid-wide sum in a word of its OWN (ADVICE r04:
    it used to share the chain kernels' word, which only a chained Gram-Schmidt step looked at - a panel (cgs) step or a
    stand-alone kh_proj_apply_complement went on with a garbage projection, and the sticky word made later chain
    kernels break out of their waits).  kh_ctx_set("proj_fault", 1) makes the next launch leave the word set and garbage
    in the vector: (i) a deflated GMRES with ortho='cgs' re-runs the step with the four-launch projector and ends on
    the undisturbed history; (ii) the stand-alone call returns the right projection; the kernel stays off afterwards."""
    import bench
    from krypy_amd import deflation, linsys, utils

    A = bench.laplace3d(130, 130, 130)
    N = A.shape[0]
    rng = np.random.default_rng(7)
    b = rng.standard_normal(N)
    U = np.linalg.qr(rng.standard_normal((N, 6)))[0]
    ls = linsys.LinearSystem(A, b, self_adjoint=True)

    def run(fault_at, ortho):
        hip.set("proj_reg", 1)

        class Faulty(deflation.DeflatedGmres):
            def _finalize_iteration(self, yk, resnorm):
                if self.iter == fault_at:
                    hip.set("proj_fault", 1)
                return super(Faulty, self)._finalize_iteration(yk, resnorm)

        r0, p0 = hip.get("n_proj_recovered"), hip.get("n_proj_reg")
        try:
            s = Faulty(ls, U=U, tol=1e-30, maxiter=24, ortho=ortho)
        except utils.ConvergenceError as e:
            s = e.solver
        return np.array(s.resnorms), np.array(s.xk), hip.get("n_proj_recovered") - r0, hip.get("n_proj_reg") - p0, hip.get("proj_reg")

    try:
        for ortho in ("cgs", "mgs"):
            good, xg, n0, used0, on0 = run(-1, ortho)
            bad, xb, n1, used1, on1 = run(6, ortho)
            assert len(bad) == len(good) == 25
            assert np.max(np.abs(bad - good) / good) < 1e-9, ortho
            assert np.linalg.norm(xb - xg) < 1e-9 * np.linalg.norm(xg), ortho
            expect_kernel(n0 == 0 and n1 == 1, "%s: recoveries 0 / 1: %r" % (ortho, (n0, n1)))
            expect_kernel(used0 >= 24 and on0 == 1 and on1 == 0, "%s: one-launch projector used / left on / switched off: %r" % (ortho, (used0, on0, on1)))
        # stand-alone
        hip.set("proj_reg", 1)
        d = 6
        Wh = np.linalg.qr(rng.standard_normal((N, d)))[0]
        Wd, Vd = hip.upload(Wh), hip.upload(Wh)
        pj = hip.proj_create(Wd, Vd, d, None, None, 2)
        a = rng.standard_normal(N)
        Ad = hip.upload(a)
        outs = []
        for fault in (0, 1):
            hip.set("proj_reg", 1)
            hip.set("proj_fault", fault)
            Z = hip.alloc(N, 1)
            r0, u0 = hip.get("n_proj_recovered"), hip.get("n_proj_reg")
            ya = hip.proj_apply_complement(pj, Ad, 0, Z, 0, want_ya=True)
            outs.append((Z.download()[:, 0], np.array(ya), hip.get("n_proj_recovered") - r0, hip.get("proj_reg"), hip.get("n_proj_reg") - u0, hip.get("proj_reg_why")))
        z = a - Wh.dot(Wh.T.dot(a))
        z = z - Wh.dot(Wh.T.dot(z))
        for zz, ya, _, _, _, _ in outs:
            assert np.linalg.norm(zz - z) < 1e-12 * np.linalg.norm(z)
            assert np.linalg.norm(ya - Wh.T.dot(a)) < 1e-12 * np.linalg.norm(a)
        expect_kernel(outs[0][2:5] == (0, 1, 1) and outs[1][2:5] == (1, 0, 1),
                      "stand-alone: (recoveries, kernel on, one-launch calls) (0, 1, 1) / (1, 0, 1): %r" % ([o[2:] for o in outs],))
    finally:
        hip.set("proj_fault", 0)
        hip.set("proj_reg", 1)
