"""Golden fixtures of BASELINE.json configs 2, 3, 4 AT THEIR STATED SIZES, from the REAL reference.

TEST INFRASTRUCTURE, build container only (needs ``/root/reference``; about half an hour of host time and 45 GB of
memory, which is why it is not part of ``python -m oracle.gen_golden``)::

    python -m oracle.gen_golden_full [2] [3] [4] [5]     # writes tests/golden/config{2,3,4}_full.npz, config5_flow.npz

What rounds 1-4 compared the GPU with at these sizes was the build's own CPU restatement (``oracle/krylov_ref.py``), run
on the GPU box for two minutes per config.  These fixtures pin the same runs to the unmodified reference itself
(``/root/reference/krypy/linsys.py:951-997`` GMRES, ``791-853`` MINRES, ``593-689`` CG): plain arrays only - residual
histories, the Hessenberg / Lanczos matrices, norms, checksums and strided samples of the big vectors (inputs are
regenerated from seeds on both sides, never stored).  ``tests/test_gpu_fullsize.py`` compares the device with them (no
oracle run on the GPU box), ``tests/test_oracle_golden.py`` compares the oracle's small invariants with them here.

Where the comparison bar depends on how far the REFERENCE's own output moves under one rounding error per datum
(un-reorthogonalised Lanczos over 60 steps; the tail of a CG residual history at 1e-8), that movement is measured here,
with the reference, and stored beside the outputs (``sens_*``)."""
import os
import sys
import time
import warnings

import numpy as np
import scipy
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refshim  # noqa: E402
from oracle.inputs import dense_spd_system_blocked  # noqa: E402
from oracle.krylov_ref import laplace2d  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NX, NY = 4000, 2500
STRIDE = 19997          # prime; 501 sampled rows of a 10^7-vector


def save(name, **arrays):
    arrays.update(numpy_version=np.__version__, scipy_version=scipy.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-24s %8.1f KB" % (name, os.path.getsize(path) / 1024.0), flush=True)


def _relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


def gen_config2(krypy):
    """One GMRES(100) cycle on the 4000 x 2500 Laplacian, b = rng(0) normal, x0 = 0 (the cycle bench.py times)."""
    A = laplace2d(NX, NY)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)
    t0 = time.perf_counter()
    try:
        s = krypy.linsys.Gmres(krypy.linsys.LinearSystem(A, b), maxiter=100, tol=1e-8, store_arnoldi=True)
        raise AssertionError("tolerance cannot be reached in one cycle at this N")
    except krypy.utils.ConvergenceError as e:
        s = e.solver
    dt = time.perf_counter() - t0
    print("reference GMRES(100) cycle at N = %d: %.1f s (%.3f it/s)" % (N, dt, 100 / dt), flush=True)
    V = s.V
    assert V.shape == (N, 101)
    save("config2_full", nx=NX, ny=NY, resnorms=np.array(s.resnorms), H=np.array(s.H), xk_norm=np.linalg.norm(s.xk),
         xk_sum=s.xk.sum(), xk_sample=np.ascontiguousarray(s.xk[::STRIDE, 0]), stride=STRIDE,
         Vsum=V.sum(axis=0), Vabssum=np.abs(V).sum(axis=0), Vsample=np.ascontiguousarray(V[::STRIDE, :]),
         v_last_head=np.ascontiguousarray(V[:4096, 100]), reference_seconds=dt)


def gen_config3(krypy, steps=60):
    """`steps` MINRES iterations with the Jacobi preconditioner on the same Laplacian (ortho='lanczos'), and how far the
    reference's own output moves when A and b are perturbed by 1e-15 relative per entry (seed 11, as
    tests/parity_cases.rounding_sensitivity does for the oracle)."""
    A = laplace2d(NX, NY)
    b = np.random.default_rng(0).standard_normal(A.shape[0])

    def run(A_, b_):
        d = A_.diagonal()
        ls = krypy.linsys.LinearSystem(A_, b_, M=sp.diags(1.0 / d).tocsr(), Minv=sp.diags(d).tocsr(), self_adjoint=True)
        try:
            s = krypy.linsys.Minres(ls, ortho="lanczos", tol=1e-8, maxiter=steps, store_arnoldi=True)
            raise AssertionError("tolerance cannot be reached in %d steps at this N" % steps)
        except krypy.utils.ConvergenceError as e:
            s = e.solver
        return s

    t0 = time.perf_counter()
    s = run(A, b)
    dt = time.perf_counter() - t0
    print("reference MINRES, %d steps: %.1f s" % (steps, dt), flush=True)
    res, H, xn = np.array(s.resnorms), np.array(s.H), float(np.linalg.norm(s.xk))
    out = dict(resnorms=res, H=H, xk_norm=xn, xk_sum=s.xk.sum(), xk_sample=np.ascontiguousarray(s.xk[::STRIDE, 0]),
               Vsum=s.V.sum(axis=0), Psum=s.P.sum(axis=0), Vsample=np.ascontiguousarray(s.V[::STRIDE, :]))
    del s
    rng = np.random.default_rng(11)
    bp = b * (1.0 + 1e-15 * rng.standard_normal(b.shape))
    Ap = A.copy().astype(float)
    Ap.data = Ap.data * (1.0 + 1e-15 * rng.standard_normal(Ap.data.shape))
    sp_ = run(Ap, bp)
    sens = dict(sens_resnorms=_relmax(np.array(sp_.resnorms)[:-1], res[:-1]), sens_H=_rel(np.array(sp_.H), H),
                sens_xnorm=abs(float(np.linalg.norm(sp_.xk)) - xn) / xn)
    print("reference's own movement under one rounding error per datum: %r" % sens, flush=True)
    save("config3_full", nx=NX, ny=NY, steps=steps, stride=STRIDE, reference_seconds=dt, **out, **sens)


def gen_config4(krypy, n=32768):
    """The whole CG solve on the dense SPD matrix of order 32768, and the reference's own movement when every row sum
    of the matrix-vector product is taken in 2 / 3 / 5 pieces (what another summation order looks like from outside)."""
    # (G G^T / n + I, b) with G, b from rng(0), the product taken in row blocks: oracle/inputs.py says why
    A, b = dense_spd_system_blocked(n)

    def solve(op):
        ls = krypy.linsys.LinearSystem(op, b, self_adjoint=True, positive_definite=True)
        return krypy.linsys.Cg(ls, tol=1e-8, maxiter=200)

    t0 = time.perf_counter()
    s = solve(A)
    dt = time.perf_counter() - t0
    res, xk = np.array(s.resnorms), s.xk[:, 0].copy()
    print("reference CG at n = %d: %d iterations, %.1f s" % (n, len(res) - 1, dt), flush=True)
    sens = xsens = 0.0
    for parts in (2, 3, 5):
        cuts = [n * i // parts for i in range(parts + 1)]

        def split_dot(X, cuts=cuts):
            Y = A[:, cuts[0]:cuts[1]].dot(X[cuts[0]:cuts[1]])
            for i in range(1, len(cuts) - 1):
                Y = Y + A[:, cuts[i]:cuts[i + 1]].dot(X[cuts[i]:cuts[i + 1]])
            return Y

        o = solve(krypy.utils.LinearOperator((n, n), float, dot=split_dot, dot_adj=split_dot))
        ores = np.array(o.resnorms)
        assert len(ores) == len(res)
        sens = max(sens, _relmax(ores, res))
        xsens = max(xsens, _rel(o.xk[:, 0], xk))
    print("reference's own movement under other summation orders: resnorms %.1e, xk %.1e" % (sens, xsens), flush=True)
    save("config4_full", n=n, resnorms=res, xk=xk, iter=s.iter, sens_resnorms=sens, sens_xk=xsens,
         A_diag_head=np.diag(A)[:64].copy(), b_head=b[:64].copy(), reference_seconds=dt)


def gen_config5_flow(krypy, n1=130, m=60, d=16):
    """BASELINE.json configs[4]'s flow at a size the reference finishes in minutes (a 130^3 grid, N = 2.2 M - long enough for
    the device's fused deflated step, one-launch projector and SpMM set-up to run as they do at 12.5 M rows per rank): plain
    GMRES(m) as `DeflatedGmres(U=None)` (recycling/linsys.py:51-103), the d Ritz vectors of smallest magnitude
    (deflation.py:738-847, factories.py:167-194), `DeflatedGmres(U)` (deflation.py:93-163) - and the reference's own movement
    when U is perturbed by one rounding error per entry (the deflated solve depends on span(U) only)."""
    from oracle.krylov_ref import laplace3d
    A = laplace3d(n1, n1, n1)
    N = A.shape[0]
    b = np.random.default_rng(0).standard_normal(N)

    def run(U):
        ls = krypy.linsys.LinearSystem(A, b, self_adjoint=True)
        try:
            return krypy.deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=m, store_arnoldi=U is None)
        except krypy.utils.ConvergenceError as e:
            return e.solver

    t0 = time.perf_counter()
    s0 = run(None)
    ritz = krypy.deflation.Ritz(s0)
    idx = np.argsort(np.abs(ritz.values))[:d]
    U = ritz.get_vectors(idx)
    assert U.shape == (N, d)
    s1 = run(U)
    dt = time.perf_counter() - t0
    print("reference flow (GMRES(%d), %d Ritz vectors, DeflatedGmres(%d)) at N = %d: %.1f s" % (m, d, m, N, dt), flush=True)
    s1p = run(U * (1.0 + 1e-15 * np.random.default_rng(1).standard_normal(U.shape)))
    w1, w1p = np.array(s1.resnorms), np.array(s1p.resnorms)
    sens = float(np.max(np.abs(w1p[:-1] - w1[:-1]) / w1[:-1]))
    print("reference's own movement under one rounding error per entry of U: %.2e" % sens, flush=True)
    rows = np.arange(0, N, 1009)            # prime stride: 2178 sampled rows of the Ritz vectors (the span test)
    save("config5_flow", n1=n1, m=m, d=d, plain_resnorms=np.array(s0.resnorms), ritz_values_abs=np.sort(np.abs(ritz.values[idx])),
         deflated_resnorms=w1, sens_deflated=sens, U_rows=rows, U_sample=np.ascontiguousarray(np.asarray(U)[rows, :]),
         U_colnorms=np.linalg.norm(np.asarray(U), axis=0), reference_seconds=dt)


def main(argv):
    warnings.simplefilter("ignore")
    os.makedirs(OUT, exist_ok=True)
    krypy = refshim.load()
    which = [int(a) for a in argv] or [2, 3, 4, 5]
    for c in which:
        {2: gen_config2, 3: gen_config3, 4: gen_config4, 5: gen_config5_flow}[c](krypy)


if __name__ == "__main__":
    main(sys.argv[1:])
