"""Generate the golden fixtures under ``tests/golden/`` from the REAL reference.

Run in the build container only (needs ``/root/reference``)::

    python -m oracle.gen_golden            # writes tests/golden/*.npz

Every fixture holds plain arrays: the outputs of the unmodified reference on
inputs that both sides regenerate from seeds (``inputs`` below), never the
reference's objects, source or bytecode.  The reference's own 18 known-answer
scalars (test/test_convenience_wrappers.py:10-12,37-39) are re-derived by
``tests/test_oracle_golden.py`` from the ``toy`` fixture and compared with the
literal numbers quoted in BASELINE.md.
"""
import os
import sys
import warnings

import numpy as np
import scipy
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refshim  # noqa: E402
from oracle.inputs import (  # noqa: E402
    toy_system, lap2d_system, minres_jacobi_system, dense_spd_system, lap3d_system,
    kernel_panel, complex_systems, complex_panel, run_solver_matrix, run_deflation_matrix,
)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def meta():
    return dict(numpy_version=np.__version__, scipy_version=scipy.__version__)


def save(name, **arrays):
    arrays.update(meta())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-40s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def col_checks(V, stride=997):
    """Per-column checksums used instead of a full basis: sum, sum|.|, strided sample."""
    return (V.sum(axis=0), np.abs(V).sum(axis=0), np.ascontiguousarray(V[::stride, :]))


def gen_toy(krypy):
    A, b = toy_system()
    out = {}
    for name, fn in (("cg", krypy.cg), ("minres", krypy.minres), ("gmres", krypy.gmres)):
        x, sol = fn(A, b)
        out[name + "_x"] = x
        out[name + "_resnorms"] = np.array(sol.resnorms)
        out[name + "_iter"] = sol.iter
        U = np.zeros(100)
        U[0] = 1.0
        x, sol = fn(A, b, U=U)
        out[name + "_defl_x"] = x
        out[name + "_defl_resnorms"] = np.array(sol.resnorms)
        out[name + "_defl_E"] = sol.E
    x, sol = krypy.gmres(A, b, store_arnoldi=True)
    out["gmres_H"] = sol.H
    out["gmres_R"] = sol.R
    out["gmres_V"] = sol.V
    # ConvergenceError semantics
    try:
        krypy.gmres(A, b, maxiter=10)
    except krypy.utils.ConvergenceError as e:
        out["gmres_m10_resnorms"] = np.array(e.solver.resnorms)
        out["gmres_m10_iter"] = e.solver.iter
        out["gmres_m10_xk"] = e.solver.xk
        out["gmres_m10_msg"] = np.array(str(e))
    save("toy", **out)


def gen_lap2d_restart(krypy, nx, rhs):
    A, b = lap2d_system(nx, rhs=rhs)
    ls = krypy.linsys.LinearSystem(A, b)
    tol, m, R = 1e-8, 100, 50
    # reference driver
    sol = krypy.linsys.RestartedGmres(ls, maxiter=m, max_restarts=R, tol=tol)
    # same loop, cycle by cycle, to capture closed-loop data
    xk = None
    cyc = {}
    c = 0
    resn = [np.inf]
    while c == 0 or resn[-1] > tol:
        try:
            s = krypy.linsys.Gmres(ls, x0=xk, maxiter=m, tol=tol, store_arnoldi=True)
        except krypy.utils.ConvergenceError as e:
            s = e.solver
        cyc["c%d_x0" % c] = np.zeros(A.shape[0]) if xk is None else xk[:, 0]
        cyc["c%d_xk" % c] = s.xk[:, 0]
        cyc["c%d_H" % c] = s.H
        cyc["c%d_resnorms" % c] = np.array(s.resnorms)
        xk = s.xk
        del resn[-1]
        resn += s.resnorms
        c += 1
    assert np.array_equal(np.array(resn), np.array(sol.resnorms))
    save("lap2d_restart_nx%d_%s" % (nx, rhs), ncycles=c, total_iters=len(sol.resnorms) - 1,
         resnorms=np.array(sol.resnorms), xk=sol.xk[:, 0], nx=nx, tol=tol, maxiter=m, **cyc)


def gen_lap2d_cycle(krypy, nx=200):
    A, b = lap2d_system(nx, rhs="rng1")
    ls = krypy.linsys.LinearSystem(A, b)
    out = {}
    for ortho in ("mgs", "dmgs"):
        try:
            s = krypy.linsys.Gmres(ls, maxiter=100, tol=1e-8, ortho=ortho, store_arnoldi=True)
        except krypy.utils.ConvergenceError as e:
            s = e.solver
        sums, asums, samp = col_checks(s.V)
        out.update({ortho + "_resnorms": np.array(s.resnorms), ortho + "_H": s.H,
                    ortho + "_xk": s.xk[:, 0], ortho + "_Vsum": sums, ortho + "_Vabssum": asums,
                    ortho + "_Vsample": samp, ortho + "_iter": s.iter})
    save("lap2d_cycle_nx%d" % nx, nx=nx, **out)


def gen_minres(krypy, nx=100):
    A, b, Mj, Minv = minres_jacobi_system(nx)
    ls = krypy.linsys.LinearSystem(A, b, M=Mj, Minv=Minv, self_adjoint=True)
    s = krypy.linsys.Minres(ls, ortho="lanczos", tol=1e-8, maxiter=2000, store_arnoldi=True)
    sums, asums, samp = col_checks(s.V, stride=499)
    psums, pasums, psamp = col_checks(s.P, stride=499)
    save("minres_jacobi_nx%d" % nx, nx=nx, resnorms=np.array(s.resnorms), H=s.H, xk=s.xk[:, 0],
         iter=s.iter, Vshape=np.array(s.V.shape), Vsum=sums, Vabssum=asums, Vsample=samp,
         Psum=psums, Pabssum=pasums, Psample=psamp)
    # unpreconditioned MINRES and CG on the same Laplacian (sparse CG path)
    ls = krypy.linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)
    s = krypy.linsys.Minres(ls, tol=1e-8, maxiter=2000)
    s2 = krypy.linsys.Cg(ls, tol=1e-8, maxiter=2000)
    save("lap2d_minres_cg_nx%d" % nx, nx=nx, minres_resnorms=np.array(s.resnorms),
         minres_xk=s.xk[:, 0], minres_iter=s.iter, cg_resnorms=np.array(s2.resnorms),
         cg_xk=s2.xk[:, 0], cg_iter=s2.iter)


def gen_cg_dense(krypy, n=512):
    A, b = dense_spd_system(n)
    x, s = krypy.cg(A, b, tol=1e-8, store_arnoldi=True)
    save("cg_dense_n%d" % n, n=n, resnorms=np.array(s.resnorms), xk=s.xk[:, 0], iter=s.iter,
         H=s.H)
    # Jacobi-preconditioned CG on the same matrix
    d = 1.0 / np.diag(A)
    M = sp.diags(d).tocsr()
    x, s = krypy.cg(A, b, M=M, tol=1e-8)
    save("cg_dense_jacobi_n%d" % n, n=n, resnorms=np.array(s.resnorms), xk=s.xk[:, 0],
         iter=s.iter)


def gen_deflation(krypy, nx=24):
    A, b = lap3d_system(nx, rhs="ones")
    ls = krypy.linsys.LinearSystem(A, b, self_adjoint=True)
    fac = krypy.recycling.factories.RitzFactorySimple(n_vectors=16, which="sm")
    rec = krypy.recycling.RecyclingGmres()
    out = {}
    sols = []
    for i in range(3):
        s = rec.solve(ls, vector_factory=fac, tol=1e-8, maxiter=300)
        sols.append(s)
        out["s%d_iters" % i] = len(s.resnorms) - 1
        out["s%d_resnorms" % i] = np.array(s.resnorms)
        out["s%d_xk" % i] = s.xk[:, 0]
        out["s%d_E" % i] = s.E
        out["s%d_C" % i] = s.C
        out["s%d_B_" % i] = s.B_
        out["s%d_H" % i] = s.H
        out["s%d_UMlr" % i] = s.UMlr
        # the vectors the factory hands to the next solve
        if i < 2:
            out["s%d_U_next" % i] = fac.get(s)
        r = krypy.deflation.Ritz(s)
        out["s%d_ritz_values" % i] = r.values
    save("deflation_lap3d_nx%d" % nx, nx=nx, **out)


def gen_kernels(krypy):
    """Kernel-level vectors (fixture F7): inner, norm, one Arnoldi step, projection, qr."""
    ku = krypy.utils
    out = {}
    for N in (1, 63, 64, 65, 4097, 100000):
        for k in (1, 2, 16, 101):
            if N * k > 2_000_000 and k == 101:
                continue
            X, w = kernel_panel(N, k, seed=N + k)
            key = "N%d_k%d_" % (N, k)
            out[key + "inner"] = ku.inner(X, w)
            out[key + "norm"] = ku.norm(w)
    # one Arnoldi advance on a Laplacian with a random start, mgs / dmgs / lanczos (+ Jacobi M)
    A, b = lap2d_system(40, rhs="rng1")
    v = b.reshape(-1, 1)
    for ortho in ("mgs", "dmgs", "lanczos"):
        ar = ku.Arnoldi(A, v, maxiter=12, ortho=ortho)
        for _ in range(12):
            ar.advance()
        out["arn_%s_H" % ortho] = ar.H
        out["arn_%s_V" % ortho] = ar.V
    # Householder Arnoldi incl. a Gmres run with it
    ar = ku.Arnoldi(A, v, maxiter=12, ortho="house")
    for _ in range(12):
        ar.advance()
    out["arn_house_H"], out["arn_house_V"] = ar.H, ar.V
    sh = krypy.linsys.Gmres(krypy.linsys.LinearSystem(A, b), ortho="house", tol=1e-9, maxiter=200)
    out["gmres_house_resnorms"], out["gmres_house_xk"] = np.array(sh.resnorms), sh.xk[:, 0]
    d = np.linspace(0.5, 1.5, A.shape[0])
    M = sp.diags(d).tocsr()
    ar = ku.Arnoldi(A, v, maxiter=12, ortho="lanczos", M=M)
    for _ in range(12):
        ar.advance()
    out["arn_lanczosM_H"], out["arn_lanczosM_V"], out["arn_lanczosM_P"] = ar.H, ar.V, ar.P
    ar = ku.Arnoldi(A, v, maxiter=12, ortho="mgs", M=M)
    for _ in range(12):
        ar.advance()
    out["arn_mgsM_H"], out["arn_mgsM_V"], out["arn_mgsM_P"] = ar.H, ar.V, ar.P
    # projection + MGS-qr (Identity *instance* as ip_B, as the deflated solvers pass it)
    X, a = kernel_panel(2000, 16, seed=7)
    Y, _ = kernel_panel(2000, 16, seed=8)
    ipI = ku.IdentityLinearOperator((2000, 2000))
    Q, R = ku.qr(X, ip_B=ipI, reorthos=1)
    out["qr_Q"], out["qr_R"] = Q, R
    Q0, R0 = ku.qr(X, ip_B=ipI, reorthos=0)
    out["qr0_Q"], out["qr0_R"] = Q0, R0
    P = ku.Projection(X, Y, ip_B=ipI)
    z, Ya = P.apply_complement(a, return_Ya=True)
    out["proj_z"], out["proj_Ya"] = z, Ya
    out["proj_apply"] = P.apply(a)
    # Givens on the real part of the reference's factor grid (test/test_utils.py:102)
    fac = [0.0, 1.0, 1e8, 1e-8, -3.0, 4.0]
    g = []
    for a_ in fac:
        for b_ in fac:
            G = ku.Givens(np.array([[a_], [b_]]))
            g.append([a_, b_, G.c, G.s, G.r])
    out["givens"] = np.array(g)
    save("kernels", **out)


def gen_ipB(krypy, nx=24):
    """Non-Euclidean inner product <x,y>_B = x^T B y with a diagonal SPD B:
    GMRES and one Arnoldi run of the reference with ip_B given as a sparse matrix."""
    A, b = lap2d_system(nx, rhs="rng1")
    N = A.shape[0]
    Bd = np.linspace(0.5, 2.0, N)
    B = sp.diags(Bd).tocsr()
    ls = krypy.linsys.LinearSystem(A, b, ip_B=B)
    s = krypy.linsys.Gmres(ls, tol=1e-10, maxiter=200, store_arnoldi=True)
    ar = krypy.utils.Arnoldi(A, b.reshape(-1, 1), maxiter=15, ortho="mgs", ip_B=B)
    for _ in range(15):
        ar.advance()
    save("ipB_lap2d_nx%d" % nx, nx=nx, resnorms=np.array(s.resnorms), xk=s.xk[:, 0], H=s.H,
         iter=s.iter, arn_H=ar.H, arn_V=ar.V)


def gen_complex(krypy, nx=24):
    """Complex (c128) runs of the reference: kernels, Arnoldi in every ortho mode,
    GMRES / MINRES / CG, mixed real-complex inputs, restarted and deflated GMRES."""
    ku, kl = krypy.utils, krypy.linsys
    c = complex_systems(nx)
    N, b, x0, U = c["N"], c["b"], c["x0"], c["U"]
    out = {}
    # kernels
    for n, k in ((1, 1), (65, 3), (4097, 16), (20000, 33)):
        X, w = complex_panel(n, k, seed=n + k)
        out["N%d_k%d_inner" % (n, k)] = ku.inner(X, w)
        out["N%d_k%d_norm" % (n, k)] = ku.norm(w)
    X, a = complex_panel(1500, 8, seed=3)
    Y, _ = complex_panel(1500, 8, seed=4)
    ipI = ku.IdentityLinearOperator((1500, 1500))
    out["qr_Q"], out["qr_R"] = ku.qr(X, ip_B=ipI, reorthos=1)
    P = ku.Projection(X, Y, ip_B=ipI)
    out["proj_z"], out["proj_Ya"] = P.apply_complement(a, return_Ya=True)
    out["proj_apply"] = P.apply(a)
    g = []
    fac = [0.0, 1.0, 1.0j, 1.0 + 1.0j, 1e8, 1e-8j, -3.0 + 4.0j]
    for a_ in fac:
        for b_ in fac:
            G = ku.Givens(np.array([[a_], [b_]]))
            g.append([a_, b_, G.c, G.s, G.r])
    out["givens"] = np.array(g, dtype=complex)
    # Arnoldi
    v = b.reshape(-1, 1)
    for ortho, A in (("mgs", c["nonh"]), ("dmgs", c["nonh"]), ("lanczos", c["hind"]), ("house", c["nonh"])):
        ar = ku.Arnoldi(A, v, maxiter=12, ortho=ortho)
        for _ in range(12):
            ar.advance()
        out["arn_%s_H" % ortho], out["arn_%s_V" % ortho] = ar.H, ar.V
    # real operator, complex start vector
    ar = ku.Arnoldi(c["L"], v, maxiter=8, ortho="mgs")
    for _ in range(8):
        ar.advance()
    out["arn_realA_H"], out["arn_realA_V"] = ar.H, ar.V
    # solvers
    def rec(tag, s):
        out[tag + "_resnorms"], out[tag + "_xk"] = np.array(s.resnorms), s.xk[:, 0]
        out[tag + "_iter"] = getattr(s, "iter", len(s.resnorms) - 1)

    s = kl.Gmres(kl.LinearSystem(c["nonh"], b), tol=1e-10, maxiter=300, store_arnoldi=True)
    rec("gmres", s)
    out["gmres_H"], out["gmres_R"] = s.H, s.R
    rec("gmres_x0", kl.Gmres(kl.LinearSystem(c["nonh"], b), x0=x0, tol=1e-10, maxiter=300))
    rec("gmres_realA", kl.Gmres(kl.LinearSystem(c["L"], b), tol=1e-10, maxiter=300))
    rec("gmres_realb", kl.Gmres(kl.LinearSystem(c["nonh"], b.real.copy()), tol=1e-10, maxiter=300))
    rec("rgmres", kl.RestartedGmres(kl.LinearSystem(c["nonh"], b), tol=1e-9, maxiter=30, max_restarts=40))
    s = kl.Minres(kl.LinearSystem(c["hind"], b, self_adjoint=True), tol=1e-10, maxiter=600,
                  store_arnoldi=True)
    rec("minres", s)
    out["minres_H"] = s.H
    d = np.asarray(c["hpd"].diagonal()).real
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    rec("cg", kl.Cg(kl.LinearSystem(c["hpd"], b, self_adjoint=True, positive_definite=True),
                    tol=1e-10, maxiter=300))
    rec("cg_jacobi", kl.Cg(kl.LinearSystem(c["hpd"], b, M=M, Minv=Minv, self_adjoint=True,
                                           positive_definite=True), tol=1e-10, maxiter=300))
    rec("minres_jacobi", kl.Minres(kl.LinearSystem(c["hind"], b, M=M, Minv=Minv, self_adjoint=True),
                                   tol=1e-10, maxiter=600))
    rec("gmres_jacobi", kl.Gmres(kl.LinearSystem(c["nonh"], b, M=M, Minv=Minv), tol=1e-10, maxiter=300))
    # deflation with a complex basis
    s = krypy.deflation.DeflatedGmres(kl.LinearSystem(c["nonh"], b), U=U, tol=1e-10, maxiter=300,
                                      store_arnoldi=True)
    rec("dgmres", s)
    out["dgmres_E"], out["dgmres_C"], out["dgmres_B_"] = s.E, s.C, s.B_
    rz = krypy.deflation.Ritz(s)
    order = np.argsort(rz.values)           # numpy sorts complex numbers lexicographically
    out["dgmres_ritz_values"] = rz.values[order]
    out["dgmres_ritz_resnorms"] = rz.resnorms[order]
    out["dgmres_ritz_explicit_resnorms"] = rz.get_explicit_resnorms()[order]
    # (on the Hermitian positive definite matrix: with the indefinite one and a random U the oblique
    # projector has norm ~50 and the reference's own Lanczos basis loses orthogonality within 15 steps)
    s = krypy.deflation.DeflatedMinres(kl.LinearSystem(c["hpd"], b, self_adjoint=True), U=U, tol=1e-10,
                                       maxiter=600)
    rec("dminres", s)
    s = krypy.deflation.DeflatedCg(kl.LinearSystem(c["hpd"], b, self_adjoint=True, positive_definite=True),
                                   U=U, tol=1e-10, maxiter=300)
    rec("dcg", s)
    # real system deflated with a complex basis
    s = krypy.deflation.DeflatedGmres(kl.LinearSystem(c["L"], b.real.copy()), U=U, tol=1e-10, maxiter=300)
    rec("dgmres_realsys", s)
    save("complex_nx%d" % nx, nx=nx, **out)


def gen_solver_matrix(krypy):
    """Outcome of every solve of the reference's solver test matrix (oracle.inputs.run_solver_matrix):
    len(resnorms) (negated when the solve ended in a ConvergenceError) and the last residual norm,
    plus a flag telling whether the reference's own outcome survives a 1e-15 relative perturbation
    of the right-hand side (some preconditioner combinations the reference's tests never really
    ran - M = Ml = Mr = inv(A) - are rounding-chaotic: there is nothing to be iterate-identical to)."""
    runs = []
    for perturb in (0.0, 1e-15):
        n_res, last = {}, {}

        def visit(idx, name, Solver, ls, params, sol, failed, A, B, M, Ml):
            n_res[idx] = -len(sol.resnorms) if failed else len(sol.resnorms)
            last[idx] = sol.resnorms[-1]

        total = run_solver_matrix(krypy.linsys, krypy.utils.ConvergenceError, visit, perturb=perturb)
        runs.append((np.array([n_res[i] for i in range(total)], dtype=np.int32),
                     np.array([last[i] for i in range(total)])))
    (n0, l0), (n1, l1) = runs
    stable = (n0 == n1) & (np.abs(l0 - l1) <= 1e-8 * np.abs(l0) + 1e-13)
    print("solver matrix: %d solves, %d stable under a 1e-15 perturbation" % (len(n0), stable.sum()))
    save("solver_matrix", n_res=n0, last=l0, stable=stable)


def gen_deflation_matrix(krypy):
    """Outcomes of the reference on its deflated-solver test matrix: len(resnorms) (negative:
    ConvergenceError), last residual norm, Frobenius norms of E, C, B_ and the Ritz values."""
    rows = {}

    def visit(idx, name, Solver, ls, sol, failed, A, B):
        n = -len(sol.resnorms) if failed else len(sol.resnorms)
        rv = np.sort(np.abs(krypy.deflation.Ritz(sol, mode="ritz").values)) if sol.H.shape[1] + \
            sol.projection.U.shape[1] > 0 else np.zeros(0)
        rows[idx] = (n, sol.resnorms[-1], np.linalg.norm(sol.E), np.linalg.norm(sol.C),
                     np.linalg.norm(sol.B_[: sol.H.shape[1]]), rv)   # (last row: a noise direction
        #                                   when the Krylov space is numerically invariant)

    total = run_deflation_matrix(krypy.linsys, krypy.deflation, krypy.utils.ConvergenceError, visit)
    ritz = np.full((total, 12), np.nan)
    for i in range(total):
        rv = rows[i][5]
        ritz[i, : len(rv)] = rv[:12]
    save("deflation_matrix", n_res=np.array([rows[i][0] for i in range(total)], dtype=np.int32),
         last=np.array([rows[i][1] for i in range(total)]),
         norms=np.array([rows[i][2:5] for i in range(total)]), ritz_abs=ritz)


def gen_api_surface(krypy):
    """Smaller pieces of the API the other fixtures do not touch: utils.arnoldi_res / orthonormality /
    norm_squared / Arnoldi.get_last, explicit_residual=True, the convenience wrappers with every keyword,
    TimedLinearSystem / ConvertedTimedLinearSystem, UnionFactory, operations()."""
    ku, kl = krypy.utils, krypy.linsys
    out = {}
    A, b = lap2d_system(20, rhs="rng1")
    N = A.shape[0]
    v = b.reshape(-1, 1)
    ar = ku.Arnoldi(A, v, maxiter=10, ortho="mgs")
    for _ in range(10):
        ar.advance()
    V, H = ar.get()
    out["arnoldi_res"] = ku.arnoldi_res(A, V, H)
    out["orthonormality"] = ku.orthonormality(V)
    out["norm_squared"] = ku.norm_squared(v)
    Vl, Hl = ar.get_last()
    out["get_last_V"], out["get_last_H"] = Vl, Hl
    Bd = np.linspace(0.5, 2.0, N)
    B = sp.diags(Bd).tocsr()
    out["arnoldi_res_B"] = ku.arnoldi_res(A, V, H, ip_B=B)
    out["orthonormality_B"] = ku.orthonormality(V, ip_B=B)
    # explicit residual in every iteration
    for name, Solver, kw in (("gmres", kl.Gmres, {}), ("minres", kl.Minres, dict(self_adjoint=True)),
                             ("cg", kl.Cg, dict(self_adjoint=True, positive_definite=True))):
        sol = Solver(kl.LinearSystem(A, b, **kw), tol=1e-9, maxiter=200, explicit_residual=True)
        out["expl_%s_resnorms" % name], out["expl_%s_xk" % name] = np.array(sol.resnorms), sol.xk[:, 0]
        out["ops_%s" % name] = np.array([Solver.operations(7)[k] for k in ("A", "M", "Ml", "Mr", "ip_B", "axpy")], dtype=float)
    # convenience wrappers with the whole keyword set
    d = np.asarray(A.diagonal())
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    x0 = 0.1 * np.ones(N)
    U = np.zeros((N, 2))
    U[0, 0] = U[1, 1] = 1.0
    exact = np.linalg.solve(A.toarray(), b)
    for name, fn, extra in (("cg", krypy.cg, {}), ("minres", krypy.minres, dict(ortho="dmgs")),
                            ("gmres", krypy.gmres, dict(ortho="dmgs"))):
        x, sol = fn(A, b, M=M, Minv=Minv, exact_solution=exact, x0=x0, U=U, tol=1e-9, maxiter=300,
                    use_explicit_residual=True, store_arnoldi=True, **extra)
        out["conv_%s_x" % name], out["conv_%s_resnorms" % name] = x, np.array(sol.resnorms)
        out["conv_%s_errnorms" % name], out["conv_%s_H" % name] = np.array(sol.errnorms), sol.H
    x, sol = krypy.gmres(A, b, inner_product=lambda x_, y_: np.dot(x_.conj(), Bd * y_), tol=1e-9, maxiter=300)
    out["conv_gmres_ip_x"], out["conv_gmres_ip_resnorms"] = x, np.array(sol.resnorms)
    # timed systems
    tls = kl.TimedLinearSystem(A, b, M=M, Minv=Minv, self_adjoint=True)
    sol = kl.Minres(tls, tol=1e-9, maxiter=300)
    out["timed_resnorms"] = np.array(sol.resnorms)
    out["timed_keys"] = np.array(sorted(k for k in tls.timings if len(tls.timings[k]) > 0))
    cls = kl.ConvertedTimedLinearSystem(kl.LinearSystem(A, b, self_adjoint=True))
    out["converted_resnorms"] = np.array(kl.Minres(cls, tol=1e-9, maxiter=300).resnorms)
    # UnionFactory of two simple Ritz factories through the recycling driver
    fac = krypy.recycling.factories.UnionFactory([
        krypy.recycling.factories.RitzFactorySimple(n_vectors=2, which="sm"),
        krypy.recycling.factories.RitzFactorySimple(n_vectors=2, which="lm")])
    rec = krypy.recycling.RecyclingMinres()
    ls = kl.LinearSystem(A, b, self_adjoint=True)
    its = []
    for _ in range(3):
        s = rec.solve(ls, vector_factory=fac, tol=1e-9, maxiter=300)
        its.append(len(s.resnorms) - 1)
    out["union_iters"], out["union_xk"] = np.array(its), s.xk[:, 0]
    save("api_surface", **out)


def gen_recycling_toy(krypy):
    """test/test_recycling.py:17-39 run on the reference: 3 recycling solvers x 7 Ritz selections x 3
    consecutive solves of the 100x100 diagonal system - iteration counts, last residual norm, and
    the recycled basis' projector U^T U deviation (orthonormality of what the factory hands on)."""
    N = 100
    d = np.linspace(1, 2, N)
    d[:5] = [1e-8, 1e-4, 1e-2, 2e-2, 3e-2]
    ls = krypy.linsys.LinearSystem(np.diag(d), np.ones((N, 1)), normal=True, self_adjoint=True,
                                   positive_definite=True)
    iters, last, ncols = [], [], []
    for Solver in (krypy.recycling.RecyclingCg, krypy.recycling.RecyclingMinres, krypy.recycling.RecyclingGmres):
        for which in ("lm", "sm", "lr", "sr", "li", "si", "smallest_res"):
            fac = krypy.recycling.factories.RitzFactorySimple(n_vectors=3, which=which)
            rs = Solver()
            for _ in range(3):
                s = rs.solve(ls, vector_factory=fac, maxiter=50, tol=1e-5, x0=None)
                iters.append(len(s.resnorms))
                last.append(s.resnorms[-1])
                ncols.append(s.projection.U.shape[1])
    save("recycling_toy", iters=np.array(iters), last=np.array(last), ncols=np.array(ncols))


def gen_estimate_time(krypy):
    """_DeflationMixin.estimate_time (deflation.py:191-233) of the three deflated solvers with a synthetic timing
    table (one distinct prime per operation, so every term of the operation-count model shows up in the number)."""
    A = np.diag(np.linspace(1.0, 2.0, 30))
    b = np.ones((30, 1))
    U = np.eye(30)[:, :3]
    prices = dict(A=2.0, M=3.0, Ml=5.0, Mr=7.0, ip_B=11.0, axpy=13.0)
    out = []
    for Solver in (krypy.deflation.DeflatedCg, krypy.deflation.DeflatedMinres, krypy.deflation.DeflatedGmres):
        tls = krypy.linsys.TimedLinearSystem(A, b, self_adjoint=True, positive_definite=True)
        s = Solver(tls, U=U, tol=1e-8)
        tls.timings.clear()
        for k, v in prices.items():
            tls.timings[k] = [v, 10.0 * v]          # Timings.get reports the minimum
        for nsteps, ndefl, w in ((7, 3, 1.0), (12, 0, 1.0), (5, 4, 2.5)):
            out.append(s.estimate_time(nsteps, ndefl, deflweight=w))
    save("estimate_time", values=np.array(out))


def gen_edge_cases(krypy):
    """Degenerate input through the solver API (oracle.inputs.edge_scenarios): status / message / length of
    resnorms / last residual / ||xk|| / last error norm as the unmodified reference produces them."""
    from oracle.inputs import run_edge_scenarios
    rows = run_edge_scenarios(krypy)
    save("edge_cases", names=np.array([r[0] for r in rows]), status=np.array([r[1] for r in rows]),
         message=np.array([r[2] for r in rows]), n_res=np.array([r[3] for r in rows], dtype=np.int64),
         last=np.array([r[4] for r in rows]), xnorm=np.array([r[5] for r in rows]), err=np.array([r[6] for r in rows]))


def main():
    warnings.simplefilter("ignore")
    os.makedirs(OUT, exist_ok=True)
    krypy = refshim.load()
    gen_toy(krypy)
    gen_kernels(krypy)
    for nx in (64, 128):
        for rhs in ("ones", "rng1"):
            gen_lap2d_restart(krypy, nx, rhs)
    gen_lap2d_cycle(krypy)
    gen_minres(krypy)
    gen_cg_dense(krypy)
    gen_deflation(krypy)
    gen_ipB(krypy)
    gen_complex(krypy)
    gen_solver_matrix(krypy)
    gen_deflation_matrix(krypy)
    gen_api_surface(krypy)
    gen_recycling_toy(krypy)
    gen_estimate_time(krypy)
    gen_edge_cases(krypy)


if __name__ == "__main__":
    main()
