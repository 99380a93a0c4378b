"""CPU tests of the product's HOST LAYER (krypy_amd.utils/linsys/deflation/_convenience).

There is no GPU in the build container: these tests install the NumPy test double
(tests/support/numpy_context.py) in place of the HIP context and run the shared parity
cases against the golden vectors of the real reference.  They prove the host logic (operator
algebra, Arnoldi bookkeeping, Givens/QR updates, restart/deflation logic, error behaviour);
the kernels themselves are proven by the same cases in tests/test_gpu_parity.py (-m gpu).
"""
import pytest

from tests import parity_cases as pc
from tests import parity_cases_complex as pcc
from tests import parity_cases_utils as pcu

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


@pytest.mark.parametrize("case", SIMPLE, ids=lambda f: f.__name__)
def test_case(cpu_double, case):
    case()


@pytest.mark.parametrize("nx,rhs,ortho", [(64, "ones", "mgs"), (64, "rng1", "mgs"),
                                           (64, "rng1", "cgs2"), (128, "ones", "mgs")])
def test_restarted_gmres(cpu_double, nx, rhs, ortho):
    pc.case_restarted_gmres(nx, rhs, ortho)


@pytest.mark.parametrize("ortho", ["mgs", "dmgs", "cgs", "cgs2"])
def test_one_cycle_nx200(cpu_double, ortho):
    pc.case_one_cycle_nx200(ortho)


def test_gmres_issues_one_device_call_per_iteration(cpu_double):
    """The hot loop crosses the C ABI once per Arnoldi step."""
    from krypy_amd import linsys
    from oracle.inputs import lap2d_system

    A, b = lap2d_system(32, rhs="rng1")
    ls = linsys.LinearSystem(A, b)
    cpu_double.calls.clear()
    try:
        linsys.Gmres(ls, maxiter=30, tol=1e-14)
    except Exception:
        pass
    assert cpu_double.calls.get("arnoldi_step") == 30
    assert cpu_double.calls.get("dot_panel", 0) == 0 and cpu_double.calls.get("axpy_panel", 0) == 0


def test_complex_deflated_gmres_projects_inside_the_step(cpu_double):
    """Complex deflated solves used to apply the projector sweep by sweep from Python (two panel products and two
    panel updates per iteration, each a C call with a host synchronisation); with kh_zproj_create the projector runs
    inside the fused complex step like the real one: one begin / end pair per iteration."""
    import numpy as np
    from krypy_amd import deflation, linsys
    from oracle.inputs import complex_systems

    c = complex_systems(24)
    U = np.linalg.qr(np.random.default_rng(2).standard_normal((576, 4)) + 1j * np.random.default_rng(3).standard_normal((576, 4)))[0]
    ls = linsys.LinearSystem(c["nonh"], c["b"])
    cpu_double.calls.clear()
    s = deflation.DeflatedGmres(ls, U=U, tol=1e-9, maxiter=300)
    n = len(s.resnorms) - 1
    assert s.xk.dtype.kind == "c" and n > 20
    assert cpu_double.calls.get("arnoldi_step") >= n          # (the look-ahead may run one step past the end)
    assert cpu_double.calls.get("dot_panel", 0) + cpu_double.calls.get("axpy_panel", 0) < 40, dict(cpu_double.calls)
    r = c["b"] - c["nonh"].dot(s.xk[:, 0])
    assert np.linalg.norm(r) <= 1.01e-9 * np.linalg.norm(c["b"])


def test_fused_cg_step_is_fenced(cpu_double, monkeypatch):
    """The fused CG step forms its step length on the device, so the host never sees a garbage ``<p, Ap>`` unless the
    library reports it: every step's scalars and sanity word land in ``solver.cg_trace``; finite data in and
    non-finite scalars out raise a ``BackendError`` that carries the trace (the iterate is left as it was) instead of
    iterating on; a divisor <= 0 (indefinite operator) is recorded only - the reference iterates on;
    ``KRYPY_AMD_CG_STEP=0`` sends CG to the step-by-step path with the same iterates."""
    import numpy as np
    import pytest
    import scipy.sparse as sp
    from krypy_amd import _hip, linsys, utils
    from oracle.inputs import complex_systems

    c = complex_systems(16)
    hpd = dict(self_adjoint=True, positive_definite=True)
    for A, b in ((c["hpd"], c["b"]), (sp.csr_matrix(c["hpd"].real + sp.identity(256) * 0.0), c["b"].real)):
        s = linsys.Cg(linsys.LinearSystem(A, b, **hpd), tol=1e-10, maxiter=300)
        assert len(s.cg_trace) == min(16, len(s.resnorms) - 1)
        assert all(t[5] == 0 and t[2] > 0 for t in s.cg_trace), list(s.cg_trace)
        monkeypatch.setenv("KRYPY_AMD_CG_STEP", "0")
        cpu_double.calls.clear()
        s0 = linsys.Cg(linsys.LinearSystem(A, b, **hpd), tol=1e-10, maxiter=300)
        monkeypatch.delenv("KRYPY_AMD_CG_STEP")
        assert "cg_step" not in cpu_double.calls and len(s0.cg_trace) == 0
        assert len(s0.resnorms) == len(s.resnorms) and np.allclose(s0.resnorms, s.resnorms, rtol=1e-9)
    # an indefinite operator: the divisor goes negative somewhere, recorded, not raised (linsys.py:640-648)
    Aind = sp.diags(np.r_[np.linspace(1, 2, 50), -np.linspace(1, 2, 50)]).tocsr()
    with pytest.warns(UserWarning):
        try:
            s = linsys.Cg(linsys.LinearSystem(Aind, np.ones(100), self_adjoint=True), tol=1e-12, maxiter=8)
        except utils.ConvergenceError as e:
            s = e.solver
    assert any(t[5] & _hip.CG_NONPOSITIVE_PAP for t in s.cg_trace), list(s.cg_trace)
    # a fault: the step hands back nan for finite input
    real_step = type(cpu_double).cg_step
    calls = []

    def faulty(self, *a):
        den, rho_new, pap, flags = real_step(self, *a)
        calls.append(1)
        if len(calls) == 3:
            return float("nan"), rho_new, pap, _hip.CG_NONFINITE_PAP
        return den, rho_new, pap, flags

    monkeypatch.setattr(type(cpu_double), "cg_step", faulty)
    with pytest.raises(_hip.BackendError) as ei:
        linsys.Cg(linsys.LinearSystem(c["hpd"], c["b"], **hpd), tol=1e-10, maxiter=300)
    assert "non-finite" in str(ei.value) and "flags 1" in str(ei.value)
    monkeypatch.setattr(type(cpu_double), "cg_step", real_step)
    # the same fault inside a run of iterations made by ONE C call (kh_cg_cycle): the iterations the call did record
    # are in the solver's state before the error leaves it - rhos, resnorms and iter agree with each other, exactly as
    # the per-step path leaves them
    from tests.support import numpy_context as nc
    Ar = sp.csr_matrix(c["hpd"].real)
    br = c["b"].real
    real_fn = nc._cg_step
    seen = {}

    class Watched(linsys.Cg):
        def _get_xk(self, yk):
            seen["solver"] = self
            return super(Watched, self)._get_xk(yk)

    for per_step in (False, True):
        count = []

        def faulty_fn(self, *a):
            den, rho_new, pap, flags = real_fn(self, *a)
            count.append(1)
            if len(count) == 4:
                return float("nan"), rho_new, pap, _hip.CG_NONFINITE_PAP
            return den, rho_new, pap, flags

        monkeypatch.setattr(nc, "_cg_step", faulty_fn)
        monkeypatch.setattr(type(cpu_double), "cg_step", faulty_fn)
        if per_step:
            monkeypatch.setenv("KRYPY_AMD_CG_CYCLE", "0")
        cpu_double.calls.clear()
        with pytest.raises(_hip.BackendError):
            Watched(linsys.LinearSystem(Ar, br, **hpd), tol=1e-10, maxiter=300)
        sv = seen.pop("solver")
        assert ("cg_cycle" in cpu_double.calls) == (not per_step)
        seen[per_step] = (sv.iter, len(sv.resnorms), list(sv.resnorms))
        assert sv.iter == 3 and len(sv.resnorms) == 4, (per_step, sv.iter, len(sv.resnorms))
    assert np.allclose(seen[False][2], seen[True][2], rtol=1e-12)
    monkeypatch.delenv("KRYPY_AMD_CG_CYCLE")
    monkeypatch.setattr(nc, "_cg_step", real_fn)
    monkeypatch.setattr(type(cpu_double), "cg_step", real_step)


def test_complex_cg_minres_gmres_with_jacobi_stay_on_the_fused_entries(cpu_double):
    """Complex CG runs one ``kh_zcg_step`` per iteration (real recurrences on the real views, the Jacobi scaling as a
    real diagonal of length 2N); complex MINRES / GMRES with a Jacobi preconditioner run the complex step with its
    complex diagonal and second block instead of a host loop over the Gram-Schmidt links, and the MINRES update is
    one ``kh_zminres_update``.  Same iterates as the CPU oracle."""
    import numpy as np
    import scipy.sparse as sp
    from krypy_amd import linsys
    from oracle import krylov_ref_c as refc
    from oracle.inputs import complex_systems

    c = complex_systems(16)
    b = c["b"]
    d = np.asarray(c["hpd"].diagonal()).real
    M, Minv = sp.diags(1.0 / d).tocsr(), sp.diags(d).tocsr()
    hpd = dict(self_adjoint=True, positive_definite=True)

    def counted(make):
        cpu_double.calls.clear()
        s = make()
        return s, dict(cpu_double.calls)

    for kw in ({}, dict(M=M, Minv=Minv)):
        s, calls = counted(lambda: linsys.Cg(linsys.LinearSystem(c["hpd"], b, **hpd, **kw), tol=1e-10, maxiter=300))
        n = len(s.resnorms) - 1
        assert s.xk.dtype.kind == "c" and calls.get("cg_step") == n, calls
        assert calls.get("dot_panel", 0) + calls.get("waxpby", 0) + calls.get("nrm2", 0) <= 6, calls
        xo, reso = refc.cg(c["hpd"], b, tol=1e-10, maxiter=300, M=kw.get("M"))
        assert len(reso) == len(s.resnorms)
        # (the last entry is the explicitly computed residual b - A x at 1e-10 ||b||: cancellation leaves it ~1e-6 relative)
        assert np.allclose(s.resnorms[:-1], reso[:-1], rtol=1e-8, atol=0) and abs(s.resnorms[-1] / reso[-1] - 1) < 1e-5
        assert np.linalg.norm(s.xk[:, 0] - xo) <= 1e-10 * np.linalg.norm(xo)
    s, calls = counted(lambda: linsys.Minres(linsys.LinearSystem(c["hind"], b, M=M, Minv=Minv, self_adjoint=True),
                                             tol=1e-10, maxiter=600))
    n = len(s.resnorms) - 1
    assert calls.get("arnoldi_step") >= n and calls.get("minres_update") == n, calls
    assert calls.get("dot_panel", 0) + calls.get("axpy_panel", 0) == 0, calls
    xo, reso = refc.minres(c["hind"], b, tol=1e-10, maxiter=600, M=M)
    assert len(reso) == len(s.resnorms) and np.allclose(s.resnorms[:-1], reso[:-1], rtol=1e-7, atol=0)
    s, calls = counted(lambda: linsys.Gmres(linsys.LinearSystem(c["nonh"], b, M=M, Minv=Minv), tol=1e-10, maxiter=300))
    n = len(s.resnorms) - 1
    assert calls.get("arnoldi_step") >= n and calls.get("dot_panel", 0) + calls.get("axpy_panel", 0) == 0, calls
    xo, reso, _, _ = refc.gmres(c["nonh"], b, tol=1e-10, maxiter=300, M=M)
    assert len(reso) == len(s.resnorms) and np.allclose(s.resnorms[:-1], reso[:-1], rtol=1e-7, atol=0)


def test_solvers_leave_no_reference_cycles(cpu_double):
    """A finished solver must die with its last reference: its basis is an (N, m+1) device block (8-10 GB at the
    benchmark sizes) that goes back to the block pool in ``DeviceVectors.__del__``.  A solver <-> operator cycle
    keeps it until the garbage collector gets round to it, and the next solve then pays a fresh ``hipMalloc``
    (measured on config 5: 570 ms instead of 330 ms per solve)."""
    import gc
    import numpy as np
    import oracle.krylov_ref as ref
    from krypy_amd import deflation, linsys, utils

    A = ref.laplace3d(10)
    b = np.random.default_rng(0).standard_normal(A.shape[0])
    ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)

    def run(cls, **kw):
        try:
            return cls(ls, tol=1e-12, maxiter=15, **kw)
        except utils.ConvergenceError as e:
            return e.solver

    s0 = run(deflation.DeflatedGmres, store_arnoldi=True)
    ritz = deflation.Ritz(s0)
    U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:3])
    del s0, ritz
    gc.collect()
    gc.disable()
    try:
        for cls, kw in ((deflation.DeflatedGmres, dict(U=U)), (deflation.DeflatedMinres, dict(U=U)),
                        (deflation.DeflatedCg, dict(U=U)), (linsys.Gmres, {}), (linsys.Minres, {}), (linsys.Cg, {}),
                        (linsys.RestartedGmres, dict(max_restarts=1))):
            s1 = run(cls, **kw)
            assert len(s1.resnorms) > 1
            del s1
            gc.set_debug(gc.DEBUG_SAVEALL)
            gc.collect()
            gc.set_debug(0)
            ours = [type(o).__name__ for o in gc.garbage if (type(o).__module__ or "").startswith("krypy_amd")]
            del gc.garbage[:]
            assert not ours, "%s left cyclic garbage: %s" % (cls.__name__, sorted(set(ours)))
    finally:
        gc.set_debug(0)
        del gc.garbage[:]
        gc.enable()


def test_reference_solver_matrix(cpu_double):
    """All 13,216 solves of the reference's solver test matrix (6 matrices, 3 of them complex, x inner
    products x right-hand sides x preconditioners x solvers x parameters) against the reference's own
    outcomes (tests/golden/solver_matrix.npz)."""
    stats = pcc.case_reference_solver_matrix()
    assert stats["n"] == 13216


def test_reference_deflation_matrix(cpu_double):
    """The reference's deflated-solver test matrix (576 solves, real + complex) against its recorded
    outcomes and the E / C / B_ / Ritz identities of test/test_deflation.py."""
    assert pcc.case_reference_deflation_matrix()["n"] == 576


@pytest.mark.parametrize("case", pcu.CASES, ids=lambda f: f.__name__)
def test_reference_utils_matrix(cpu_double, case):
    """The reference's utils test matrix (test/test_utils.py: House, Givens, Projection, qr, angles,
    hegedus, Arnoldi in every ortho mode, Ritz pairs) on real AND complex matrices."""
    assert case() > 20


def test_solve_fuzz_host_layer_against_the_oracle(cpu_double):
    """The random solves of tools/solve_fuzz.py through the host layer on the NumPy double (sizes up to 20,000)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import solve_fuzz
    for seed in range(24):
        solve_fuzz.one_solve(seed, max_n=20_000)
        solve_fuzz.one_solve_extra(seed, max_n=20_000)


def test_minres_cycle_bookkeeping_equals_the_per_step_loop(cpu_double, monkeypatch):
    """Minres._run_cycle (a run of iterations through Context.minres_cycle - on the GPU kh_minres_cycle, here the NumPy
    restatement of its contract) against the per-step loop: every observable equal - residual history, Lanczos matrix,
    iterate, iteration counters - through window slides (150 steps, 66 columns), Jacobi, a stored basis, maxiter = 2 and 3
    (no room for the call), convergence inside a call."""
    import numpy as np
    import scipy.sparse as sp
    from krypy_amd import linsys, utils
    from oracle.inputs import lap2d_system

    A, b = lap2d_system(24, rhs="rng1")
    M = sp.diags(1.0 / A.diagonal()).tocsr()

    def run(**kw):
        ls = linsys.LinearSystem(A, b, M=kw.pop("M", None), self_adjoint=True)
        try:
            return linsys.Minres(ls, **kw)
        except utils.ConvergenceError as e:
            return e.solver

    for kw, min_calls in ((dict(tol=1e-10, maxiter=400), 1), (dict(tol=1e-10, maxiter=400, M=M), 1),
                          (dict(tol=1e-30, maxiter=150), 3), (dict(tol=1e-8, maxiter=300, store_arnoldi=True), 1),
                          (dict(tol=1e-30, maxiter=2), 0), (dict(tol=1e-30, maxiter=3), 1)):
        n0 = cpu_double.calls.get("minres_cycle", 0)
        s1 = run(**dict(kw))
        calls = cpu_double.calls.get("minres_cycle", 0) - n0
        monkeypatch.setenv("KRYPY_AMD_MINRES_CYCLE", "0")
        s0 = run(**dict(kw))
        assert cpu_double.calls.get("minres_cycle", 0) - n0 == calls
        monkeypatch.delenv("KRYPY_AMD_MINRES_CYCLE")
        assert calls >= min_calls, (kw, calls)
        assert s1.resnorms == s0.resnorms and s1.iter == s0.iter and s1.lanczos.iter == s0.lanczos.iter, kw
        k = s1.lanczos.iter
        assert np.array_equal(s1.lanczos.H[: k + 2, : k + 1], s0.lanczos.H[: k + 2, : k + 1]), kw
        assert np.array_equal(s1.xk, s0.xk), kw


def test_cg_cycle_bookkeeping_equals_the_per_step_loop(cpu_double, monkeypatch):
    """Cg._solve through Context.cg_cycle (on the GPU kh_cg_cycle, here the NumPy restatement of its contract) against the
    per-step loop: residual history, rhos, the trace of the last sixteen steps, iterate and counters equal - converging
    solves with and without Jacobi, an exhausted maxiter, maxiter = 2 (no room for the call) and 3, the default
    tolerance."""
    import numpy as np
    import scipy.sparse as sp
    from krypy_amd import linsys, utils
    from oracle.inputs import lap2d_system

    A, b = lap2d_system(24, rhs="rng1")
    M = sp.diags(1.0 / A.diagonal()).tocsr()

    def run(**kw):
        ls = linsys.LinearSystem(A, b, M=kw.pop("M", None), self_adjoint=True, positive_definite=True)
        try:
            return linsys.Cg(ls, **kw)
        except utils.ConvergenceError as e:
            return e.solver

    for kw, min_calls in ((dict(tol=1e-10, maxiter=400), 1), (dict(tol=1e-10, maxiter=400, M=M), 1), (dict(tol=1e-30, maxiter=40), 1),
                          (dict(tol=1e-30, maxiter=3), 1), (dict(tol=1e-30, maxiter=2), 0), (dict(tol=1e-3), 1)):
        n0 = cpu_double.calls.get("cg_cycle", 0)
        s1 = run(**dict(kw))
        calls = cpu_double.calls.get("cg_cycle", 0) - n0
        monkeypatch.setenv("KRYPY_AMD_CG_CYCLE", "0")
        s0 = run(**dict(kw))
        assert cpu_double.calls.get("cg_cycle", 0) - n0 == calls
        monkeypatch.delenv("KRYPY_AMD_CG_CYCLE")
        assert calls >= min_calls, (kw, calls)
        assert s1.resnorms == s0.resnorms and s1.rhos == s0.rhos and s1.iter == s0.iter, kw
        assert list(s1.cg_trace) == list(s0.cg_trace), kw
        assert np.array_equal(s1.xk, s0.xk), kw


def test_committed_bench_line_fractions_are_fractions():
    """profiles/r02_bench.json (a bench.py line from the MI355X): no roofline fraction above 1, each one a single
    division of numbers in the same object, the traffic from a stamped PMC profile, the contract's fields present."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r02_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "iterations/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["config"]["iterations_timed"] / (d["ms_per_step"] * d["steps"] / 1e3) - d["value"]) < 1e-6 * d["value"]

    def walk(x, path):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "/" + str(k))
        elif isinstance(x, (int, float)) and not isinstance(x, bool) and "frac" in path.rsplit("/", 1)[-1]:
            assert 0.0 < x <= 1.0, (path, x)
    walk(d, "")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["frac_compulsory"] - r["compulsory_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
    assert r["traffic"] is None or (r["traffic"] == r["bytes_per_launch"] and r["traffic"] >= r["compulsory_bytes_per_launch"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s"


def test_round4_bench_line_says_what_was_measured():
    """profiles/r04_bench.json (round 4's bench.py line from the MI355X; the round-3 verdict's item 2): `bytes_per_launch`
    is the ALGORITHMIC figure (every basis column once, the operator's arrays once, v_{k+1} out), `frac` is that figure
    over the HIP-event launch time over the peak, `traffic` is the PMC fabric traffic of the same launches (never less
    than the algorithmic bytes) with `traffic_over_bytes` beside it, the kernel timed is the SOLVER's instantiation
    (operator in the prologue), and the general-CSR / separate-SpMV rates of the reference order are on the line."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "iterations/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-9 * r["achieved"]
    assert abs(r["frac"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-12
    assert 0.0 < r["frac"] <= 1.0
    assert "prologue" in r["kernel"] and "k_mgs_chain_lds<40,false,false,5>" in r["kernel"]
    assert r["traffic"] is not None and r["traffic"] >= r["bytes_per_launch"]
    assert abs(r["traffic_over_bytes"] - r["traffic"] / r["bytes_per_launch"]) < 1e-12
    assert 0.0 < r["frac_traffic"] <= 1.0 and "fabric" in r["traffic_source"].lower()
    assert "hbm_bytes_pmc" not in json.dumps(d)
    modes = d["other_modes"]
    assert any("spmv_dia = 0" in k for k in modes) and any("chain_spmv = 0" in k for k in modes)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s"


def test_givens_scalars_equal_the_givens_class():
    """utils.givens_scalars is what the MINRES / GMRES loops call per iteration (the (2, 1)-array constructor of
    utils.Givens costs 10 us, an iteration at short vectors is one 18 us launch): the same BLAS call on the same values -
    bit for bit for real pairs (zero pairs and complex dtypes with zero imaginary parts included); for complex pairs
    zrotg hands back c with an uninitialised imaginary part (denormal garbage that differs from call to call in the
    reference as well), everything else is identical."""
    import numpy as np
    from krypy_amd import utils

    rng = np.random.default_rng(0)
    for t in range(3000):
        kind = t % 6
        a, b = float(rng.standard_normal()), float(rng.standard_normal())
        if kind == 1:
            a, b = complex(a, rng.standard_normal()), complex(b, rng.standard_normal())
        elif kind == 2:
            b = complex(b, rng.standard_normal())
        elif kind == 3:
            a, b = complex(a, 0.0), complex(b, 0.0)
        elif kind == 4:
            b = 0.0
        elif kind == 5:
            a, b = 0.0, (0.0 if t % 12 == 5 else b)
        g = utils.Givens(np.array([[a], [b]]))
        c, s, r = utils.givens_scalars(a, b)
        if kind in (1, 2):
            assert complex(g.c).real == complex(c).real and complex(g.s) == complex(s)
            assert abs(complex(g.r) - complex(r)) < 1e-300
        else:
            assert type(c) is type(g.c) and (g.c, g.s, g.r) == (c, s, r), (a, b)


def test_blocked_one_ahead_recurrence_is_the_reference_loop():
    """The recurrence of k_mgs_chain_blk (chain_blk.h, blk_alphas) restated in NumPy: the coefficients of a block of four
    columns are taken against a w that neither the block itself NOR the block before it has updated, and corrected with
    the Gram entries (the table's rows: the four columns of the previous block, the earlier columns of the own block).  In
    exact arithmetic that is the reference's sequential loop (krypy/utils.py:1012-1029); here: a basis that is orthonormal
    only to 1e-4, so the corrections are anything but negligible, every ragged length of the last block."""
    import numpy as np

    rng = np.random.default_rng(7)
    n, bc = 400, 4
    for k in (8, 9, 10, 11, 12, 23):
        Q, _ = np.linalg.qr(rng.standard_normal((n, k)))
        V = Q + 1e-4 * rng.standard_normal((n, k))
        w0 = rng.standard_normal(n)
        # the reference's loop
        w, ref_alpha = w0.copy(), np.zeros(k)
        for j in range(k):
            ref_alpha[j] = V[:, j] @ w
            w = w - ref_alpha[j] * V[:, j]
        # the kernel's: c of block i + 1 against the w that block i has not updated
        G = V.T @ V
        nblk = (k + bc - 1) // bc
        w_upd = w0.copy()                     # updated by the blocks < i
        alpha, aprev = np.zeros(k), np.zeros(bc)
        c_next = V[:, 0:bc].T @ w_upd         # block 0 against the initial w
        for i in range(nblk):
            cols = list(range(i * bc, min((i + 1) * bc, k)))
            c = c_next
            if i + 1 < nblk:                  # (under the exchange of block i's sums)
                c_next = V[:, (i + 1) * bc: min((i + 2) * bc, k)].T @ w_upd
            a_blk = np.zeros(bc)
            for l, j in enumerate(cols):
                a = c[l]
                if i > 0:
                    for m in range(bc):
                        a -= aprev[m] * G[(i - 1) * bc + m, j]
                for m in range(l):
                    a -= a_blk[m] * G[cols[m], j]
                a_blk[l] = a
                alpha[j] = a
            for l, j in enumerate(cols):
                w_upd = w_upd - a_blk[l] * V[:, j]
            aprev = a_blk
        assert np.max(np.abs(alpha - ref_alpha)) < 1e-12 * np.max(np.abs(ref_alpha)), k
        assert np.linalg.norm(w_upd - w) < 1e-12 * np.linalg.norm(w0), k


def test_blk2_recurrence_on_row_slabs_is_the_reference_loop():
    """The recurrence of k_mgs_chain_blk2 (chain_blk2.h; the two-block and the ONE-block shapes both run it) restated in NumPy on
    THREE row slabs: per block of four columns the slabs' partial dots are added in rank order (the cross-rank stage), the
    coefficients are corrected with the OWN block's Gram entries only, the four updates follow in the reference's order; the
    dots of the next block are taken after the update.  The tail's sum carries the norm and the new column's table row (dots of
    the last block's columns with the final w, over h).  In exact arithmetic that is the reference's sequential loop
    (krypy/utils.py:1012-1045); here: a basis that is orthonormal only to 1e-4, every ragged length of the last block, and the
    table row the step leaves behind is the Gram row the NEXT step's recurrence needs."""
    import numpy as np

    rng = np.random.default_rng(11)
    n, bc = 300, 4
    slabs = [slice(0, 90), slice(90, 217), slice(217, n)]

    def xsum(parts):                      # rank-ordered sum of the slabs' contributions
        t = 0.0
        for p in parts:
            t = t + p
        return t

    worst = 0.0
    for total in (1, 3, 4, 5, 8, 9, 10, 11, 12, 23):
        Q, _ = np.linalg.qr(rng.standard_normal((n, total)))
        V = Q + 1e-4 * rng.standard_normal((n, total))
        w0 = rng.standard_normal(n)
        # the reference's loop
        w, ref_alpha = w0.copy(), np.zeros(total)
        for j in range(total):
            ref_alpha[j] = V[:, j] @ w
            w = w - ref_alpha[j] * V[:, j]
        ref_h = np.linalg.norm(w)
        ref_v = w / ref_h
        # the kernel's
        G = V.T @ V                       # (only own-block entries are read below)
        wk, alpha = w0.copy(), np.zeros(total)
        for i in range((total + bc - 1) // bc):
            cols = list(range(i * bc, min((i + 1) * bc, total)))
            c = [xsum([V[s, j] @ wk[s] for s in slabs]) for j in cols]
            a_blk = []
            for l, j in enumerate(cols):
                a = c[l]
                for m in range(l):
                    assert cols[m] // bc == j // bc
                    a -= a_blk[m] * G[cols[m], j]
                a_blk.append(a)
                alpha[j] = a
            for l, j in enumerate(cols):
                wk = wk - a_blk[l] * V[:, j]
        pnew = total % bc
        last = list(range(total - pnew, total)) if pnew else []
        tail = [xsum([wk[s] @ wk[s] for s in slabs])] + [xsum([V[s, m] @ wk[s] for s in slabs]) for m in last]
        h = np.sqrt(tail[0])
        vnew = wk / h
        row = [t / h for t in tail[1:]]
        scale = np.max(np.abs(ref_alpha))
        worst = max(worst, np.max(np.abs(alpha - ref_alpha)) / scale, abs(h - ref_h) / ref_h, np.max(np.abs(vnew - ref_v)))
        assert np.max(np.abs(alpha - ref_alpha)) < 1e-12 * scale, total
        assert abs(h - ref_h) < 1e-12 * ref_h and np.max(np.abs(vnew - ref_v)) < 1e-12, total
        # the row left behind = the new column against the earlier columns of ITS block (none when it opens a block)
        assert len(row) == (total % bc)
        for t, m in enumerate(last):
            assert abs(row[t] - V[:, m] @ vnew) < 1e-14, (total, m)
    print("blk2 restatement on three slabs against the reference loop: worst deviation %.1e" % worst)


def test_enable_xr_is_all_or_nothing():
    """`dist.enable_xr`: whether the sums across the ranks run as mailbox kernels or as ncclAllReduce changes what a peer has to
    take part in, so it must come out the same on EVERY rank - whatever fails on whichever rank (the export, the attach, the
    self-test of a few sums).  Three ranks as threads, a rendezvous in memory, contexts that fail where they are told to."""
    import threading

    import numpy as np
    from krypy_amd import _hip, dist

    world = 3

    class Rdv(object):
        lock, cond = threading.Lock(), None
        slots, gen = {}, [0]

        def __init__(self, rank):
            self.rank, self.world = rank, world

        def _exchange(self, value):
            cls = Rdv
            with cls.cond:
                g = cls.gen[0]
                cls.slots.setdefault(g, {})[self.rank] = value
                if len(cls.slots[g]) == world:
                    cls.gen[0] += 1
                    cls.cond.notify_all()
                else:
                    while cls.gen[0] == g:
                        cls.cond.wait(timeout=30)
                return [cls.slots[g][r] for r in range(world)]

        def allgather_bytes(self, data):
            return self._exchange(bytes(data))

        def allreduce_min(self, x):
            return min(self._exchange(float(x)))

    Rdv.cond = threading.Condition(Rdv.lock)

    class Ctx(object):
        def __init__(self, rank, fail):
            self.rank, self.nranks, self.fail, self.on, self.detached = 0, 1, fail, False, False
            self.kv = {}

        def xr_export(self):
            if self.fail == "export":
                raise _hip.BackendError("no fine-grained memory")
            return bytes([self._r]) * 64

        def xr_attach(self, rank, nranks, handles):
            assert len(handles) == 64 * nranks and all(handles[64 * q] in (q, 0) for q in range(nranks))
            if self.fail == "attach":
                raise _hip.BackendError("cannot map")

        def xr_enable(self, rank, nranks):
            self.on, self.rank, self.nranks = True, rank, nranks

        def xr_detach(self):
            self.detached = True

        def set(self, key, value):
            self.kv[key] = value
            if key == "xr" and value == 0:
                self.on = False

        def allreduce_host(self, vals):
            # the self-test's sums: every rank can work the answer out itself (rng streams by rank) - unless told to fail
            if self.fail == "selftest":
                raise _hip.BackendError("rank %d's contribution did not arrive" % self.rank)
            count = len(vals)
            t = self._t
            self._t += 1
            out = np.random.default_rng(1000 * t + 0).standard_normal(count)
            for r in range(1, world):
                out = out + np.random.default_rng(1000 * t + r).standard_normal(count)
            return out

    for failing_rank, fail in ((None, None), (1, "export"), (2, "attach"), (0, "selftest")):
        ctxs, res = [], [None] * world

        def work(r):
            c = Ctx(r, fail if r == failing_rank else None)
            c._r, c._t = r, 0
            ctxs.append((r, c))
            res[r] = dist.enable_xr(c, Rdv(r))

        ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert all(x is not None for x in res), (fail, res)
        assert len(set(res)) == 1, (fail, res)                     # every rank the same answer
        assert res[0] is (fail is None), (fail, res)
        for r, c in ctxs:
            assert c.on == (fail is None), (fail, r)
            if fail is not None and c.on is False and fail != "export":
                assert c.detached or r == failing_rank or fail == "attach", (fail, r)
