"""NumPy test double of ``krypy_amd._hip.Context`` - TEST INFRASTRUCTURE ONLY.

The build container has no GPU, so the CPU test-suite (``-m "not gpu"``) drives the
product's *host layer* (operator algebra, Arnoldi bookkeeping, Givens/QR updates, restart
and deflation logic, error behaviour) with this stand-in for the device library.  It
implements the same method set as ``Context`` with plain NumPy, one method per C entry
point, following the documented semantics of ``include/krylov_hip.h``.

It is installed explicitly by ``tests/conftest.py`` through
``krypy_amd._hip._install_context_for_testing``; the package never imports it, ships no
alternative backend, and the ``-m gpu`` tests never use it (they run the real HIP library).
"""
import numpy as np
import scipy.sparse as sp

from krypy_amd._hip import BackendError


def _bdt(dtype):
    return np.dtype(np.complex128) if np.dtype(dtype).kind == "c" else np.dtype(np.float64)


def _same(what, *blocks):
    """As strict as the HIP library: real and complex blocks never mix in one call."""
    dt = blocks[0].dtype
    for b in blocks[1:]:
        if b.dtype != dt:
            raise BackendError("%s: real and complex device blocks mixed" % what)
    return dt.kind == "c"


def _coef(h, block, what):
    h = np.asarray(h)
    if block.dtype.kind != "c" and h.dtype.kind == "c":
        if np.any(h.imag != 0):
            raise BackendError("%s: complex coefficient for real device blocks" % what)
        h = h.real
    return h


class NumpyVectors(object):
    def __init__(self, ctx, n, ncols, dtype=float):
        self.ctx, self.n, self.ncols = ctx, int(n), int(ncols)
        self.dtype = _bdt(dtype)
        self.a = np.zeros((self.n, self.ncols), order="F", dtype=self.dtype)
        self.handle = self

    def upload(self, col0, arr):
        a = np.asarray(arr)
        if self.dtype.kind != "c" and a.dtype.kind == "c":
            raise BackendError("upload: complex data into a real device block")
        a = np.asarray(a, dtype=self.dtype)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        assert a.shape[0] == self.n
        self.a[:, col0: col0 + a.shape[1]] = a
        return self

    def download(self, col0=0, ncols=None):
        ncols = self.ncols - col0 if ncols is None else ncols
        return np.array(self.a[:, col0: col0 + ncols], order="F", copy=True)

    def zero(self, col0=0, ncols=None):
        ncols = self.ncols - col0 if ncols is None else ncols
        self.a[:, col0: col0 + ncols] = 0.0

    def copy_from(self, dcol, src, scol, ncols=1):
        _same("copy_from", self, src)
        assert src.n == self.n
        self.a[:, dcol: dcol + ncols] = src.a[:, scol: scol + ncols]

    def get(self, col, i0, count=1):
        return self.a[i0: i0 + count, col].copy()

    def set(self, col, i0, values):
        v = _coef(np.asarray(values).reshape(-1), self, "set")
        self.a[i0: i0 + v.size, col] = v

    def zero_range(self, col, i0, count):
        self.a[i0: i0 + count, col] = 0.0


class NumpyMatrix(object):
    def __init__(self, ctx, kind, mat, shape):
        self.ctx, self.kind, self.mat, self.shape = ctx, kind, mat, shape
        self.dtype = _bdt(mat.dtype)
        self.handle = self
        self.nnz = getattr(mat, "nnz", np.size(mat))


class NumpyContext(object):
    """Same public surface as ``krypy_amd._hip.Context``; counts calls for host-logic tests."""

    def __init__(self, comm=None):
        self.calls = {}
        self.rank, self.nranks = 0, 1
        self._comm = comm           # optional object with allreduce(np.ndarray) (gloo tests)
        self.device = -1
        self._alive = True
        self._t0 = None

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def _allreduce(self, x):
        if self._comm is None:
            return x
        x = np.asarray(x)
        if np.iscomplexobj(x):       # (re, im) pairs, like the RCCL all-reduce of complex panels
            return self._comm.allreduce(x.real.copy()) + 1j * self._comm.allreduce(x.imag.copy())
        return self._comm.allreduce(np.asarray(x, dtype=float))

    # bookkeeping
    def sync(self):
        pass

    def close(self):
        pass

    def counters(self):
        return dict(chain=0, chain_lds=0, chain_fused=0, cgs_register=0)

    # kh_ctx_get / kh_ctx_set: the double has no kernels to select - it remembers what it is told and counts nothing
    def get(self, key):
        return self.__dict__.setdefault("_kv", {}).get(key, 0)

    def set(self, key, value):
        self.__dict__.setdefault("_kv", {})[key] = int(value)

    def info(self):
        return dict(compute_units=0, mem_total=0, mem_free=0, reduce_blocks=0)

    def tune(self, reduce_blocks=0, spmv_tile=0):
        pass

    def timer_start(self):
        import time
        self._t0 = time.perf_counter()

    def timer_stop(self):
        import time
        return (time.perf_counter() - self._t0) * 1e3

    def allreduce_host(self, vals):
        return self._allreduce(np.asarray(vals, dtype=float))

    # launcher-side protocol of the real context (kh_comm_unique_id / kh_comm_init): rank 0 makes an id, every rank
    # joins with it.  Here the communicator is torch.distributed's default (gloo) group, created by the launcher.
    def comm_unique_id(self):
        return b"gloo-test-double".ljust(128, b"\0")

    def comm_init(self, rank, nranks, unique_id):
        assert len(unique_id) == 128 and unique_id.startswith(b"gloo-test-double"), "every rank must get rank 0's id"
        self.rank, self.nranks = rank, nranks
        if nranks > 1:
            # the double's "RCCL" is torch.distributed's gloo group; a launcher that does not use torch itself
            # (bench.py: krypy_amd.dist.TcpRendezvous) has not created one
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=rank, world_size=nranks)
        self._comm = GlooComm(rank, nranks) if nranks > 1 else None

    # allocation
    def alloc(self, n, ncols=1, dtype=float, zero=True):
        v = NumpyVectors(self, n, ncols, dtype)
        if not zero:
            v.a[:] = np.nan      # a recycled block holds garbage: whoever reads before writing shows up
        return v

    def upload(self, arr, dtype=None):
        a = np.asarray(arr)
        dt = _bdt(a.dtype if dtype is None else np.result_type(a.dtype, dtype))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return NumpyVectors(self, a.shape[0], a.shape[1], dt).upload(0, a)

    def promote(self, X, xcol, Z, zcol, ncols=1):
        if X.dtype.kind == "c" or Z.dtype.kind != "c":
            raise BackendError("promote: real source and complex destination expected")
        Z.a[:, zcol: zcol + ncols] = X.a[:, xcol: xcol + ncols]

    def csr(self, A, n_cols=None, dtype=None):
        A = sp.csr_matrix(A)
        A = A.astype(_bdt(A.dtype if dtype is None else np.result_type(A.dtype, dtype)))
        return NumpyMatrix(self, "csr", A, A.shape)

    def dense(self, A, dtype=None):
        A = np.asarray(A)
        A = np.ascontiguousarray(A, dtype=_bdt(A.dtype if dtype is None else np.result_type(A.dtype, dtype)))
        return NumpyMatrix(self, "dense", A, A.shape)

    def diag(self, d, dtype=None):
        d = np.asarray(d)
        d = np.ascontiguousarray(d, dtype=_bdt(d.dtype if dtype is None else np.result_type(d.dtype, dtype)))
        return NumpyMatrix(self, "diag", d, (d.size, d.size))

    # numerics
    def _matvec(self, A, x):
        if A.kind == "diag":
            return A.mat * x
        if getattr(A, "halo", None) is not None:
            x = A.halo(x)
        return A.mat.dot(x)

    def apply(self, A, X, xcol, Y, ycol, ncols=1):
        self._count("apply")
        if (A.dtype.kind == "c") != _same("apply", X, Y):
            raise BackendError("apply: %s operator on %s blocks" % (A.dtype, X.dtype))
        for c in range(ncols):
            Y.a[:, ycol + c] = self._matvec(A, X.a[:, xcol + c])

    def dot_panel(self, V, j0, ncols, W, wcol):
        self._count("dot_panel")
        _same("dot_panel", V, W)
        return self._allreduce(V.a[:, j0: j0 + ncols].T.conj().dot(W.a[:, wcol]))

    def gemm_tn(self, X, x0, nx, Y, y0, ny):
        self._count("gemm_tn")
        _same("gemm_tn", X, Y)
        out = X.a[:, x0: x0 + nx].T.conj().dot(Y.a[:, y0: y0 + ny])
        return self._allreduce(out.ravel()).reshape(nx, ny)

    def axpy_panel(self, V, j0, ncols, h, W, wcol):
        self._count("axpy_panel")
        _same("axpy_panel", V, W)
        h = _coef(np.asarray(h).reshape(-1), W, "axpy_panel")
        for j in range(ncols):
            W.a[:, wcol] = W.a[:, wcol] - h[j] * V.a[:, j0 + j]

    def gemm_nn(self, X, x0, k, C, alpha, beta, Y, y0):
        self._count("gemm_nn")
        _same("gemm_nn", X, Y)
        C = _coef(C, Y, "gemm_nn")
        alpha, beta = _coef(alpha, Y, "gemm_nn"), _coef(beta, Y, "gemm_nn")
        if C.ndim == 1:
            C = C.reshape(-1, 1)
        for c in range(C.shape[1]):
            y = np.zeros(Y.n, dtype=Y.dtype) if beta == 0.0 else beta * Y.a[:, y0 + c]
            for i in range(k):
                y = y + (alpha * C[i, c]) * X.a[:, x0 + i]
            Y.a[:, y0 + c] = y

    def nrm2(self, W, wcol):
        self._count("nrm2")
        w = W.a[:, wcol]
        return float(np.sqrt(self._allreduce(np.array([np.vdot(w, w).real]))[0]))

    def waxpby(self, Z, zcol, alpha, X, xcol, beta, Y, ycol):
        self._count("waxpby")
        _same("waxpby", Z, X, Y)
        alpha, beta = _coef(alpha, Z, "waxpby"), _coef(beta, Z, "waxpby")
        a = X.a[:, xcol] if alpha == 1.0 else alpha * X.a[:, xcol]
        if beta == 0.0:
            Z.a[:, zcol] = a
        else:
            Z.a[:, zcol] = a + (Y.a[:, ycol] if beta == 1.0 else beta * Y.a[:, ycol])

    def vdiv(self, Z, zcol, X, xcol, s):
        self._count("vdiv")
        _same("vdiv", Z, X)
        Z.a[:, zcol] = X.a[:, xcol] / _coef(s, Z, "vdiv")

    def arnoldi_step(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1=0.0):
        """Semantics of kh_arnoldi_step (include/krylov_hip.h)."""
        self._count("arnoldi_step")
        B = P if P is not None else V
        cplx = _same("arnoldi_step", V, W)
        if cplx and ((Md is None) != (P is None) or (Md is not None and Md.dtype.kind != "c")):
            raise BackendError("arnoldi_step: the complex step takes a complex operator Md with its block P")
        hcol = np.zeros(k + 2, dtype=V.dtype)
        if A is not None:
            if (A.dtype.kind == "c") != cplx:
                raise BackendError("arnoldi_step: operator / block dtype mismatch")
            W.a[:, wcol] = self._matvec(A, V.a[:, k])
        w = W.a[:, wcol]
        if start > 0 and start == k:
            w = w - h_km1 * B.a[:, k - 1]
        for _ in range(sweeps):
            if gs_mode == 0:
                for j in range(start, k + 1):
                    alpha = self._allreduce(np.array([np.vdot(V.a[:, j], w)]))[0]
                    hcol[j] += alpha
                    w = w - alpha * B.a[:, j]
            else:
                h = self._allreduce(V.a[:, start: k + 1].T.conj().dot(w))
                hcol[start: k + 1] += h
                for j in range(start, k + 1):
                    w = w - h[j - start] * B.a[:, j]
        W.a[:, wcol] = w        # (before the store of v_{k+1}: W may alias the basis block, in-place QR)
        if Md is not None:
            mw = self._matvec(Md, w)      # Jacobi diagonal, or the SPD matrix of a non-Euclidean inner product
            W.a[:, wcol + 1] = mw
            hn = float(np.sqrt(abs(self._allreduce(np.array([np.vdot(w, mw).real]))[0])))
            with np.errstate(divide="ignore", invalid="ignore"):
                P.a[:, k + 1] = w / hn
                V.a[:, k + 1] = mw / hn
        else:
            hn = float(np.sqrt(self._allreduce(np.array([np.vdot(w, w).real]))[0]))
            with np.errstate(divide="ignore", invalid="ignore"):
                V.a[:, k + 1] = w / hn
        hcol[k + 1] = hn
        return hcol

    def arnoldi_step_begin(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1, slot,
                           proj=None):
        if not hasattr(self, "_slots"):
            self._slots = {}
        if h_km1 != h_km1:      # NaN: H[k,k-1] of the step begun just before (device-side value)
            h_km1 = float(np.real(self._slots[(slot - 1) % 4][k]))
        if proj is None:
            self._slots[slot] = self.arnoldi_step(A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1)
            return
        # projected operator: w = (I - P) A v_k, then the Gram-Schmidt part with A = None
        W.a[:, wcol] = self._matvec(A, V.a[:, k])
        ya = self.proj_apply_complement(proj, W, wcol, W, wcol, want_ya=True)
        hcol = self.arnoldi_step(None, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1)
        self._slots[slot] = np.concatenate([hcol, ya])

    def proj_create(self, W, V, d, T, WRH, iterations):
        cplx = _same("proj_create", W, V)     # (kh_zproj_create for complex blocks)
        if cplx and d > 512:
            raise BackendError("kh_zproj_create: at most 512 vectors")
        class _P(object):
            pass
        p = _P()
        p.W, p.V, p.d, p.iterations, p.handle, p.cplx = W, V, d, iterations, None, cplx
        dt = complex if cplx else float
        p.T = None if T is None else np.array(T, dtype=dt)
        p.WRH = None if WRH is None else np.array(WRH, dtype=dt)
        return p

    def proj_apply_complement(self, proj, A, acol, Z, zcol, want_ya=False):
        self._count("proj_apply_complement")
        d = proj.d
        z = A.a[:, acol].copy()
        ya = None
        for it in range(proj.iterations):
            c = self._allreduce(proj.W.a[:, :d].T.conj().dot(z))
            if it == 0 and want_ya:
                ya = c.copy() if proj.WRH is None else proj.WRH.dot(c)
            c = c if proj.T is None else proj.T.dot(c)
            for j in range(d):
                z = z - c[j] * proj.V.a[:, j]
        Z.a[:, zcol] = z
        return ya

    def arnoldi_step_end(self, slot, count, cplx=False):
        return self._slots[slot][:count].copy()

    def residual(self, A, B, bcol, X, xcol, R, rcol):
        self._count("residual")
        if (A.dtype.kind == "c") != _same("residual", B, X, R):
            raise BackendError("residual: operator / block dtype mismatch")
        r = B.a[:, bcol] - self._matvec(A, X.a[:, xcol])
        R.a[:, rcol] = r
        return float(np.sqrt(self._allreduce(np.array([np.vdot(r, r).real]))[0]))

    def minres_flush(self):
        """kh_minres_flush (the double applies every update at once: nothing is ever pending)"""
        self._count("minres_flush")

    def minres_update(self, V, k, Wk, slot, r0, r1, r2, y0, YK, ycol, defer=False):
        self._count("minres_update")
        _same("minres_update", V, Wk, YK)
        r0, r1, r2, y0 = (_coef(t, V, "minres_update") for t in (r0, r1, r2, y0))
        z = ((V.a[:, k] - r0 * Wk.a[:, slot]) - r1 * Wk.a[:, 1 - slot]) / r2
        Wk.a[:, slot] = z
        YK.a[:, ycol] = YK.a[:, ycol] + y0 * z

    def minres_cycle(self, A, Md, V, P, W, k0, k_stop, k_last, base, enq, tol, bnorm, H, Wm, wslot, YK, ycol, st, h2,
                     resn):
        """Semantics of kh_minres_cycle (include/krylov_hip.h): Lanczos steps with one step of look-ahead through the
        four slots, the QR update with the two remembered rotations, the recurrence update, stop reasons."""
        from scipy.linalg import blas
        self._count("minres_cycle")
        if not (base >= 0 and k0 >= base and k0 <= k_stop <= k_last + 1 and k_last + 2 - base <= V.ncols):
            raise BackendError("minres_cycle: step range")
        g1, g2, nrot, y0, y1 = (st[0], st[1]), (st[2], st[3]), int(st[4]), float(st[5]), float(st[6])
        reason, k = 0, k0
        while k < k_stop:
            last = min(k + 1, k_last)
            while enq <= last:
                e = enq
                h_km1 = 0.0
                if e > 0:
                    h_km1 = float(H[e, e - 1]) if e <= k else float("nan")
                self.arnoldi_step_begin(A, Md, V, P, W, 0, e - base, e - base if e > 0 else 0, 1, 0, h_km1, e % 4)
                enq += 1
            kp = k - base
            col = self.arnoldi_step_end(k % 4, kp + 2)
            alpha, hn = float(col[kp]), float(col[kp + 1])
            hkm = float(H[k, k - 1]) if k > 0 else 0.0
            c2 = (hkm * hkm if k > 0 else 0.0) + alpha * alpha + hn * hn
            fro = np.sqrt(h2 + c2)
            if not (fro > 0.0) or not (hn / fro > 1e-14) or not np.isfinite(fro):
                reason = 2
                break
            h2 += c2
            if k > 0:
                H[k - 1, k] = hkm
            H[k, k] += alpha
            H[k + 1, k] = hn
            R0, R1 = 0.0, (hkm if k > 0 else 0.0)
            if nrot >= 2:
                R0, R1 = g1[0] * R0 + g1[1] * R1, -g1[1] * R0 + g1[0] * R1
            R2, R3 = float(H[k, k]), hn
            if nrot >= 1:
                R1, R2 = g2[0] * R1 + g2[1] * R2, -g2[1] * R1 + g2[0] * R2
            g1 = g2
            c, s = blas.drotg(R2, R3)
            c, s = float(c), float(s)
            g2 = (c, s)
            nrot = min(nrot + 1, 2)
            R2 = c * R2 + s * R3
            y0, y1 = c * y0 + s * y1, -s * y0 + c * y1
            self.minres_update(V, kp, Wm, wslot, R0, R1, R2, y0, YK, ycol, defer=True)
            wslot = 1 - wslot
            y0, y1 = y1, 0.0
            resn[k] = abs(y0)
            if not (resn[k] / bnorm > tol):
                k += 1
                reason = 1
                break
            k += 1
        st[0:2], st[2:4], st[4], st[5], st[6] = g1, g2, nrot, y0, y1
        return k, enq, h2, wslot, reason

    def cg_update(self, alpha, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Md, Z, zcol):
        self._count("cg_update")
        cplx = _same("cg_update", Pd, AP, YK, R)
        YK.a[:, ycol] = YK.a[:, ycol] + alpha * Pd.a[:, pcol]
        r = R.a[:, rcol] - alpha * AP.a[:, apcol]
        R.a[:, rcol] = r
        z = r
        if Md is not None:
            z = _jacobi(Md, r, cplx, "cg_update") * r
            Z.a[:, zcol] = z
        return float(self._allreduce(np.array([np.vdot(r, z).real]))[0])


def _jacobi(Md, r, cplx, what):
    """The diagonal kh_cg_update / kh_cg_step scale with: real of length N for real blocks, and for complex blocks
    the REAL diagonal of length 2N (every entry twice) that acts on the interleaved real view."""
    if Md.kind != "diag" or Md.dtype.kind == "c" or Md.mat.size != (2 if cplx else 1) * r.size:
        raise BackendError("%s: Md must be a real diagonal of the real view's length" % what)
    if cplx:
        assert np.array_equal(Md.mat[0::2], Md.mat[1::2])
        return Md.mat[0::2]
    return Md.mat


def _cg_step(self, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, first, omega, rho):
    """Semantics of kh_cg_step (include/krylov_hip.h)."""
    self._count("cg_step")
    cplx = _same("cg_step", Pd, AP, YK, R)
    if cplx != (A.dtype.kind == "c"):
        raise BackendError("cg_step: operator / block dtype mismatch")
    z = Z.a[:, zcol] if Md is not None else R.a[:, rcol]
    if not first:
        Pd.a[:, pcol] = z + omega * Pd.a[:, pcol]
    p = Pd.a[:, pcol]
    ap = self._matvec(A, p)
    AP.a[:, apcol] = ap
    pap = self._allreduce(np.array([np.vdot(p, ap)]))[0]
    if cplx:                      # rho / den = Re(rho / <p, Ap>)  (kh_zcg_step)
        re, im = float(pap.real), float(pap.imag)
        den = re + im * (im / re) if abs(re) >= abs(im) else (re * (re / im) + im) / (re / im)
        pap = complex(pap)
    else:
        den = pap = float(pap)
    with np.errstate(all="ignore"):
        alpha = np.float64(rho) / np.float64(den)
    if not np.isfinite(alpha):        # (the device leaves yk and r untouched and reports it in the sanity word)
        alpha = 0.0
    YK.a[:, ycol] = YK.a[:, ycol] + alpha * p
    r = R.a[:, rcol] - alpha * ap
    R.a[:, rcol] = r
    zz = r
    if Md is not None:
        zz = _jacobi(Md, r, cplx, "cg_step") * r
        Z.a[:, zcol] = zz
    rho_new = float(self._allreduce(np.array([np.vdot(r, zz).real]))[0])
    flags = 0                     # KH_CG_* of include/krylov_hip.h
    if not (np.isfinite(den) and np.isfinite(complex(pap).real) and np.isfinite(complex(pap).imag)):
        flags |= 1
    elif not den > 0.0:
        flags |= 2
    if not np.isfinite(rho_new):
        flags |= 4
    elif rho_new < 0.0:
        flags |= 8
    return den, rho_new, pap, flags


NumpyContext.cg_step = _cg_step


def _cg_cycle(self, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, k0, k_stop, tol, bnorm, rhos, trace):
    """Semantics of kh_cg_cycle (include/krylov_hip.h): iterations of the fused step with omega, rho and the convergence
    test formed here; real data."""
    self._count("cg_cycle")
    if _same("cg_cycle", Pd, AP, YK, R):
        raise BackendError("cg_cycle: real blocks expected")
    reason, k = 0, k0
    while k < k_stop:
        rho = float(rhos[k])
        omega = rho / float(rhos[k - 1]) if k > 0 else 0.0
        den, rho_new, pap, flags = _cg_step(self, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, k == 0, omega, rho)
        trace[6 * k: 6 * k + 6] = (rho, den, pap, rho_new, flags, 0.0)
        if (flags & (1 | 4 | 16)) and np.isfinite(rho):
            reason = 2
            break
        nrm = np.sqrt(abs(rho_new))
        trace[6 * k + 5] = nrm
        rhos[k + 1] = nrm ** 2          # (a NumPy scalar: libm's pow, as in Cg._solve)
        if not (nrm / bnorm > tol):
            k += 1
            reason = 1
            break
        k += 1
    return k, reason


NumpyContext.cg_cycle = _cg_cycle


class GlooComm(object):
    """torch.distributed (gloo, CPU) stand-in for the RCCL calls of libkrylov_hip: sum
    all-reduce of small panels and the nearest-neighbour halo exchange.  world_size-2 tests only."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def allreduce(self, x):
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(np.array(x, dtype=np.float64, copy=True).reshape(-1))
        dist.all_reduce(t)
        return t.numpy().reshape(np.shape(x))

    def exchange(self, x, nsend_prev, nsend_next, nrecv_prev, nrecv_next):
        import torch
        import torch.distributed as dist

        if np.iscomplexobj(x):            # (re, im) pairs, like the RCCL exchange of a complex block
            xr = np.ascontiguousarray(x, dtype=complex).view(float)
            gp, gn = self.exchange(xr, 2 * nsend_prev, 2 * nsend_next, 2 * nrecv_prev, 2 * nrecv_next)
            return gp.view(complex), gn.view(complex)
        reqs = []
        gp = torch.zeros(nrecv_prev, dtype=torch.float64)
        gn = torch.zeros(nrecv_next, dtype=torch.float64)
        if self.rank > 0:
            if nsend_prev:
                reqs.append(dist.isend(torch.from_numpy(np.array(x[:nsend_prev])), self.rank - 1))
            if nrecv_prev:
                reqs.append(dist.irecv(gp, self.rank - 1))
        if self.rank + 1 < self.world:
            if nsend_next:
                reqs.append(dist.isend(torch.from_numpy(np.array(x[len(x) - nsend_next:])),
                                       self.rank + 1))
            if nrecv_next:
                reqs.append(dist.irecv(gn, self.rank + 1))
        for r in reqs:
            r.wait()
        return gp.numpy(), gn.numpy()


def _set_ghost(self, A, values):
    """kh_mat_set_ghost: ghost entries by hand (one process checking a slab against the global operator)"""
    A._ghost = np.array(values, dtype=A.dtype)


def _set_halo(self, A, nsend_prev, nsend_next, nrecv_prev, nrecv_next):
    comm = self._comm

    def halo(x):
        if comm is None:        # one rank: nothing to exchange (the device library skips the hook too) ...
            if nrecv_prev or nrecv_next:      # ... ghost entries written by hand (kh_mat_set_ghost)
                g = getattr(A, "_ghost", None)
                assert g is not None and g.size == nrecv_prev + nrecv_next, "set_ghost first"
                return np.concatenate([x, g])
            return x
        gp, gn = comm.exchange(x, nsend_prev, nsend_next, nrecv_prev, nrecv_next)
        return np.concatenate([x, gp, gn])

    A.halo = halo


NumpyContext.set_halo = _set_halo
NumpyContext.set_ghost = _set_ghost
