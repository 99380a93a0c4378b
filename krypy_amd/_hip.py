"""ctypes binding of ``libkrylov_hip.so`` (C ABI: ``include/krylov_hip.h``).

This is the only place that talks to the device library.  There is no CPU
fallback: if the library is missing, cannot be loaded, or no MI355X is
visible, :class:`BackendError` is raised as soon as a context is needed.

The Python classes here are thin: :class:`Context` owns a ``kh_ctx`` (one HIP
stream, reduction scratch), :class:`DeviceVectors` a column-major block of
fp64 column vectors resident in HBM, :class:`DeviceMatrix` an operator (CSR /
dense / diagonal).  Every numeric method maps 1:1 onto a C entry point.
"""
import ctypes
import os

import numpy

__all__ = ["BackendError", "Context", "DeviceVectors", "DeviceMatrix", "get_context",
           "library_path", "load_library", "GS_MGS", "GS_CGS"]

GS_MGS = 0
GS_CGS = 1

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_H = ctypes.c_void_p   # opaque handles
_I64 = ctypes.c_int64
_D = ctypes.c_double
_INT = ctypes.c_int


# sanity word of the fused CG step (KH_CG_* of include/krylov_hip.h)
CG_NONFINITE_PAP, CG_NONPOSITIVE_PAP, CG_NONFINITE_RHO, CG_NEGATIVE_RHO = 1, 2, 4, 8
CG_STEP_CLAMPED = 16
# stop reasons of kh_gmres_cycle (KH_CYCLE_* of the header)
CYCLE_LIMIT, CYCLE_TOL, CYCLE_CHECK = 0, 1, 2


class BackendError(RuntimeError):
    """HIP / RCCL failure, or the device library is unavailable.

    The reference has no counterpart (it never leaves NumPy); error codes of the
    C ABI are mapped onto this exception, carrying ``kh_last_error()``.
    """


# name -> (argtypes); every function returns int status except kh_last_error/kh_version
_SIGNATURES = {
    "kh_device_count": [ctypes.POINTER(_INT)],
    "kh_ctx_create": [_INT, ctypes.POINTER(_H)],
    "kh_ctx_destroy": [_H],
    "kh_ctx_sync": [_H],
    "kh_ctx_info": [_H, _c_int64_p],
    "kh_ctx_counters": [_H, _c_int64_p],
    "kh_ctx_tune": [_H, _INT, _INT],
    "kh_ctx_set": [_H, ctypes.c_char_p, _I64],
    "kh_ctx_get": [_H, ctypes.c_char_p, _c_int64_p],
    "kh_timer_start": [_H],
    "kh_timer_stop": [_H, _c_double_p],
    "kh_comm_unique_id": [ctypes.c_char_p],
    "kh_comm_init": [_H, _INT, _INT, ctypes.c_char_p],
    "kh_comm_destroy": [_H],
    "kh_comm_allreduce_host": [_H, _c_double_p, _I64],
    "kh_xr_export": [_H, ctypes.c_char_p],
    "kh_xr_attach": [_H, _INT, _INT, ctypes.c_char_p],
    "kh_xr_detach": [_H],
    "kh_mat_xh_export": [_H, _H, ctypes.c_char_p],
    "kh_mat_xh_attach": [_H, _H, ctypes.c_char_p, _I64, _I64, ctypes.c_char_p, _I64, _INT],
    "kh_mat_xh_enable": [_H, _H, _INT],
    "kh_mat_xh_detach": [_H, _H],
    "kh_mat_set_halo": [_H, _H, _I64, _I64, _I64, _I64],
    "kh_mat_set_rows_max": [_H, _I64],
    "kh_mat_set_ghost": [_H, _c_double_p, _I64],
    "kh_mat_get_ghost": [_H, _c_double_p, _I64],
    "kh_vec_alloc": [_H, _I64, _I64, ctypes.POINTER(_H)],
    "kh_vec_free": [_H],
    "kh_vec_shape": [_H, _c_int64_p, _c_int64_p, _c_int64_p],
    "kh_vec_upload": [_H, _I64, _I64, _c_double_p, _I64],
    "kh_vec_download": [_H, _I64, _I64, _c_double_p, _I64],
    "kh_vec_zero": [_H, _I64, _I64],
    "kh_vec_copy": [_H, _I64, _H, _I64, _I64],
    "kh_vec_get": [_H, _I64, _I64, _I64, _c_double_p],
    "kh_vec_set": [_H, _I64, _I64, _I64, _c_double_p],
    "kh_vec_zero_range": [_H, _I64, _I64, _I64],
    "kh_csr_upload": [_H, _I64, _I64, _I64, _c_int32_p, _c_int32_p, _c_double_p,
                      ctypes.POINTER(_H)],
    "kh_dense_upload": [_H, _I64, _I64, _c_double_p, _I64, ctypes.POINTER(_H)],
    "kh_dense_from_block": [_H, _H, _I64, _I64, ctypes.c_double, ctypes.c_double, ctypes.POINTER(_H)],
    "kh_diag_upload": [_H, _I64, _c_double_p, ctypes.POINTER(_H)],
    "kh_mat_free": [_H],
    "kh_mat_diagonals": [_H],
    "kh_apply": [_H, _H, _H, _I64, _H, _I64, _I64],
    "kh_dot_panel": [_H, _H, _I64, _I64, _H, _I64, _c_double_p],
    "kh_gemm_tn": [_H, _H, _I64, _I64, _H, _I64, _I64, _c_double_p],
    "kh_axpy_panel": [_H, _H, _I64, _I64, _c_double_p, _H, _I64],
    "kh_gemm_nn": [_H, _H, _I64, _I64, _c_double_p, _I64, _D, _D, _H, _I64],
    "kh_nrm2": [_H, _H, _I64, _c_double_p],
    "kh_waxpby": [_H, _H, _I64, _D, _H, _I64, _D, _H, _I64],
    "kh_vdiv": [_H, _H, _I64, _H, _I64, _D],
    "kh_arnoldi_step": [_H, _H, _H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _D, _c_double_p],
    "kh_arnoldi_step_begin": [_H, _H, _H, _H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _D, _INT],
    "kh_proj_create": [_H, _H, _H, _I64, _c_double_p, _c_double_p, _INT, ctypes.POINTER(_H)],
    "kh_proj_free": [_H],
    "kh_proj_apply_complement": [_H, _H, _H, _I64, _H, _I64, _c_double_p],
    "kh_arnoldi_step_end": [_H, _INT, _I64, _c_double_p],
    "kh_residual": [_H, _H, _H, _I64, _H, _I64, _H, _I64, _c_double_p],
    "kh_gmres_cycle": [_H, _H, _H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _c_int64_p, _D, _D, _c_double_p, _I64,
                       _c_double_p, _I64, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_int64_p,
                       ctypes.POINTER(ctypes.c_int)],
    "kh_ctx_set_rotg": [_H, ctypes.c_void_p],
    "kh_minres_cycle": [_H, _H, _H, _H, _H, _H, _I64, _I64, _I64, _I64, _c_int64_p, _D, _D, _c_double_p, _I64, _H,
                        ctypes.POINTER(ctypes.c_int), _H, _I64, _c_double_p, _c_double_p, _c_double_p, _c_int64_p,
                        ctypes.POINTER(ctypes.c_int)],
    "kh_minres_update": [_H, _H, _I64, _H, _INT, _D, _D, _D, _D, _H, _I64],
    "kh_minres_update_deferred": [_H, _H, _I64, _H, _INT, _D, _D, _D, _D, _H, _I64],
    "kh_minres_flush": [_H],
    "kh_cg_update": [_H, _D, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _H, _H, _I64, _c_double_p],
    "kh_cg_step": [_H, _H, _H, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _INT, _D, _D,
                   _c_double_p],
    "kh_cg_cycle": [_H, _H, _H, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _I64, _I64, _D, _D, _c_double_p,
                    _c_double_p, _c_int64_p, ctypes.POINTER(ctypes.c_int)],
    "kh_bench_kernel": [_H, _INT, _H, _H, _INT, _c_double_p],
    "kh_bench_arnoldi": [_H, _H, _H, _H, _I64, _INT, _INT, _c_double_p],
    # complex (c128) side: vectors are real blocks of length 2N (interleaved re, im)
    "kh_zcsr_upload": [_H, _I64, _I64, _I64, _c_int32_p, _c_int32_p, _c_double_p,
                       ctypes.POINTER(_H)],
    "kh_zdense_upload": [_H, _I64, _I64, _c_double_p, _I64, ctypes.POINTER(_H)],
    "kh_zdiag_upload": [_H, _I64, _c_double_p, ctypes.POINTER(_H)],
    "kh_zfrom_real": [_H, _H, _I64, _H, _I64, _I64],
    "kh_zdot_panel": [_H, _H, _I64, _I64, _H, _I64, _c_double_p],
    "kh_zaxpy_panel": [_H, _H, _I64, _I64, _c_double_p, _H, _I64],
    "kh_zgemm_nn": [_H, _H, _I64, _I64, _c_double_p, _I64, _D, _H, _I64],
    "kh_zwaxpby": [_H, _H, _I64, _c_double_p, _H, _I64, _c_double_p, _H, _I64],
    "kh_zarnoldi_step": [_H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _c_double_p, _c_double_p],
    "kh_zarnoldi_step_begin": [_H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _c_double_p, _INT],
    "kh_zarnoldi_step_begin_proj": [_H, _H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _c_double_p, _INT],
    "kh_zarnoldi_step_begin_md": [_H, _H, _H, _H, _H, _H, _H, _I64, _I64, _I64, _INT, _INT, _c_double_p, _INT],
    "kh_zminres_update": [_H, _H, _I64, _H, _INT, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _H, _I64],
    "kh_zcg_step": [_H, _H, _H, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _H, _I64, _INT, _D, _D, _c_double_p],
    "kh_zproj_create": [_H, _H, _H, _I64, _c_double_p, _c_double_p, _INT, ctypes.POINTER(_H)],
    "kh_zproj_apply_complement": [_H, _H, _H, _I64, _H, _I64, _c_double_p],
}

_lib = None


def library_path():
    """Location of the in-tree shared library (``KRYPY_AMD_LIB`` overrides the path of the
    *same* HIP library, e.g. an experimental build; there is no alternative backend)."""
    env = os.environ.get("KRYPY_AMD_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libkrylov_hip.so")


def load_library():
    """dlopen ``libkrylov_hip.so`` and declare the C ABI; raises BackendError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise BackendError(
            "libkrylov_hip.so not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C krypy_amd/csrc`. "
            "krypy_amd has no CPU fallback." % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise BackendError("cannot load %s: %s" % (path, exc))
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here == ABI/header mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.kh_last_error.restype = ctypes.c_char_p
    lib.kh_last_error.argtypes = []
    lib.kh_version.restype = ctypes.c_int
    lib.kh_version.argtypes = []
    _lib = lib
    return lib


def exported_symbols():
    """Names the header declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGNATURES) + ["kh_last_error", "kh_version"])


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.kh_last_error()
        raise BackendError("%s failed (status %d): %s" % (
            what, rc, msg.decode("utf-8", "replace") if msg else "?"))


def _dptr(a):
    return a.ctypes.data_as(_c_double_p)


_F64 = numpy.dtype(numpy.float64)
_C128 = numpy.dtype(numpy.complex128)


def _block_dtype(dtype):
    """float64 for every real input type, complex128 for every complex one."""
    return _C128 if numpy.dtype(dtype).kind == "c" else _F64


def _zarr(values):
    """contiguous complex128 copy of ``values`` and a pointer to its (re, im) doubles"""
    a = numpy.ascontiguousarray(values, dtype=numpy.complex128)
    return a, a.ctypes.data_as(_c_double_p)


def _real_scalar(x, what):
    if numpy.imag(x) != 0:
        raise BackendError("%s: complex coefficient for real device blocks" % what)
    return float(numpy.real(x))


def _real_coeffs(h, what):
    h = numpy.asarray(h)
    if h.dtype.kind == "c":
        if numpy.any(h.imag != 0):
            raise BackendError("%s: complex coefficients for real device blocks" % what)
        h = h.real
    return numpy.ascontiguousarray(h, dtype=numpy.float64)


def _same_dtype(what, *blocks):
    dt = blocks[0].dtype
    for b in blocks[1:]:
        if b.dtype != dt:
            raise BackendError("%s: real and complex device blocks mixed (%s)" % (
                what, ", ".join(str(b.dtype) for b in blocks)))
    return dt == _C128


class DeviceMatrix(object):
    """An operator resident on the device (``kh_mat``)."""

    def __init__(self, ctx, handle, kind, shape, nnz=0, dtype=_F64):
        self.ctx, self.handle, self.kind, self.shape, self.nnz = ctx, handle, kind, shape, nnz
        self.dtype = numpy.dtype(dtype)

    @property
    def diagonals(self):
        """Diagonals of the banded copy the library built for a CSR operator (0: CSR kernel)."""
        if self.kind != "csr" or self.dtype != _F64:
            return 0
        return int(self.ctx._lib.kh_mat_diagonals(self.handle))

    def __del__(self):
        try:
            if self.handle is not None and self.ctx._alive:
                self.ctx._lib.kh_mat_free(self.handle)
        except Exception:
            pass
        self.handle = None


class DeviceProjector(object):
    """Device image of a deflation projector (``kh_proj``); keeps its basis blocks alive."""

    def __init__(self, ctx, handle, d, keep):
        self.ctx, self.handle, self.d, self._keep = ctx, handle, d, keep

    def __del__(self):
        try:
            if self.handle is not None and self.ctx._alive:
                self.ctx._lib.kh_proj_free(self.handle)
        except Exception:
            pass
        self.handle = None


class DeviceVectors(object):
    """A block of ``ncols`` column vectors of length ``n`` in HBM (``kh_vec``), fp64 or c128.

    Column-contiguous, 256-byte aligned columns: the device image of the
    reference's ``(N, k)`` ndarrays.  A complex block is a real ``kh_vec`` of length ``2n``
    (interleaved re, im): the real-linear kernels run on it unchanged.
    """

    def __init__(self, ctx, n, ncols, dtype=_F64, zero=True):
        """``zero=False``: the caller writes every column before it reads it (a Krylov basis), so a
        recycled block is handed out as it is - only its padding is guaranteed to be zero (no kernel
        ever writes there).  A fresh allocation is zero-filled either way."""
        self.ctx, self.n, self.ncols = ctx, int(n), int(ncols)
        self.dtype = _block_dtype(dtype)
        self._w = 2 if self.dtype == _C128 else 1      # doubles per entry
        self._rn = self.n * self._w                    # length of the real kh_vec
        h = ctx._pool_take(self._rn, self.ncols, zero)
        if h is None:
            h = _H()
            rc = ctx._with_memory(lambda: ctx._lib.kh_vec_alloc(ctx._h, self._rn, self.ncols, ctypes.byref(h)))
            _check(ctx._lib, rc, "kh_vec_alloc(%d x %d)" % (self._rn, self.ncols))
        self.handle = h

    def __del__(self):
        try:
            if self.handle is not None and self.ctx._alive:
                self.ctx._pool_give(self._rn, self.ncols, self.handle)
        except Exception:
            pass
        self.handle = None

    @property
    def ld(self):
        """Leading dimension (doubles) of the real ``kh_vec`` behind this block."""
        a, b, c = _I64(0), _I64(0), _I64(0)
        _check(self.ctx._lib, self.ctx._lib.kh_vec_shape(self.handle, ctypes.byref(a), ctypes.byref(b),
                                                         ctypes.byref(c)), "kh_vec_shape")
        return c.value

    def upload(self, col0, arr):
        """Copy a host ``(n, k)`` (or ``(n,)``) array into columns ``col0 ..``."""
        a = numpy.asarray(arr)
        if self._w == 1 and a.dtype.kind == "c":
            raise BackendError("upload: complex data into a real device block")
        a = numpy.asarray(a, dtype=self.dtype)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        if a.shape[0] != self.n:
            raise BackendError("upload: %d rows into a block of length %d" % (a.shape[0], self.n))
        a = numpy.asfortranarray(a)
        k = a.shape[1]
        ld = max(a.strides[1] // 8, self._rn) if k > 1 else max(self._rn, 1)
        _check(self.ctx._lib, self.ctx._lib.kh_vec_upload(self.handle, col0, k, _dptr(a), ld),
               "kh_vec_upload")
        return self

    def download(self, col0=0, ncols=None):
        """Return columns ``[col0, col0+ncols)`` as a fresh Fortran-ordered ``(n, ncols)`` array."""
        ncols = self.ncols - col0 if ncols is None else ncols
        out = numpy.empty((self.n, ncols), dtype=self.dtype, order="F")
        if self.n and ncols:
            _check(self.ctx._lib, self.ctx._lib.kh_vec_download(self.handle, col0, ncols, _dptr(out),
                                                                max(self._rn, 1)), "kh_vec_download")
        return out

    def zero(self, col0=0, ncols=None):
        ncols = self.ncols - col0 if ncols is None else ncols
        _check(self.ctx._lib, self.ctx._lib.kh_vec_zero(self.handle, col0, ncols), "kh_vec_zero")

    def copy_from(self, dcol, src, scol, ncols=1):
        _same_dtype("copy_from", self, src)
        _check(self.ctx._lib, self.ctx._lib.kh_vec_copy(self.handle, dcol, src.handle, scol, ncols),
               "kh_vec_copy")

    def get(self, col, i0, count=1):
        """A few consecutive entries of one column as a host array."""
        out = numpy.empty(max(count, 1), dtype=self.dtype)
        _check(self.ctx._lib, self.ctx._lib.kh_vec_get(self.handle, col, i0 * self._w,
                                                       count * self._w, _dptr(out)), "kh_vec_get")
        return out[:count]

    def set(self, col, i0, values):
        v = numpy.asarray(values)
        if self._w == 1 and v.dtype.kind == "c":
            raise BackendError("set: complex values into a real device block")
        a = numpy.ascontiguousarray(v, dtype=self.dtype).reshape(-1)
        _check(self.ctx._lib, self.ctx._lib.kh_vec_set(self.handle, col, i0 * self._w,
                                                       a.size * self._w, _dptr(a)), "kh_vec_set")

    def zero_range(self, col, i0, count):
        _check(self.ctx._lib, self.ctx._lib.kh_vec_zero_range(self.handle, col, i0 * self._w,
                                                              count * self._w), "kh_vec_zero_range")


def _blas_drotg_address():
    """Address of the Fortran ``drotg`` behind ``scipy.linalg.blas.drotg`` (f2py keeps it in a capsule), or None."""
    try:
        from scipy.linalg import blas
        cap = blas.drotg._cpointer
        api = ctypes.pythonapi
        api.PyCapsule_GetName.restype = ctypes.c_char_p
        api.PyCapsule_GetName.argtypes = [ctypes.py_object]
        api.PyCapsule_GetPointer.restype = ctypes.c_void_p
        api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
        return api.PyCapsule_GetPointer(cap, api.PyCapsule_GetName(cap)) or None
    except Exception:
        return None


class Context(object):
    """One GPU, one HIP stream (``kh_ctx``).  Not thread-safe, like the reference."""

    def __init__(self, device=0, lib=None):
        self._lib = load_library() if lib is None else lib
        self._alive = False
        n = _INT(0)
        rc = self._lib.kh_device_count(ctypes.byref(n))
        if rc != 0 or n.value == 0:
            msg = self._lib.kh_last_error()
            raise BackendError("no HIP device visible (%s); krypy_amd needs an MI355X and has no "
                               "CPU fallback" % (msg.decode() if msg else "count=0"))
        h = _H()
        _check(self._lib, self._lib.kh_ctx_create(device, ctypes.byref(h)), "kh_ctx_create")
        self._h = h
        self._alive = True
        self.device = device
        self.rank, self.nranks = 0, 1
        self._pool, self._pool_bytes = {}, 0
        # the C host loops (kh_gmres_cycle, kh_minres_cycle) generate their Givens rotations with the SAME BLAS drotg the
        # per-step loops of the host layer call (utils.givens_scalars -> scipy.linalg.blas.drotg): its address goes
        # across the ABI once.  Without it they use the reference BLAS formula (last-bit differences).
        fn = _blas_drotg_address()
        if fn:
            _check(self._lib, self._lib.kh_ctx_set_rotg(self._h, fn), "kh_ctx_set_rotg")

    def close(self):
        if self._alive:
            self._pool_flush()
            self._alive = False
            self._lib.kh_ctx_destroy(self._h)

    # ---- block pool: a restarted solver allocates and drops an (N, m+1) basis every cycle
    # (8 GB at N = 10^7, m = 100); hipMalloc/hipFree of that size costs tens of ms, so released
    # blocks are parked (at most _POOL_PER_SHAPE per shape, _POOL_FRACTION of device memory in
    # total) and handed out again zero-filled, which is what kh_vec_alloc guarantees. ----
    _POOL_PER_SHAPE = 2          # big blocks (a basis)
    _POOL_PER_SHAPE_MEDIUM = 6   # blocks below _POOL_MEDIUM_BYTES: a deflated solve holds U, AU and the projector's two bases of one
    _POOL_MEDIUM_BYTES = 8 << 30 # shape (N x 16: 1.6 GB at config 5's slab) - with two parked, the next solve allocated two afresh
    _POOL_PER_SHAPE_SMALL = 12   # blocks below _POOL_SMALL_BYTES (single vectors, W pairs, panels):
    _POOL_SMALL_BYTES = 1 << 30  # a cycle allocates and drops about ten of them
    _POOL_FRACTION = 0.6         # (0.35 until round 6: config 5 at N = 10^8 on one device - an 80.8 GB basis beside 34 GB of parked
                                 #  projector blocks - fell over the cap and paid 4 x 0.86 s of hipMalloc per solve, 3.4 of 8.4 s;
                                 #  an allocation that fails flushes the pool and is tried again, so a generous cap costs nothing)

    def _pool_take(self, n, ncols, zero=True):
        lst = self.__dict__.setdefault("_pool", {}).get((n, ncols))
        if not lst:
            return None
        h = lst.pop()
        self._pool_bytes -= 8 * n * max(ncols, 1)
        if zero:
            _check(self._lib, self._lib.kh_vec_zero(h, 0, ncols), "kh_vec_zero(pool)")
        return h

    def _pool_give(self, n, ncols, h):
        pool = self.__dict__.setdefault("_pool", {})
        self.__dict__.setdefault("_pool_bytes", 0)
        if "_pool_cap" not in self.__dict__:
            try:
                self._pool_cap = int(self.info()["mem_total"] * self._POOL_FRACTION)
            except Exception:
                self._pool_cap = 0
        nbytes = 8 * n * max(ncols, 1)
        lst = pool.setdefault((n, ncols), [])
        cap = (self._POOL_PER_SHAPE_SMALL if nbytes < self._POOL_SMALL_BYTES
               else self._POOL_PER_SHAPE_MEDIUM if nbytes < self._POOL_MEDIUM_BYTES else self._POOL_PER_SHAPE)
        if len(lst) < cap and self._pool_bytes + nbytes <= self._pool_cap:
            lst.append(h)
            self._pool_bytes += nbytes
        else:
            self._lib.kh_vec_free(h)

    def _with_memory(self, call):
        """Run an allocating C call; on failure release what the process is merely holding on to - blocks of
        solvers that are garbage but sit in reference cycles (a solver and the operators it built refer to each
        other), then the block pool - and try once more."""
        rc = call()
        if rc == -3:           # KH_ERR_NOMEM
            import gc
            gc.collect()
            self._pool_flush()
            rc = call()
        return rc

    def _pool_flush(self):
        pool = self.__dict__.get("_pool") or {}
        n = 0
        for lst in pool.values():
            while lst:
                self._lib.kh_vec_free(lst.pop())
                n += 1
        self._pool_bytes = 0
        return n

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- bookkeeping ----
    def sync(self):
        _check(self._lib, self._lib.kh_ctx_sync(self._h), "kh_ctx_sync")

    def info(self):
        buf = (ctypes.c_int64 * 4)()
        _check(self._lib, self._lib.kh_ctx_info(self._h, buf), "kh_ctx_info")
        return dict(compute_units=buf[0], mem_total=buf[1], mem_free=buf[2], reduce_blocks=buf[3])

    def counters(self):
        """Launch counts of the Gram-Schmidt kernel families (``kh_ctx_counters``)."""
        buf = (ctypes.c_int64 * 4)()
        _check(self._lib, self._lib.kh_ctx_counters(self._h, buf), "kh_ctx_counters")
        return dict(chain=buf[0], chain_lds=buf[1], chain_fused=buf[2], cgs_register=buf[3])

    def tune(self, reduce_blocks=0, spmv_tile=0):
        _check(self._lib, self._lib.kh_ctx_tune(self._h, reduce_blocks, spmv_tile), "kh_ctx_tune")

    def set(self, key, value):
        """Named switch of the context (``kh_ctx_set``): spmv_dia, chain, chain_lds, chain_spmv."""
        _check(self._lib, self._lib.kh_ctx_set(self._h, key.encode(), int(value)), "kh_ctx_set(%s)" % key)

    def get(self, key):
        """A switch or a counter of the context (``kh_ctx_get``), e.g. ``n_spmm``, ``n_chain_recovered``."""
        v = _I64(0)
        _check(self._lib, self._lib.kh_ctx_get(self._h, key.encode(), ctypes.byref(v)), "kh_ctx_get(%s)" % key)
        return v.value

    def timer_start(self):
        _check(self._lib, self._lib.kh_timer_start(self._h), "kh_timer_start")

    def timer_stop(self):
        ms = _D(0.0)
        _check(self._lib, self._lib.kh_timer_stop(self._h, ctypes.byref(ms)), "kh_timer_stop")
        return ms.value

    # ---- multi-GPU ----
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        _check(self._lib, self._lib.kh_comm_unique_id(buf), "kh_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        _check(self._lib, self._lib.kh_comm_init(self._h, rank, nranks, unique_id), "kh_comm_init")
        self.rank, self.nranks = rank, nranks

    # xr: cross-rank sums through IPC-mapped mailboxes (csrc/xr.hip); see krypy_amd.dist.enable_xr for the whole protocol
    def xr_export(self):
        """This rank's mailbox as a 64-byte ``hipIpcMemHandle_t`` (``kh_xr_export``)."""
        buf = ctypes.create_string_buffer(64)
        _check(self._lib, self._lib.kh_xr_export(self._h, buf), "kh_xr_export")
        return buf.raw

    def xr_attach(self, rank, nranks, handles):
        """Map every peer's mailbox: ``handles`` = the ``nranks`` handles in rank order, 64 bytes each."""
        handles = bytes(handles)
        if len(handles) != 64 * nranks:
            raise BackendError("xr_attach: %d bytes of handles for %d ranks" % (len(handles), nranks))
        _check(self._lib, self._lib.kh_xr_attach(self._h, rank, nranks, handles), "kh_xr_attach")

    def xr_enable(self, rank, nranks):
        """All-reduces take the mailboxes from now on (every rank must do the same: ``dist.enable_xr``)."""
        self.set("xr", 1)
        self.rank, self.nranks = rank, nranks

    def xr_detach(self):
        self._lib.kh_xr_detach(self._h)

    # xh: the halo of a shard through IPC-mapped granules inside the banded SpMV's launch (krypy_amd.dist.ShardedCSROperator)
    def xh_export(self, A):
        buf = ctypes.create_string_buffer(64)
        _check(self._lib, self._lib.kh_mat_xh_export(self._h, A.handle, buf), "kh_mat_xh_export")
        return buf.raw

    def xh_attach(self, A, prev, prev_ng, prev_off, next_, next_ng, self_loop=False):
        _check(self._lib, self._lib.kh_mat_xh_attach(self._h, A.handle, prev, int(prev_ng), int(prev_off), next_, int(next_ng),
                                                      1 if self_loop else 0), "kh_mat_xh_attach")

    def xh_detach(self, A):
        _check(self._lib, self._lib.kh_mat_xh_detach(self._h, A.handle), "kh_mat_xh_detach")

    def xh_enable(self, A, on=True):
        _check(self._lib, self._lib.kh_mat_xh_enable(self._h, A.handle, 1 if on else 0), "kh_mat_xh_enable")

    def allreduce_host(self, vals):
        a = numpy.ascontiguousarray(vals, dtype=numpy.float64)
        _check(self._lib, self._lib.kh_comm_allreduce_host(self._h, _dptr(a), a.size),
               "kh_comm_allreduce_host")
        return a

    def set_ghost(self, A, values):
        """Diagnostic: the ghost entries a halo exchange would deliver (``kh_mat_set_ghost``)."""
        if A.dtype == _C128:          # (re, im) pairs, like the halo exchange of a complex shard delivers them
            v = numpy.ascontiguousarray(values, dtype=numpy.complex128)
            count = 2 * v.size
        else:
            if numpy.iscomplexobj(values) and numpy.any(numpy.imag(values)):
                raise BackendError("set_ghost: complex ghost entries for a real operator")
            v = numpy.ascontiguousarray(numpy.real(values), dtype=numpy.float64)
            count = v.size
        _check(self._lib, self._lib.kh_mat_set_ghost(A.handle, _dptr(v), count), "kh_mat_set_ghost")

    def get_ghost(self, A, count):
        """Diagnostic: the ``count`` ghost entries the last halo exchange delivered (``kh_mat_get_ghost``)."""
        cplx = A.dtype == _C128
        out = numpy.empty(2 * count if cplx else count, dtype=numpy.float64)
        _check(self._lib, self._lib.kh_mat_get_ghost(A.handle, _dptr(out), out.size), "kh_mat_get_ghost")
        return out.view(numpy.complex128) if cplx else out

    def set_rows_max(self, A, rows_max):
        """The longest slab of the run a sharded operator belongs to (``kh_mat_set_rows_max``: every rank the same number)."""
        _check(self._lib, self._lib.kh_mat_set_rows_max(A.handle, int(rows_max)), "kh_mat_set_rows_max")

    def set_halo(self, A, nsend_prev, nsend_next, nrecv_prev, nrecv_next):
        _check(self._lib, self._lib.kh_mat_set_halo(self._h, A.handle, nsend_prev, nsend_next,
                                                    nrecv_prev, nrecv_next), "kh_mat_set_halo")

    # ---- allocation / transfer ----
    def alloc(self, n, ncols=1, dtype=_F64, zero=True):
        return DeviceVectors(self, n, ncols, dtype, zero)

    def upload(self, arr, dtype=None):
        """Host array -> device block; ``dtype`` forces a complex block for real data."""
        a = numpy.asarray(arr)
        dt = _block_dtype(a.dtype if dtype is None else numpy.result_type(a.dtype, dtype))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return DeviceVectors(self, a.shape[0], a.shape[1], dt).upload(0, a)

    def promote(self, X, xcol, Z, zcol, ncols=1):
        """Z[:, zcol:zcol+ncols] (complex) = X[:, xcol:xcol+ncols] (real) + 0i, on the device."""
        if X.dtype != _F64 or Z.dtype != _C128:
            raise BackendError("promote: real source and complex destination expected")
        _check(self._lib, self._lib.kh_zfrom_real(self._h, X.handle, xcol, Z.handle, zcol, ncols),
               "kh_zfrom_real")

    def csr(self, A, n_cols=None, dtype=None):
        """Upload a SciPy CSR matrix (int32 indices, fp64 / c128 data) without reordering rows."""
        indptr = numpy.ascontiguousarray(A.indptr, dtype=numpy.int32)
        indices = numpy.ascontiguousarray(A.indices, dtype=numpy.int32)
        dt = _block_dtype(A.dtype if dtype is None else numpy.result_type(A.dtype, dtype))
        data = numpy.ascontiguousarray(A.data, dtype=dt)
        n_rows = A.shape[0]
        n_cols = A.shape[1] if n_cols is None else n_cols
        h = _H()
        fn = self._lib.kh_zcsr_upload if dt == _C128 else self._lib.kh_csr_upload
        _check(self._lib, self._with_memory(lambda: fn(
            self._h, n_rows, n_cols, data.size, indptr.ctypes.data_as(_c_int32_p),
            indices.ctypes.data_as(_c_int32_p), _dptr(data), ctypes.byref(h))), "kh_csr_upload")
        return DeviceMatrix(self, h, "csr", (n_rows, n_cols), data.size, dt)

    def dense(self, A, dtype=None):
        A = numpy.asarray(A)
        dt = _block_dtype(A.dtype if dtype is None else numpy.result_type(A.dtype, dtype))
        a = numpy.ascontiguousarray(A, dtype=dt)
        h = _H()
        fn = self._lib.kh_zdense_upload if dt == _C128 else self._lib.kh_dense_upload
        _check(self._lib, self._with_memory(lambda: fn(self._h, a.shape[0], a.shape[1], _dptr(a), a.shape[1],
                                                       ctypes.byref(h))), "kh_dense_upload")
        return DeviceMatrix(self, h, "dense", a.shape, a.size, dt)

    def dense_from_block(self, X, col0, nrows, alpha=1.0, beta=0.0):
        """Dense operator alpha X[:, col0 : col0 + nrows]^T + beta I from a real block that is already on the device."""
        if X.dtype != _F64:
            raise BackendError("dense_from_block: a real block expected")
        h = _H()
        _check(self._lib, self._with_memory(lambda: self._lib.kh_dense_from_block(
            self._h, X.handle, col0, nrows, float(alpha), float(beta), ctypes.byref(h))), "kh_dense_from_block")
        return DeviceMatrix(self, h, "dense", (nrows, X.n), nrows * X.n, _F64)

    def diag(self, d, dtype=None):
        d = numpy.asarray(d)
        dt = _block_dtype(d.dtype if dtype is None else numpy.result_type(d.dtype, dtype))
        d = numpy.ascontiguousarray(d, dtype=dt)
        h = _H()
        fn = self._lib.kh_zdiag_upload if dt == _C128 else self._lib.kh_diag_upload
        _check(self._lib, fn(self._h, d.size, _dptr(d), ctypes.byref(h)), "kh_diag_upload")
        return DeviceMatrix(self, h, "diag", (d.size, d.size), d.size, dt)

    # ---- numerics (each is one C entry point; complex blocks go to the kh_z* twin) ----
    def apply(self, A, X, xcol, Y, ycol, ncols=1):
        if (A.dtype == _C128) != _same_dtype("apply", X, Y):
            raise BackendError("apply: %s operator on %s blocks" % (A.dtype, X.dtype))
        _check(self._lib, self._lib.kh_apply(self._h, A.handle, X.handle, xcol, Y.handle, ycol,
                                             ncols), "kh_apply")

    def dot_panel(self, V, j0, ncols, W, wcol):
        if _same_dtype("dot_panel", V, W):
            out = numpy.empty(max(ncols, 1), dtype=numpy.complex128)
            _check(self._lib, self._lib.kh_zdot_panel(self._h, V.handle, j0, ncols, W.handle, wcol,
                                                      _dptr(out)), "kh_zdot_panel")
            return out[:ncols]
        out = numpy.empty(max(ncols, 1), dtype=numpy.float64)
        for c0 in range(0, max(ncols, 1), 1024):      # the C entry point takes at most 1024 columns per call
            m = min(1024, ncols - c0)
            _check(self._lib, self._lib.kh_dot_panel(self._h, V.handle, j0 + c0, m, W.handle, wcol,
                                                     _dptr(out[c0:])), "kh_dot_panel")
        return out[:ncols]

    def gemm_tn(self, X, x0, nx, Y, y0, ny):
        if _same_dtype("gemm_tn", X, Y):
            out = numpy.empty((nx, ny), dtype=numpy.complex128)
            for c in range(ny):
                done = 0
                while done < nx:          # kh_zdot_panel takes at most 512 columns
                    m = min(512, nx - done)
                    out[done:done + m, c] = self.dot_panel(X, x0 + done, m, Y, y0 + c)
                    done += m
            return out
        out = numpy.empty((nx, ny), dtype=numpy.float64)
        if nx and ny:
            for r0 in range(0, nx, 1024):             # at most 1024 rows per call
                m = min(1024, nx - r0)
                blk = numpy.empty((m, ny), dtype=numpy.float64)
                _check(self._lib, self._lib.kh_gemm_tn(self._h, X.handle, x0 + r0, m, Y.handle, y0, ny,
                                                       _dptr(blk)), "kh_gemm_tn")
                out[r0:r0 + m] = blk
        return out

    def axpy_panel(self, V, j0, ncols, h, W, wcol):
        if _same_dtype("axpy_panel", V, W):
            h, hp = _zarr(numpy.asarray(h).reshape(-1))
            _check(self._lib, self._lib.kh_zaxpy_panel(self._h, V.handle, j0, ncols, hp, W.handle,
                                                       wcol), "kh_zaxpy_panel")
            return
        h = _real_coeffs(h, "axpy_panel").reshape(-1)
        for c0 in range(0, max(ncols, 1), 1024):      # at most 1024 columns per call, left to right
            m = min(1024, ncols - c0)
            _check(self._lib, self._lib.kh_axpy_panel(self._h, V.handle, j0 + c0, m, _dptr(h[c0:]), W.handle,
                                                      wcol), "kh_axpy_panel")

    def gemm_nn(self, X, x0, k, C, alpha, beta, Y, y0):
        if _same_dtype("gemm_nn", X, Y):
            C = numpy.asarray(C)
            if C.ndim == 1:
                C = C.reshape(-1, 1)
            if numpy.imag(beta) != 0:
                raise BackendError("gemm_nn: beta must be real")
            C = numpy.asarray(C * alpha, dtype=numpy.complex128)
            done, b = 0, float(numpy.real(beta))
            while True:                       # kh_zgemm_nn takes at most 512 columns of X per call
                m = min(512, k - done)
                Cc, cp = _zarr(C[done:done + m])
                _check(self._lib, self._lib.kh_zgemm_nn(self._h, X.handle, x0 + done, m, cp, C.shape[1],
                                                        b, Y.handle, y0), "kh_zgemm_nn")
                done, b = done + m, 1.0
                if done >= k:
                    break
            return
        C = _real_coeffs(C, "gemm_nn")
        if C.ndim == 1:
            C = C.reshape(-1, 1)
        _check(self._lib, self._lib.kh_gemm_nn(self._h, X.handle, x0, k, _dptr(C), C.shape[1],
                                               _real_scalar(alpha, "gemm_nn"),
                                               _real_scalar(beta, "gemm_nn"), Y.handle, y0),
               "kh_gemm_nn")

    def nrm2(self, W, wcol):
        # complex: the 2-norm of the (re, im) view is the complex 2-norm
        out = _D(0.0)
        _check(self._lib, self._lib.kh_nrm2(self._h, W.handle, wcol, ctypes.byref(out)), "kh_nrm2")
        return out.value

    def waxpby(self, Z, zcol, alpha, X, xcol, beta, Y, ycol):
        if _same_dtype("waxpby", Z, X, Y) and (numpy.imag(alpha) != 0 or numpy.imag(beta) != 0):
            a, ap = _zarr([alpha])
            b, bp = _zarr([beta])
            _check(self._lib, self._lib.kh_zwaxpby(self._h, Z.handle, zcol, ap, X.handle, xcol, bp,
                                                   Y.handle, ycol), "kh_zwaxpby")
            return
        _check(self._lib, self._lib.kh_waxpby(self._h, Z.handle, zcol, _real_scalar(alpha, "waxpby"),
                                              X.handle, xcol, _real_scalar(beta, "waxpby"),
                                              Y.handle, ycol), "kh_waxpby")

    def vdiv(self, Z, zcol, X, xcol, s):
        if _same_dtype("vdiv", Z, X) and numpy.imag(s) != 0:
            return self.waxpby(Z, zcol, 1.0 / complex(s), X, xcol, 0.0, X, xcol)
        _check(self._lib, self._lib.kh_vdiv(self._h, Z.handle, zcol, X.handle, xcol,
                                            _real_scalar(s, "vdiv")), "kh_vdiv")

    def arnoldi_step(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1=0.0):
        if _same_dtype("arnoldi_step", V, W):
            if Md is not None or P is not None:       # (complex diagonal Md with its block P: the general entry)
                self.arnoldi_step_begin(A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1, 0)
                return self.arnoldi_step_end(0, k + 2, cplx=True)
            out = numpy.empty(k + 2, dtype=numpy.complex128)
            hk, hkp = _zarr([h_km1])
            _check(self._lib, self._lib.kh_zarnoldi_step(
                self._h, A.handle if A is not None else None, V.handle, W.handle, wcol, k, start,
                sweeps, gs_mode, hkp, _dptr(out)), "kh_zarnoldi_step")
            return out
        out = numpy.empty(k + 2, dtype=numpy.float64)
        _check(self._lib, self._lib.kh_arnoldi_step(
            self._h, A.handle if A is not None else None, Md.handle if Md is not None else None,
            V.handle, P.handle if P is not None else None, W.handle, wcol, k, start, sweeps,
            gs_mode, h_km1, _dptr(out)), "kh_arnoldi_step")
        return out

    def arnoldi_step_begin(self, A, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1, slot,
                           proj=None):
        if _same_dtype("arnoldi_step_begin", V, W):
            if (Md is None) != (P is None) or (Md is not None and Md.dtype != _C128):
                raise BackendError("arnoldi_step_begin: the complex step takes a complex operator Md with its block P")
            hk = numpy.array([numpy.real(h_km1), numpy.imag(h_km1)], dtype=numpy.float64)
            _check(self._lib, self._lib.kh_zarnoldi_step_begin_md(
                self._h, A.handle if A is not None else None, proj.handle if proj is not None else None,
                Md.handle if Md is not None else None, V.handle, P.handle if P is not None else None,
                W.handle, wcol, k, start, sweeps, gs_mode, _dptr(hk), slot), "kh_zarnoldi_step_begin")
            return
        _check(self._lib, self._lib.kh_arnoldi_step_begin(
            self._h, A.handle if A is not None else None,
            proj.handle if proj is not None else None, Md.handle if Md is not None else None,
            V.handle, P.handle if P is not None else None, W.handle, wcol, k, start, sweeps,
            gs_mode, h_km1, slot), "kh_arnoldi_step_begin")

    def proj_create(self, W, V, d, T, WRH, iterations):
        """Device image of a Projection (``kh_proj``): T = R^{-1} Q^H, WRH = WR^H (d x d) or None."""
        cplx = _same_dtype("proj_create", W, V)

        def mat(M):
            if M is None:
                return None, None
            M = numpy.ascontiguousarray(M, dtype=numpy.complex128 if cplx else numpy.float64)
            return M, M.ctypes.data_as(_c_double_p)
        Tm, Tp = mat(T)
        Wm, Wp = mat(WRH)
        h = _H()
        fn = self._lib.kh_zproj_create if cplx else self._lib.kh_proj_create
        _check(self._lib, fn(self._h, W.handle, V.handle, d, Tp, Wp, iterations, ctypes.byref(h)), "kh_proj_create")
        p = DeviceProjector(self, h, d, (W, V))
        p.cplx = cplx
        return p

    def proj_apply_complement(self, proj, A, acol, Z, zcol, want_ya=False):
        cplx = getattr(proj, "cplx", False)
        ya = numpy.empty(proj.d, dtype=numpy.complex128 if cplx else numpy.float64) if want_ya else None
        fn = self._lib.kh_zproj_apply_complement if cplx else self._lib.kh_proj_apply_complement
        _check(self._lib, fn(self._h, proj.handle, A.handle, acol, Z.handle, zcol,
                             ya.ctypes.data_as(_c_double_p) if want_ya else None), "kh_proj_apply_complement")
        return ya

    def arnoldi_step_end(self, slot, count, cplx=False):
        """The H column of the step begun in ``slot``: ``count`` numbers (complex ones with ``cplx``)."""
        out = numpy.empty(count, dtype=numpy.complex128 if cplx else numpy.float64)
        _check(self._lib, self._lib.kh_arnoldi_step_end(self._h, slot, count * (2 if cplx else 1),
                                                        _dptr(out)), "kh_arnoldi_step_end")
        return out

    def residual(self, A, B, bcol, X, xcol, R, rcol):
        if _same_dtype("residual", B, X, R):     # complex: operator, real-linear axpby, norm
            self.apply(A, X, xcol, R, rcol, 1)
            self.waxpby(R, rcol, 1.0, B, bcol, -1.0, R, rcol)
            return self.nrm2(R, rcol)
        out = _D(0.0)
        _check(self._lib, self._lib.kh_residual(self._h, A.handle, B.handle, bcol, X.handle, xcol,
                                                R.handle, rcol, ctypes.byref(out)), "kh_residual")
        return out.value

    def minres_update(self, V, k, Wk, slot, r0, r1, r2, y0, YK, ycol, defer=False):
        """``defer``: the (real) update may wait for the next Lanczos launch to carry it
        (``kh_minres_update_deferred``); :meth:`minres_flush` before ``yk`` is read."""
        if defer and V.dtype == _F64:
            _same_dtype("minres_update", V, Wk, YK)
            _check(self._lib, self._lib.kh_minres_update_deferred(self._h, V.handle, k, Wk.handle, slot, r0, r1,
                                                                  r2, y0, YK.handle, ycol), "kh_minres_update_deferred")
            return
        if _same_dtype("minres_update", V, Wk, YK):
            # z = (V_k - r0 W0 - r1 W1)/r2 written over W0 (= column `slot`);  yk += y0 z: one pass, complex scalars
            c = [_zarr([x]) for x in (r0, r1, r2, y0)]
            _check(self._lib, self._lib.kh_zminres_update(self._h, V.handle, k, Wk.handle, slot, c[0][1], c[1][1],
                                                          c[2][1], c[3][1], YK.handle, ycol), "kh_zminres_update")
            return
        _check(self._lib, self._lib.kh_minres_update(self._h, V.handle, k, Wk.handle, slot, r0, r1,
                                                     r2, y0, YK.handle, ycol), "kh_minres_update")

    def gmres_cycle(self, A, Md, V, P, W, k0, k_stop, k_last, sweeps, gs_mode, enq, tol, bnorm, H, R, cs, y, h2, resn):
        """Steps ``k0 .. k_stop-1`` of a GMRES cycle in one call (``kh_gmres_cycle``): Arnoldi with look-ahead on the
        device, Givens QR / residual recurrence on the host in C.  ``H``, ``R``: C-ordered float64 2-D arrays of the
        solver, ``cs`` (2 per step), ``y``, ``resn``: float64 1-D arrays, all updated in place.  Returns
        ``(k_done, enq, h2, reason)``."""
        for a in (H, R, cs, y, resn):
            if a.dtype != numpy.float64 or not a.flags.c_contiguous:
                raise BackendError("gmres_cycle: C-ordered float64 arrays needed")
        if H.ndim != 2 or R.ndim != 2 or H.shape[0] < k_stop + 1 or R.shape[0] < k_stop + 1:
            raise BackendError("gmres_cycle: H and R need k_stop + 1 = %d rows" % (k_stop + 1))
        enq_ = ctypes.c_int64(int(enq))
        h2_ = _D(float(h2))
        kd = ctypes.c_int64(0)
        why = ctypes.c_int(0)
        _check(self._lib, self._lib.kh_gmres_cycle(
            self._h, A.handle, Md.handle if Md is not None else None, V.handle, P.handle if P is not None else None,
            W.handle, k0, k_stop, k_last, sweeps, gs_mode, ctypes.byref(enq_), float(tol), float(bnorm), _dptr(H), H.shape[1],
            _dptr(R), R.shape[1], _dptr(cs), _dptr(y), ctypes.byref(h2_), _dptr(resn), ctypes.byref(kd),
            ctypes.byref(why)), "kh_gmres_cycle")
        return kd.value, enq_.value, h2_.value, why.value

    def minres_cycle(self, A, Md, V, P, W, k0, k_stop, k_last, base, enq, tol, bnorm, H, Wm, wslot, YK, ycol, st, h2,
                     resn):
        """Iterations ``k0 .. k_stop-1`` of MINRES in one call (``kh_minres_cycle``): Lanczos steps with look-ahead on
        the device, the QR update with the two remembered rotations and the deferred vector recurrences in C.  ``H``:
        the C-ordered float64 Lanczos matrix of the solver, ``st`` (7 doubles: rotations, their count, rotated
        right-hand side) and ``resn``: float64 1-D arrays, updated in place.  Returns
        ``(k_done, enq, h2, wslot, reason)``."""
        for a in (H, st, resn):
            if a.dtype != numpy.float64 or not a.flags.c_contiguous:
                raise BackendError("minres_cycle: C-ordered float64 arrays needed")
        if H.ndim != 2 or H.shape[0] < k_stop + 1 or st.size < 7 or resn.size < k_stop:
            raise BackendError("minres_cycle: H needs k_stop + 1 = %d rows, st 7 entries, resn k_stop" % (k_stop + 1))
        _same_dtype("minres_cycle", V, W, Wm, YK)
        enq_ = ctypes.c_int64(int(enq))
        h2_ = _D(float(h2))
        kd = ctypes.c_int64(0)
        why = ctypes.c_int(0)
        ws = ctypes.c_int(int(wslot))
        _check(self._lib, self._lib.kh_minres_cycle(
            self._h, A.handle, Md.handle if Md is not None else None, V.handle, P.handle if P is not None else None,
            W.handle, k0, k_stop, k_last, base, ctypes.byref(enq_), float(tol), float(bnorm), _dptr(H), H.shape[1],
            Wm.handle, ctypes.byref(ws), YK.handle, ycol, _dptr(st), ctypes.byref(h2_), _dptr(resn), ctypes.byref(kd),
            ctypes.byref(why)), "kh_minres_cycle")
        return kd.value, enq_.value, h2_.value, ws.value, why.value

    def minres_flush(self):
        """Run a deferred MINRES update now (``kh_minres_flush``)."""
        _check(self._lib, self._lib.kh_minres_flush(self._h), "kh_minres_flush")

    def cg_update(self, alpha, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Md, Z, zcol):
        """``yk += alpha p; r -= alpha Ap; z = Md r; return <r, z>`` in one pass.  The recurrences have real
        coefficients, so complex blocks run as their real views of length 2N; ``Md`` is then a REAL diagonal of
        length 2N (every Jacobi entry twice) and the result the real part of ``<r, z>``."""
        if _same_dtype("cg_update", Pd, AP, YK, R) and Md is not None and (Md.dtype != _F64 or Md.shape[0] != 2 * R.n):
            raise BackendError("cg_update on complex blocks: Md must be the real diagonal of length 2N")
        out = _D(0.0)
        _check(self._lib, self._lib.kh_cg_update(
            self._h, alpha, Pd.handle, pcol, AP.handle, apcol, YK.handle, ycol, R.handle, rcol,
            Md.handle if Md is not None else None, Z.handle if Z is not None else None, zcol,
            ctypes.byref(out)), "kh_cg_update")
        return out.value


    def cg_cycle(self, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, k0, k_stop, tol, bnorm, rhos, trace):
        """Iterations ``k0 .. k_stop-1`` of CG in one call (``kh_cg_cycle``; real data): ``rhos`` (float64, at least
        ``k_stop + 1`` entries, ``rhos[i]`` = rho after i iterations) and ``trace`` (six doubles per iteration) are
        updated in place.  Returns ``(k_done, reason)``."""
        for a in (rhos, trace):
            if a.dtype != numpy.float64 or not a.flags.c_contiguous:
                raise BackendError("cg_cycle: C-ordered float64 arrays needed")
        if rhos.size < k_stop + 1 or trace.size < 6 * k_stop:
            raise BackendError("cg_cycle: rhos needs k_stop + 1 entries, trace 6 k_stop")
        if _same_dtype("cg_cycle", Pd, AP, YK, R):
            raise BackendError("cg_cycle: real blocks expected")
        kd = ctypes.c_int64(0)
        why = ctypes.c_int(0)
        _check(self._lib, self._lib.kh_cg_cycle(
            self._h, A.handle, Md.handle if Md is not None else None, Pd.handle, pcol, AP.handle, apcol, YK.handle, ycol,
            R.handle, rcol, Z.handle if Z is not None else None, zcol, k0, k_stop, float(tol), float(bnorm), _dptr(rhos),
            _dptr(trace), ctypes.byref(kd), ctypes.byref(why)), "kh_cg_cycle")
        return kd.value, why.value

    def cg_step(self, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, first, omega, rho):
        """One CG iteration in one call; returns ``(d, rho_new, <p, Ap>, flags)`` (``flags``: the ``KH_CG_*`` sanity
        word of the header, 0 = fine) where ``alpha = rho / d`` is the step
        the device took: ``d = <p, Ap>`` for real data, ``rho / d = Re(rho / <p, Ap>)`` for complex data (then ``A``
        is a complex operator and ``Md`` a REAL diagonal of length 2N, every Jacobi entry twice)."""
        if _same_dtype("cg_step", Pd, AP, YK, R):
            if A.dtype != _C128 or (Md is not None and (Md.dtype != _F64 or Md.shape[0] != 2 * R.n)):
                raise BackendError("cg_step on complex blocks: complex operator and real diagonal of length 2N needed")
            out = numpy.empty(5, dtype=numpy.float64)
            _check(self._lib, self._lib.kh_zcg_step(
                self._h, A.handle, Md.handle if Md is not None else None, Pd.handle, pcol, AP.handle,
                apcol, YK.handle, ycol, R.handle, rcol, Z.handle if Z is not None else None, zcol,
                1 if first else 0, omega, rho, _dptr(out)), "kh_zcg_step")
            return float(out[0]), float(out[1]), complex(out[2], out[3]), int(out[4])
        out = numpy.empty(3, dtype=numpy.float64)
        _check(self._lib, self._lib.kh_cg_step(
            self._h, A.handle, Md.handle if Md is not None else None, Pd.handle, pcol, AP.handle,
            apcol, YK.handle, ycol, R.handle, rcol, Z.handle if Z is not None else None, zcol,
            1 if first else 0, omega, rho, _dptr(out)), "kh_cg_step")
        return float(out[0]), float(out[1]), float(out[0]), int(out[2])

    def bench_kernel(self, which, V, W, reps):
        ms = _D(0.0)
        _check(self._lib, self._lib.kh_bench_kernel(self._h, which, V.handle, W.handle, reps,
                                                    ctypes.byref(ms)), "kh_bench_kernel")
        return ms.value

    def bench_arnoldi(self, A, V, W, m, gs_mode, reps):
        """Average duration (ms) of one Arnoldi step over ``reps`` sequences k = 0 .. m-1 (``kh_bench_arnoldi``)."""
        ms = _D(0.0)
        _check(self._lib, self._lib.kh_bench_arnoldi(self._h, A.handle, V.handle, W.handle, int(m), int(gs_mode),
                                                     int(reps), ctypes.byref(ms)), "kh_bench_arnoldi")
        return ms.value


_default_ctx = None


def device_count():
    """GPUs visible to this process (``kh_device_count``); 0 when HIP reports none.  Raises :class:`BackendError` when the
    library itself is missing - a launcher must know how many ranks it can start before it starts them."""
    lib = load_library()
    n = _INT(0)
    rc = lib.kh_device_count(ctypes.byref(n))
    return int(n.value) if rc == 0 else 0


def get_context():
    """The process-wide device context (device = ``LOCAL_RANK`` or ``KRYPY_AMD_DEVICE`` or 0).

    Raises :class:`BackendError` when the HIP library or the GPU is missing.
    """
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("KRYPY_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(dev)
    return _default_ctx


def _install_context_for_testing(ctx):
    """Swap the process-wide context.  Used ONLY by the CPU test-suite to drive the host
    layer with the NumPy test double in ``tests/support`` (there is no GPU in the build
    container); the product never calls this and ships no alternative context."""
    global _default_ctx
    old, _default_ctx = _default_ctx, ctx
    return old
