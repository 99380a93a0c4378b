"""Block-row sharding of the Krylov path over the GPUs of one node (one process per GPU).

The reference is single-process; this module is the new component SURVEY.md 8(e) describes:
matrix rows and every N-vector are split into contiguous slabs, one per rank; ``H``, the Givens
rotations and all small factors stay replicated on every rank's host.  Two exchanges exist and
both run inside ``libkrylov_hip.so`` over RCCL/xGMI:

* all-reduce(sum, fp64) of dot-product panels (inside ``kh_dot_panel``, ``kh_nrm2``,
  ``kh_arnoldi_step``, ``kh_residual``, ``kh_cg_update``; complex data: ``kh_zdot_panel`` and the complex step) -
  the host layer is unchanged, it just
  sees the local slab length as ``N``;
* a nearest-neighbour halo exchange before each sharded SpMV (``kh_mat_set_halo``): for a banded
  matrix (stencils) a rank needs the last ``h`` entries of the previous slab and the first ``h``
  of the next one.

Usage (every rank)::

    ctx = krypy_amd._hip.get_context(); ctx.comm_init(rank, nranks, unique_id)
    op = ShardedCSROperator(A_rows, row0, n_global, ctx)   # A_rows: CSR of rows [row0, row1),
    ls = LinearSystem(op, b_local)                         #         global column indices
    RestartedGmres(ls, maxiter=100, max_restarts=R, ortho="cgs2")
"""
import os
import socket
import struct
import time

import numpy
import scipy.sparse

from . import _hip, utils

__all__ = ["slab_cuts", "localize_columns", "ShardedCSROperator", "TcpRendezvous", "enable_xr"]


class TcpRendezvous(object):
    """What a launcher of one process per GPU needs besides RCCL itself: hand rank 0's ``ncclUniqueId`` to the other
    ranks, a barrier, and the maximum of one double over the ranks (the timing contract of ``bench.py``).  Plain TCP
    sockets in a star around rank 0 - no PyTorch, no MPI on the host side (north_star).  The environment is the one
    ``python -m torch.distributed.run`` (or any other launcher) provides: ``MASTER_ADDR`` and ``MASTER_PORT``; the
    port itself belongs to the launcher's own store, so rank 0 listens on the first free one of the eight ports
    behind it and the others find it there (a handshake word tells a foreign listener from rank 0).  Everything that
    moves data between GPUs - halo exchange, all-reduces - runs inside ``libkrylov_hip.so`` over RCCL."""

    MAGIC = b"krypy_amd.rdv.1\0"

    def __init__(self, rank, world, addr=None, port=None, timeout=600.0):
        self.rank, self.world = int(rank), int(world)
        self._peers = []          # rank 0: sockets of ranks 1 .. world-1 (by rank); others: [socket to rank 0]
        if self.world <= 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else os.environ.get("MASTER_PORT", "29511"))
        ports = [base + 1 + i for i in range(8)]
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("TcpRendezvous: no free port in %s on %s" % (ports, addr))
            srv.listen(self.world)
            got = {}
            while len(got) < self.world - 1:
                srv.settimeout(max(1.0, deadline - time.time()))
                c, _ = srv.accept()
                c.settimeout(30.0)
                try:
                    hello = self._recv_exact(c, len(self.MAGIC) + 8)
                except (OSError, RuntimeError):
                    c.close()
                    continue
                r, w = struct.unpack("<ii", hello[len(self.MAGIC):])
                if hello[: len(self.MAGIC)] != self.MAGIC or w != self.world or not (0 < r < self.world) or r in got:
                    c.close()
                    continue
                c.sendall(self.MAGIC)
                c.settimeout(None)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                got[r] = c
            srv.close()
            self._peers = [got[r] for r in range(1, self.world)]
        else:
            sock = None
            while sock is None:
                for p in ports:
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                        c.sendall(self.MAGIC + struct.pack("<ii", self.rank, self.world))
                        if self._recv_exact(c, len(self.MAGIC)) == self.MAGIC:
                            c.settimeout(None)
                            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            sock = c
                            break
                        c.close()
                    except (OSError, RuntimeError):
                        pass
                if sock is None:
                    if time.time() > deadline:
                        raise RuntimeError("TcpRendezvous: rank %d found no rank 0 at %s:%s" % (self.rank, addr, ports))
                    time.sleep(0.05)
            self._peers = [sock]

    @staticmethod
    def _recv_exact(sock, n):
        buf = b""
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("TcpRendezvous: peer closed the connection")
            buf += chunk
        return buf

    def _send(self, sock, payload):
        sock.sendall(struct.pack("<I", len(payload)) + payload)

    def _recv(self, sock):
        (n,) = struct.unpack("<I", self._recv_exact(sock, 4))
        return self._recv_exact(sock, n)

    def broadcast_bytes(self, data):
        """``data`` of rank 0 on every rank."""
        if self.world <= 1:
            return data
        if self.rank == 0:
            for s in self._peers:
                self._send(s, bytes(data))
            return data
        return self._recv(self._peers[0])

    def _reduce(self, x, fn):
        if self.world <= 1:
            return x
        if self.rank == 0:
            vals = [x] + [struct.unpack("<d", self._recv(s))[0] for s in self._peers]
            out = fn(vals)
            for s in self._peers:
                self._send(s, struct.pack("<d", out))
            return out
        self._send(self._peers[0], struct.pack("<d", float(x)))
        return struct.unpack("<d", self._recv(self._peers[0]))[0]

    def allreduce_max(self, x):
        """max over the ranks of one double (every rank gets it)."""
        return self._reduce(float(x), max)

    def allreduce_min(self, x):
        """min over the ranks of one double (every rank gets it): "did EVERY rank succeed?"."""
        return self._reduce(float(x), min)

    def allgather_bytes(self, data):
        """Every rank's ``data`` (equal lengths), in rank order, on every rank."""
        data = bytes(data)
        if self.world <= 1:
            return [data]
        if self.rank == 0:
            parts = [data] + [self._recv(s) for s in self._peers]
            blob = b"".join(parts)
            for s in self._peers:
                self._send(s, blob)
        else:
            self._send(self._peers[0], data)
            blob = self._recv(self._peers[0])
        n = len(data)
        if len(blob) != n * self.world:
            raise RuntimeError("TcpRendezvous.allgather_bytes: contributions of unequal length")
        return [blob[i * n:(i + 1) * n] for i in range(self.world)]

    def barrier(self):
        self._reduce(0.0, max)

    def close(self):
        for s in self._peers:
            try:
                s.close()
            except OSError:
                pass
        self._peers = []

    destroy_process_group = close        # (the name the launcher-side code knew the torch.distributed module by)


def enable_xr(ctx, rdv, rank=None, world=None):
    """Switch the cross-rank sums of ``ctx`` to the IPC mailboxes of ``csrc/xr.hip`` - on EVERY rank or on none.

    Every rank exports its mailbox (``kh_xr_export``), the handles are gathered over the launcher's rendezvous ``rdv``
    (``allgather_bytes`` / ``allreduce_min``: :class:`TcpRendezvous` or anything with those two methods), every rank
    maps its peers' mailboxes (``kh_xr_attach``), and only when ALL ranks report success is the transport switched on
    (``kh_ctx_set "xr"``): whether a sum runs as a mailbox kernel or as ``ncclAllReduce`` must come out the same on
    every rank.  Works without an RCCL communicator too (then ``rank`` / ``world`` define the ranks: sums cross them, halos
    do not).  ``KRYPY_AMD_XR=0`` keeps RCCL.  Returns True when the transport is on."""
    rank = rdv.rank if rank is None else rank
    world = rdv.world if world is None else world
    want = (os.environ.get("KRYPY_AMD_XR", "1") != "0" and hasattr(ctx, "xr_export") and world <= 16)
    ok, handle = 0.0, b"\0" * 64
    if want:
        try:
            handle = ctx.xr_export()
            ok = 1.0
        except _hip.BackendError:
            ok = 0.0
    handles = rdv.allgather_bytes(handle)
    if rdv.allreduce_min(ok) < 1.0:
        return False
    try:
        ctx.xr_attach(rank, world, b"".join(handles))
        ok = 1.0
    except _hip.BackendError:
        ok = 0.0
    if rdv.allreduce_min(ok) < 1.0:
        if ok:
            ctx.xr_detach()
        return False
    before = (ctx.rank, ctx.nranks)
    ctx.xr_enable(rank, world)
    # a short self-test before anything depends on it: a few sums whose results every rank can work out by itself, with
    # a short timeout.  Stale or invisible stores (a platform where fine-grained IPC memory does not behave as assumed)
    # show up here as a timeout or a wrong sum - then EVERY rank goes back to RCCL, together.
    good = 1.0
    try:
        ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_XR_SELFTEST_S", "8")) * 1e3))
        for t, count in enumerate((1, 37, 512, 700, 5, 1, 1, 64)):
            def contrib(r):
                return numpy.random.default_rng(1000 * t + r).standard_normal(count)
            expect = contrib(0)
            for r in range(1, world):
                expect = expect + contrib(r)                  # rank order, like the kernel
            got = ctx.allreduce_host(contrib(rank))
            if not numpy.array_equal(got, expect):
                good = 0.0
    except _hip.BackendError:
        good = 0.0
    if rdv.allreduce_min(good) < 1.0:
        ctx.set("xr", 0)
        ctx.xr_detach()
        ctx.rank, ctx.nranks = before
        return False
    ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_XR_TIMEOUT_S", "60")) * 1e3))
    return True


def slab_cuts(n, nranks, align=1):
    """Row boundaries ``cuts[p] .. cuts[p+1]`` of ``nranks`` near-equal contiguous slabs whose
    sizes are multiples of ``align`` (e.g. one grid line), except possibly the last."""
    units = (n + align - 1) // align
    cuts = [min(n, ((units * p) // nranks) * align) for p in range(nranks + 1)]
    cuts[-1] = n
    return cuts


def localize_columns(A_rows, row0, n_global):
    """Remap the global column indices of the row slab ``[row0, row0+nloc)`` to the local layout
    ``[local rows | ghosts from the previous slab | ghosts from the next slab]``.

    Returns ``(A_local, nrecv_prev, nrecv_next)`` where ``A_local`` is CSR with
    ``nloc + nrecv_prev + nrecv_next`` columns and row order / in-row order untouched (the
    per-row summation order of the SpMV is the same as for the unsharded matrix)."""
    A = scipy.sparse.csr_matrix(A_rows)
    nloc = A.shape[0]
    row1 = row0 + nloc
    idx = A.indices.astype(numpy.int64)
    below = idx < row0
    above = idx >= row1
    nrecv_prev = int(row0 - idx[below].min()) if below.any() else 0
    nrecv_next = int(idx[above].max() - row1 + 1) if above.any() else 0
    new = idx - row0
    new[below] = nloc + (idx[below] - (row0 - nrecv_prev))
    new[above] = nloc + nrecv_prev + (idx[above] - row1)
    A_local = scipy.sparse.csr_matrix(
        (A.data, new.astype(numpy.int32), A.indptr), shape=(nloc, nloc + nrecv_prev + nrecv_next))
    return A_local, nrecv_prev, nrecv_next


class ShardedCSROperator(utils.LinearOperator):
    """The local row slab of a block-row-sharded CSR matrix, as a krypy ``LinearOperator``.

    ``shape`` is ``(nloc, nloc)``: from the host layer's point of view the problem is the
    local slab; halo exchange and all-reduces happen below the C ABI.
    """

    def __init__(self, A_rows, row0, n_global, ctx=None, self_loop=False):
        ctx = _hip.get_context() if ctx is None else ctx
        self._ctx = ctx
        A_local, nrp, nrn = localize_columns(A_rows, row0, n_global)
        nloc = A_local.shape[0]
        self._self_loop = bool(self_loop)
        if self._self_loop:
            # measurement / test aid on ONE rank in forced multi-rank mode: the slab is a ring's only member - its own previous
            # and next neighbour (what a middle rank of N does, kernel for kernel: boundary rows out, ghost rows in; the operator
            # becomes the slab periodic across its cut)
            if ctx.nranks != 1 or not (nrp > 0 and nrn > 0):
                raise utils.ArgumentError("self_loop: one rank, and a slab with rows above and below it")
            ctx.set("halo_loopback", 1)
        # every rank learns its neighbours' halo widths: what rank p receives from p-1 is what
        # p-1 has to send "next", and vice versa
        table = numpy.zeros(3 * ctx.nranks)
        table[3 * ctx.rank] = nrp
        table[3 * ctx.rank + 1] = nrn
        table[3 * ctx.rank + 2] = nloc
        table = ctx.allreduce_host(table)
        nsend_prev = int(table[3 * (ctx.rank - 1) + 1]) if ctx.rank > 0 else 0
        nsend_next = int(table[3 * (ctx.rank + 1)]) if ctx.rank + 1 < ctx.nranks else 0
        if self._self_loop:
            nsend_prev, nsend_next = nrn, nrp         # what goes to the previous slab is ITS "next" ghost region
        # the LONGEST slab of the run: kernel choices that change the pattern of all-reduces (the one-reduction form of
        # reference-order Gram-Schmidt) must come out the same on every rank, so they are made for that length
        self._rows_max = int(table[2::3].max())
        if hasattr(ctx, "set") and ctx.nranks > 1:
            ctx.set("lowsync_rows", (int(nloc) << 32) | self._rows_max)
        if nsend_prev > nloc or nsend_next > nloc:
            raise utils.ArgumentError("halo wider than the local slab: use fewer ranks")
        self._A_local = A_local
        self.halo = (nsend_prev, nsend_next, nrp, nrn)
        self.row0, self.n_global = row0, n_global
        # device images by block dtype: the matrix' own, and - like MatrixLinearOperator - a c128 copy of a real
        # matrix the first time it meets complex vectors (complex halo entries travel as (re, im) pairs)
        self._dmats = {}
        dt = numpy.dtype(complex if numpy.dtype(A_local.dtype).kind == "c" else float)
        self._dmat = self._image(dt)
        self.halo_in_launch = False if dt.kind == "c" else self._xh_setup(self._dmat, table)
        self._xh_ready = self.halo_in_launch           # (halo_via may switch it off and on again)
        super(ShardedCSROperator, self).__init__((nloc, nloc), dt, self._dot_host)

    def _image(self, dt):
        kind = numpy.dtype(dt).kind
        dm = self._dmats.get(kind)
        if dm is None:
            dm = self._ctx.csr(self._A_local, n_cols=self._A_local.shape[1], dtype=dt)
            self._ctx.set_halo(dm, *self.halo)
            if hasattr(self._ctx, "set_rows_max"):      # (the operator carries the run's longest slab: ADVICE r05)
                self._ctx.set_rows_max(dm, self._rows_max)
            self._dmats[kind] = dm
        return dm

    def _xh_setup(self, dm, table):
        """The halo of this operator inside its SpMV's own launch (``kh_mat_xh_*``: IPC-mapped ghost granules, no RCCL kernel) -
        on EVERY rank or on none: each rank exports its granules, the 64-byte handles are gathered through the context's own
        cross-rank sums (one byte per double: exact), every rank maps its two neighbours', and only when all report success is
        it switched on.  Needs the xr transport (the sums) and a banded shard on every rank (the banded kernel carries it);
        ``KRYPY_AMD_XH=0`` keeps the RCCL exchange.  Returns True when on."""
        ctx = self._ctx
        nr, r = ctx.nranks, ctx.rank
        if not (hasattr(ctx, "xh_export") and hasattr(ctx, "get") and (nr > 1 or self._self_loop) and ctx.get("xr") == 1
                and os.environ.get("KRYPY_AMD_XH", "1") != "0"):
            return False
        ok, handle = 1.0, b"\0" * 64
        try:
            if dm.diagonals > 0:
                handle = ctx.xh_export(dm)
            else:
                ok = 0.0
        except _hip.BackendError:
            ok = 0.0
        tab = numpy.zeros(65 * nr)
        tab[65 * r: 65 * r + 64] = numpy.frombuffer(handle, dtype=numpy.uint8)
        tab[65 * r + 64] = ok
        tab = ctx.allreduce_host(tab)
        if tab[64::65].min() < 1.0:
            return False

        def hbytes(q):
            return bytes(bytearray(int(v) for v in tab[65 * q: 65 * q + 64]))

        ng = [int(table[3 * q] + table[3 * q + 1]) for q in range(nr)]
        ok = 1.0
        try:
            if self._self_loop:
                ctx.xh_attach(dm, None, 0, 0, None, 0, self_loop=True)
            else:
                ctx.xh_attach(dm, hbytes(r - 1) if r > 0 else None, ng[r - 1] if r > 0 else 0, int(table[3 * (r - 1)]) if r > 0 else 0,
                              hbytes(r + 1) if r + 1 < nr else None, ng[r + 1] if r + 1 < nr else 0)
        except _hip.BackendError:
            ok = 0.0
        if ctx.allreduce_host(numpy.array([1.0 - ok]))[0] > 0.0:
            if ok and hasattr(ctx, "xh_detach"):          # (some rank could not attach: the ranks that did let go again)
                ctx.xh_detach(dm)
            return False
        ctx.xh_enable(dm, True)
        # one application with a known answer before anything depends on it, under a short timeout: x = 1 on every rank, so
        # every ghost entry is 1 and the product is this slab's row sums (the banded kernel adds a row in SciPy's order: the same
        # bits).  A neighbour's stores that do not become visible here show up as a timeout or a wrong boundary row - then EVERY
        # rank goes back to the RCCL exchange, together.
        good = 1.0
        try:
            ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_XR_SELFTEST_S", "8")) * 1e3))
            nloc = self._A_local.shape[0]
            Xd = ctx.upload(numpy.ones((nloc, 1)))
            Yd = ctx.alloc(nloc, 1)
            ctx.apply(dm, Xd, 0, Yd, 0, 1)
            got = Yd.download()[:, 0]
            want = self._A_local.dot(numpy.ones(self._A_local.shape[1]))
            if not numpy.allclose(got, want, rtol=1e-12, atol=1e-12 * max(1.0, float(numpy.abs(want).max()))):
                good = 0.0
        except _hip.BackendError:
            good = 0.0
        finally:
            ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_XR_TIMEOUT_S", "60")) * 1e3))
        if ctx.allreduce_host(numpy.array([1.0 - good]))[0] > 0.0:
            ctx.xh_enable(dm, False)
            if hasattr(ctx, "xh_detach"):
                ctx.xh_detach(dm)
            return False
        return True

    def halo_through_rccl(self):
        """Back to the grouped ``ncclSend`` / ``ncclRecv`` exchange (every rank must do the same: a launcher whose self-test
        of the mailboxes failed calls this on all of them)."""
        self.halo_via("rccl")

    def halo_via(self, mode):
        """``"rccl"``: the grouped ``ncclSend`` / ``ncclRecv`` exchange; ``"in-launch"``: back inside the banded SpMV's own
        launch - only where ``_xh_setup`` had brought it up on every rank (the neighbours' granules stay mapped while it is
        off).  EVERY rank must make the same call: a launcher that times its candidates over both transports does."""
        want = (mode == "in-launch") and self._xh_ready
        if want != self.halo_in_launch:
            self._ctx.xh_enable(self._dmat, want)
            self.halo_in_launch = want

    def _device_matrix(self, ctx=None, dtype=None):
        if dtype is not None and numpy.dtype(dtype).kind == "c":
            return self._image(numpy.dtype(complex))
        return self._dmat

    def _apply_dev(self, X, xcol, Y, ycol, ncols=1):
        self._ctx.apply(self._device_matrix(dtype=X.dtype), X, xcol, Y, ycol, ncols)

    def _dot_host(self, X):
        X = numpy.asarray(X)
        dt = utils._bdt(self.dtype, X.dtype)
        Xd = self._ctx.upload(X, dtype=dt)
        Yd = self._ctx.alloc(self.shape[0], X.shape[1], dtype=dt)
        self._apply_dev(Xd, 0, Yd, 0, X.shape[1])
        return numpy.ascontiguousarray(Yd.download())
