#!/usr/bin/env python
"""bench.py - GMRES(100) iterations/sec on the 2-D 5-point Laplacian, N = 10^7, fp64.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W`` prints ONE
JSON line on rank 0.  A *step* is one GMRES(100) restart cycle (100 Arnoldi iterations + the
cycle end: triangular solve, x update, explicit residual) of
``RestartedGmres(ls, maxiter=100, max_restarts=K-1, tol=1e-8)`` on

    A = kron(I_2500, T_4000) + kron(T_2500, I_4000)   (N = 10,000,000, nnz = 49,987,000, CSR)
    b = numpy.random.default_rng(0).standard_normal(N),  x0 = 0

(BASELINE.json configs[1]; tolerance is not reached at this N - SURVEY.md section 0 - so the
timed region is a fixed number of cycles and the expected ConvergenceError is caught).
Inputs (CSR matrix, b) are resident in HBM before the timed region starts.

value = K*100 iterations / wall time (max over ranks, barrier + device sync on both sides).
For N > 1 the matrix rows and all vectors are sharded in contiguous slabs, one process per GPU.  ``python bench.py
--gpus N`` with no RANK / WORLD_SIZE in the environment starts the N rank processes ITSELF (`_launch`: LOCAL_RANK = i,
KRYPY_AMD_DEVICE = i, a free MASTER_PORT on 127.0.0.1; fewer than N visible devices is an error, never a silent N = 1);
under torch.distributed.run or any launcher that sets RANK / WORLD_SIZE / MASTER_* it is one of those ranks; halo
exchange + dot-product all-reduces go through RCCL inside libkrylov_hip.so; the ncclUniqueId reaches the
ranks, and the timing barrier / max-over-ranks run, over krypy_amd.dist.TcpRendezvous (plain sockets: no
PyTorch on the host side).  Total work is fixed -> "scaling": "strong".

Extra objects on the JSON line:
  roofline     the dominant kernel (Gram-Schmidt link / panel kernels), timed live with HIP
               events on the library's stream; algorithmic bytes per SURVEY.md 8(d).
  cpu_baseline the CPU oracle (NumPy/SciPy restatement of the reference) timed on a bounded
               sample of the same workload on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def laplace2d(nx, ny, row0=None, row1=None):
    """5-point Laplacian kron(I_ny,T_nx)+kron(T_ny,I_nx) as sorted int32 CSR, built with array
    arithmetic (no Python loops); optionally only grid rows [row0, row1) of the ny dimension."""
    import scipy.sparse as sp

    r0 = 0 if row0 is None else row0
    r1 = ny if row1 is None else row1
    jj, ii = np.meshgrid(np.arange(r0, r1, dtype=np.int64), np.arange(nx, dtype=np.int64),
                         indexing="ij")
    row = (jj * nx + ii).ravel()
    cols = [row - nx, row - 1, row, row + 1, row + nx]
    ok = [(jj > 0).ravel(), (ii > 0).ravel(), np.ones(row.size, bool), (ii < nx - 1).ravel(),
          (jj < ny - 1).ravel()]
    vals = [-1.0, -1.0, 4.0, -1.0, -1.0]
    C = np.stack(cols, axis=1)
    K = np.stack(ok, axis=1)
    Vv = np.broadcast_to(np.array(vals), C.shape)
    counts = K.sum(axis=1)
    indptr = np.zeros(row.size + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = C[K]
    data = Vv[K].astype(np.float64)
    A = sp.csr_matrix((data, indices.astype(np.int64), indptr), shape=(row.size, nx * ny))
    return A


def laplace3d(nx, ny, nz, z0=None, z1=None, chunk=8):
    """7-point Laplacian on an nx x ny x nz grid (x fastest, z slowest) as sorted int32 CSR with GLOBAL column
    indices; optionally only the rows of the planes [z0, z1) (a rank's slab).  Built plane-chunk by plane-chunk with
    array arithmetic straight into the CSR arrays (698.7 M entries at 500 x 500 x 400: no 7 x N temporaries)."""
    import scipy.sparse as sp

    z0 = 0 if z0 is None else z0
    z1 = nz if z1 is None else z1
    plane = nx * ny
    N = plane * nz
    nrow = plane * (z1 - z0)
    ii = np.arange(plane, dtype=np.int64) % nx
    jj = np.arange(plane, dtype=np.int64) // nx
    in_plane_ok = [None, jj > 0, ii > 0, None, ii < nx - 1, jj < ny - 1, None]
    offs = [-plane, -nx, -1, 0, 1, nx, plane]
    vals = np.array([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0])
    per_plane = np.full(plane, 7, dtype=np.int64) - (jj == 0) - (ii == 0) - (ii == nx - 1) - (jj == ny - 1)
    counts = np.tile(per_plane, z1 - z0)
    if z0 == 0:
        counts[:plane] -= 1
    if z1 == nz:
        counts[nrow - plane:] -= 1
    indptr = np.zeros(nrow + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    del counts
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    data = np.empty(int(indptr[-1]), dtype=np.float64)
    for c0 in range(z0, z1, chunk):
        c1 = min(z1, c0 + chunk)
        kk = np.repeat(np.arange(c0, c1, dtype=np.int64), plane)
        row = kk * plane + np.tile(np.arange(plane, dtype=np.int64), c1 - c0)
        K = np.ones((row.size, 7), dtype=bool)
        for d in (1, 2, 4, 5):
            K[:, d] = np.tile(in_plane_ok[d], c1 - c0)
        K[:, 0] = kk > 0
        K[:, 6] = kk < nz - 1
        C = row[:, None] + np.array(offs, dtype=np.int64)[None, :]
        lo, hi = indptr[(c0 - z0) * plane], indptr[(c1 - z0) * plane]
        indices[lo:hi] = C[K]
        data[lo:hi] = np.broadcast_to(vals, C.shape)[K]
        del K, C, row, kk
    return sp.csr_matrix((data, indices, indptr), shape=(nrow, N))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="BASELINE.json config: 2 = GMRES(100) on the 2-D Laplacian N = 10^7 (the metric; default), "
                         "5 = DeflatedGmres with 16 recycled Ritz vectors on the 3-D 7-point Laplacian "
                         "500 x 500 x 400 (N = 10^8), z-slabs over the ranks")
    ap.add_argument("--nz", type=int, default=400, help="config 5: planes of the grid (nx, ny default to 500 there)")
    ap.add_argument("--defl", type=int, default=16, help="config 5: recycled Ritz vectors")
    ap.add_argument("--steps", type=int, default=8, help="timed GMRES(100) restart cycles")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up cycles")
    ap.add_argument("--nx", type=int, default=4000)
    ap.add_argument("--ny", type=int, default=2500)
    ap.add_argument("--restart", type=int, default=100)
    ap.add_argument("--ortho", default=os.environ.get("KRYPY_AMD_BENCH_ORTHO", "auto"),
                    help="Gram-Schmidt variant of the timed region.  auto = mgs on one GPU (the "
                         "reference's sequential order, register-resident chain kernel) and cgs on "
                         "several GPUs (panel classical GS: one all-reduce per step instead of k+1). "
                         "Also: mgs | dmgs | cgs | cgs2.  All pass the 1e-10 parity tests.")
    ap.add_argument("--other-modes", default="cgs,cgs2",
                    help="comma list of further variants measured AFTER the timed region (N=1 only)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="testing: take the multi-GPU code path (gloo init, RCCL communicator, "
                         "ShardedCSROperator, all-reduced reductions) even with one rank")
    ap.add_argument("--loop-halo", action="store_true",
                    help="with --force-sharded: the one rank's slab is a MIDDLE slab of the problem split over N ranks, with itself as "
                         "previous and next neighbour (periodic across its cuts): boundary rows out and ghost rows in exactly as a "
                         "middle rank has them - through the in-launch exchange of the banded SpMV when the mailboxes are on")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "xr"),
                    help="N > 1 ranks.  rccl (default): an RCCL communicator, with the sums across the ranks (and a banded shard's "
                         "halo) moved to the IPC mailboxes of csrc/xr.hip when every rank can use them.  xr: NO RCCL communicator "
                         "at all - sums, halos and the in-launch exchange of the blocked kernel through the mailboxes alone (a run "
                         "that cannot do that fails)")
    ap.add_argument("--share-devices", action="store_true",
                    help="testing: --gpus N ranks on FEWER than N devices (rank r on device r mod the visible ones); only with "
                         "--transport xr (RCCL refuses two ranks on one device).  The line then reports n_gpus = the devices "
                         "really used and config.ranks = N.  For small problems: the ranks' kernels wait for each other inside "
                         "their launches, so all of them must fit on the shared device together (at the benchmark's size one "
                         "rank's blocked kernel fills it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the kernel micro-benchmarks (profiling runs that want the solver's kernels only)")
    ap.add_argument("--cpu-budget-s", type=float, default=130.0,
                    help="seconds of CPU work the oracle's timed GMRES(100) cycle may take before it is cut "
                         "and extrapolated (a full cycle at N = 10^7 takes about two minutes)")
    args = ap.parse_args()
    if args.loop_halo and not (args.force_sharded and args.gpus == 1):
        ap.error("--loop-halo: with --force-sharded on one rank (--gpus 1)")
    return args


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([int(p.get("num_threads", 1)) for p in threadpool_info()] or [1])
    except Exception:
        return None


def _oracle_cycle(ref, A, b, m, budget_s):
    """One GMRES(m) cycle of oracle Arnoldi steps, timed step by step; cut at budget_s seconds."""
    st = ref.arnoldi_init(A, b, m)          # warm-up: page in V, first touches
    for _ in range(2):
        ref.arnoldi_step(st)
    del st
    st = ref.arnoldi_init(A, b, m)
    ts = []
    t0 = time.perf_counter()
    for _ in range(m):
        t1 = time.perf_counter()
        ref.arnoldi_step(st)
        ts.append(time.perf_counter() - t1)
        if time.perf_counter() - t0 > budget_s:
            break
    total = time.perf_counter() - t0
    done = len(ts)
    if done == m:
        return m / total, done, "one fully timed %d-step cycle, %.1f s" % (m, total)
    k = np.arange(done)
    c, a = np.polyfit(k[1:], np.array(ts)[1:], 1) if done > 2 else (0.0, float(np.mean(ts)))
    cycle = float(np.sum(ts) + np.sum(a + c * np.arange(done, m)))
    return m / cycle, done, ("first %d of %d steps timed (%.1f s, budget %.0f s), the rest extrapolated from the "
                            "fit a + c k (a = %.3f s, c = %.4f s)" % (done, m, total, budget_s, a, c))


def cpu_baseline(A, b, m, budget_s):
    """The CPU oracle (oracle/krylov_ref.py, the NumPy/SciPy restatement of the reference) on the SAME
    inputs, on this node's host cores (SURVEY 8d / BASELINE.md section 4): after a 2-step warm-up ONE
    GMRES(m) cycle of Arnoldi steps, timed step by step, (i) with one BLAS thread - the whole cycle, cut at
    `budget_s` seconds - and (ii) with the thread pool NumPy picks by itself, on a sample of a fifth of that
    budget (extrapolated with the linear fit of the per-step times: an MGS step costs a + c k).  `value`
    is the faster of the two and `cores` the threads it used.  SpMV: 20 repetitions (SciPy's csr_matvec
    is single-threaded)."""
    from oracle import krylov_ref as ref

    nthr = _blas_threads()
    runs = {}
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            runs[1] = _oracle_cycle(ref, A, b, m, budget_s)
    except ImportError:
        pass
    if not runs or (nthr or 1) > 1:
        runs[nthr or 1] = _oracle_cycle(ref, A, b, m, budget_s if not runs else budget_s / 5.0)
    best = max(runs, key=lambda t: runs[t][0])
    xs = np.ascontiguousarray(b)
    A.dot(xs)
    t2 = time.perf_counter()
    for _ in range(20):
        A.dot(xs)
    spmv = (time.perf_counter() - t2) / 20
    desc = "; ".join("%d BLAS thread%s: %.3f it/s (%s)" % (t, "" if t == 1 else "s", runs[t][0], runs[t][2])
                     for t in sorted(runs))
    return {
        "value": runs[best][0], "unit": "iterations/s", "cores": best, "kind": "port",
        "sample": ("GMRES(%d) Arnoldi cycle at N=%d with the NumPy/SciPy oracle (oracle/krylov_ref.py), same A and "
                   "b as the GPU run, each after a 2-step warm-up - %s; os.cpu_count() = %s, OMP_NUM_THREADS=%s.  "
                   "Context: the unmodified reference itself measured 0.2-0.35 it/s at N = 10^7 on 8 vCPUs in the "
                   "build container (BASELINE.md section 2)"
                   % (m, A.shape[0], desc, os.cpu_count(), os.environ.get("OMP_NUM_THREADS", "unset"))),
        "runs": dict((str(t), {"iterations_per_s": runs[t][0], "steps_timed": runs[t][1]}) for t in runs),
        "cpu_count": os.cpu_count(), "blas_threads_default": nthr,
        "spmv_ms_1thread": spmv * 1e3,
        "spmv_gbs_1thread": (12.0 * A.nnz + 4.0 * (A.shape[0] + 1) + 16.0 * A.shape[0]) / spmv / 1e9,
    }


class _StdoutToStderr(object):
    """Everything written to fd 1 while this is active goes to stderr (RCCL and gloo print banners
    on stdout); the contract is ONE JSON line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _free_port():
    import socket
    best = None
    for _ in range(16):          # TcpRendezvous listens on one of the eight ports BEHIND MASTER_PORT
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        best = sk.getsockname()[1]
        sk.close()
        if best < 65000:
            break
    return best


def _visible_devices():
    """Number of GPUs this process can see (kh_device_count of the HIP library; built first if missing)."""
    from krypy_amd import _hip
    if not os.path.exists(_hip.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _hip.device_count()


def _launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) here, relay rank 0's
    ONE JSON line and the ranks' exit codes.  The children are this very command line again (sys.argv) with RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT and KRYPY_AMD_DEVICE set - what torch.distributed.run would set."""
    import subprocess
    n = args.gpus
    have = _visible_devices()
    share = args.share_devices and args.transport == "xr" and have >= 1
    if have < n and not share:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU%s visible: refusing to run (a smaller run would be "
                         "reported as n_gpus=%d)\n" % (n, have, "" if have == 1 else "s", n))
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KRYPY_AMD_DEVICE=str(r % have if share else r),
                   KRYPY_AMD_BENCH_DEVICES=str(min(n, have)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL between processes)
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0 prints the one JSON line; it is read on a thread of its own so that a rank that dies while rank 0 still waits
    # for it (in the rendezvous, in a collective) is seen by the loop below, which then stops the others
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    deadline = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) or rcs[0] == 0:
            # a rank failed (its peers would wait in a collective for ever), or rank 0 is done: give the others a
            # grace period, then stop them - by PID, these are our own children
            if deadline is None:
                deadline = time.time() + (60.0 if all(rc in (None, 0) for rc in rcs) else 10.0)
            elif time.time() > deadline:
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        p.kill()
                        rcs[r] = -9
        time.sleep(0.05)
    reader.join(10.0)
    line = got[0] if got else b""
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    sys.stdout.write(line.decode() if isinstance(line, bytes) else line)
    sys.stdout.flush()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(_launch(args))
    with _StdoutToStderr():
        out, rank, dist = _run()
    if rank == 0:
        print(json.dumps(out), flush=True)
    # from here on nothing may reach stdout any more: RCCL prints its version banner from a library destructor at
    # process exit, i.e. AFTER the JSON line (seen with --force-sharded: "Extra data" for a JSON parser)
    sys.stdout.flush()
    os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.close()


def _world(args):
    """(rank, world, local_rank) of this process; --gpus must be the number of ranks that really run."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the line would not say what ran" % (args.gpus, world))
    if not (0 <= rank < world):
        raise SystemExit("bench.py: RANK=%d outside WORLD_SIZE=%d" % (rank, world))
    return rank, world, local_rank


def _run():
    args = parse_args()
    if args.config == 5:
        return _run_config5(args)
    rank, world, local_rank = _world(args)
    os.environ.setdefault("KRYPY_AMD_DEVICE", str(local_rank))

    sharded = world > 1 or args.force_sharded
    if args.force_sharded:
        os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    dist = None
    if sharded:
        from krypy_amd.dist import TcpRendezvous      # plumbing only: unique-id broadcast, barrier, max over ranks
        dist = TcpRendezvous(rank, world)

    import krypy_amd
    from krypy_amd import _hip, linsys, utils

    if not os.path.exists(_hip.library_path()):     # fresh checkout: compile the HIP library once
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    ctx = _hip.get_context()
    nx, ny, m = args.nx, args.ny, args.restart
    N = nx * ny
    ortho = args.ortho
    auto_sharded = sharded and ortho == "auto"
    if ortho == "auto":
        ortho = "cgs" if sharded else "mgs"
    if sharded:
        from krypy_amd import dist as kdist
        if args.transport == "rccl":
            uid = dist.broadcast_bytes(ctx.comm_unique_id() if rank == 0 else None)
            ctx.comm_init(rank, world, uid)
        # the sums across the ranks through IPC-mapped mailboxes (csrc/xr.hip) when every rank can map every peer's and a
        # self-test of a few sums passes on all of them - otherwise ncclAllReduce, on every rank alike (KRYPY_AMD_XR=0: RCCL)
        xr_on = kdist.enable_xr(ctx, dist)
        if args.transport == "xr" and not xr_on:
            raise SystemExit("bench.py --transport xr: the mailboxes did not come up on every rank, and there is no RCCL communicator")
        # contiguous slabs of grid rows (y index): every shard holds whole x-lines
        cuts = [(ny * p) // world for p in range(world + 1)]
        if args.loop_halo:
            if not (args.force_sharded and world == 1):
                raise SystemExit("bench.py --loop-halo: with --force-sharded on one rank")
            Aloc = laplace2d(nx, 3 * ny, ny, 2 * ny)
            op = kdist.ShardedCSROperator(Aloc, ny * nx, 3 * N, ctx, self_loop=True)
        else:
            Aloc = laplace2d(nx, ny, cuts[rank], cuts[rank + 1])
            op = kdist.ShardedCSROperator(Aloc, cuts[rank] * nx, N, ctx)
        b_full = np.random.default_rng(0).standard_normal(N)
        b = b_full[cuts[rank] * nx: cuts[rank + 1] * nx].copy()
        A_for_ls = op
        nnz_global = 5 * N - 2 * nx - 2 * ny
    else:
        A_for_ls = laplace2d(nx, ny)
        nnz_global = A_for_ls.nnz
        b = np.random.default_rng(0).standard_normal(N)

    ls = linsys.LinearSystem(A_for_ls, b)
    xr_on = bool(locals().get("xr_on", False))

    cycle_marks = []

    class StampedGmres(linsys.Gmres):
        """linsys.Gmres that notes the host time when its cycle ends (diagnostics only: one
        perf_counter call per 100 iterations)."""

        def _finalize(self):
            super(StampedGmres, self)._finalize()
            cycle_marks.append(time.perf_counter())

    def run_cycles(ncyc, x0, ortho=None):
        try:
            # RestartedGmres == _RestartedSolver(Gmres, ...) (linsys.py:1075-1081)
            sol = linsys._RestartedSolver(StampedGmres, ls, x0=x0, maxiter=m,
                                          max_restarts=ncyc - 1, tol=1e-8, ortho=ortho or ortho_timed[0])
        except utils.ConvergenceError as e:
            sol = e.solver
        return sol

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- N > 1 ranks, --ortho auto: which Gram-Schmidt form is faster on THIS node is a property of its links - the panel
    # form needs two sums across the ranks per step and reads the local basis twice; the reference order runs (xr transport
    # on, slabs of up to 6 rows per lane) as the blocked kernel with one exchange per four columns INSIDE the launch and
    # reads the basis once, else as the one-reduction form.  One untimed cycle of each, the max over the ranks decides - on
    # every rank alike.  Both forms pass the 1e-10 parity tests.  A failure of the candidate (its in-launch sums have never
    # run between two GPUs) must not cost the run its line: every rank then goes back to RCCL and the panel form, together.
    ortho_timed = [ortho]
    auto_report = None
    if auto_sharded and hasattr(ctx, "get"):
        auto_report = {}
        probe_ok = 1.0
        for cand in ("cgs", "mgs"):
            try:
                if cand == "mgs" and xr_on:
                    ctx.set("xr_timeout_ms", 10000)
                barrier()
                b2 = ctx.get("n_chain_blk2")
                t1 = time.perf_counter()
                run_cycles(1, None, ortho=cand)
                ctx.sync()
                d1 = time.perf_counter() - t1
                auto_report[cand] = {"ms_per_cycle": dist.allreduce_max(d1) * 1e3,
                                     "sums_inside_the_launch": bool(ctx.get("n_chain_blk2") > b2)}
            except _hip.BackendError as exc:
                probe_ok = 0.0
                auto_report[cand] = {"error": str(exc)[:300]}
            if cand == "mgs":
                all_ok = dist.allreduce_min(probe_ok) >= 1.0
                if not all_ok and args.transport == "xr":
                    raise SystemExit("bench.py --transport xr: the reference-order candidate failed on some rank and there is no RCCL "
                                     "communicator to go back to: %r" % (auto_report,))
                if not all_ok:
                    # some rank's candidate failed: the mailboxes' epochs may no longer agree - off with them, everywhere
                    ctx.set("chain_blk2", 0)
                    if hasattr(A_for_ls, "halo_through_rccl"):
                        A_for_ls.halo_through_rccl()
                    if xr_on:
                        ctx.set("xr", 0)
                        ctx.xr_detach()
                        xr_on = False
                    auto_report.setdefault("mgs", {})["disabled"] = "a rank failed: every rank back to ncclAllReduce and the panel form"
                    auto_report["mgs"].pop("ms_per_cycle", None)
                elif xr_on:
                    ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_XR_TIMEOUT_S", "60")) * 1e3))
        if "ms_per_cycle" in auto_report.get("mgs", {}) and auto_report["mgs"]["ms_per_cycle"] < auto_report["cgs"]["ms_per_cycle"]:
            ortho_timed[0] = "mgs"
        ortho = ortho_timed[0]
        auto_report["chosen"] = ortho

    x0 = None
    sol = None
    if args.warmup > 0:
        sol = run_cycles(args.warmup, None)
        x0 = sol.__dict__["_xk_dev"]
    barrier()
    del cycle_marks[:]
    t0 = time.perf_counter()
    sol = run_cycles(args.steps, x0)
    ctx.sync()
    dt = time.perf_counter() - t0
    cycle_ms = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + cycle_marks[:-1], cycle_marks)]
    if dist is not None:
        dt = dist.allreduce_max(dt)
    n_iters = len(sol.resnorms) - 1
    assert n_iters == args.steps * m, (n_iters, args.steps, m)
    its = n_iters / dt

    # ---- loss of orthogonality of one cycle's basis, ||V^T V - I||_F over all 101 columns (outside the timed
    # region): the evidence that the panel form the sharded runs default to is as good a basis as the
    # reference-order MGS at this size (all-reduced over the ranks like every other inner product) ----
    def basis_orthogonality(mode):
        try:
            s1 = linsys.Gmres(ls, x0=x0, maxiter=m, tol=1e-8, ortho=mode, store_arnoldi=True)
        except utils.ConvergenceError as e:
            s1 = e.solver
        Vb = s1.arnoldi._V
        G = ctx.gemm_tn(Vb, 0, m + 1, Vb, 0, m + 1)
        return float(np.linalg.norm(G - np.eye(m + 1)))

    orth = {}
    try:
        orth[ortho] = basis_orthogonality(ortho)
        if not sharded:          # one GPU: both the reference order and the panel form the N > 1 runs default to
            for mode in ("mgs", "cgs"):
                if mode not in orth:
                    orth[mode] = basis_orthogonality(mode)
    except Exception as exc:
        orth["error"] = repr(exc)

    # ---- the other Gram-Schmidt variants on the same inputs (outside the timed region) ----
    others = {}
    if not sharded and args.other_modes:
        for mode in [t for t in args.other_modes.split(",") if t and t != ortho and t != "none"]:
            barrier()
            t1 = time.perf_counter()
            s2 = run_cycles(args.steps, x0, ortho=mode)
            ctx.sync()
            d2 = time.perf_counter() - t1
            others[mode] = {"iterations_per_s": (len(s2.resnorms) - 1) / d2,
                            "ms_per_cycle": d2 / args.steps * 1e3,
                            "final_relres": float(s2.resnorms[-1])}

    # sharded runs: the reference order next to the panel form.  On N ranks ortho='mgs' takes all k + 1 coefficients of a
    # step from one pass and ONE all-reduce (Gram-table correction, krylov_hip.hip: try_lowsync_mgs) - two all-reduces per
    # step like 'cgs', the reference's recurrence in exact arithmetic; all-reduces per Arnoldi step are counted
    # (on one rank in forced mode by default; on a real N-rank run only with KRYPY_AMD_BENCH_SHARDED_EXTRAS=1 - the one-reduction form
    # has run through a 1-rank communicator only, and an untimed extra must not be able to cost the run its line)
    if (sharded and args.other_modes and hasattr(ctx, "get")
            and (world == 1 or os.environ.get("KRYPY_AMD_BENCH_SHARDED_EXTRAS", "0") == "1")):
        for mode in [t for t in ("cgs", "mgs") if t != ortho]:
            barrier()
            a0 = ctx.get("n_allreduce")
            t1 = time.perf_counter()
            s2 = run_cycles(args.steps, x0, ortho=mode)
            ctx.sync()
            d2 = time.perf_counter() - t1
            n2 = max(len(s2.resnorms) - 1, 1)
            others[mode] = {"iterations_per_s": (len(s2.resnorms) - 1) / d2, "ms_per_cycle": d2 / args.steps * 1e3,
                            "final_relres": float(s2.resnorms[-1]),
                            "allreduces_per_iteration": (ctx.get("n_allreduce") - a0) / float(n2)}
            try:
                orth[mode] = basis_orthogonality(mode)
            except Exception as exc:
                orth[mode] = repr(exc)

    # the reference-order solver on a GENERAL CSR operator (the CSR-stream SpMV kernel + the chain kernel: what a matrix
    # that is not a stencil gets), and with the banded SpMV as a launch of its own (no operator in the chain's prologue)
    if not sharded and args.other_modes and ortho == "mgs" and hasattr(ctx, "set"):
        for label, key in (("mgs, general CSR SpMV kernel + chain kernel (spmv_dia = 0, as KRYPY_AMD_SPMV_DIA=0)", "spmv_dia"),
                           ("mgs, banded SpMV launch + chain kernel (chain_spmv = 0, as KRYPY_AMD_CHAIN_SPMV=0)", "chain_spmv")):
            barrier()
            ctx.set(key, 0)
            try:
                t1 = time.perf_counter()
                s2 = run_cycles(args.steps, x0, ortho="mgs")
                ctx.sync()
                d2 = time.perf_counter() - t1
            finally:
                ctx.set(key, 1)
            others[label] = {"iterations_per_s": (len(s2.resnorms) - 1) / d2, "ms_per_cycle": d2 / args.steps * 1e3,
                             "final_relres": float(s2.resnorms[-1])}

    # ---- roofline of the dominant kernel, timed live with HIP events on the library stream ----
    nloc = ls.N
    roof = None
    extra = {}
    try:
        from krypy_amd import _bench
        if args.no_roofline:
            raise RuntimeError("skipped (--no-roofline)")
        import glob
        # HBM bytes per launch from PMC passes of this same command (tools/profile.sh + tools/summarize_prof.py;
        # bench.py cannot collect counters itself): attached only if the file carries the stamp of the kernel
        # sources that have just been timed, otherwise `traffic` stays null and the byte model is used
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        roof, extra = _bench.roofline(ctx, ls, ortho, HBM_PEAK_GBS, traffic_files=tfiles, m=m)
    except Exception as exc:   # never lose the headline number to the instrumentation
        extra = {"roofline_error": repr(exc)}

    out = {
        "metric": "GMRES iterations/sec + SpMV HBM GB/s, n=10^7 5-pt Laplacian fp64",
        "value": its, "unit": "iterations/s", "n_gpus": int(os.environ.get("KRYPY_AMD_BENCH_DEVICES", world)), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "GMRES(%d) restart cycles, 2-D 5-pt Laplacian %dx%d CSR (N=%d, nnz=%d), "
                               "b=rng(0) normal, x0=0, tol=1e-8 (BASELINE.json configs[1])"
                               % (m, nx, ny, N, nnz_global),
                   "n": N, "ortho": ortho, "restart": m, "iterations_timed": n_iters,
                   "parallelism": "1 GPU" if not sharded else "row-sharded x%d (%s)" % (world, "RCCL" if args.transport == "rccl" else "mailboxes only"),
                   "ranks": world,
                   # sums across the ranks: "xr" = one kernel of system-scope stores into the peers' IPC-mapped mailboxes
                   # per panel (csrc/xr.hip), "rccl" = ncclAllReduce; the halo exchange is RCCL point-to-point either way
                   "cross_rank_sums": None if not sharded else ("xr" if xr_on else "rccl"),
                   # the halo of the sharded SpMV: "in-launch" = boundary rows stored into the neighbours' IPC-mapped ghost granules
                   # by the banded kernel itself (kh_mat_xh_*), "rccl" = grouped ncclSend / ncclRecv on the communication stream
                   "halo": None if not sharded else (("in-launch" if getattr(A_for_ls, "halo_in_launch", False) else "rccl") +
                                                    (" (the slab is its own neighbour)" if args.loop_halo else "")),
                   "ortho_auto": auto_report,
                   "final_relres": float(sol.resnorms[-1]), "cycle_ms": cycle_ms,
                   # guard against a slow first cycle / a box in a low power state: `value` is K cycles over their
                   # total time (the contract); the median cycle says what a typical one took
                   "median_cycle_ms": float(np.median(cycle_ms)) if cycle_ms else None,
                   "iterations_per_s_at_median_cycle": (m / float(np.median(cycle_ms)) * 1e3) if cycle_ms else None,
                   "basis_orthogonality_fro": orth},
        "roofline": roof,
    }
    out.update(extra)
    if others:
        out["other_modes"] = others
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(A_for_ls, b, m, args.cpu_budget_s)
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}
    return out, rank, dist


def _run_config5(args):
    """BASELINE.json configs[4]: 3-D 7-point Laplacian on 500 x 500 x 400 points (N = 10^8, nnz = 698,700,000), rows in
    z-slabs over the ranks (400 / 8 = 50 planes = 12.5 M rows per GPU), b = rng(0) normal.  Solve 1: plain GMRES(m)
    (DeflatedGmres without U) to harvest the `defl` smallest-magnitude Ritz vectors ON THE DEVICE (every rank keeps its
    slab of them; the small eigenproblem is replicated host work); solve 2: DeflatedGmres(U, maxiter=m), timed - a
    *step* is one such solve of m iterations from x0 = 0 (reference flow: recycling/linsys.py:51-103,
    deflation.py:93-163).  Same JSON contract as config 2; `scaling` is "strong" (the grid is fixed)."""
    rank, world, local_rank = _world(args)
    os.environ.setdefault("KRYPY_AMD_DEVICE", str(local_rank))
    sharded = world > 1 or args.force_sharded
    if args.force_sharded:
        os.environ["KRYPY_AMD_FORCE_MULTI"] = "1"
    dist = None
    if sharded:
        from krypy_amd.dist import TcpRendezvous      # plumbing only: unique-id broadcast, barrier, max over ranks
        dist = TcpRendezvous(rank, world)
    from krypy_amd import _hip, deflation, linsys, utils

    if not os.path.exists(_hip.library_path()):
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if dist is not None:
            dist.barrier()
    ctx = _hip.get_context()
    nx = 500 if args.nx == 4000 else args.nx          # (4000 / 2500 are config 2's defaults)
    ny = 500 if args.ny == 2500 else args.ny
    nz, m, d = args.nz, args.restart, args.defl
    plane = nx * ny
    N = plane * nz
    ortho = args.ortho
    if ortho == "auto":
        ortho = "cgs" if sharded else "mgs"
    b_rng = np.random.default_rng(0)
    if sharded:
        from krypy_amd import dist as kdist
        if args.transport == "rccl":
            uid = dist.broadcast_bytes(ctx.comm_unique_id() if rank == 0 else None)
            ctx.comm_init(rank, world, uid)
        xr_on = kdist.enable_xr(ctx, dist)
        if args.transport == "xr" and not xr_on:
            raise SystemExit("bench.py --transport xr: the mailboxes did not come up on every rank, and there is no RCCL communicator")
        cuts = [(nz * p) // world for p in range(world + 1)]          # whole planes per rank
        z0, z1 = cuts[rank], cuts[rank + 1]
        Aloc = laplace3d(nx, ny, nz, z0, z1)
        op = kdist.ShardedCSROperator(Aloc, z0 * plane, N, ctx)
        del Aloc
        # every rank draws the same stream and keeps its slab (the right-hand side of the unsharded run)
        b = None
        for p in range(world):
            seg = b_rng.standard_normal(plane * (cuts[p + 1] - cuts[p]))
            if p == rank:
                b = seg
        A_for_ls = op
    else:
        A_for_ls = laplace3d(nx, ny, nz)
        b = b_rng.standard_normal(N)
    nnz_global = 7 * N - 2 * (nx * ny + ny * nz + nx * nz)
    ls = linsys.LinearSystem(A_for_ls, b, self_adjoint=True)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    def solve(U):
        try:
            return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=m, ortho=ortho, store_arnoldi=U is None)
        except utils.ConvergenceError as e:
            return e.solver

    # solve 1 (untimed): harvest the Ritz vectors on the device
    s0 = solve(None)
    ritz = deflation.Ritz(s0)
    U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:d])
    plain_relres = float(s0.resnorms[-1])
    ritz_values = np.sort(np.abs(ritz.values))[:d]
    del s0, ritz
    for _ in range(args.warmup):
        solve(U)
    barrier()
    cycle_marks = []
    t0 = time.perf_counter()
    n_iters = 0
    for _ in range(args.steps):
        s1 = solve(U)
        n_iters += len(s1.resnorms) - 1
        cycle_marks.append(time.perf_counter())
    ctx.sync()
    dt = time.perf_counter() - t0
    cycle_ms = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + cycle_marks[:-1], cycle_marks)]
    if dist is not None:
        dt = dist.allreduce_max(dt)
    assert n_iters == args.steps * m, (n_iters, args.steps, m)
    its = n_iters / dt
    nloc = ls.N
    # bytes one deflated iteration has to move at least: the operator, the Gram-Schmidt columns once per
    # use, the projector's two sweeps over the 2 d vectors (560 N for d = 16)
    it_bytes = (12.0 * nnz_global + 4.0 * (N + 1) + 16.0 * N) + 16.0 * N * (m + 1) / 2.0 + 48.0 * N + \
        (32.0 * d + 48.0) * N
    out = {
        "metric": "DeflatedGmres iterations/sec, 3-D 7-pt Laplacian n=10^8 row-sharded, 16 recycled Ritz vectors, fp64",
        "value": its, "unit": "iterations/s", "n_gpus": int(os.environ.get("KRYPY_AMD_BENCH_DEVICES", world)), "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "DeflatedGmres(%d) solves with %d recycled Ritz vectors, 3-D 7-pt Laplacian %dx%dx%d CSR "
                               "(N=%d, nnz=%d), b=rng(0) normal, x0=0 (BASELINE.json configs[4]); Ritz vectors harvested "
                               "from one plain GMRES(%d) solve, untimed" % (m, d, nx, ny, nz, N, nnz_global, m),
                   "n": N, "rows_per_gpu": nloc, "ortho": ortho, "restart": m, "deflation_vectors": d,
                   "iterations_timed": n_iters,
                   "parallelism": "1 GPU" if not sharded else "z-slabs x%d (%s)" % (world, "RCCL" if args.transport == "rccl" else "mailboxes only"),
                   "ranks": world,
                   "cross_rank_sums": None if not sharded else ("xr" if locals().get("xr_on") else "rccl"),
                   "halo": None if not sharded else (("in-launch" if getattr(A_for_ls, "halo_in_launch", False) else "rccl") +
                                                    (" (the slab is its own neighbour)" if args.loop_halo else "")),
                   "plain_relres": plain_relres, "deflated_relres": float(s1.resnorms[-1]),
                   "smallest_ritz_values": [float(v) for v in ritz_values[:4]],
                   "cycle_ms": cycle_ms, "median_cycle_ms": float(np.median(cycle_ms))},
        "roofline": {"bound": "hbm", "achieved": it_bytes / world * its / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": it_bytes / world * its / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "bytes_source": "SURVEY 8(d) algorithmic bytes of one deflated iteration per GPU (operator as CSR, "
                                     "every basis column twice, projector 560 N for d = 16), whole-solve average - not "
                                     "a per-kernel figure"},
    }
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        out["cpu_baseline"] = {"value": None, "note": "config 2 carries the CPU baseline (bench.py without --config)"}
    return out, rank, dist


if __name__ == "__main__":
    main()
