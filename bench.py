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
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json config: 2 = GMRES(100) on the 2-D Laplacian N = 10^7 (the metric; default), "
                         "3 = MINRES + Jacobi on the same matrix (one MI355X; a step = one solve of --iters iterations), "
                         "4 = CG on the dense SPD matrix n = 32768 (one MI355X; a step = one whole solve to tol 1e-8), "
                         "5 = DeflatedGmres with 16 recycled Ritz vectors on the 3-D 7-point Laplacian "
                         "500 x 500 x 400 (N = 10^8), z-slabs over the ranks")
    ap.add_argument("--iters", type=int, default=200, help="config 3: MINRES iterations per step (SURVEY 8d: 200)")
    ap.add_argument("--dense-n", type=int, default=32768, help="config 4: order of the dense SPD matrix")
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
    # the whole run is bounded: a collective that never returns (a link that never delivers) must end in a non-zero exit,
    # not hold the driver until ITS timeout
    overall = time.time() + float(os.environ.get("KRYPY_AMD_BENCH_DEADLINE_S", "1500"))
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if time.time() > overall:
            sys.stderr.write("bench.py: the ranks did not finish within KRYPY_AMD_BENCH_DEADLINE_S; stopping them\n")
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()
                    rcs[r] = -9
            break
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


def _probe_sharded(ctx, dist, args, op, xr_on, run_one, xr_timeout_ms=10000):
    """N > 1 ranks, --ortho auto: which Gram-Schmidt form over which transport is faster on THIS node is a property of its
    links.  Candidates, one untimed pass of `run_one(ortho)` each, the max over the ranks decides, on every rank alike:

      cgs / rccl   the panel form, two ncclAllReduce per step, halo as grouped ncclSend / ncclRecv (the only path that has no
                   kernel of ours waiting for another GPU)
      cgs / xr     the same with the sums (and a banded shard's halo) through the IPC mailboxes
      mgs / xr     the reference order with the cross-rank sums INSIDE the launch: the blocked kernel up to 2.5 M rows per rank,
                   the register-resident chain kernel beyond (the local basis is read once), else the one-reduction form
      mgs / rccl   (no mailboxes) the one-reduction form / per-link all-reduces

    EVERY rank runs the same sequence of host collectives whatever happened locally (a rank whose candidate
    raised used to skip one and the star around rank 0 went off by one).  A candidate that fails over the mailboxes - its
    in-launch sums have never run between two GPUs - takes the mailboxes off the table for the rest of the run, on every rank
    together (the epochs may no longer agree).  Returns (ortho, transport, xr_on, report)."""
    from krypy_amd import _hip
    have_rccl = args.transport == "rccl"
    cands = []
    if have_rccl:
        cands.append(("cgs", "rccl"))
    if xr_on:
        cands += [("cgs", "xr"), ("mgs", "xr")]
    elif have_rccl:
        cands.append(("mgs", "rccl"))
    report = {"candidates": []}
    state = {"xr_on": xr_on, "xr_usable": xr_on}

    def set_transport(tr):
        """sums (and the banded halo) through the mailboxes or through RCCL - every rank makes the same call"""
        if not state["xr_usable"] or not hasattr(ctx, "set"):
            return
        want = tr == "xr"
        ctx.set("xr", 1 if want else 0)
        if hasattr(op, "halo_via"):
            op.halo_via("in-launch" if want else "rccl")
        state["xr_on"] = want

    best = None
    for ortho, tr in cands:
        entry = {"ortho": ortho, "transport": tr}
        report["candidates"].append(entry)
        if tr == "xr" and not state["xr_usable"]:
            entry["skipped"] = "the mailboxes were taken off after an earlier candidate failed"
            continue
        ok, d1, err = 1.0, 1e30, None
        try:
            set_transport(tr)
            if tr == "xr" and hasattr(ctx, "set"):
                ctx.set("xr_timeout_ms", int(xr_timeout_ms))
            ctx.sync()
            dist.barrier()
            c0 = dict((k, ctx.get(k)) for k in ("n_chain_blk2", "n_chain_xr", "n_lowsync", "n_allreduce", "n_xr")) if hasattr(ctx, "get") else {}
            t1 = time.perf_counter()
            n_it = run_one(ortho)
            ctx.sync()
            d1 = time.perf_counter() - t1
            if c0:
                per = float(max(n_it, 1))
                entry["per_iteration"] = {"allreduce_calls": (ctx.get("n_allreduce") - c0["n_allreduce"]) / per,
                                          "mailbox_sums": (ctx.get("n_xr") - c0["n_xr"]) / per}
                entry["kernels"] = {"blocked_in_launch_sums": ctx.get("n_chain_blk2") - c0["n_chain_blk2"],
                                    "chain_in_launch_sums": ctx.get("n_chain_xr") - c0["n_chain_xr"],
                                    "one_reduction_steps": ctx.get("n_lowsync") - c0["n_lowsync"]}
        except _hip.BackendError as exc:
            ok, err = 0.0, str(exc)[:300]
        # ---- the same two collectives on every rank, whatever happened above ----
        all_ok = dist.allreduce_min(ok) >= 1.0
        dmax = dist.allreduce_max(d1 if ok else 1e30)
        if err is not None:
            entry["error"] = err
        if all_ok:
            entry["ms"] = dmax * 1e3
            if best is None or dmax < best[0]:
                best = (dmax, ortho, tr)
        else:
            entry["failed_on_some_rank"] = True
            if tr == "xr" and state["xr_usable"]:
                if not have_rccl:
                    raise SystemExit("bench.py --transport xr: a candidate failed on some rank and there is no RCCL communicator to go "
                                     "back to: %r" % (report,))
                # off with the mailboxes, everywhere: blocked / chain kernels with in-launch sums, in-launch halo, mailbox sums
                if hasattr(ctx, "set"):
                    ctx.set("chain_blk2", 0)
                    ctx.set("chain_xr", 0)
                if hasattr(op, "halo_via"):
                    op.halo_via("rccl")
                if hasattr(ctx, "set"):
                    ctx.set("xr", 0)
                if hasattr(ctx, "xr_detach"):
                    ctx.xr_detach()
                state["xr_usable"] = state["xr_on"] = False
                entry["disabled"] = "a rank failed: every rank back to ncclAllReduce / ncclSend / ncclRecv and the panel form"
            else:
                entry["disabled"] = "a rank failed: the candidate is not timed (every rank keeps the panel form over RCCL)"
    if best is None:
        raise SystemExit("bench.py: no Gram-Schmidt candidate ran on every rank: %r" % (report,))
    set_transport(best[2])
    if best[2] == "xr" and hasattr(ctx, "set"):
        ctx.set("xr_timeout_ms", int(float(os.environ.get("KRYPY_AMD_BENCH_XR_TIMEOUT_S", "15")) * 1e3))
    report["chosen"] = {"ortho": best[1], "transport": best[2]}
    return best[1], best[2], state["xr_on"], report


def _rank_table(dist, value):
    """one double per rank, in rank order, on every rank"""
    import struct
    if dist is None:
        return [float(value)]
    return [struct.unpack("<d", x)[0] for x in dist.allgather_bytes(struct.pack("<d", float(value)))]


def _spmv_us(ctx, opnd, n, reps=20):
    """this rank's sharded (or plain) SpMV, microseconds per product by HIP events on the library's stream"""
    X, Y = ctx.alloc(n, 1), ctx.alloc(n, 1)
    X.upload(0, np.ones(n))
    dm = opnd._device_matrix()
    for _ in range(3):
        ctx.apply(dm, X, 0, Y, 0, 1)
    ctx.timer_start()
    for _ in range(reps):
        ctx.apply(dm, X, 0, Y, 0, 1)
    return ctx.timer_stop() / reps * 1e3


def _run():
    args = parse_args()
    if args.config == 5:
        return _run_config5(args)
    if args.config in (3, 4):
        if args.gpus != 1 or args.force_sharded:
            raise SystemExit("bench.py --config %d: a one-GPU configuration (BASELINE.json: \"1 MI355X\")" % args.config)
        return _run_config3(args) if args.config == 3 else _run_config4(args)
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

    # ---- N > 1 ranks, --ortho auto: one untimed cycle per (Gram-Schmidt form, transport) candidate, the max over the ranks
    # decides, on every rank alike (_probe_sharded).  All forms pass the 1e-10 parity tests. ----
    ortho_timed = [ortho]
    auto_report = None
    transport_used = None if not sharded else ("xr" if xr_on else "rccl")
    if auto_sharded:
        def one_cycle(cand):
            return len(run_cycles(1, None, ortho=cand).resnorms) - 1
        ortho, transport_used, xr_on, auto_report = _probe_sharded(ctx, dist, args, A_for_ls, xr_on, one_cycle)
        ortho_timed[0] = ortho
        # (the per-form keys round 5's line carried, for whoever compares the two)
        for c in auto_report["candidates"]:
            e = auto_report.setdefault(c["ortho"], {})
            if "ms" in c and c["ms"] < e.get("ms_per_cycle", 1e30):
                e["ms_per_cycle"] = c["ms"]
            for key in ("error", "disabled"):
                if key in c:
                    e[key] = c[key]
        for form in ("cgs", "mgs"):
            if "disabled" in auto_report.get(form, {}) and not any(
                    c["ortho"] == form and "ms" in c for c in auto_report["candidates"]):
                auto_report[form].pop("ms_per_cycle", None)
        auto_report["chosen"] = ortho
        auto_report["chosen_transport"] = transport_used

    fault_once = [os.environ.get("KRYPY_AMD_BENCH_FAULT", "0") == "1" and sharded and hasattr(ctx, "set")]

    def timed_region():
        x0_ = None
        if args.warmup > 0:
            x0_ = run_cycles(args.warmup, None).__dict__["_xk_dev"]
        if fault_once[0]:
            # test hook (tests/test_gpu_multirank.py): the next launch of a kernel with in-launch sums behaves like one whose sum
            # timed out - on a communicator that is KH_ERR_COMM at the step's end, i.e. a failure INSIDE the timed region
            fault_once[0] = False
            ctx.set("chain_fault", 1)
        barrier()
        del cycle_marks[:]
        t0_ = time.perf_counter()
        sol_ = run_cycles(args.steps, x0_)
        ctx.sync()
        return sol_, x0_, t0_, time.perf_counter() - t0_

    # A failure INSIDE the timed region (a sum over the mailboxes that times out: KH_ERR_COMM after KRYPY_AMD_BENCH_XR_TIMEOUT_S,
    # 15 s) must not cost the run its line where there is something to go back to: every rank reports whether its
    # region completed - the same host collective on every rank - and if one did not, all of them switch to RCCL and the panel
    # form together and the region is timed again, once.  The line then says so (`timed_region_fallback`).
    region_fallback = None
    ok, err = 1.0, None
    try:
        sol, x0, t0, dt = timed_region()
    except _hip.BackendError as exc:
        if not sharded:
            raise
        ok, err = 0.0, str(exc)[:300]
    if sharded and dist.allreduce_min(ok) < 1.0:
        if args.transport != "rccl" or (not xr_on and ortho_timed[0] == "cgs"):
            raise SystemExit("bench.py: the timed region failed on some rank (%s) and there is no other path to go back to" % (err,))
        region_fallback = {"reason": err or "another rank's timed region failed",
                           "from": {"ortho": ortho_timed[0], "transport": transport_used}}
        if hasattr(ctx, "set"):
            ctx.set("chain_blk2", 0)
            ctx.set("chain_xr", 0)
        if hasattr(A_for_ls, "halo_via"):
            A_for_ls.halo_via("rccl")
        if xr_on:
            ctx.set("xr", 0)
            ctx.xr_detach()
            xr_on = False
        ortho = ortho_timed[0] = "cgs"
        transport_used = "rccl"
        sol, x0, t0, dt = timed_region()
    cycle_ms = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + cycle_marks[:-1], cycle_marks)]
    if dist is not None:
        dt = dist.allreduce_max(dt)
    n_iters = len(sol.resnorms) - 1
    assert n_iters == args.steps * m, (n_iters, args.steps, m)
    its = n_iters / dt

    # ---- what a first run on real links needs to be read: per-rank SpMV time, sums per iteration and what carried them ----
    shard_diag = None
    if sharded:
        keys = ("n_allreduce", "n_xr", "n_halo_exchange", "n_halo_xh", "n_chain_blk2", "n_chain_xr", "n_lowsync")
        try:
            c0 = dict((k, ctx.get(k)) for k in keys) if hasattr(ctx, "get") else {}
            barrier()
            one = run_cycles(1, x0)
            ctx.sync()
            per = float(max(len(one.resnorms) - 1, 1))
            shard_diag = {"per_iteration": dict((k, (ctx.get(k) - c0[k]) / per) for k in c0)}
            mine = _spmv_us(ctx, A_for_ls, ls.N) if hasattr(A_for_ls, "_device_matrix") else -1.0
        except _hip.BackendError as exc:
            shard_diag = {"error": str(exc)[:300]}
            mine = -1.0
        shard_diag["spmv_us_per_rank"] = _rank_table(dist, mine)
        shard_diag["rows_per_rank"] = _rank_table(dist, ls.N)

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
                   # CSR in, DIA streamed: the operator arrives as SciPy's CSR arrays and is uploaded as they are; the
                   # banded structure is detected at upload and the solver's steps stream a diagonal-major copy built on the device
                   # (8 B per slot, no index stream; bit-identical products).  The CSR kernel north_star names is timed beside it:
                   # `spmv` (stand-alone) and `other_modes["mgs, general CSR SpMV kernel + chain kernel ..."]` (the whole run).
                   "operator_format": (("CSR in (int32 indptr / indices, fp64 data, as SciPy holds them); the timed iterations stream the "
                                        "diagonal-major copy of the %d diagonals built at upload" % ls.A._device_matrix().diagonals)
                                       if (hasattr(ls.A, "_device_matrix") and getattr(ls.A._device_matrix(), "diagonals", 0)) else
                                       "CSR in, CSR streamed (k_spmv_stream)"),
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
                   # a failure inside the timed region that every rank recovered from together (RCCL + the panel form), or null
                   "timed_region_fallback": region_fallback,
                   # N > 1 diagnostics (one extra untimed cycle): collectives per iteration by kind and which kernels served the
                   # steps, every rank's SpMV time (HIP events) and slab length
                   "sharded_diagnostics": shard_diag,
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
    if not sharded and os.environ.get("KRYPY_AMD_BENCH_SECONDARY", "1") != "0" and hasattr(ctx, "get"):
        # one secondary configuration driver-timed in every round: BASELINE.json configs[2], MINRES + Jacobi
        # on this very matrix, 2 x 200 iterations after one warm-up solve (< 2 s); `bench.py --config 3` is the full leg
        try:
            its3, dt3, n3, det3 = minres_jacobi_leg(ctx, A_for_ls, b, 200, 2, 1)
            out["secondary"] = {"config3_minres_jacobi": {"iterations_per_s": its3, "iterations_timed": n3, "ms_per_iteration": dt3 / n3 * 1e3,
                                                          "device_ms_per_iteration_hip_events": det3["device_ms_hip_events"] / n3,
                                                          "lanczos_fused_launches": det3["lanczos_fused_launches"],
                                                          "note": "BASELINE.json configs[2] on the timed run's matrix and right-hand side; "
                                                                  "`python bench.py --config 3` carries its roofline and CPU baseline"}}
        except Exception as exc:
            out["secondary"] = {"config3_minres_jacobi": {"error": repr(exc)[:300]}}
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(A_for_ls, b, m, args.cpu_budget_s)
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}
    return out, rank, dist


def _stamped_traffic(key, n=None):
    """HBM-side bytes per launch of kernel `key` from a profiles/*_traffic.json of THIS source tree (tools/profile_config.sh
    writes them from separate --pmc FETCH_SIZE / WRITE_SIZE passes; 2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction).
    None unless a file carries the stamp of the kernel sources that have just been timed."""
    import glob
    from krypy_amd import _bench
    stamp = _bench.source_stamp()
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(fn))
            if tj.get("source_stamp") != stamp or key not in tj:
                continue
            if n is not None and int(tj.get("n", n)) != int(n):
                continue
            e = tj[key]
            return e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"], os.path.basename(fn)
        except Exception:
            continue
    return None, None


def _roof(kernel, bytes_per_launch, avg_launch_ms, traffic_key, n, bytes_source, note):
    """The `roofline` object of a secondary configuration: algorithmic bytes per launch / the HIP-event launch time."""
    from krypy_amd import _bench
    ach = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    traffic, tfile = _stamped_traffic(traffic_key, n)
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": traffic, "avg_launch_ms": avg_launch_ms, "bytes_per_launch": bytes_per_launch, "bytes_source": bytes_source,
         "source_stamp": _bench.source_stamp(), "note": note}
    if traffic is not None:
        r["traffic_over_bytes"] = traffic / bytes_per_launch
        r["traffic_source"] = "rocprofv3 PMC, separate passes: 2 x FETCH_SIZE + WRITE_SIZE, " + tfile
    else:
        r["traffic_source"] = "null: no profiles/*_traffic.json carries the stamp of these kernel sources for this kernel"
    return r


def minres_jacobi_leg(ctx, A, b, iters, steps, warmup):
    """BASELINE.json configs[2] on an operator already built: MINRES (ortho='lanczos') with the Jacobi preconditioner
    M = diag(A)^-1 (linsys.py:791-853), `steps` solves of `iters` iterations from x0 = 0 (the tolerance is not reached), after
    `warmup` untimed ones.  Returns (iterations/s by the host clock, dict of details, roofline inputs)."""
    import scipy.sparse as sp
    from krypy_amd import linsys, utils
    d = A.diagonal()
    ls = linsys.LinearSystem(A, b, M=sp.diags(1.0 / d).tocsr(), Minv=sp.diags(d).tocsr(), self_adjoint=True)

    def solve():
        try:
            return linsys.Minres(ls, ortho="lanczos", tol=1e-12, maxiter=iters)
        except utils.ConvergenceError as e:
            return e.solver
    for _ in range(max(warmup, 1)):
        solve()
    ctx.sync()
    f0, r0 = ctx.get("n_lanczos_fused"), ctx.get("n_minres_rides")
    ctx.timer_start()
    t0 = time.perf_counter()
    n_it, sol = 0, None
    for _ in range(steps):
        sol = None          # (the previous solver's blocks go back to the pool BEFORE the next one allocates: no second set)
        sol = solve()
        n_it += len(sol.resnorms) - 1
    ctx.sync()
    dt = time.perf_counter() - t0
    ev_ms = ctx.timer_stop()
    fused, rides = ctx.get("n_lanczos_fused") - f0, ctx.get("n_minres_rides") - r0
    return n_it / dt, dt, n_it, {"final_relres": float(sol.resnorms[-1]), "lanczos_fused_launches": fused,
                                 "minres_updates_carried": rides, "device_ms_hip_events": ev_ms}


def _run_config3(args):
    """BASELINE.json configs[2]: the 2-D 5-pt Laplacian of config 2 (N = 10^7), MINRES + Jacobi M, one MI355X."""
    os.environ.setdefault("KRYPY_AMD_DEVICE", "0")
    from krypy_amd import _hip
    if not os.path.exists(_hip.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    ctx = _hip.get_context()
    nx, ny = args.nx, args.ny
    N = nx * ny
    A = laplace2d(nx, ny)
    b = np.random.default_rng(0).standard_normal(N)
    its, dt, n_it, det = minres_jacobi_leg(ctx, A, b, args.iters, args.steps, args.warmup)
    assert n_it == args.steps * args.iters, (n_it, args.steps, args.iters)
    fused = det["lanczos_fused_launches"] >= n_it - args.steps
    # every datum once (lanczos.h): 5 diagonals 40 N + v_k 8 N + p_{k-1} 8 N | p_k 8 N + D 8 N | two stores 16 N = 88 N, and the
    # MINRES recurrences of iteration k - 2 riding along: v, W0, W1, yk in, z and yk out = 48 N
    bpl = (88.0 if fused else 104.0) * N + 48.0 * N
    launches = max(det["lanczos_fused_launches"], 1) if fused else n_it
    roof = _roof("k_lanczos_fused<40,5,JAC,MR>: one Lanczos step in three passes, the MINRES recurrences of iteration k-2 riding along"
                 if fused else "per-kernel Lanczos + MINRES update path",
                 bpl, det["device_ms_hip_events"] / launches, "k_lanczos_fused", N,
                 "algorithmic bytes of one iteration, every datum once: 88 N (Lanczos launch: 5 diagonals of the banded copy, "
                 "v_k, p_{k-1}, p_k, D, two stores) + 48 N (MINRES recurrences)",
                 "avg_launch_ms = HIP-event time of the timed region (kh_timer_*, the library's stream) / launches of the fused "
                 "kernel in it - the few other launches of a solve (initial residual, the final explicit residual) are inside, "
                 "so this is an upper bound on the kernel's own duration; profiles/r06_config3.md has the kernel trace")
    out = {
        "metric": "MINRES+Jacobi iterations/sec, n=10^7 5-pt Laplacian fp64",
        "value": its, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "MINRES (ortho=lanczos) + Jacobi M = diag(A)^-1, %d iterations per solve from x0 = 0, 2-D 5-pt "
                               "Laplacian %dx%d CSR (N=%d, nnz=%d), b=rng(0) normal (BASELINE.json configs[2])"
                               % (args.iters, nx, ny, N, A.nnz),
                   "n": N, "iterations_timed": n_it, "parallelism": "1 GPU", "ranks": 1, **det},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = _cpu_baseline_minres(A, b, min(args.cpu_budget_s, 40.0))
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}
    return out, 0, None


def _cpu_baseline_minres(A, b, budget_s):
    """The CPU oracle's MINRES + Jacobi (oracle/krylov_ref.py: minres) on the same A and b: as many iterations as fit the
    budget (an iteration costs the same whatever k: one SpMV + about twelve vector passes)."""
    import scipy.sparse as sp
    from oracle import krylov_ref as ref
    d = A.diagonal()
    M = sp.diags(1.0 / d).tocsr()
    t0 = time.perf_counter()
    ref.minres(A, b, tol=1e-30, maxiter=3, M=M)          # warm-up and a first estimate
    per = (time.perf_counter() - t0) / 3.0
    iters = int(max(5, min(200, budget_s / max(per, 1e-3))))
    t0 = time.perf_counter()
    res = ref.minres(A, b, tol=1e-30, maxiter=iters, M=M)
    dt = time.perf_counter() - t0
    done = len(res["resnorms"]) - 1
    return {"value": done / dt, "unit": "iterations/s", "cores": _blas_threads() or 1, "kind": "port",
            "sample": "%d MINRES + Jacobi iterations at N=%d with the NumPy/SciPy oracle (oracle/krylov_ref.py: minres), same A and b, "
                      "%.1f s, set-up and the final explicit residual included; BLAS threads as NumPy picks them (%s), SciPy's "
                      "csr_matvec is single-threaded" % (done, A.shape[0], dt, _blas_threads())}


def _run_config4(args):
    """BASELINE.json configs[3]: dense random SPD n = 32768 fp64, CG, the product A p as the row-per-wave GEMV kernel (the panel
    A = G G^T is formed ON THE DEVICE by the FP64 MFMA kernel, k_gemm_dense_mfma, through kh_apply's panel path - the host never
    multiplies).  A step = one whole solve to tol 1e-8 from x0 = 0 (linsys.py:593-689)."""
    os.environ.setdefault("KRYPY_AMD_DEVICE", "0")
    from krypy_amd import _hip, linsys, utils
    if not os.path.exists(_hip.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    ctx = _hip.get_context()
    n = args.dense_n
    rng = np.random.default_rng(0)
    t_setup = time.perf_counter()
    G = rng.standard_normal((n, n))                 # the same stream as oracle/inputs.py: dense_spd_system (G first, then b)
    b = rng.standard_normal(n)
    # Y = G G^T: G as a dense operator applied to the block whose columns are the rows of G; the columns of the symmetric
    # Y are the rows of the row-major operator A = Y / n + I (kh_dense_from_block) - all on the device
    Gop = ctx.dense(G)
    Gt = ctx.alloc(n, n, zero=False)
    step = 2048
    for c0 in range(0, n, step):
        Gt.upload(c0, np.ascontiguousarray(G[c0:c0 + step].T))
    del G
    Y = ctx.alloc(n, n, zero=False)
    ctx.sync()
    t_mm = time.perf_counter()
    ctx.apply(Gop, Gt, 0, Y, 0, n)
    ctx.sync()
    t_mm = time.perf_counter() - t_mm
    del Gop, Gt
    Aop = ctx.dense_from_block(Y, 0, n, 1.0 / n, 1.0)
    del Y
    import gc
    gc.collect()
    ctx._pool_flush()
    t_setup = time.perf_counter() - t_setup
    A = utils.DeviceOperator(Aop) if hasattr(utils, "DeviceOperator") else None
    if A is None:
        raise SystemExit("bench.py --config 4: krypy_amd.utils.DeviceOperator missing")
    ls = linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True)

    def solve():
        return linsys.Cg(ls, tol=1e-8, maxiter=200)
    for _ in range(max(args.warmup, 1)):
        solve()
    ctx.sync()
    t0 = time.perf_counter()
    n_it = 0
    for _ in range(args.steps):
        sol = solve()
        n_it += sol.iter
    ctx.sync()
    dt = time.perf_counter() - t0
    its = n_it / dt
    # the dominant kernel, timed by itself with HIP events on the library's stream: the GEMV of one CG iteration
    X, Yv = ctx.upload(b), ctx.alloc(n, 1)
    for _ in range(3):
        ctx.apply(Aop, X, 0, Yv, 0, 1)
    reps = 30
    ctx.timer_start()
    for _ in range(reps):
        ctx.apply(Aop, X, 0, Yv, 0, 1)
    gemv_ms = ctx.timer_stop() / reps
    roof = _roof("k_gemv_dense (row-major GEMV, four rows per wave: one launch per CG iteration)", 8.0 * n * n + 16.0 * n, gemv_ms,
                 "k_gemv_dense", n, "the matrix once (8 n^2) + x in, y out (16 n)",
                 "avg_launch_ms = %d back-to-back launches between two HIP events on the library's stream (kh_timer_*); a CG "
                 "iteration is this launch + two small fused vector launches (linsys.py:655-665)" % reps)
    out = {
        "metric": "CG iterations/sec, dense random SPD n=32768 fp64",
        "value": its, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "CG to tol 1e-8 from x0 = 0 on the dense SPD matrix A = G G^T / n + I, G = rng(0) normal (n=%d), "
                               "b = the same stream's next n normals (BASELINE.json configs[3]); a step = one whole solve" % n,
                   "n": n, "iterations_timed": n_it, "iterations_per_solve": sol.iter, "final_relres": float(sol.resnorms[-1]),
                   "parallelism": "1 GPU", "ranks": 1,
                   "setup_s": t_setup, "gg_t_on_device_s": t_mm,
                   "gg_t_tflops": 2.0 * n * n * n / t_mm / 1e12,
                   "setup_note": "A formed on the device: G uploaded once as an operator and once as a block, Y = G G^T by "
                                 "k_gemm_dense_mfma (16 columns per launch, FP64 MFMA), A = Y^T / n + I by kh_dense_from_block; "
                                 "sums taken in the MFMA kernel's order, i.e. A equals the NumPy matrix to rounding"},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = _cpu_baseline_cg(n, min(args.cpu_budget_s, 40.0))
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}
    return out, 0, None


def _cpu_baseline_cg(n, budget_s):
    """The CPU oracle's CG (oracle/krylov_ref.py: cg) on a BOUNDED sample: the same construction at n_s = 8192 (the full
    matrix costs a 70-TFLOP host dgemm before the first iteration), whole solves; an iteration is one dense GEMV, i.e.
    n^2 flops and 8 n^2 bytes, so the rate at n is the measured one times (n_s / n)^2."""
    from oracle import krylov_ref as ref
    from oracle.inputs import dense_spd_system
    ns = min(n, 8192)
    A, b = dense_spd_system(ns)
    ref.cg(A, b, tol=1e-8, maxiter=200)
    t0 = time.perf_counter()
    done, solves = 0, 0
    while time.perf_counter() - t0 < budget_s / 4.0 or solves == 0:
        res = ref.cg(A, b, tol=1e-8, maxiter=200)
        done += len(res["resnorms"]) - 1
        solves += 1
    dt = time.perf_counter() - t0
    scale = (float(ns) / n) ** 2
    return {"value": done / dt * scale, "unit": "iterations/s", "cores": _blas_threads() or 1, "kind": "port",
            "measured_at_sample_size": done / dt,
            "sample": "%d whole CG solves (%d iterations, %.1f s) of the NumPy oracle (oracle/krylov_ref.py: cg) on the same "
                      "construction at n = %d; `value` = that rate x (%d / %d)^2 (an iteration is one dense GEMV: 8 n^2 bytes), "
                      "BLAS threads as NumPy picks them (%s)" % (solves, done, dt, ns, ns, n, _blas_threads())}


def _run_config5(args):
    """BASELINE.json configs[4]: 3-D 7-point Laplacian on 500 x 500 x 400 points (N = 10^8, nnz = 698,700,000), rows in
    z-slabs over the ranks (400 / 8 = 50 planes = 12.5 M rows per GPU), b = rng(0) normal.  Solve 1: plain GMRES(m)
    (DeflatedGmres without U) to harvest the `defl` smallest-magnitude Ritz vectors ON THE DEVICE (every rank keeps its
    slab of them; the small eigenproblem is replicated host work); solve 2: DeflatedGmres(U, maxiter=m), timed - a
    *step* is one such solve of m iterations from x0 = 0 (reference flow: recycling/linsys.py:51-103,
    deflation.py:93-163).  Same JSON contract as config 2; `scaling` is "strong" (the grid is fixed).  On one GPU the whole
    problem (N = 10^8: the basis 80.8 GB, the projector's four bases 51 GB, the operator 14 GB of the 288 GB) runs as it is."""
    import gc
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
    auto = ortho == "auto"
    if auto:
        ortho = "cgs" if sharded else "mgs"
    b_rng = np.random.default_rng(0)
    xr_on = False
    t_setup = time.perf_counter()
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
        if args.loop_halo:
            if not (args.force_sharded and world == 1):
                raise SystemExit("bench.py --loop-halo: with --force-sharded on one rank")
            Aloc = laplace3d(nx, ny, 3 * nz, nz, 2 * nz)
            op = kdist.ShardedCSROperator(Aloc, nz * plane, 3 * N, ctx, self_loop=True)
        else:
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
    t_setup = time.perf_counter() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    mode = [ortho]

    def solve(U, mode_=None):
        try:
            return deflation.DeflatedGmres(ls, U=U, tol=1e-12, maxiter=m, ortho=mode_ or mode[0], store_arnoldi=U is None)
        except utils.ConvergenceError as e:
            return e.solver

    # solve 1 (untimed): harvest the Ritz vectors on the device.  (A solver and the operators it builds refer to each other:
    # its 80 GB basis is garbage only to the cycle collector - collect before the next one is allocated.)
    s0 = solve(None)
    ctx.sync()
    t_ritz = time.perf_counter()
    ritz = deflation.Ritz(s0)
    U = ritz._get_vectors_dev(np.argsort(np.abs(ritz.values))[:d])
    ctx.sync()
    t_ritz = time.perf_counter() - t_ritz
    plain_relres = float(s0.resnorms[-1])
    ritz_values = np.sort(np.abs(ritz.values))[:d]
    del s0, ritz
    gc.collect()

    def one_solve(cand):
        s_ = solve(U, cand)
        n_ = len(s_.resnorms) - 1
        del s_
        gc.collect()
        return n_

    # --ortho auto: one untimed deflated solve per candidate, the faster one is timed (N ranks: per transport as well, the max
    # over the ranks decides on every rank alike; a candidate that fails over the mailboxes sends every rank back to RCCL)
    auto_report = None
    transport_used = None if not sharded else ("xr" if xr_on else "rccl")
    if auto and sharded:
        ortho, transport_used, xr_on, auto_report = _probe_sharded(ctx, dist, args, A_for_ls, xr_on, one_solve)
        mode[0] = ortho
    elif auto:
        auto_report = {"candidates": []}
        best = None
        for cand in ("mgs", "cgs"):
            ctx.sync()
            t1 = time.perf_counter()
            one_solve(cand)
            ctx.sync()
            d1 = time.perf_counter() - t1
            auto_report["candidates"].append({"ortho": cand, "ms": d1 * 1e3})
            if best is None or d1 < best[0]:
                best = (d1, cand)
        ortho = mode[0] = best[1]
        auto_report["chosen"] = {"ortho": ortho}

    gc_ms = []

    def timed_region():
        for _ in range(args.warmup):
            one_solve(None)
        # everything alive now (modules, the operator, U) stays: out of the collector's way, so that the collection after each
        # solve - which is what hands the solver's basis back to the block pool - walks the solve's own objects only
        gc.collect()
        gc.freeze()
        barrier()
        marks = []
        t0_ = time.perf_counter()
        n_, last = 0, None
        for _ in range(args.steps):
            s1 = solve(U)
            n_ += len(s1.resnorms) - 1
            last = float(s1.resnorms[-1])
            del s1
            tg = time.perf_counter()
            gc.collect()
            gc_ms.append((time.perf_counter() - tg) * 1e3)
            marks.append(time.perf_counter())
        ctx.sync()
        return n_, last, t0_, marks, time.perf_counter() - t0_

    region_fallback = None
    ok, err = 1.0, None
    try:
        n_iters, deflated_relres, t0, cycle_marks, dt = timed_region()
    except _hip.BackendError as exc:
        if not sharded:
            raise
        ok, err = 0.0, str(exc)[:300]
    if sharded and dist.allreduce_min(ok) < 1.0:
        if args.transport != "rccl" or (not xr_on and mode[0] == "cgs"):
            raise SystemExit("bench.py: the timed region failed on some rank (%s) and there is no other path to go back to" % (err,))
        region_fallback = {"reason": err or "another rank's timed region failed", "from": {"ortho": mode[0], "transport": transport_used}}
        if hasattr(ctx, "set"):
            ctx.set("chain_blk2", 0)
            ctx.set("chain_xr", 0)
        if hasattr(A_for_ls, "halo_via"):
            A_for_ls.halo_via("rccl")
        if xr_on:
            ctx.set("xr", 0)
            ctx.xr_detach()
            xr_on = False
        ortho = mode[0] = "cgs"
        transport_used = "rccl"
        n_iters, deflated_relres, t0, cycle_marks, dt = timed_region()
    cycle_ms = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + cycle_marks[:-1], cycle_marks)]
    if dist is not None:
        dt = dist.allreduce_max(dt)
    assert n_iters == args.steps * m, (n_iters, args.steps, m)
    its = n_iters / dt
    nloc = ls.N

    shard_diag = None
    if sharded:
        keys = ("n_allreduce", "n_xr", "n_halo_exchange", "n_halo_xh", "n_chain_blk2", "n_chain_xr", "n_lowsync")
        try:
            c0 = dict((k, ctx.get(k)) for k in keys) if hasattr(ctx, "get") else {}
            barrier()
            per = float(max(one_solve(None), 1))
            shard_diag = {"per_iteration": dict((k, (ctx.get(k) - c0[k]) / per) for k in c0)}
            mine = _spmv_us(ctx, A_for_ls, nloc) if hasattr(ctx, "get") else -1.0
        except _hip.BackendError as exc:
            shard_diag = {"error": str(exc)[:300]}
            mine = -1.0
        shard_diag["spmv_us_per_rank"] = _rank_table(dist, mine)
        shard_diag["rows_per_rank"] = _rank_table(dist, nloc)

    # ---- per-kernel rooflines on this rank's slab (one rank; N ranks keep the whole-iteration average below: the
    # micro-launches have no cross-rank stage) ----
    # bytes one deflated iteration has to move at least on this rank: the operator (banded copy when there is one), w out
    # and in again around the projector, the projector's two sweeps over its 2 d columns, every Gram-Schmidt column once
    # per use of the form that ran, v_{k+1} out
    dm = ls.A._device_matrix() if hasattr(ls.A, "_device_matrix") else None
    nd = int(getattr(dm, "diagonals", 0) or 0)
    op_bytes = (8.0 * nd * nloc + 16.0 * nloc) if nd else (12.0 * nnz_global / world + 4.0 * (nloc + 1) + 16.0 * nloc)
    uses = 1.0 if (ortho == "mgs" and nloc <= 14680064) else 2.0          # register-resident chain: a column is read once
    it_bytes = op_bytes + 32.0 * nloc + 32.0 * d * nloc + uses * 8.0 * nloc * (m + 1) / 2.0 + 8.0 * nloc
    roof = {"bound": "hbm", "kernel": "whole deflated iteration (all kernels)", "achieved": it_bytes * its / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": it_bytes * its / 1e9 / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": it_bytes,
            "bytes_source": "algorithmic bytes of one deflated iteration on one rank, every datum once per use: operator %s, the "
                            "projector's two sweeps over 2 x %d columns (32 d N), %s, w out / in around the projector, v_{k+1} out"
                            % ("as its banded copy (8 nd N + 16 N)" if nd else "as CSR", d,
                               "the Gram-Schmidt columns once (register-resident chain)" if uses == 1.0 else
                               "the Gram-Schmidt columns twice (dots pass + update pass: w does not fit the register file, or the panel form)")}
    extra = {}
    if world == 1 and not args.no_roofline and hasattr(ctx, "bench_kernel"):
        try:
            from krypy_amd import _bench
            gc.collect()
            kroof, kextra = _bench.roofline(ctx, ls, ortho, HBM_PEAK_GBS, reps=20, m=m, solver_steps=False)
            kernels = kextra.get("kernels", {})
            # the projector, as the solver applies it (d columns, two sweeps): HIP events over 10 applications
            pj_ms, pj_bytes, pj_name = _bench.projector_probe(ctx, nloc, d)
            kernels[pj_name] = {"avg_ms": pj_ms, "compulsory_bytes": pj_bytes, "compulsory_gbs": pj_bytes / pj_ms / 1e6,
                                "frac_compulsory": pj_bytes / pj_ms / 1e6 / HBM_PEAK_GBS}
            # shares of an iteration: Gram-Schmidt ((m + 1) / 2 links or columns on average), projector, operator
            gs_ms = kroof["avg_launch_ms"] * kroof.get("links_or_columns_per_iteration", (m + 1) / 2.0) / kroof.get("links_or_columns_per_launch", 1.0)
            sp = kextra.get("spmv", {})
            sp_ms = min([v["avg_ms"] for v in sp.values()] or [0.0]) if nd else max([v["avg_ms"] for v in sp.values()] or [0.0])
            share = {"gram_schmidt_ms": gs_ms, "projector_ms": pj_ms, "operator_ms": sp_ms, "measured_iteration_ms": dt / n_iters * 1e3}
            whole = roof
            roof = dict(kroof)
            if pj_ms > gs_ms:           # the projector dominates (short restart lengths): its roofline is the line's
                roof.update(kernel=pj_name, achieved=pj_bytes / pj_ms / 1e6, frac=pj_bytes / pj_ms / 1e6 / HBM_PEAK_GBS,
                            avg_launch_ms=pj_ms, bytes_per_launch=pj_bytes, traffic=None)
            t_key = "k_proj" if pj_ms > gs_ms else kroof.get("traffic_key")
            if t_key:
                tr, tf = _stamped_traffic(t_key, nloc)
                if tr is not None:
                    roof["traffic"], roof["traffic_over_bytes"], roof["traffic_source"] = tr, tr / roof["bytes_per_launch"], "rocprofv3 PMC passes, " + tf
            roof["iteration_shares_ms"] = share
            extra = {"kernels": kernels, "whole_iteration": whole, "attainable": kextra.get("attainable"), "spmv": sp}
        except Exception as exc:
            extra = {"roofline_error": repr(exc)[:400]}
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
                   "cross_rank_sums": None if not sharded else ("xr" if xr_on else "rccl"),
                   "halo": None if not sharded else (("in-launch" if getattr(A_for_ls, "halo_in_launch", False) else "rccl") +
                                                    (" (the slab is its own neighbour)" if args.loop_halo else "")),
                   "ortho_auto": auto_report, "timed_region_fallback": region_fallback, "sharded_diagnostics": shard_diag,
                   "plain_relres": plain_relres, "deflated_relres": deflated_relres,
                   "smallest_ritz_values": [float(v) for v in ritz_values[:4]],
                   "ritz_harvest_ms": t_ritz * 1e3,      # deflation.Ritz + the vectors [V_n, U] @ coeffs (deflation.py:738-847), untimed

                   "operator_diagonals": nd, "setup_s": t_setup,
                   "cycle_ms": cycle_ms, "median_cycle_ms": float(np.median(cycle_ms)),
                   "host_gc_ms_per_solve": [round(g, 2) for g in gc_ms]},
        "roofline": roof,
    }
    out.update(extra)
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        out["cpu_baseline"] = {"value": None, "note": "config 2 carries the CPU baseline (bench.py without --config)"}
    return out, rank, dist


if __name__ == "__main__":
    main()
