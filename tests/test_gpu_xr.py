"""The xr transport (krypy_amd/csrc/xr.hip): sums across the ranks of one node through IPC-mapped mailboxes instead of
ncclAllReduce kernels (the inner products that are summed: /root/reference/krypy/utils.py:182-183).

What can run on a box with ONE GPU runs here: (i) one rank in loopback - the exchange kernels, the fused
reduce-and-exchange kernel, tags and parities - against the RCCL path of the same context, bit for bit; (ii) TWO PROCESSES
on the one device, joined through hipIpcGetMemHandle / hipIpcOpenMemHandle with no RCCL communicator at all, summing panels
of every size class and running whole solves of a block-diagonal system against one process that solves the whole
system; (iii) a peer that does not arrive: an error after the timeout, never a hang and never a fallback.  With two or
more GPUs on the box the same two processes run on two devices - the first exchange over a real xGMI link.  The latency
of a system-scope store between two GPUs is unmeasured on today's boxes."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref
from tests.support.kernel_expect import expect_kernel

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("KRYPY_AMD_XR", "1") == "0",
                                 reason="KRYPY_AMD_XR=0: the transport these tests are about is switched off (the RCCL path is what every "
                                        "other multi-rank test of the suite runs)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def forced_ctx(hip):
    """A context with a 1-rank RCCL communicator in forced multi-rank mode (every inner product is all-reduced)."""
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


def test_one_rank_loopback_equals_the_rccl_path(forced_ctx):
    """One rank is its own only peer: every sum still runs through the mailbox kernels (publish to 'every' rank, poll,
    add in rank order) and must leave exactly the value RCCL's 1-rank all-reduce leaves - whole solves bit for bit,
    with the fused reduce-and-exchange kernel doing the Gram-Schmidt panels and the norms."""
    from krypy_amd import dist as kdist, linsys, utils

    ctx = forced_ctx
    ctx.set("chain_blk2", 0)      # (with the transport on 'mgs' would take the blocked kernel with the sums inside the launch -
    #                                not a bit-for-bit kernel, tests/test_gpu_blk2.py; here the exchange KERNELS are compared with RCCL)
    A = ref.laplace2d(300, 260)
    b = np.random.default_rng(3).standard_normal(A.shape[0])

    def solves():
        out = []
        for ortho in ("mgs", "cgs", "cgs2"):
            try:
                s = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=25, max_restarts=3, tol=1e-12, ortho=ortho)
            except utils.ConvergenceError as e:
                s = e.solver
            out.append(np.array(s.resnorms))
        try:
            m = linsys.Minres(linsys.LinearSystem(A, b, self_adjoint=True), tol=1e-10, maxiter=150)
        except utils.ConvergenceError as e:
            m = e.solver
        out.append(np.array(m.resnorms))
        return out

    a0 = ctx.get("n_allreduce")
    want = solves()
    n_rccl = ctx.get("n_allreduce") - a0
    rdv = kdist.TcpRendezvous(0, 1)
    assert kdist.enable_xr(ctx, rdv) is True
    x0, f0, a0 = ctx.get("n_xr"), ctx.get("n_xr_fused"), ctx.get("n_allreduce")
    got = solves()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    for count in (1, 5, 512, 513, 3000):
        v = np.random.default_rng(count).standard_normal(count)
        assert np.array_equal(ctx.allreduce_host(v.copy()), v)
    expect_kernel(ctx.get("n_allreduce") - a0 == n_rccl + 5, "the same number of cross-rank sums as through RCCL")
    expect_kernel(ctx.get("n_xr") - x0 >= n_rccl and ctx.get("n_xr_fused") - f0 > 100,
                  "every sum ran as a mailbox kernel, the panels in the fused form: %r" % ((ctx.get("n_xr") - x0, ctx.get("n_xr_fused") - f0, n_rccl),))


def _two_ranks(devices, timeout_test=True):
    out = tempfile.mkdtemp(prefix="xr_")
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(2):
        env = dict((k, v) for k, v in os.environ.items() if k not in ("KRYPY_AMD_FORCE_MULTI", "LOCAL_RANK"))
        env.update(RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   KRYPY_AMD_DEVICE=str(devices[r]), XR_OUT=out, XR_TIMEOUT_TEST="1" if timeout_test else "0",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "support", "xr_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode()[-3000:])
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, logs[r])
    return out


@pytest.mark.parametrize("placement", ["one_gpu", "two_gpus"])
def test_two_processes_sum_through_ipc_mailboxes(hip, placement):
    from krypy_amd import _hip, linsys
    from tests.support.xr_worker import block

    if placement == "two_gpus" and _hip.device_count() < 2:
        pytest.skip("one GPU on this box: both processes ran on it in the other case")
    out = _two_ranks((0, 0) if placement == "one_gpu" else (0, 1))
    r = [np.load(os.path.join(out, "rank%d.npz" % k)) for k in range(2)]
    (A0, b0), (A1, b1) = block(0), block(1)
    A = sp.block_diag([A0, A1]).tocsr()
    b = np.concatenate([b0, b1])
    n0 = A0.shape[0]
    for ortho in ("mgs", "cgs"):
        s = linsys.RestartedGmres(linsys.LinearSystem(A, b), maxiter=30, max_restarts=40, tol=1e-9, ortho=ortho)
        want = np.array(s.resnorms)
        for k in range(2):
            got = r[k]["gmres_%s_resnorms" % ortho]
            assert len(got) == len(want), (ortho, k, len(got), len(want))                 # the same iteration count
            assert np.max(np.abs(got[:31] - want[:31]) / want[:31]) < 1e-10, (ortho, k)       # first cycle: 1e-10
            assert np.max(np.abs(got - want) / want) < 1e-6, (ortho, k)
        assert np.array_equal(r[0]["gmres_%s_resnorms" % ortho], r[1]["gmres_%s_resnorms" % ortho])      # replicated scalars: the same bits
        x = np.concatenate([r[0]["gmres_%s_x" % ortho], r[1]["gmres_%s_x" % ortho]])
        assert np.linalg.norm(A.dot(x) - b) <= 1.0001e-9 * np.linalg.norm(b)
        assert np.linalg.norm(x - s.xk[:, 0]) < 1e-7 * np.linalg.norm(s.xk)
    c = linsys.Cg(linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True), tol=1e-9, maxiter=500)
    want = np.array(c.resnorms)
    assert len(r[0]["cg_resnorms"]) == len(want) and np.array_equal(r[0]["cg_resnorms"], r[1]["cg_resnorms"])
    assert np.max(np.abs(r[0]["cg_resnorms"][:40] - want[:40]) / want[:40]) < 1e-9
    x = np.concatenate([r[0]["cg_x"], r[1]["cg_x"]])
    assert np.linalg.norm(x - c.xk[:, 0]) < 1e-7 * np.linalg.norm(c.xk)
    # the coupled Laplacian in two slabs: halos inside the SpMV launches, sums through the mailboxes, no RCCL anywhere
    Ac = ref.laplace2d(120, 96)
    bc = np.random.default_rng(77).standard_normal(Ac.shape[0])
    coupled = int(r[0]["coupled"]) == 1 and int(r[1]["coupled"]) == 1
    expect_kernel(coupled, "the coupled solves ran (a banded shard whose halo goes into the SpMV's launch)")
    for ortho in (("mgs", "cgs") if coupled else ()):
        s = linsys.RestartedGmres(linsys.LinearSystem(Ac, bc), maxiter=40, max_restarts=40, tol=1e-9, ortho=ortho)
        want = np.array(s.resnorms)
        for k in range(2):
            got = r[k]["coupled_%s_resnorms" % ortho]
            assert len(got) == len(want), (ortho, k, len(got), len(want))
            assert np.max(np.abs(got[:41] - want[:41]) / want[:41]) < 1e-10, (ortho, k)
        assert np.array_equal(r[0]["coupled_%s_resnorms" % ortho], r[1]["coupled_%s_resnorms" % ortho])
        x = np.concatenate([r[0]["coupled_%s_x" % ortho], r[1]["coupled_%s_x" % ortho]])
        assert np.linalg.norm(Ac.dot(x) - bc) <= 1.0001e-9 * np.linalg.norm(bc)
        assert np.linalg.norm(x - s.xk[:, 0]) < 1e-7 * np.linalg.norm(s.xk)
    expect_kernel((not coupled) or int(r[0]["n_halo_xh"]) > 100 and int(r[0]["n_halo_exchange"]) == 0 and int(r[1]["n_halo_exchange"]) == 0,
                  "every halo travelled inside an SpMV launch: %r" % ((int(r[0]["n_halo_xh"]), int(r[0]["n_halo_exchange"])),))
    assert int(r[0]["panels_checked"]) == int(r[1]["panels_checked"]) == 420
    expect_kernel(int(r[0]["n_xr"]) == int(r[1]["n_xr"]) and int(r[0]["n_xr"]) > 1000, "both ranks issued the same exchanges: %r" % ((int(r[0]["n_xr"]), int(r[1]["n_xr"])),))
    expect_kernel(int(r[0]["n_xr_fused"]) > 100, "the panel form took the fused reduce-and-exchange kernel")
    expect_kernel(int(r[0]["n_chain_blk2"]) == int(r[1]["n_chain_blk2"]) and int(r[0]["n_chain_blk2"]) > 30,
                  "the reference order took the blocked kernel with the cross-rank sums inside the launch: %r" % ((int(r[0]["n_chain_blk2"]), int(r[1]["n_chain_blk2"])),))
    # slabs beyond the blocked kernel's range: the chain kernels with the cross-rank stage (24 and 48 rows per lane), two
    # processes' launches exchanging every link's sum - against ONE process solving the block-diagonal system
    from krypy_amd import utils
    for tag, nxx in (("24", 300), ("48", 430)):
        (X0, y0), (X1, y1) = block(0, nxx), block(1, nxx)
        Ax, bx = sp.block_diag([X0, X1]).tocsr(), np.concatenate([y0, y1])
        try:
            s = linsys.RestartedGmres(linsys.LinearSystem(Ax, bx), maxiter=30, max_restarts=5, tol=1e-9, ortho="mgs")
        except utils.ConvergenceError as e:
            s = e.solver
        want = np.array(s.resnorms)
        for k in range(2):
            got = r[k]["chainxr_%s_resnorms" % tag]
            assert len(got) == len(want), (tag, k, len(got), len(want))
            assert np.max(np.abs(got[:31] - want[:31]) / want[:31]) < 1e-10, (tag, k)
            assert np.max(np.abs(got - want) / want) < 1e-6, (tag, k)
        assert np.array_equal(r[0]["chainxr_%s_resnorms" % tag], r[1]["chainxr_%s_resnorms" % tag])
        x = np.concatenate([r[0]["chainxr_%s_x" % tag], r[1]["chainxr_%s_x" % tag]])
        assert np.linalg.norm(x - s.xk[:, 0]) < 1e-7 * np.linalg.norm(s.xk)
        # (one step more than iterations: the look-ahead step begun behind the last one)
        expect_kernel(int(r[0]["chainxr_%s_launches" % tag]) == int(r[1]["chainxr_%s_launches" % tag]) >= len(want) - 1,
                      "every step of the %s-row solves took the chain kernel with the cross-rank stage: %r" % (tag, (int(r[0]["chainxr_%s_launches" % tag]), len(want) - 1)))
    # the peer that did not arrive: rank 0's sum ended in an error that says so
    assert open(os.path.join(out, "rank0.done")).read().strip() == "1"
