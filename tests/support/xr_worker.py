"""One rank of a cross-rank-sum test on REAL devices - TEST INFRASTRUCTURE (tests/test_gpu_xr.py starts two of these).

Two PROCESSES, each with its own HIP context - on one GPU (today's boxes: ``KRYPY_AMD_DEVICE=0`` for both) or on two -
join through the xr transport alone (csrc/xr.hip: IPC-mapped mailboxes, no RCCL communicator: RCCL refuses two ranks on
one device) and run, through the product's own entry points,

1. panels of every size class through ``kh_comm_allreduce_host`` (one lane group, several workgroups, chunked) against
   sums both ranks can work out for themselves - hundreds of them back to back (epochs, both parities);
2. whole solves of a BLOCK-DIAGONAL system (each rank holds one diagonal block: every inner product crosses the ranks,
   no halo does): restarted GMRES in the reference order (`mgs`: one sum per Gram-Schmidt link - the most sums per step
   any path issues), the panel form (`cgs`: the fused reduce-and-exchange kernel) and CG, results written to
   ``$XR_OUT/rank<r>.npz`` for the test to compare with ONE process solving the whole system;
3. the timeout path: rank 1 leaves one sum out, rank 0's wait ends in KH_ERR_COMM after ``KRYPY_AMD_XR_TIMEOUT_S``.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def block(rank, nx=150):
    """Diagonal block `rank` of the test system: a shifted 2-D Laplacian (different spectra on the two ranks)."""
    from oracle import krylov_ref as ref
    import scipy.sparse as sp
    A = (ref.laplace2d(nx, nx - 10 * rank) + sp.identity(nx * (nx - 10 * rank)) * (0.05 + 0.1 * rank)).tocsr()
    b = np.random.default_rng(40 + rank).standard_normal(A.shape[0])
    return A, b


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = os.environ["XR_OUT"]
    from krypy_amd import _hip, dist as kdist, linsys, utils

    rdv = kdist.TcpRendezvous(rank, world)
    ctx = _hip.get_context()
    on = kdist.enable_xr(ctx, rdv)
    assert on, "the xr transport did not come up"
    assert ctx.nranks == world and ctx.rank == rank and ctx.get("xr") == 1
    # 1. panels
    checked = 0
    for rep in range(60):
        for count in (1, 7, 64, 65, 512, 513, 1500):
            def contrib(r, rep=rep, count=count):
                return np.random.default_rng(7919 * rep + 31 * count + r).standard_normal(count)
            want = contrib(0)
            for r in range(1, world):
                want = want + contrib(r)
            got = ctx.allreduce_host(contrib(rank))
            assert np.array_equal(got, want), (rep, count)
            checked += 1
    # 2. solves
    A, b = block(rank)
    # (what dist.ShardedCSROperator announces for a sharded operator: the longest slab of the run - the reference-order step
    # then takes the blocked kernel with the cross-rank sums inside the launch, csrc/chain_blk2.h)
    nmax = int(rdv.allreduce_max(float(A.shape[0])))
    ctx.set("lowsync_rows", (int(A.shape[0]) << 32) | nmax)
    res = {}
    for ortho in ("mgs", "cgs"):
        ls = linsys.LinearSystem(A, b)
        s = linsys.RestartedGmres(ls, maxiter=30, max_restarts=40, tol=1e-9, ortho=ortho)
        res["gmres_%s_resnorms" % ortho] = np.array(s.resnorms)
        res["gmres_%s_x" % ortho] = s.xk[:, 0].copy()
    c = linsys.Cg(linsys.LinearSystem(A, b, self_adjoint=True, positive_definite=True), tol=1e-9, maxiter=500)
    res["cg_resnorms"], res["cg_x"] = np.array(c.resnorms), c.xk[:, 0].copy()
    # 2b. a COUPLED system: one 2-D Laplacian in two slabs - every operator application exchanges a halo, and with no RCCL
    # communicator that halo can only travel inside the banded SpMV's own launch (kh_mat_xh_*: the neighbour's ghost granules)
    from oracle import krylov_ref as ref
    nxc = 120
    Ac = ref.laplace2d(nxc, 96)
    bc = np.random.default_rng(77).standard_normal(Ac.shape[0])
    cuts = kdist.slab_cuts(Ac.shape[0], world, align=nxc)
    r0, r1 = cuts[rank], cuts[rank + 1]
    op = kdist.ShardedCSROperator(Ac[r0:r1], r0, Ac.shape[0], ctx)
    # (no diagonal-major copy of the shard - KRYPY_AMD_SPMV_DIA=0 - or KRYPY_AMD_XH=0: nothing can carry a halo between two ranks
    # that have no RCCL communicator; the coupled solves are left out then and the test says so)
    res["coupled"] = int(bool(op.halo_in_launch))
    for ortho in (("mgs", "cgs") if op.halo_in_launch else ()):
        sc = linsys.RestartedGmres(linsys.LinearSystem(op, bc[r0:r1]), maxiter=40, max_restarts=40, tol=1e-9, ortho=ortho)
        res["coupled_%s_resnorms" % ortho] = np.array(sc.resnorms)
        res["coupled_%s_x" % ortho] = sc.xk[:, 0].copy()
    # 2c. slabs beyond the blocked kernel's range: the register-resident chain kernels with the cross-rank stage inside every
    # grid-wide sum (csrc/chain_xr.hip).  Two processes share this device, so the shapes are chosen for FOUR compute units
    # (kh_ctx_set "chain_xr_cus"): 24 rows per lane at 90,000 / 87,000 rows, 48 (the last 8 rows of w in LDS) at 184,900 / 180,600
    ctx.set("chain_xr_cus", 4)
    ctx.set("chain_blk2", 0)
    for tag, nxx in (("24", 300), ("48", 430)):
        Ax, bx = block(rank, nxx)
        nmax = int(rdv.allreduce_max(float(Ax.shape[0])))
        ctx.set("lowsync_rows", (int(Ax.shape[0]) << 32) | nmax)
        c0 = ctx.get("n_chain_xr")
        try:
            sx = linsys.RestartedGmres(linsys.LinearSystem(Ax, bx), maxiter=30, max_restarts=5, tol=1e-9, ortho="mgs")
        except utils.ConvergenceError as e:
            sx = e.solver
        res["chainxr_%s_resnorms" % tag] = np.array(sx.resnorms)
        res["chainxr_%s_x" % tag] = sx.xk[:, 0].copy()
        res["chainxr_%s_launches" % tag] = ctx.get("n_chain_xr") - c0
    ctx.set("chain_xr_cus", 0)
    ctx.set("chain_blk2", 1)
    res["n_halo_xh"], res["n_halo_exchange"] = ctx.get("n_halo_xh"), ctx.get("n_halo_exchange")
    res["n_xr"], res["n_xr_fused"], res["panels_checked"] = ctx.get("n_xr"), ctx.get("n_xr_fused"), checked
    res["n_chain_blk2"] = ctx.get("n_chain_blk2")
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
    rdv.barrier()
    # 3. a peer that does not arrive
    timed_out = -1
    if os.environ.get("XR_TIMEOUT_TEST", "1") == "1":
        ctx.set("xr_timeout_ms", 1500)
        if rank == 0:
            try:
                ctx.allreduce_host(np.ones(3))
                timed_out = 0
            except _hip.BackendError as e:
                timed_out = 1 if "did not arrive" in str(e) else 0
        rdv.barrier()
    with open(os.path.join(out, "rank%d.done" % rank), "w") as fh:
        fh.write("%d\n" % timed_out)
    rdv.close()
    os._exit(0)          # (rank 0's context holds a sum that will never complete: no destructors)


if __name__ == "__main__":
    main()
