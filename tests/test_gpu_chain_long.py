"""k_mgs_chain_long (krypy_amd/csrc/chain_long.h): the Gram-Schmidt chain at 48 rows per lane (10.49 M ... 12.58 M rows per GPU:
config 5's 12.5 M-row slabs) with a third of every basis column kept on the chip between its dot and its update - same
arithmetic in the same order as k_mgs_chain<48> (/root/reference/krypy/utils.py:1012-1029), so THE SAME BITS, with and without
the banded operator in the prologue."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_ref as ref
from tests.support.kernel_expect import expect_kernel

pytestmark = pytest.mark.gpu


def _steps(ctx, A, v, m, sweeps_at=None):
    Ad = ctx.csr(A)
    n = A.shape[0]
    V, W = ctx.alloc(n, m + 1), ctx.alloc(n, 2)
    V.upload(0, v)
    H = np.zeros((m + 1, m))
    for k in range(m):
        H[: k + 2, k] = ctx.arnoldi_step(Ad, None, V, None, W, 0, k, 0, 2 if k == sweeps_at else 1, 0)
    return H, V.download()


@pytest.mark.parametrize("kind", ["tridiagonal (SpMV launch + chain)", "5-point stencil (operator in the prologue)",
                                  "7-point stencil (operator in the prologue)"])
def test_same_bits_as_the_kernel_with_both_reads_from_memory(hip, kind):
    """Seven Arnoldi steps (one of them with two sweeps) at 12 M rows through k_mgs_chain_long and through k_mgs_chain<48>
    (kh_ctx_set "chain_long", 0): H and the basis bit for bit; the Arnoldi relation and orthogonality at 1e-12."""
    if kind.startswith("tridiagonal"):
        n = 12_000_000
        A = sp.diags([np.full(n - 1, -1.0), np.linspace(2.0, 3.0, n), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr()
    elif kind.startswith("5"):
        A = ref.laplace2d(4000, 3000)
    else:
        A = ref.laplace3d(500, 500, 48)
    n = A.shape[0]
    m = 7
    v = np.random.default_rng(48).standard_normal(n)
    v /= np.linalg.norm(v)
    out = {}
    for on in (1, 0):
        hip.set("chain_long", on)
        try:
            c0, f0 = hip.get("n_chain_long"), hip.counters()["chain_fused"]
            out[on] = _steps(hip, A, v, m, sweeps_at=3) + (hip.get("n_chain_long") - c0, hip.counters()["chain_fused"] - f0)
        finally:
            hip.set("chain_long", 1)
    (H1, V1, used1, fused1), (H0, V0, used0, fused0) = out[1], out[0]
    assert np.array_equal(H1, H0) and np.array_equal(V1, V0)
    assert np.linalg.norm(A.dot(V1[:, :m]) - V1.dot(H1)) < 1e-12 * np.linalg.norm(H1)
    assert np.linalg.norm(V1.T.dot(V1) - np.eye(m + 1)) < 1e-12
    # (the very first step of a banded operator is the three-pass Lanczos kernel: one link)
    expect_kernel(used1 >= m - 1 and used0 == 0, "launches of the long kernel with it on / off: %r" % ((used1, used0),))
    expect_kernel((fused1 > 0) == (not kind.startswith("tridiagonal")) and fused1 == fused0,
                  "operator in the prologue: %r" % ((fused1, fused0),))
