#!/usr/bin/env python
"""Where a launch of the blocked Gram-Schmidt kernel (chain_blk.h) spends its time: 64-link micro-launches
(kh_bench_kernel 20 .. 24: 64 columns, one sweep) as they are, without the exchange between workgroups (every workgroup uses its
own partial sums), and without the column stream; next to the per-column kernel (k_mgs_chain_small).
    python tools/blk_prof.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(sizes):
    import numpy as np
    from krypy_amd import _hip

    ctx = _hip.get_context()
    rng = np.random.default_rng(0)
    for n in sizes:
        V, W = ctx.alloc(n, 66), ctx.alloc(n, 2)
        for j in range(66):
            V.upload(j, rng.standard_normal(n) / np.sqrt(n))
        W.upload(0, rng.standard_normal(n))
        out = []
        for which, label in ((20, "blocked"), (21, "blocked, no exchange"), (22, "blocked, no stream"), (23, "blocked, neither"),
                             (24, "per-column sums")):
            try:
                ctx.bench_kernel(which, V, W, 5)
                ms = min(ctx.bench_kernel(which, V, W, 40) for _ in range(3))
                out.append("%s: %.1f us (%.2f us/link)" % (label, ms * 1e3, ms * 1e3 / 64))
            except Exception as exc:
                out.append("%s: %r" % (label, exc))
        ctx.set("chain_blk", 1)
        print("N = %8d: %s" % (n, "; ".join(out)), flush=True)
        del V, W


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [140000, 250000, 500000, 1000000])
