"""The environment switches of krypy_amd, as ONE table.

Every ``KRYPY_AMD_*`` variable any source file of the package, ``bench.py`` or the test-suite reads is a row here; INTEGRATION.md
section 4 is this table rendered (``python -m krypy_amd._switches`` prints it) and ``tests/test_abi.py`` holds the three to each
other: a variable read somewhere but missing here, a row missing from INTEGRATION.md, or a ``kernel-path`` switch whose
off-setting is not exercised by the fallback suites of ``tools/r05_evidence.sh`` fails the CPU suite.

kinds
  config       where things are (device, library)
  kernel-path  selects a different kernel / host loop for the SAME result: each off-setting is a tested path
               (``tools/r05_evidence.sh fallback`` runs the GPU parity files under it)
  tuning       a threshold or size inside one kernel path (measurement; results do not depend on it beyond rounding)
  bench        read by ``bench.py`` only
  test         read by the test-suite / test hooks only
"""

SWITCHES = [
    # name, default, kind, off (the setting the fallback suites run, kernel-path only), meaning
    ("KRYPY_AMD_DEVICE", "LOCAL_RANK or 0", "config", None, "HIP device of the process-wide context"),
    ("KRYPY_AMD_LIB", "in-tree krypy_amd/lib/libkrylov_hip.so", "config", None,
     "path of the (same) HIP library, e.g. an experimental build; there is no alternative backend"),
    ("KRYPY_AMD_MGS_CHAIN", "1", "kernel-path", "0",
     "0: reference-order MGS through one launch per column (`k_gs_link`) instead of the register-resident chain kernels"),
    ("KRYPY_AMD_SPMV_DIA", "4", "kernel-path", "0",
     "banded (diagonal-major) SpMV for CSR operators with <= 32 well-filled diagonals: row pairs per lane (1, 2, 4); 0: always the CSR "
     "kernel.  Same bits either way"),
    ("KRYPY_AMD_CHAIN_SPMV", "1", "kernel-path", "0",
     "0: banded operators are applied by a separate SpMV launch instead of in the chain kernel's prologue.  Same bits either way"),
    ("KRYPY_AMD_CHAIN_LDS", "1", "kernel-path", "0", "0: plain chain kernel (no LDS / register-ring reuse of the column)"),
    ("KRYPY_AMD_CHAIN_PF", "1", "kernel-path", "0",
     "0: never use `k_mgs_chain_pf` (LDS-DMA re-reads, next column prefetched through the update phase; used up to 24 rows per lane); "
     "2: use it at 32 / 40 rows per lane as well"),
    ("KRYPY_AMD_GMRES_CYCLE", "1", "kernel-path", "0",
     "0: `Gmres` takes every iteration through the per-step Python loop instead of `kh_gmres_cycle`"),
    ("KRYPY_AMD_CG_CYCLE", "1", "kernel-path", "0",
     "0: `Cg` takes every iteration through the per-step Python loop instead of `kh_cg_cycle`.  Same bits either way"),
    ("KRYPY_AMD_MINRES_CYCLE", "1", "kernel-path", "0",
     "0: `Minres` takes every iteration through the per-step Python loop instead of `kh_minres_cycle`.  Same bits either way"),
    ("KRYPY_AMD_CHAIN_BLK", "1", "kernel-path", "0",
     "0: short vectors (4 rows per lane, 8 or more Gram-Schmidt links in the step) take the per-column ring kernel instead of the "
     "blocked kernel (`csrc/chain_blk.h`: one grid-wide sum per four columns)"),
    ("KRYPY_AMD_CHAIN_BLK2", "1", "kernel-path", "0",
     "0: vectors of 5 / 6 rows per lane (1.05 M ... 1.57 M rows) keep the per-column ring kernel on one GPU, and on N ranks with the xr "
     "transport on `ortho='mgs'` keeps the one-reduction form (two cross-rank sums per step, the local basis read twice) instead of the "
     "eight-wave blocked kernel with the cross-rank sums INSIDE the launch (`csrc/chain_blk2.h`: the basis read once, no all-reduce call)"),
    ("KRYPY_AMD_CHAIN_LONG", "1", "kernel-path", "0",
     "0: 48 rows per lane (10.49 M ... 12.58 M rows per GPU / rank) keep `k_mgs_chain<48>` - both reads of every basis column from memory "
     "- instead of `k_mgs_chain_long` (`csrc/chain_long.h`: two batches parked in LDS, the last two still in the register ring: 16 of 48 "
     "rows never leave the chip between a column's dot and its update).  Same bits either way; counter `n_chain_long`"),
    ("KRYPY_AMD_GRAM_MFMA", "1", "kernel-path", "0",
     "0: an inner product with a block on both sides (`kh_gemm_tn` with two or more columns on the right: `<W, V>` of the deflation "
     "projector's set-up, `<U, AU>`, the Ritz set-up's `<V, AU>`) keeps one `k_multidot` launch and one host round trip per column of "
     "the right block - the left block read once PER column - instead of `k_gram_mfma` (`csrc/kernels.h`: both blocks read once per "
     "16 x 16 tile, FP64 matrix cores), and a block times a small matrix with 2 ... 16 output columns (`kh_gemm_nn`: the Ritz vectors "
     "`[V_n, U] @ coeffs`) keeps one `k_multiaxpy` pass over the block per output column instead of `k_panel_gemm_mfma` (the block "
     "read once).  Another summation order, the same bits from run to run; counters `n_gram_mfma`, `n_panel_gemm`"),
    ("KRYPY_AMD_CHAIN_XR", "1", "kernel-path", "0",
     "0: on N ranks with the xr transport on, slabs beyond the blocked kernel's 2.5 M rows keep the one-reduction form / the panel "
     "kernels (the local basis read twice, two sums across the ranks per step) instead of the register-resident chain kernels with the "
     "cross-rank stage inside every grid-wide sum (`csrc/chain_xr.hip`: 16 ... 56 rows per lane, up to 14.68 M rows per rank, the basis "
     "read once, no all-reduce call; counter `n_chain_xr`)"),
    ("KRYPY_AMD_BLK2_CW", "1", "kernel-path", "0",
     "0: every wave of the eight-wave blocked kernel carries rows (512 lanes, 4 ... 6 rows per lane, up to 1.57 M rows) instead of wave 0 "
     "being a communication wave without rows (448 lanes with rows, 4 ... 7 rows per lane, up to 1.6 M rows: no register spills up to 6 "
     "rows, the total of a sum seen without waiting for the wave's own rows); 2: the communication wave up to 6 rows per lane, 512 lanes "
     "beyond (the A / B of the 7-row shape)"),
    ("KRYPY_AMD_BLK2_ONE", "2", "kernel-path", "0",
     "the eight-wave blocked kernel's shapes with ONE block of four columns in registers (8 ... 11 rows per lane, vectors / slabs of "
     "1.6 ... 2.5 M rows: a rank's share of the benchmark problem on four devices): 2 = wherever they fit, 1 = on a communicator only "
     "(the sums cross the ranks inside the launch, the basis is read once where the panel forms read it twice), 0 = never"),
    ("KRYPY_AMD_BLK_ONEX_MAXN", "70000", "tuning", None,
     "vectors longer than this run the blocked kernel spread over the chip instead of on one XCD"),
    ("KRYPY_AMD_BLK_NX", "8", "kernel-path", "0",
     "workgroups WITHOUT rows in front of a blocked launch that is spread over the chip: they gather the grid-wide sums on compute "
     "units that carry no column stream (0: none; `kh_ctx_set \"blk_nx\"`, counter `n_blk_rowless`)"),
    ("KRYPY_AMD_MGS_LOWSYNC", "1", "kernel-path", "0",
     "0: on N ranks `ortho='mgs'` takes one cross-rank sum per Gram-Schmidt link (k + 2 per step) instead of all coefficients from one "
     "pass and ONE sum with the Gram-table correction (2 per step; shards up to 24 rows per lane, up to 128 basis columns)"),
    ("KRYPY_AMD_PROJ_REG", "1", "kernel-path", "0",
     "0: the deflation projector inside a deflated Arnoldi step runs as four launches per sweep instead of one launch with the vector "
     "in registers (`csrc/proj_reg.h`; one GPU, 16 or more rows per lane, up to 16 deflation vectors)"),
    ("KRYPY_AMD_PROJ_PANEL", "1", "kernel-path", "0",
     "0: where the one-launch projector does not apply (N ranks) the projector's passes over its two bases run through the chunked "
     "kernels for long vectors too (`k_multidot<16>` / `k_multiaxpy<16>`) instead of the register-resident panel kernels"),
    ("KRYPY_AMD_CHAIN_ONEX", "1", "kernel-path", "0",
     "0: short vectors (4 ... 32 workgroups) keep their chain-kernel workgroups spread over the eight XCDs instead of on one"),
    ("KRYPY_AMD_CHAIN_SMALL", "1", "kernel-path", "0",
     "0: short vectors without a preconditioner take the general chain kernels instead of the column-ring kernel (`k_mgs_chain_small`)"),
    ("KRYPY_AMD_TAG_WAIT", "1", "kernel-path", "0",
     "0: every Arnoldi step records an event and `kh_arnoldi_step_end` waits for it; 1: the chain kernels write a completion tag "
     "behind the H column in pinned memory and the host polls that word"),
    ("KRYPY_AMD_LANCZOS_FUSED", "1", "kernel-path", "0",
     "0: a step with one Gram-Schmidt link (every Lanczos / MINRES step) runs the general chain kernel (six passes) instead of the "
     "three-pass kernel of `csrc/lanczos.h`.  Same bits either way"),
    ("KRYPY_AMD_CG_STEP", "1", "kernel-path", "0",
     "0: CG runs operator, inner product and updates as separate calls with the step length formed on the host instead of one fused "
     "`kh_cg_step` / `kh_zcg_step` per iteration"),
    ("KRYPY_AMD_SPMV_WIN", "1", "kernel-path", "0",
     "0: the CSR-stream SpMV gathers every entry of x from global memory instead of from an LDS window loaded once per row block "
     "(operators whose row blocks touch at most 4096 neighbouring columns)"),
    ("KRYPY_AMD_SPMV_SPLIT", "1", "kernel-path", "0",
     "0: a sharded SpMV waits for its halo exchange and multiplies all rows in one launch instead of overlapping the exchange with the "
     "interior rows"),
    ("KRYPY_AMD_XR", "1", "kernel-path", "0",
     "0: the sums across the ranks of a node stay `ncclAllReduce` calls instead of the IPC-mailbox kernels of `csrc/xr.hip` "
     "(`krypy_amd.dist.enable_xr`: switched on only when EVERY rank could map every peer's mailbox and a self-test passed)"),
    ("KRYPY_AMD_XH", "1", "kernel-path", "0",
     "0: the halo of a banded shard stays a grouped `ncclSend` / `ncclRecv` on the communication stream around split SpMV launches "
     "instead of travelling INSIDE the banded kernel's one launch through IPC-mapped ghost granules (`kh_mat_xh_*`; needs the xr "
     "transport; `ShardedCSROperator` switches it on only when every rank could map its neighbours' granules)"),
    ("KRYPY_AMD_XR_TIMEOUT_S", "60", "tuning", None,
     "seconds a cross-rank sum waits for a peer's contribution before the next host synchronisation reports `KH_ERR_COMM`"),
    ("KRYPY_AMD_XR_SELFTEST_S", "8", "tuning", None, "the same timeout during `enable_xr`'s self-test of a few sums"),
    ("KRYPY_AMD_ROCTX", "0", "config", None,
     "1: `roctxRangePush/Pop` around `kh_arnoldi_step_begin/_end`, `kh_gemm_nn`, `kh_residual`, `kh_cg_step` (libroctx64 is `dlopen`ed "
     "then; `rocprofv3 --marker-trace`)"),
    ("KRYPY_AMD_CGS_REVERSE", "1", "kernel-path", "0", "0: the panel update pass walks the columns first to last"),
    ("KRYPY_AMD_CGS_NT_GB", "0.75", "tuning", None,
     "panel size (GB) above which the panel dots pass streams the basis with non-temporal loads"),
    ("KRYPY_AMD_FORCE_MULTI", "0", "test", None,
     "1: run the multi-rank code path (cross-rank sums, halo hook) on a 1-rank communicator"),
    ("KRYPY_AMD_BENCH_SHARDED_EXTRAS", "0", "bench", None,
     "1: a `bench.py` run on N > 1 ranks also times the other Gram-Schmidt order (`mgs` beside the `cgs` default) and reports its "
     "cross-rank sums per iteration and orthogonality, outside the timed region (on one rank in `--force-sharded` mode it always does)"),
    ("KRYPY_AMD_BENCH_DEVICES", "unset", "bench", None,
     "set by `bench.py`'s own launcher for the rank processes it starts: the number of distinct devices the run uses (`n_gpus` of the line; "
     "differs from the number of ranks only in the `--share-devices` test mode)"),
    ("KRYPY_AMD_BENCH_ORTHO", "auto", "bench", None,
     "`bench.py --ortho` default (`auto` = `mgs` on 1 GPU; on N ranks one untimed cycle per (form, transport) candidate, the fastest "
     "that ran on every rank is timed)"),
    ("KRYPY_AMD_BENCH_SECONDARY", "1", "bench", None,
     "0: the default `bench.py` line does not carry `secondary.config3_minres_jacobi` (MINRES + Jacobi on the timed run's matrix, "
     "2 x 200 iterations after the timed region)"),
    ("KRYPY_AMD_BENCH_XR_TIMEOUT_S", "15", "bench", None,
     "seconds a cross-rank sum over the mailboxes may wait for a peer inside `bench.py`'s timed region before the run goes back to RCCL "
     "and the panel form on every rank (`timed_region_fallback` of the line); the probe of the candidates uses 10 s"),
    ("KRYPY_AMD_BENCH_FAULT", "0", "test", None,
     "1: a sharded `bench.py` run fakes a timed-out in-launch sum in the first launch of its timed region (`kh_ctx_set \"chain_fault\"`): "
     "the test of the timed region's way back to RCCL and the panel form on the real context (`tests/test_gpu_multirank.py`)"),
    ("KRYPY_AMD_BENCH_DEADLINE_S", "1500", "bench", None,
     "seconds after which `bench.py --gpus N`'s own launcher stops its rank processes and exits non-zero (a collective that never returns "
     "must not hold the driver for ever)"),
    ("KRYPY_AMD_TEST_RLIMIT_GB", "96", "test", None,
     "cap (GB) on the host memory of the GPU test session (`RLIMIT_DATA`, set once the HIP context exists): a test asking for absurd "
     "memory dies with a `MemoryError` instead of taking the box down; 0: off"),
    ("KRYPY_AMD_TEST_FORCE_MULTI", "0", "test", None,
     "1: the WHOLE GPU suite through the multi-rank code path on a 1-rank communicator (`tests/conftest.py`)"),
    ("KRYPY_AMD_PARITY_LOG", "unset", "test", None,
     "file the full-size parity tests append their measured deviations to (`profiles/r05_fullsize_parity.log` comes from such a run)"),
    ("KRYPY_AMD_FULLSIZE_ORACLE", "0", "test", None,
     "1: `tests/test_oracle_golden.py` also runs the CPU oracle at N = 10^7 / n = 32768 against the full-size fixtures (ten minutes)"),
]


def names():
    return [row[0] for row in SWITCHES]


def kernel_path_switches():
    """(name, off-setting) of every switch that selects another kernel or host loop for the same result."""
    return [(row[0], row[3]) for row in SWITCHES if row[2] == "kernel-path"]


def markdown():
    lines = ["| variable | default | kind | meaning |", "|---|---|---|---|"]
    for name, default, kind, off, text in SWITCHES:
        lines.append("| `%s` | %s | %s | %s |" % (name, ("`%s`" % default) if " " not in default else default, kind, text))
    return "\n".join(lines)


if __name__ == "__main__":
    print(markdown())
