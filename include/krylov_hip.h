/*
 * krylov_hip.h - C ABI of libkrylov_hip.so, the MI355X (gfx950) Krylov solver core.
 *
 * The reference (andrenarchy/krypy, pure Python) has no FFI: every flop of its
 * Arnoldi/Lanczos hot path is a NumPy/SciPy/OpenBLAS call.  This header is the
 * boundary the device core exposes in place of those calls; each entry point
 * names the reference call site it replaces (file:line under /root/reference).
 * The only caller is krypy_amd/_hip.py (ctypes); see INTEGRATION.md for the
 * binding a KryPy maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *  - every function returns 0 on success, a negative kh_status on failure and
 *    stores a message retrievable with kh_last_error() (thread-local).
 *  - handles are opaque; host pointers are caller-owned; device buffers are
 *    library-owned and released by kh_*_free / kh_ctx_destroy.
 *  - one HIP stream per context; a context is single-threaded by contract (like
 *    the reference).  Calls that return host scalars synchronise the stream
 *    before returning; all others are asynchronous on the context's stream.
 *  - all storage is fp64.  A kh_vec is a block of `ncols` column vectors of
 *    length n, each column contiguous and 256-byte aligned (leading dimension
 *    rounded up to 32 doubles): the reference's (N, k) ndarrays, stored
 *    column-major so that basis vectors stream at full HBM width.  A COMPLEX
 *    (c128) N-vector block is a kh_vec of length 2N (re, im interleaved) handed
 *    to the kh_z* entry points, which take complex coefficients as (re, im) pairs.
 *  - reductions are deterministic: fixed grid, fixed-order tree, no float atomics.
 */
#ifndef KRYLOV_HIP_H
#define KRYLOV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kh_ctx_s* kh_ctx;
typedef struct kh_mat_s* kh_mat;
typedef struct kh_vec_s* kh_vec;
typedef struct kh_proj_s* kh_proj;

typedef enum {
    KH_OK = 0,
    KH_ERR_HIP = -1,      /* a HIP runtime call failed (message has hipGetErrorString) */
    KH_ERR_ARG = -2,      /* invalid argument (shape mismatch, bad index, NULL handle) */
    KH_ERR_NOMEM = -3,    /* device or pinned-host allocation failed */
    KH_ERR_COMM = -4,     /* RCCL call failed */
    KH_ERR_UNSUPPORTED = -5
} kh_status;

/* Gram-Schmidt variants of kh_arnoldi_step */
typedef enum {
    KH_GS_MGS = 0,   /* reference order: for j: alpha=<v_j,w>; w-=alpha*b_j  (utils.py:1012-1029) */
    KH_GS_CGS = 1    /* panel form: h=V^T w; w-=B h  (one reduction per sweep)                     */
} kh_gs_mode;

/* ---- library / context ------------------------------------------------------------ */
const char* kh_last_error(void);
int kh_version(void);
int kh_device_count(int* count);
/* create a context on HIP device `device` (one stream, scratch for reductions).  One context = one device = one
 * process: N GPUs are N processes that join with kh_comm_init (the blueprint sketched (ndev, devs) here; with one
 * process per GPU a context never spans devices). */
int kh_ctx_create(int device, kh_ctx* out);
int kh_ctx_destroy(kh_ctx ctx);
int kh_ctx_sync(kh_ctx ctx);
/* info[0]=compute units, info[1]=total device memory (bytes), info[2]=free bytes, info[3]=reduction grid */
int kh_ctx_info(kh_ctx ctx, int64_t info[4]);
/* which Gram-Schmidt kernels ran so far on this context: [0] chain launches (k_mgs_chain*), [1] of
 * those with the column head parked in LDS, [2] of those with the operator fused into the prologue,
 * [3] register-resident panel sweeps (k_cgs_dots + k_cgs_update) */
int kh_ctx_counters(kh_ctx ctx, int64_t out[4]);
/* tuning knobs (0 keeps the default): reduction grid size, SpMV LDS tile (nnz) */
int kh_ctx_tune(kh_ctx ctx, int reduce_blocks, int spmv_tile);
/* named switches of a context (1 = on, the default; the environment variables of INTEGRATION.md set the
 * initial values): "spmv_dia" banded SpMV for stencil CSR operators, "chain" register-resident MGS chain,
 * "chain_lds" column head parked in LDS, "chain_spmv" operator fused into the chain prologue, "chain_onex" short vectors
 * on one XCD, "chain_small" the column-ring kernel for them, "lanczos_fused" the three-pass Lanczos kernel, "tag_wait"
 * completion tags in pinned memory instead of an event per Arnoldi step.  bench.py
 * uses it to time the CSR-stream and the banded SpMV kernel on the same operator.  Test-only switches (0 by
 * default): "chain_fault" the next chain launch fakes a timeout, "halo_loopback" a 1-rank communicator exchanges
 * the halo of a sharded operator with ITSELF (grouped ncclSend / ncclRecv to its own rank: the slab of an operator
 * that is periodic across the slab boundary) - the real exchange on one GPU. */
int kh_ctx_set(kh_ctx ctx, const char* key, int64_t value);
/* read a switch back, or a counter: "n_spmm" panel applications of a CSR operator that streamed the matrix
 * once (k_spmm_stream / k_spmm_dia), "n_chain_recovered" Arnoldi steps re-run on the per-column kernels after
 * a timeout of the chain kernel's grid-wide reduction, "n_spmv_split" sharded SpMVs run as interior / boundary
 * launches around the halo exchange, "n_halo_exchange" grouped ncclSend / ncclRecv exchanges issued */
int kh_ctx_get(kh_ctx ctx, const char* key, int64_t* value);
/* event timing on the context's stream (for bench.py's per-kernel roofline numbers) */
int kh_timer_start(kh_ctx ctx);
int kh_timer_stop(kh_ctx ctx, double* elapsed_ms);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) ------------------------------- */
/* 128-byte ncclUniqueId; rank 0 creates it, the launcher broadcasts it to all ranks */
int kh_comm_unique_id(unsigned char id[128]);
int kh_comm_init(kh_ctx ctx, int rank, int nranks, const unsigned char id[128]);
int kh_comm_destroy(kh_ctx ctx);
/* in-place sum all-reduce of `count` host doubles through a device staging buffer (setup paths) */
int kh_comm_allreduce_host(kh_ctx ctx, double* vals, int64_t count);
/* xr - sums across the ranks of ONE node without a library call (csrc/xr.hip; replaces the ncclAllReduce of the inner
 * products of /root/reference/krypy/utils.py:182-183 on N ranks).  kh_xr_export allocates this rank's mailbox in
 * fine-grained device memory and returns its 64-byte hipIpcMemHandle_t; the launcher gathers the handles of all ranks
 * (rank order) and hands the 64 * nranks bytes to kh_xr_attach, which maps every peer's mailbox.  After EVERY rank has
 * attached successfully the launcher sets kh_ctx_set(ctx, "xr", 1) on all of them (the choice must be the same on every
 * rank): all-reduces of panels then run as one kernel of tagged 8-byte system-scope stores into the peers' mailboxes and
 * a rank-ordered sum (the same bits on every rank).  Works with or without an RCCL communicator (without one: sums
 * cross the ranks, halos do not).  A peer that does not arrive within KRYPY_AMD_XR_TIMEOUT_S (60) seconds is reported as
 * KH_ERR_COMM by the next call that synchronises with the host. */
int kh_xr_export(kh_ctx ctx, unsigned char handle[64]);
int kh_xr_attach(kh_ctx ctx, int rank, int nranks, const unsigned char* handles);
int kh_xr_detach(kh_ctx ctx);
/* xh - the halo of a block-row shard through the same kind of mailboxes: after kh_mat_set_halo, kh_mat_xh_export allocates this
 * shard's ghost GRANULES in fine-grained device memory and returns their 64-byte IPC handle; the launcher hands every rank its two
 * neighbours' handles (kh_mat_xh_attach: `prev_ng` / `next_ng` = the ghost entries nrecv_prev + nrecv_next of THEIR boxes, `prev_off`
 * = the previous rank's nrecv_prev; NULL = no such neighbour; self_loop = 1: the slab of an operator that is periodic across the slab
 * boundary, the rank is its own neighbour) and, after EVERY rank has attached, switches it on everywhere (kh_mat_xh_enable).  The
 * banded SpMV of the shard (the operator of /root/reference/krypy/utils.py:1593-1594 on this rank's rows) then stores its boundary
 * rows into the neighbours' granules and polls its own INSIDE its one launch: no ncclSend / ncclRecv kernel, no second stream. */
int kh_mat_xh_export(kh_ctx ctx, kh_mat A, unsigned char handle[64]);
int kh_mat_xh_attach(kh_ctx ctx, kh_mat A, const unsigned char* prev, int64_t prev_ng, int64_t prev_off, const unsigned char* next,
                     int64_t next_ng, int self_loop);
int kh_mat_xh_enable(kh_ctx ctx, kh_mat A, int on);
/* unmap the neighbours' granules again (every rank, when the collective decision about the in-launch halo is "off" after
 * some ranks had attached): a later kh_mat_xh_attach starts from scratch.  New component. */
int kh_mat_xh_detach(kh_ctx ctx, kh_mat A);
/* describe the halo of a block-row-sharded matrix: this rank sends `nsend_*` of its first/last
 * local rows to the previous/next rank and receives as many ghost entries from them.  After
 * this call kh_apply() on `A` exchanges halos (ncclSend/ncclRecv) before the local SpMV; the
 * matrix' column indices must address [local rows | ghosts from prev | ghosts from next]. */
int kh_mat_set_halo(kh_ctx ctx, kh_mat A, int64_t nsend_prev, int64_t nsend_next,
                    int64_t nrecv_prev, int64_t nrecv_next);
/* The LONGEST slab of the run this shard belongs to (every rank passes the same number).  Kernel choices of an Arnoldi step that
 * change the PATTERN of sums across the ranks - the one-reduction form, the blocked and the chain kernels with the sums inside
 * their launch - are made for it, so that every rank decides alike whatever the length of its own slab; keyed on the operator
 * (steps without an operator: kh_ctx_set "lowsync_rows" by local length).  New component: the reference is
 * single-process. */
int kh_mat_set_rows_max(kh_mat A, int64_t rows_max);
/* diagnostic: write the nrecv_prev + nrecv_next ghost entries directly (what the halo exchange would
 * deliver); lets a single process check a shard's SpMV against the global operator */
int kh_mat_set_ghost(kh_mat A, const double* values, int64_t count);
/* diagnostic: read the ghost entries back (what the last halo exchange delivered) */
int kh_mat_get_ghost(kh_mat A, double* values, int64_t count);

/* ---- vectors ---------------------------------------------------------------------- */
int kh_vec_alloc(kh_ctx ctx, int64_t n, int64_t ncols, kh_vec* out);   /* zero-filled */
int kh_vec_free(kh_vec v);
int kh_vec_shape(kh_vec v, int64_t* n, int64_t* ncols, int64_t* ld);
/* host <-> device; `host` is column-major with leading dimension host_ld (>= n) */
int kh_vec_upload(kh_vec v, int64_t col0, int64_t ncols, const double* host, int64_t host_ld);
int kh_vec_download(kh_vec v, int64_t col0, int64_t ncols, double* host, int64_t host_ld);
int kh_vec_zero(kh_vec v, int64_t col0, int64_t ncols);
int kh_vec_copy(kh_vec dst, int64_t dcol, kh_vec src, int64_t scol, int64_t ncols);
/* a few consecutive entries of one column (Householder Arnoldi reads/writes single elements:
 * utils.py:349-375, 973-983); synchronous */
int kh_vec_get(kh_vec v, int64_t col, int64_t i0, int64_t count, double* out);
int kh_vec_set(kh_vec v, int64_t col, int64_t i0, int64_t count, const double* in);
/* zero entries [i0, i0+count) of one column */
int kh_vec_zero_range(kh_vec v, int64_t col, int64_t i0, int64_t count);

/* ---- operators (replace MatrixLinearOperator._dot -> A.dot(X), utils.py:1593-1594) --- */
/* CSR as SciPy holds it: int32 indptr[n_rows+1], int32 indices[nnz] (sorted or not), fp64 data.
 * n_cols may exceed n_rows for a sharded matrix with ghost columns. */
int kh_csr_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* indptr,
                  const int32_t* indices, const double* data, kh_mat* out);
/* dense row-major (C-ordered ndarray), leading dimension lda */
int kh_dense_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, const double* a, int64_t lda,
                    kh_mat* out);
/* dense operator from a block already on the device: A = alpha X[:, col0 : col0 + n_rows]^T + beta I (row i of the
 * operator = column col0 + i of X).  With Y = G G^T formed by kh_apply's panel path the columns of the symmetric Y are
 * the rows of the row-major operator: BASELINE config 4's A = G G^T / n + I without a trip over the host
 * ("build on device or in blocks"; the reference gets its ndarray through get_linearoperator,
 * utils.py:241-259).  Synchronous. */
int kh_dense_from_block(kh_ctx ctx, kh_vec X, int64_t col0, int64_t n_rows, double alpha, double beta,
                        kh_mat* out);
/* diagonal operator (Jacobi M / Minv given as scipy.sparse.diags) */
int kh_diag_upload(kh_ctx ctx, int64_t n, const double* d, kh_mat* out);
int kh_mat_free(kh_mat A);
/* number of diagonals of the banded copy kh_csr_upload built for this operator (square, sorted
 * columns, no stored zeros, <= 32 diagonals at least 70 % full; KRYPY_AMD_SPMV_DIA=0 disables it),
 * 0 when the CSR kernel serves.  Same results bit for bit either way. */
int kh_mat_diagonals(kh_mat A);
/* Y[:, ycol:ycol+nc] = A * X[:, xcol:xcol+nc].  CSR rows are summed left to right in storage
 * order with separate multiply and add, i.e. bit-identical to scipy's csr_matvec(s) - for every row that fits the
 * kernel's LDS tile (2048 entries; kh_ctx_tune).  A longer row gets a workgroup of its own that adds its products as a
 * fixed tree: the same sum to rounding (tested at 1e-12), not the same bits. */
int kh_apply(kh_ctx ctx, kh_mat A, kh_vec X, int64_t xcol, kh_vec Y, int64_t ycol, int64_t ncols);

/* ---- inner products, norms, updates (utils.inner/norm utils.py:160-238) ------------- */
/* out[j] = <V[:, j0+j], W[:, wcol]>, j < ncols   (numpy.dot(X.T.conj(), Y), utils.py:183) */
int kh_dot_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, kh_vec W, int64_t wcol,
                 double* out);
/* out (row-major nx*ny) = X[:, x0:x0+nx]^T Y[:, y0:y0+ny]   (utils.inner with a block on both sides, utils.py:160-193).
 * ny >= 2: 16 x 16 tiles on the FP64 matrix cores, both blocks read once per tile (k_gram_mfma; kh_ctx_set "gram_mfma" 0 /
 * KRYPY_AMD_GRAM_MFMA=0: one kh_dot_panel per column of Y).  Either way a fixed summation order: the same bits from run to run. */
int kh_gemm_tn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t nx, kh_vec Y, int64_t y0, int64_t ny,
               double* out);
/* W[:, wcol] -= sum_j h[j] * V[:, j0+j], applied left to right, multiply then subtract
 * (the `Av -= alpha * V[:, [j]]` of utils.py:1027-1029) */
int kh_axpy_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* h, kh_vec W,
                  int64_t wcol);
/* Y[:, y0:y0+nc] = beta*Y[:, ...] + alpha * X[:, x0:x0+k] @ C  with C (k x nc) row-major.
 * (V[:, :k].dot(yy) linsys.py:947;  V.dot(c) utils.py:549;  Ritz.get_vectors deflation.py:845)
 * nc = 1: one k_multiaxpy pass, additions left to right.  nc = 2 ... 16: the block is read ONCE for all output columns on the FP64
 * matrix cores (k_panel_gemm_mfma, passes of 64 columns; kh_ctx_set "gram_mfma" 0: one pass over the block per output column). */
int kh_gemm_nn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t k, const double* C, int64_t nc,
               double alpha, double beta, kh_vec Y, int64_t y0);
/* out = ||W[:, wcol]||_2   (numpy.linalg.norm(x, 2), utils.py:226) */
int kh_nrm2(kh_ctx ctx, kh_vec W, int64_t wcol, double* out);
/* Z[:, zcol] = alpha*X[:, xcol] + beta*Y[:, ycol]   (Z may alias X or Y) */
int kh_waxpby(kh_ctx ctx, kh_vec Z, int64_t zcol, double alpha, kh_vec X, int64_t xcol,
              double beta, kh_vec Y, int64_t ycol);
/* Z[:, zcol] = X[:, xcol] / s   (true division, as `V[:, [k+1]] = Av / H[k+1, k]` utils.py:1045) */
int kh_vdiv(kh_ctx ctx, kh_vec Z, int64_t zcol, kh_vec X, int64_t xcol, double s);

/* ---- fused hot path ------------------------------------------------------------------ */
/* One Arnoldi.advance() (utils.py:954-1048, mgs/dmgs/lanczos branches) entirely on the device:
 *   w = A V[:,k]           (skipped when A == NULL: W[:, wcol] already holds the operator result)
 *   lanczos (start==k>0):  w -= h_km1 * B[:,k-1]                      (utils.py:1000-1009)
 *   `sweeps` times, j=start..k:  alpha=<V_j,w>; hcol[j]+=alpha; w -= alpha*B_j   (1012-1029)
 *   hcol[k+1] = ||w||  or sqrt(<w, Md w>) when Md != NULL             (1030-1034)
 *   V[:,k+1] = (Md w | w)/hcol[k+1],  P[:,k+1] = w/hcol[k+1]          (1041-1045)
 * with B = P when P != NULL (preconditioned: V = M P) else V.  Md is a diagonal kh_mat or NULL.
 * hcol_out receives k+2 doubles (entries below `start` are zero).  The invariance test and the
 * Hessenberg/Givens algebra stay on the host (O(k)).  MW is an optional work column for Md*w.
 */
int kh_arnoldi_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t wcol,
                    int64_t k, int64_t start, int sweeps, int gs_mode, double h_km1,
                    double* hcol_out);
/* ---- deflation projector (utils.Projection._apply / apply_complement, utils.py:522-627) ------- */
/* Device image of a Projection with orthonormalised bases: W, V are (N, d) blocks, T = R^{-1} Q^H
 * and WRH = WR^H are d x d row-major host matrices (NULL = identity), copied to the device.  One
 * application z = a - P a is, per sweep, c = W^T z (panel product, all-reduced), c' = T c (tiny
 * device kernel), z -= V c' (panel update); <Y,a> = WRH c of the FIRST sweep is what the reference
 * returns as Ya.  Nothing goes through the host. */
int kh_proj_create(kh_ctx ctx, kh_vec W, kh_vec V, int64_t d, const double* T, const double* WRH,
                   int iterations, kh_proj* out);
int kh_proj_free(kh_proj p);
/* Z[:, zcol] = complement projection of A[:, acol]  (Z may be the same column: in place);
 * ya_out (d doubles, may be NULL) receives <Y, a>; synchronises only when ya_out != NULL */
int kh_proj_apply_complement(kh_ctx ctx, kh_proj p, kh_vec A, int64_t acol, kh_vec Z, int64_t zcol,
                             double* ya_out);

/* The same step split in two so that the host can process step k's Hessenberg column while the
 * device already runs step k+1 (whose kernels depend on device data only): _begin enqueues the
 * whole step plus an asynchronous copy of the H column into pinned slot `slot` (0..3) and records
 * an event; _end waits for that event only and returns `count` (= k+2) doubles.  With a
 * projector `proj` (deflated solvers: operator = (I - P) A, deflation.py:127-143) the projection is
 * applied to A v_k on the device and its d values <U, A v_k> (the new column of C) follow the H
 * column: count = k+2+d.  Steps begun in
 * order on one context execute in order.  A speculative step past convergence/invariance only
 * writes V[:,k+1] (P[:,k+1]) and the work vector; the caller discards it. */
int kh_arnoldi_step_begin(kh_ctx ctx, kh_mat A, kh_proj proj, kh_mat Md, kh_vec V, kh_vec P,
                          kh_vec W, int64_t wcol, int64_t k, int64_t start, int sweeps, int gs_mode,
                          double h_km1, int slot);
int kh_arnoldi_step_end(kh_ctx ctx, int slot, int64_t count, double* hcol_out);

/* A run of GMRES iterations in ONE call (krypy/linsys.py:951-997; SURVEY 8b "fused cycle"): Arnoldi steps
 * k0 .. k_stop-1 with look-ahead on the device (kh_arnoldi_step_begin / _end, slots k mod 4), and on the host - in C,
 * not in the caller's interpreter - what the reference does between two steps: the new Hessenberg column through the
 * previous Givens rotations and its own (BLAS drotg convention), R, the rotated right-hand side y and the residual
 * recurrence |y[k+1]|.  H and R are (ldh = ldr >= k_stop) row-major like the reference's arrays; cs holds (c, s) per
 * step; *h2_io the running squared Frobenius norm of H; *enq_io the number of steps begun on the device (in: steps the
 * caller has already begun, out: k_done + what is still in flight - at most step k_last is ever begun).
 * Stops  KH_CYCLE_TOL    after the step whose |y[k+1]| / bnorm <= tol (that step IS recorded),
 *        KH_CYCLE_CHECK  at a step whose H[k+1,k] / ||H||_F is not > 1e-14: maybe an invariant subspace - the step is
 *                        NOT recorded (its column stays in its slot for kh_arnoldi_step_end), the caller decides,
 *        KH_CYCLE_LIMIT  at k_stop.
 * *k_done = number of recorded steps (counted from 0). */
#define KH_CYCLE_LIMIT 0
#define KH_CYCLE_TOL 1
#define KH_CYCLE_CHECK 2
int kh_gmres_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t k0, int64_t k_stop,
                   int64_t k_last, int sweeps, int gs_mode, int64_t* enq_io, double tol, double bnorm, double* H, int64_t ldh,
                   double* R, int64_t ldr, double* cs, double* y, double* h2_io, double* resn, int64_t* k_done,
                   int* reason);

/* The Givens rotation the C host loops (kh_gmres_cycle, kh_minres_cycle) generate: by default the reference BLAS formula;
 * a caller whose own per-step loop uses its BLAS library's drotg (krypy/utils.py:426-427 through scipy.linalg.blas)
 * hands that function over - Fortran convention drotg(a, b, c, s), a and b overwritten - and gets the same bits from
 * both loops.  NULL restores the built-in formula. */
int kh_ctx_set_rotg(kh_ctx ctx, void (*drotg)(double* a, double* b, double* c, double* s));

/* A run of MINRES iterations in ONE call (krypy/linsys.py:791-853): Lanczos steps k0 .. k_stop-1 with look-ahead on the
 * device (kh_arnoldi_step_begin / _end, slots k mod 4; the basis may be a sliding WINDOW whose column 0 is logical
 * column `base`), and on the host in C what the reference does between two steps: the symmetric fill H[k-1,k] = H[k,k-1],
 * the new column through the two remembered Givens rotations and its own, the rotated right-hand side, and the vector
 * recurrences z = (v_k - R0 W0 - R1 W1)/R2; W <- [W1, z]; yk += y0 z as a DEFERRED update (kh_minres_update_deferred:
 * the next Lanczos launch carries it; call kh_minres_flush before yk is read).  H is (ldh >= k_stop) row-major like the
 * reference's array and receives the three entries of each recorded column.  st[0..3] = the two remembered rotations
 * (c, s) older first, st[4] = how many of them exist (0, 1, 2), st[5..6] = the rotated right-hand side (y[0], y[1]);
 * *wslot_io = the column of Wm that holds W0; *h2_io the running squared Frobenius norm of H; *enq_io as for
 * kh_gmres_cycle.  resn[k] = |y[k+1]| of every recorded step.  Stop reasons and *k_done as for kh_gmres_cycle. */
int kh_minres_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t k0, int64_t k_stop,
                    int64_t k_last, int64_t base, int64_t* enq_io, double tol, double bnorm, double* H, int64_t ldh,
                    kh_vec Wm, int* wslot_io, kh_vec YK, int64_t ycol, double* st, double* h2_io, double* resn,
                    int64_t* k_done, int* reason);

/* r = b - A x fused with its squared norm: R[:, rcol] = B[:, bcol] - A X[:, xcol]; *nrm = ||r||_2
 * (LinearSystem.get_residual linsys.py:156-160 for M = Ml = identity) */
int kh_residual(kh_ctx ctx, kh_mat A, kh_vec B, int64_t bcol, kh_vec X, int64_t xcol, kh_vec R,
                int64_t rcol, double* nrm);

/* One MINRES vector update (linsys.py:844-846):
 *   z = (V[:,k] - r0*W0 - r1*W1)/r2;   W0 <- W1;  W1 <- z;   yk += y0*z
 * Wk holds the two columns W0|W1 addressed through `slot` (the column that is W0 now and
 * receives z): no copies. */
int kh_minres_update(kh_ctx ctx, kh_vec V, int64_t k, kh_vec Wk, int slot, double r0, double r1,
                     double r2, double y0, kh_vec YK, int64_t ycol);
/* The same update, DEFERRED: it is carried by the next Lanczos step launch (kh_arnoldi_step_begin with a banded
 * operator: krypy_amd/csrc/lanczos.h - six independent streams in the shadow of that launch's last pass) or, failing
 * that, run by kh_minres_flush / the next kh_minres_update[_deferred] call.  Updates take effect in the order given.
 * Nothing but these entries reads W or yk in between: call kh_minres_flush before yk is used (krypy/linsys.py:844-847:
 * yk is needed only when an iterate is formed). */
int kh_minres_update_deferred(kh_ctx ctx, kh_vec V, int64_t k, kh_vec W, int slot, double r0, double r1, double r2,
                              double y0, kh_vec YK, int64_t ycol);
int kh_minres_flush(kh_ctx ctx);

/* One CG step after Ap is known (linsys.py:655-665), fused:
 *   yk += alpha*p;  r -= alpha*Ap;  z = Md r (or r);  *rho_new = <r, z>
 *   and, when beta_valid, nothing else; the direction update p = z + (rho_new/rho_old) p is
 *   kh_waxpby. */
int kh_cg_update(kh_ctx ctx, double alpha, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol,
                 kh_vec YK, int64_t ycol, kh_vec R, int64_t rcol, kh_mat Md, kh_vec Z, int64_t zcol,
                 double* rho_new);

/* One whole CG iteration (linsys.py:622-665) in one call, one host synchronisation:
 *   p = z + omega*p (skipped when `first`);  Ap = A p;  pAp = <p, Ap>;  alpha = rho / pAp (on the
 *   device);  yk += alpha p;  r -= alpha Ap;  z = Md r (z is r when Md == NULL);  rho_new = <r, z>
 * out[0] = <p, Ap>, out[1] = rho_new, out[2] = sanity word (KH_CG_* bits, 0 = fine): the step length is formed
 * on the device, so a divisor that is not a positive finite number is reported with the scalars; with a step length
 * that is not finite yk and r are left untouched.  omega = rho/rho_prev and rho are the host's values (the host
 * may have replaced rho by an explicit residual, linsys.py:667-669). */
#define KH_CG_NONFINITE_PAP 1     /* <p, Ap> (or the divisor formed from it) is inf / nan */
#define KH_CG_NONPOSITIVE_PAP 2   /* Re <p, Ap> <= 0: not a positive definite operator in this inner product */
#define KH_CG_NONFINITE_RHO 4     /* <r, z> is inf / nan */
#define KH_CG_NEGATIVE_RHO 8      /* <r, z> < 0: the preconditioner is not positive definite */
#define KH_CG_STEP_CLAMPED 16     /* rho / <p, Ap> is not finite although both are (a zero divisor): the device took NO step */
int kh_cg_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol,
               kh_vec YK, int64_t ycol, kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol, int first,
               double omega, double rho, double* out);

/* A run of CG iterations in ONE call (krypy/linsys.py:622-690): iterations k0 .. k_stop-1 of kh_cg_step with what the
 * reference does between two of them on the host in C - omega = rho_k / rho_{k-1}, rho_{k+1} = (sqrt |<r, z>|)^2 as the
 * caller forms it, the convergence test.  rhos[i] = rho after i iterations (in: rhos[k0] and, for k0 > 0, rhos[k0 - 1];
 * out: rhos[k0 + 1 ..]); trace receives six doubles per iteration: rho, d, <p, Ap>, <r, z>, the KH_CG_* sanity word,
 * sqrt(|<r, z>|).  (rho_{k+1} is libm's pow(sqrt |<r, z>|, 2.0), which is what `norm ** 2` on a NumPy scalar evaluates.)
 * Stops  KH_CYCLE_TOL    after the iteration whose sqrt(|<r, z>|) / bnorm <= tol (recorded; the caller finalises it),
 *        KH_CYCLE_CHECK  at an iteration that returned non-finite scalars from finite input (NOT recorded; yk and r are
 *                        as before it, the caller raises with the trace),
 *        KH_CYCLE_LIMIT  at k_stop.        *k_done = iterations recorded (counted from 0). */
int kh_cg_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK, int64_t ycol,
                kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol, int64_t k0, int64_t k_stop, double tol, double bnorm,
                double* rhos, double* trace, int64_t* k_done, int* reason);

/* ---- complex (c128) twin of the hot path ------------------------------------------------- */
/* KryPy's kernels are dtype-generic NumPy (H/V are allocated with the common dtype of A, v, M:
 * utils.py:893-905, complex inner products are X^H Y: utils.py:183).  A complex N-vector block is
 * a REAL kh_vec of length 2N (interleaved re, im), allocated / copied / zeroed / normed / scaled
 * by a real with the entry points above; the entry points below are what is genuinely complex.
 * Complex scalars cross the ABI as (re, im) pairs of doubles. */
/* complex operators; kh_apply dispatches on the handle (X, Y are 2N-real views).  A square complex CSR operator whose
 * entries sit on 5 or 7 well-filled diagonals (a shifted stencil matrix) also gets a diagonal-major copy of (re, im) pairs:
 * the complex Arnoldi / Lanczos step then forms w = A v_k inside its Gram-Schmidt kernel - same bits as the SpMV launch. */
int kh_zcsr_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* indptr,
                   const int32_t* indices, const double* data_re_im, kh_mat* out);
int kh_zdense_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, const double* a_re_im, int64_t lda,
                     kh_mat* out);
int kh_zdiag_upload(kh_ctx ctx, int64_t n, const double* d_re_im, kh_mat* out);
/* Z[:, zcol..] (complex view) = X[:, xcol..] (real) + 0i: NumPy's upcast of a real operand */
int kh_zfrom_real(kh_ctx ctx, kh_vec X, int64_t xcol, kh_vec Z, int64_t zcol, int64_t ncols);
/* out[2j], out[2j+1] = <V[:, j0+j], W[:, wcol]> = conj(v)^T w   (utils.py:183, ncols <= 512) */
int kh_zdot_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, kh_vec W, int64_t wcol, double* out);
/* W[:, wcol] -= sum_j h[j] V[:, j0+j], complex h, left to right (utils.py:1029) */
int kh_zaxpy_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* h, kh_vec W,
                   int64_t wcol);
/* Y[:, y0..y0+nc) = beta*Y + X[:, x0..x0+k) C, C complex k x nc row-major (linsys.py:947) */
int kh_zgemm_nn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t k, const double* C, int64_t nc, double beta,
                kh_vec Y, int64_t y0);
/* z = alpha*x + beta*y with complex alpha, beta */
int kh_zwaxpby(kh_ctx ctx, kh_vec Z, int64_t zcol, const double alpha[2], kh_vec X, int64_t xcol,
               const double beta[2], kh_vec Y, int64_t ycol);
/* Arnoldi.advance for complex data (utils.py:954-1048; mgs / dmgs / lanczos / panel CGS, Euclidean
 * inner product, no preconditioner): hcol_out receives k+2 complex numbers, the last (H[k+1,k], 0). */
int kh_zarnoldi_step(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t wcol, int64_t k, int64_t start,
                     int sweeps, int gs_mode, const double h_km1[2], double* hcol_out);
/* ... and split for look-ahead like kh_arnoldi_step_begin: collect the column with
 * kh_arnoldi_step_end(ctx, slot, 2*(k+2), out).  h_km1[0] = NaN takes the Lanczos coefficient from
 * the previous slot's device-side column. */
int kh_zarnoldi_step_begin(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t wcol, int64_t k,
                           int64_t start, int sweeps, int gs_mode, const double h_km1[2], int slot);
/* the complex twins of kh_proj_create / kh_proj_apply_complement (utils.py:604-627 with complex data) and the
 * complex step with the deflation projector inside it (deflation.py:127-143): T = R^{-1} Q^H and WRH = WR^H are
 * d x d row-major arrays of (re, im) pairs (NULL: identity); the step returns <U, A v_k> (d complex numbers) behind
 * the H column, like the real one */
int kh_zproj_create(kh_ctx ctx, kh_vec W, kh_vec V, int64_t d, const double* T, const double* WRH, int iterations,
                    kh_proj* out);
int kh_zproj_apply_complement(kh_ctx ctx, kh_proj p, kh_vec A, int64_t acol, kh_vec Z, int64_t zcol, double* ya_out);
int kh_zarnoldi_step_begin_proj(kh_ctx ctx, kh_mat A, kh_proj proj, kh_vec V, kh_vec W, int64_t wcol, int64_t k,
                                int64_t start, int sweeps, int gs_mode, const double h_km1[2], int slot);
/* The general complex step: Md = the Jacobi preconditioner as a complex diagonal (kh_zdiag_upload) with its second
 * block P (V = Md P, krypy/utils.py:1026-1045), W with two columns then; Md = P = NULL: as above. */
int kh_zarnoldi_step_begin_md(kh_ctx ctx, kh_mat A, kh_proj proj, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t wcol,
                              int64_t k, int64_t start, int sweeps, int gs_mode, const double h_km1[2], int slot);
/* Complex kh_minres_update (krypy/linsys.py:844-846): z = (v_k - r0 W0 - r1 W1)/r2; W <- [W1, z]; yk += y0 z with
 * complex coefficients ((re, im) pairs), one pass. */
int kh_zminres_update(kh_ctx ctx, kh_vec V, int64_t k, kh_vec Wk, int slot, const double r0[2], const double r1[2],
                      const double r2[2], const double y0[2], kh_vec YK, int64_t ycol);
/* Complex kh_cg_step (krypy/linsys.py:622-665), one host synchronisation per iteration.  A complex; the vectors are
 * complex blocks (real kh_vec of length 2N); Md NULL or a REAL diagonal of length 2N (each Jacobi entry twice).
 * out[0] = d with step length alpha = rho / d = Re(rho / <p, Ap>), out[1] = <r, z>, out[2..3] = <p, Ap>,
 * out[4] = sanity word (KH_CG_* bits as for kh_cg_step). */
int kh_zcg_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK,
                int64_t ycol, kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol, int first, double omega, double rho,
                double* out);

/* ---- measurement ----------------------------------------------------------------------- */
/* bench.py's roofline numbers: average duration (ms) of `reps` back-to-back launches of one hot
 * kernel, HIP events on the context's stream.  which: 0 Gram-Schmidt link (axpy+dot),
 * 1 multidot<16>, 2 multiaxpy<16>, 3 link with norm tail, 4 scale-and-store, 5 register-resident
 * MGS chain (16 columns x 4 sweeps = 64 links per launch; 6/7: without the grid reduction / the
 * reduction alone), 8 register-resident panel GS over 16 columns (k_cgs_dots + k_reduce_partials +
 * k_cgs_update).  V needs >= 17 columns, W 2 columns. */
int kh_bench_kernel(kh_ctx ctx, int which, kh_vec V, kh_vec W, int reps, double* avg_ms);
/* The solver's own launch sequence (what Gmres._solve / kh_gmres_cycle enqueue, utils.py:954-1048 once per k):
 * `reps` times the Arnoldi steps k = 0 .. m-1 on V (>= m+1 columns, column 0 a unit vector), one step of look-ahead,
 * columns fetched in order.  avg_step_ms = HIP-event time / (reps * m).  With a banded operator and vectors that
 * fit the register file each step is ONE launch of the fused chain kernel: its average duration over k+1 = 1 .. m
 * Gram-Schmidt links is the `roofline.avg_launch_ms` of bench.py. */
int kh_bench_arnoldi(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t m, int gs_mode, int reps,
                     double* avg_step_ms);

#ifdef __cplusplus
}
#endif
#endif /* KRYLOV_HIP_H */
