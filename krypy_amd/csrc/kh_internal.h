// Internal definitions of libkrylov_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "krylov_hip.h"

namespace kh {

constexpr int BS = 256;          // threads per workgroup of every vector kernel (4 wave64)
constexpr int MAXC = 16;         // widest column panel one multidot / multiaxpy launch handles
constexpr int NB_MAX = 4096;
constexpr int KH_GRAM_NB = 1024; // workgroups of k_gram_mfma (kernels.h)     // upper bound of the reduction grid
constexpr int SCAL_CAP = 8192;   // device scalar slots (panel coefficients, norms)
#define KH_NSLOT 4               // Arnoldi steps that may be in flight (H-column slots)
#define KH_CHAIN_REARM_STEPS 100  // clean per-column steps after which a timed-out chain family is tried again
#define KH_CHAIN_MAX_RECOVERIES 3

extern thread_local std::string g_err;

int fail(int code, const char* fmt, ...);

#define KH_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return kh::fail(e_ == hipErrorOutOfMemory ? KH_ERR_NOMEM : KH_ERR_HIP,        \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                            __FILE__, __LINE__);                                          \
    } while (0)

#define KH_ARG(cond, ...)                                     \
    do {                                                      \
        if (!(cond)) return kh::fail(KH_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define KH_TRY(call)            \
    do {                        \
        int rc_ = (call);       \
        if (rc_ != 0) return rc_; \
    } while (0)

}  // namespace kh

// what kh_arnoldi_step_begin / kh_zarnoldi_step_begin enqueued in an H-column slot: enough to run the
// step again on the per-column kernels when the chain kernel reports a timeout (kh_arnoldi_step_end)
struct kh_step_s {
    int kind = 0;            // 0 = nothing, 1 = real step, 2 = complex step
    kh_mat A = nullptr, Md = nullptr;
    kh_proj proj = nullptr;
    kh_vec V = nullptr, P = nullptr, W = nullptr;
    int64_t wcol = 0, k = 0, start = 0;
    int sweeps = 1, gs_mode = 0;
    double h_km1[2] = {0.0, 0.0};
};

struct kh_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    int ncu = 0;
    int nb = 1024;          // reduction grid (workgroups of the streaming vector kernels)
    int spmv_tile = 2048;   // nnz staged in LDS per SpMV workgroup
    double* part = nullptr; // [MAXC + 4][NB_MAX] partial sums
    double* scal = nullptr; // SCAL_CAP device scalars
    double* hpin = nullptr; // pinned host staging, SCAL_CAP doubles
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* hslot_dev[KH_NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    double* hslot_pin[KH_NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t hev[KH_NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    int64_t hcap = 0;
    // register-resident MGS chain (chain.h)
    int chain_enabled = 1;
    int64_t n_chain = 0, n_chain_lds = 0, n_chain_fused = 0, n_cgs_reg = 0;   // launch counters (kh_ctx_counters)
    int64_t n_chain_recovered = 0;   // Arnoldi steps re-run on the link kernels after a chain timeout
    // A timed-out chain launch switches the chain family off (chain_enabled = 0) and the step is re-run on the
    // per-column kernels.  A timeout is a transient of a shared GPU, so the switch is not for life: after
    // KH_CHAIN_REARM_STEPS clean steps on the per-column kernels, or when a new basis starts (k == 0), the configured
    // value comes back.  The third recovery in one context leaves it off and says so once on stderr.
    int chain_configured = 1;        // what KRYPY_AMD_MGS_CHAIN / kh_ctx_set("chain") asked for
    int chain_recoveries = 0;        // timeouts recovered in this context (kh_ctx_get "chain_recoveries")
    int chain_clean_steps = 0;       // steps begun on the per-column kernels since the last timeout
    int chain_in_recovery = 0;       // kh_arnoldi_step_end is re-running a step: no re-arming from inside it
    int64_t n_chain_rearmed = 0;     // times the chain family was switched on again (kh_ctx_get "n_chain_rearmed")
    int64_t chain_refused_n = -1;    // vector length whose chain launch the runtime refused (occupancy): that shape only
    int64_t n_spmm = 0;     // panel applications of a CSR operator that streamed the matrix once
    int chain_spmv = 1;     // banded operators: w = A v_k in the chain kernel's prologue (KRYPY_AMD_CHAIN_SPMV)
    int chain_pf = 1;       // ... and keep HBM busy through the update phase (k_mgs_chain_pf; KRYPY_AMD_CHAIN_PF)
    int64_t n_chain_pf = 0;
    int chain_onex = 1;     // short vectors: all working workgroups of the chain kernel on one XCD (KRYPY_AMD_CHAIN_ONEX)
    int64_t n_chain_onex = 0;
    int chain_small = 1;    // short vectors without a preconditioner: the column-ring kernel k_mgs_chain_small (KRYPY_AMD_CHAIN_SMALL)
    int64_t n_chain_small = 0;
    // ... and its blocked form: one grid-wide sum per block of 4 columns (chain_blk.h; KRYPY_AMD_CHAIN_BLK)
    int chain_blk = 1;
    int blk_nx = 8;                   // workgroups without rows in front of a blocked launch that is spread over the chip (they
                                      // gather the sums on compute units that carry no column stream; KRYPY_AMD_BLK_NX, 0: none)
    int64_t blk_onex_maxn = 70000;    // vectors longer than this run the blocked kernel spread over the chip, not on one XCD: the
                                      // one XCD's memory port is the bound from there on (15.6k vs 15.1k it/s at 62,500 rows, 13.8k vs
                                      // 15.0k at 80,089, 12.9k vs 14.0k at 10^5; KRYPY_AMD_BLK_ONEX_MAXN)
    int64_t n_chain_blk = 0;
    unsigned long long* blk_gran = nullptr;   // granules + per-XCD totals of the blocked kernel's sums + its Gram table (chain_blk.hip)
    const void* blk_V = nullptr;     // the basis block whose Arnoldi sequence owns the Gram table ...
    int64_t blk_next = -1;           // ... and the step that finds it valid (-1: nobody)
    int blk_kind = 1;                // whose table it is: 1 = k_mgs_chain_blk (entries of the block before + the own block), 2 =
                                     // k_mgs_chain_blk2 (own block only)
    int chain_blk2 = 1;              // KRYPY_AMD_CHAIN_BLK2: the eight-wave blocked kernel (4 ... 6 rows per lane; on N ranks with the
                                     // cross-rank sums inside the launch)
    int blk2_cw = 1;                 // KRYPY_AMD_BLK2_CW: wave 0 of the eight-wave blocked kernel owns no rows where that shape fits (448 lanes with rows)
    int blk2_one = 2;                // the one-block shapes of chain_blk2.h: 0 never, 1 on a communicator only, 2 on one GPU too (default; KRYPY_AMD_BLK2_ONE)
    int gemv_rows = 0;               // rows per wave of k_gemv_dense (0: chosen by size)
    int blk2_cw_maxrows = 7;         // rows per lane up to which the communication-wave shape is used (KRYPY_AMD_BLK2_CW=2: 6, for A / B)
    int64_t n_chain_blk2 = 0;
    // N ranks, slabs beyond the blocked kernel's 2.5 M rows: the register-resident chain kernels with the cross-rank stage inside
    // every grid-wide sum (chain_xr.hip; KRYPY_AMD_CHAIN_XR) - the local basis read once, no all-reduce call in the step
    int chain_long = 1;              // KRYPY_AMD_CHAIN_LONG: 48 rows per lane take k_mgs_chain_long (chain_long.h: a third of every column
                                     // stays on the chip between its dot and its update) instead of k_mgs_chain<48> (both reads from memory)
    int64_t n_chain_long = 0;
    int gram_mfma = 1;               // KRYPY_AMD_GRAM_MFMA: kh_gemm_tn with 2 ... columns on the right takes k_gram_mfma (both blocks read
                                     // once per 16 x 16 tile) instead of one k_multidot launch and one host round trip per column
    int64_t n_gram_mfma = 0;         // 16 x 16 tiles computed by it
    int64_t n_panel_gemm = 0;        // passes of k_panel_gemm_mfma (kh_gemm_nn with 2 ... 16 output columns; the same switch)
    double* gram_part = nullptr;     // [256][KH_GRAM_NB] workgroup partials of k_gram_mfma (allocated at first use)
    int64_t n_zspmv_dia = 0;         // products of a banded complex operator through its diagonal-major copy (zpath.h: k_zspmv_dia)
    int chain_xr = 1;
    int chain_xr_cus = 0;            // tests: the compute units the shape is chosen for (0: all; two processes share one device)
    int64_t n_chain_xr = 0;
    int64_t blk2_refused_n = -1;
    int64_t n_blk_rowless = 0;       // blocked launches with workgroups without rows in front (chain_blk.h, BlkBufs::nx)
    int64_t blk_refused_n = -1;      // vector length whose blocked launch was refused (occupancy ...): not tried - nor its table rebuilt - again
    int64_t n_blk_rebuild = 0;       // times the Gram table was rebuilt from the basis (a sequence's first blocked step)
    // reference-order Gram-Schmidt with one reduction per step on N ranks (krylov_hip.hip: try_lowsync_mgs; KRYPY_AMD_MGS_LOWSYNC)
    int mgs_lowsync = 1;
    double* ls_tab = nullptr;        // [LS_MAXCOL][LS_MAXCOL] Gram table of the current Arnoldi sequence: [m][j] = <v_m, v_j>, m < j
    const void* ls_V = nullptr;      // the basis block whose sequence owns the table ...
    int64_t ls_next = -1;            // ... and the step that finds columns 0 .. k-1 of it valid
    int64_t n_lowsync = 0, n_ls_rebuild = 0;
    // N ranks: (local slab length, longest slab of the run) pairs announced by the host layer - the form is chosen for the longest
    int64_t ls_rows_local[4] = {-1, -1, -1, -1}, ls_rows_max[4] = {0, 0, 0, 0};
    int ls_rows_n = 0;
    // the deflation projector with the vector in registers, one launch (proj_reg.h; KRYPY_AMD_PROJ_REG)
    int proj_reg = 1;
    unsigned long long* proj_gran = nullptr;   // granules, per-XCD totals and leader stamps of its 16-value sums
    unsigned proj_epoch = 1;
    int64_t n_proj_reg = 0;
    int* proj_err = nullptr;         // its OWN device error word (a timed-out sum), mirrored to pinned memory behind every launch:
    int* proj_err_pin = nullptr;     // kh_arnoldi_step_end / kh_proj_apply_complement re-run without this kernel
    int proj_fault = 0;              // tests: the next launch behaves like a timed-out one (kh_ctx_set "proj_fault")
    int64_t n_proj_recovered = 0;
    int proj_reg_why = -1;           // why proj_reg_apply declined last time (1: switches / shape of the projector, 2: short vector, 3: blocks
                                     // not padded alike, 4: the launch failed; 0: it ran)
    int proj_panel = 1;              // ... or, on N ranks, its passes through the register-resident panel kernels (KRYPY_AMD_PROJ_PANEL)
    int64_t n_proj_panel = 0;        // sweeps that took them
    int64_t n_cycle_steps = 0;   // GMRES iterations recorded by kh_gmres_cycle
    int64_t n_minres_cycle_steps = 0;   // MINRES iterations recorded by kh_minres_cycle
    int64_t n_cg_cycle_steps = 0;       // CG iterations recorded by kh_cg_cycle
    void (*rotg)(double*, double*, double*, double*) = nullptr;   // the host layer's BLAS drotg (kh_ctx_set_rotg), or NULL
    std::vector<double> cyc_col;        // H-column scratch of the C host loops
    unsigned* onex_ticket = nullptr;   // 256 rotating ticket words
    int lanczos_fused = 1;  // steps with one Gram-Schmidt link: the three-pass kernel of lanczos.h (KRYPY_AMD_LANCZOS_FUSED)
    int64_t n_lanczos_fused = 0;
    int mr_taken = 0;       // the last chain launch carried a MINRES recurrence job (lanczos.h)
    // a MINRES recurrence update waiting for the next Lanczos launch to carry it (kh_minres_update_deferred)
    struct {
        int on = 0;
        kh_vec V = nullptr, W = nullptr, YK = nullptr;
        int64_t vcol = 0, ycol = 0;
        int slot = 0;
        double r0 = 0, r1 = 0, r2 = 0, y0 = 0;
    } mr_pending;
    int64_t n_minres_rides = 0;   // deferred MINRES updates that went along with a Lanczos launch
    int chain_lds = 1;      // park the head of every column in LDS (k_mgs_chain_lds; KRYPY_AMD_CHAIN_LDS)
    int spmv_win = 1;       // CSR-stream kernel: gather x from an LDS window where a row block's columns fit one (KRYPY_AMD_SPMV_WIN)
    int64_t n_spmv_win = 0;
    int spmv_dia = 1;       // use the banded copy of a CSR operator when it has one (kh_ctx_set "spmv_dia")
    unsigned long long* chain_gran = nullptr;
    unsigned long long* chain_xcc = nullptr;   // per-XCD result granules + leader stamps of the grid-wide sums
    int* chain_err = nullptr;        // device error word
    int* chain_err_pin[KH_NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    // completion tags: the last kernel of a chained step writes done_seq[slot] to the slot's pinned word behind the H
    // column and the host polls it - an event record between two launches costs 2.8 us of queue time
    // (tools/probe/gap_probe.hip); steps whose last kernel does not write the tag keep the event
    int* done_pin[KH_NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    int done_seq[KH_NSLOT] = {0, 0, 0, 0};
    bool wait_tag[KH_NSLOT] = {false, false, false, false};
    int done_counter = 0;
    int tag_wait = 1;       // KRYPY_AMD_TAG_WAIT=0: events for every step
    int64_t n_tag_waits = 0;
    unsigned chain_epoch = 1;
    int64_t n_epoch_wraps = 0;       // times the epoch counter was brought back to 1 (granules zeroed, Gram-table key withdrawn)
    int chain_debug = 0;
    int roctx = 0;          // KRYPY_AMD_ROCTX=1: roctx ranges around the entry points of the hot loop
    int chain_fault = 0;    // tests: the next chain launch reports a timeout and leaves garbage behind
    kh_step_s step[KH_NSLOT];
#ifdef KH_CHAIN_TRACE
    unsigned long long* chain_trace = nullptr;   // diagnostic build: phase stamps of the next chain launch
#endif
    double* cgs_part = nullptr;     // [CGS_MAXCOL][wave partials] of the register-resident panel GS
    // RCCL (resolved lazily with dlopen so that single-GPU runs never load librccl)
    void* comm = nullptr;
    int rank = 0, nranks = 1;
    int force_multi = 0;   // tests: run the multi-rank code path on a 1-rank communicator
    // sharded SpMV: the halo exchange runs on its own stream while the interior rows are multiplied
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_x = nullptr, ev_halo = nullptr;
    int spmv_split = 1;
    int64_t n_spmv_split = 0;
    int halo_loopback = 0;          // tests: a 1-rank communicator exchanges its halo with itself (periodic slab)
    int64_t n_halo_exchange = 0;    // grouped ncclSend / ncclRecv exchanges issued
    int64_t n_halo_xh = 0;          // sharded SpMVs whose halo travelled inside their own launch (kh_mat_xh_*)
    int64_t n_allreduce = 0;        // ncclAllReduce calls issued (kh_ctx_get "n_allreduce")
    double* commbuf = nullptr;  // device staging for host all-reduces
    // ---- xr: sums across the ranks of ONE node without a library call (xr.hip) ----
    // Every rank owns a mailbox in fine-grained device memory, exported with hipIpcGetMemHandle and mapped by its peers
    // (kh_xr_export / kh_xr_attach; the handles travel over the launcher's rendezvous).  A sum is then ONE small kernel:
    // system-scope stores of tagged 8-byte granules into every peer's mailbox (over xGMI), a poll of the own mailbox,
    // the contributions added in rank order - the same bits on every rank - instead of an ncclAllReduce kernel per panel.
    unsigned long long* xr_box = nullptr;              // my mailbox
    unsigned long long* xr_peer[16] = {nullptr};       // every rank's mailbox as mapped here (xr_peer[rank] == xr_box)
    int xr_rank = 0, xr_nranks = 0;                    // as attached (0: not attached)
    int xr_on = 0;                                     // kh_ctx_set "xr": all-reduces take this path (the host layer switches it
                                                       // on after EVERY rank has attached: the choice must be the same everywhere)
    int xr_own_comm = 0;                               // no RCCL communicator: rank / nranks come from the attach
    unsigned xr_epoch = 1;                             // one per exchange kernel; tags the granules
    int* xr_err_pin = nullptr;                         // mapped pinned word a timed-out exchange writes (checked where the host syncs)
    int64_t xr_timeout_ms = 0;                         // how long a gather waits for a peer (0: KRYPY_AMD_XR_TIMEOUT_S or 60 s)
    int64_t n_xr = 0;                                  // exchanges issued
    int64_t n_xr_fused = 0;                            // ... of which in the same launch as the reduction of the partial sums
};

struct kh_vec_s {
    kh_ctx ctx;
    int64_t n, ncols, ld;
    double* d;
    double* col(int64_t j) const { return d + j * ld; }
};

enum { KH_MAT_CSR = 0, KH_MAT_DENSE = 1, KH_MAT_DIAG = 2,
       KH_MAT_ZCSR = 3, KH_MAT_ZDENSE = 4, KH_MAT_ZDIAG = 5 };   // complex operators (zpath.h)

struct kh_mat_s {
    kh_ctx ctx;
    int kind;
    int64_t n_rows, n_cols, nnz;
    // CSR
    int32_t* indptr = nullptr;
    int32_t* indices = nullptr;
    double* data = nullptr;
    int32_t* rowblk = nullptr;  // row-block boundaries of the CSR-stream kernel
    int64_t rows_max = 0;       // a block-row shard: the LONGEST slab of the run (kh_mat_set_rows_max; 0: not announced) - kernel choices
                                // that change the pattern of cross-rank sums are made for it, so that every rank decides alike
    int32_t* blkwin = nullptr;  // [nblk][cmin, span]: the columns a row block touches (k_spmv_stream<.., WIN>)
    int win_cap = 0;            // LDS window (entries of x) of that kernel for this operator; 0: not worth it
    int nblk = 0;
    int tile = 0;
    double* part = nullptr;     // max(nblk, dia_nblk) partial sums for the fused dot / norm epilogues
    // banded copy of a CSR operator whose entries sit on <= KH_DIA_MAX diagonals (k_spmv_dia)
    double* dia = nullptr;      // [dia_nd][dia_ld], 0.0 = no entry
    int64_t dia_ld = 0;
    int dia_nd = 0, dia_nblk = 0, dia_rpt = 0;
    int dia_off[32] = {0};
    // the same for a complex CSR operator (round 4): zdia[d][i] as (re, im) pairs, leading dimension in complex entries;
    // dia_nd / dia_off describe it; used by the complex chain kernels' prologue (the stand-alone SpMV stays CSR-stream)
    double* zdia = nullptr;
    int64_t zdia_ld = 0;
    // dense
    double* a = nullptr;
    int64_t lda = 0;
    // diag
    double* diag = nullptr;
    // halo of a block-row shard
    int64_t nsend_prev = 0, nsend_next = 0, nrecv_prev = 0, nrecv_next = 0;
    double* ghost = nullptr;    // nrecv_prev + nrecv_next doubles
    double* ghost_panel = nullptr;   // [ghost_panel_cols][nrecv_prev + nrecv_next]: ghost entries of a block of vectors (sharded SpMM)
    int64_t ghost_panel_cols = 0;
    // row blocks [b0, b1) of the banded / the CSR-stream kernel touch no ghost column (interior of the slab)
    int dia_b0 = 0, dia_b1 = 0, csr_b0 = 0, csr_b1 = 0;
    // xh: the halo through IPC-mapped mailboxes (xr.hip, kh_mat_xh_*): the banded SpMV itself stores this slab's boundary rows
    // into the neighbours' ghost granules and polls its own - ONE launch per sharded SpMV, no RCCL kernel, no second stream
    unsigned long long* xh_box = nullptr;      // my ghost granules: [2 parities][nrecv_prev + nrecv_next][lo, hi]
    unsigned long long* xh_prev = nullptr;     // the previous / next rank's box as mapped here (loopback: my own)
    unsigned long long* xh_next = nullptr;
    int64_t xh_prev_ng = 0, xh_next_ng = 0;    // ghost entries of the neighbours' boxes (the stride of a parity)
    int64_t xh_prev_off = 0;                   // where my first rows go in the previous rank's box: behind ITS ghosts from its prev
    unsigned xh_epoch = 1;                     // one per sharded SpMV of this operator; tags the granules
    int xh_on = 0;                             // the host layer switches it on after EVERY rank has attached (kh_mat_xh_enable)
    int xh_self = 0;                           // loopback: the neighbours' boxes are my own (nothing to close)
};

// true when reductions must be all-reduced / halos exchanged (several ranks, or a 1-rank
// communicator in forced mode: the multi-rank code path is then testable on a single GPU)
static inline bool kh_multi(const kh_ctx_s* ctx) { return ctx->nranks > 1 || ctx->force_multi; }

struct kh_proj_s {
    kh_ctx ctx;
    kh_vec W, V;
    int64_t d;
    int iterations;
    int cplx = 0;            // W, V, T, WRH and the coefficient buffers hold complex numbers (kh_zproj_create)
    double* T = nullptr;     // d x d row-major, device (nullptr: identity)
    double* WRH = nullptr;   // d x d row-major, device (nullptr: identity)
    double* c0 = nullptr;    // d
    double* c1 = nullptr;    // d
    double* ya = nullptr;    // d
};

namespace kh {
// comm.hip
int comm_allreduce_dev(kh_ctx ctx, double* dev, int64_t count);
int comm_halo_exchange(kh_ctx ctx, kh_mat A, const double* x, hipStream_t stream, int width = 1);
int comm_halo_exchange_panel(kh_ctx ctx, kh_mat A, const double* X, int64_t ldx, int64_t ncols, hipStream_t stream);
// xr.hip
constexpr int XR_MAXV = 512;        // values per exchange kernel (larger panels go in chunks)
constexpr int XR_MAXRANKS = 16;
int xr_allreduce_dev(kh_ctx ctx, double* dev, int64_t count);
// out[c] = sum over the ranks of the fixed-order sum of part[c * pstride + 0 .. nb): k_reduce_partials (mode 0) and the
// all-reduce in ONE launch when the xr transport is on and count <= XR_MAXV; otherwise the reduction launch followed by
// comm_allreduce_dev.  `multi` false: the reduction alone.
int reduce_partials_allreduce(kh_ctx ctx, const double* part, int nb, int pstride, double* out, int count, bool multi);
void xr_free(kh_ctx ctx);
void xh_free(kh_mat A);
int xr_check(kh_ctx ctx);           // KH_ERR_COMM when an exchange has timed out since the last look
// krylov_hip.hip
int dia_rebuild_for_halo(kh_ctx ctx, kh_mat A);
// chain_blk.hip
struct ChainArgs;
hipError_t chain_blk_launch(kh_ctx ctx, int r2, int G, bool onex, bool padded, int fnd, ChainArgs& a, const void* V, int* nsums);
double* chain_blk_table(kh_ctx ctx);
bool chain_blk_shape_ok(int r2, int G, const ChainArgs& a, int fnd);
#ifndef KH_BLK_BC_CFG
#define KH_BLK_BC_CFG 4
#endif
constexpr int KH_BLK_BC = KH_BLK_BC_CFG;   // (= BLK_BC of chain_blk.h) columns per block
constexpr int KH_BLK_TABCOLS = 4096;       // (= BLK_TABCOLS of chain_blk.h: static_assert there) basis columns the Gram table has rows for
constexpr int KH_BLK_TW = 2 * KH_BLK_BC;   // (= BLK_TW) entries per row of the Gram table: the block before the column's, then its own
// an entry point writes to block v: the Gram table of an Arnoldi sequence on it (chain_blk.hip) is no longer vouched for
static inline void chain_blk_touch(kh_ctx ctx, const void* v) {
    if (ctx->blk_V == v) ctx->blk_next = -1;
    if (ctx->ls_V == v) ctx->ls_next = -1;       // (the one-reduction form's table, krylov_hip.hip: same rule)
}
hipError_t chain_blk_reset(kh_ctx ctx);
// chain_blk2.hip: the eight-wave blocked kernel with the cross-rank stage
// one_slot: may the shapes with ONE block in registers (8 ... 11 rows per lane, up to 2.5 M rows) be offered
bool chain_blk2_shape(kh_ctx ctx, int64_t n, int* r2_out, int* g_out, int* cw_out = nullptr, bool one_slot = false);
int chain_blk2_step(kh_ctx ctx, kh_vec V, const double* w, int64_t wld, int64_t k, double* hdev, int slot, double* hpin, int hcount,
                    bool multi);
// chain_xr.hip: the register-resident chain kernels (16 ... 56 rows per lane) with the cross-rank stage in every sum
bool chain_xr_shape(kh_ctx ctx, int64_t n, int* r2_out, int* g_out);
int chain_xr_step(kh_ctx ctx, kh_vec V, const double* w, int64_t wld, int64_t k, double* hdev, int slot, double* hpin, int hcount);
// krylov_hip.hip: the epoch counter of the grid-wide sums brought back to 1 when it nears its wrap; <V[:, j0 .. j0+ncols), w> on the device
int chain_epoch_check(kh_ctx ctx);
int dot_panel_raw(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* w, double* out_dev);
void chain_blk_free(kh_ctx ctx);
// proj_reg.hip
int proj_reg_apply(kh_ctx ctx, kh_proj p, double* z, int64_t zld, int r2, int G, double* ya_dev);
void proj_reg_free(kh_ctx ctx);
}  // namespace kh
