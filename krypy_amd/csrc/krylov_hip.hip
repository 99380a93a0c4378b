// libkrylov_hip: host side of the C ABI declared in include/krylov_hip.h.
// Launch logic only; the kernels live in kernels.h.  gfx950 only, no compatibility paths.
#include <stdarg.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "kernels.h"
#include "chain.h"
#include "chain_long.h"
#include "lanczos.h"
#include "krylov_steps.h"
#include <stdlib.h>

extern "C" {
static int dot_panel_dev(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* w, double* out_dev, int rmode);
}

namespace kh {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    // a refused allocation stays behind as the runtime's "last error": the host layer frees what it was merely holding on
    // to and allocates again (_hip.py: _with_memory), and the next launch check (hipGetLastError) would report the stale
    // "out of memory" of the attempt that was recovered from (seen at N = 10^8: config 5 on one device)
    if (code == KH_ERR_NOMEM) (void)hipGetLastError();
    return code;
}

// ---- optional roctx ranges (rocprofv3 --marker-trace): KRYPY_AMD_ROCTX=1 ----------------------
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
};
static Roctx g_roctx;
void roctx_push(kh_ctx ctx, const char* fmt, long long a, long long b) {
    if (!ctx->roctx) return;
    if (!g_roctx.tried) {
        g_roctx.tried = true;
        void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
            g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        }
    }
    if (g_roctx.push) {
        char buf[96];
        snprintf(buf, sizeof(buf), fmt, a, b);
        g_roctx.push(buf);
    }
}
void roctx_pop(kh_ctx ctx) {
    if (ctx->roctx && g_roctx.pop) g_roctx.pop();
}

int grid_for(kh_ctx ctx, int64_t n) {
    // enough workgroups to cover n/2 double2 elements, capped at the fixed reduction grid
    int64_t need = ((n >> 1) + BS - 1) / BS;
    if (need < 1) need = 1;
    return (int)std::min<int64_t>(need, ctx->nb);
}

// partial-sum slots inside ctx->part
double* part_slot(kh_ctx ctx, int slot) { return ctx->part + (int64_t)slot * NB_MAX; }

// H-column slots: up to NSLOT Arnoldi steps can be in flight on the stream (the host processes
// step k's column while steps k+1.. already run), each with its own device column, pinned copy
// and completion event.
int ensure_hcap(kh_ctx ctx, int64_t need) {
    if (need <= ctx->hcap) return 0;
    // a larger basis begins a step: the slots grow, and a column that is complete but not fetched yet
    // (another basis' look-ahead step) moves into the new buffers - nothing in flight is dropped
    KH_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t cap = std::max<int64_t>(need * 2, 1024);
    for (int s = 0; s < KH_NSLOT; ++s) {
        double* dev = nullptr;
        double* pin = nullptr;
        KH_HIP(hipMalloc(&dev, sizeof(double) * cap));
        hipError_t e = hipHostMalloc(&pin, sizeof(double) * cap, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipFree(dev);
            return fail(KH_ERR_NOMEM, "ensure_hcap: hipHostMalloc(%lld doubles): %s", (long long)cap,
                        hipGetErrorString(e));
        }
        if (ctx->hslot_dev[s]) {
            e = hipMemcpy(dev, ctx->hslot_dev[s], sizeof(double) * ctx->hcap, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {       // (the old buffers stay in place: slots already moved are simply larger)
                (void)hipFree(dev);
                (void)hipHostFree(pin);
                return fail(KH_ERR_HIP, "ensure_hcap: hipMemcpy: %s", hipGetErrorString(e));
            }
            (void)hipFree(ctx->hslot_dev[s]);
        }
        if (ctx->hslot_pin[s]) {
            memcpy(pin, ctx->hslot_pin[s], sizeof(double) * ctx->hcap);
            (void)hipHostFree(ctx->hslot_pin[s]);
        }
        ctx->hslot_dev[s] = dev;
        ctx->hslot_pin[s] = pin;
        if (!ctx->hev[s]) KH_HIP(hipEventCreateWithFlags(&ctx->hev[s], hipEventDisableTiming));
    }
    ctx->hcap = cap;
    return 0;
}

int check_vec(kh_vec v, int64_t col, int64_t ncols, const char* what) {
    KH_ARG(v != nullptr, "%s: NULL vector handle", what);
    KH_ARG(col >= 0 && ncols >= 0 && col + ncols <= v->ncols,
           "%s: columns [%lld, %lld) out of range (ncols=%lld)", what, (long long)col,
           (long long)(col + ncols), (long long)v->ncols);
    return 0;
}

int fetch_scalars(kh_ctx ctx, const double* dev, int64_t count, double* out) {
    KH_ARG(count <= SCAL_CAP, "fetch_scalars: %lld > capacity", (long long)count);
    KH_HIP(hipMemcpyAsync(ctx->hpin, dev, count * sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
    KH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->hpin, count * sizeof(double));
    return xr_check(ctx);          // (a cross-rank sum that timed out upstream of these scalars: an error, xr.hip)
}

int push_scalars(kh_ctx ctx, const double* host, int64_t count, double* dev) {
    // pinned staging is reused: wait for earlier consumers of hpin first
    KH_ARG(count <= SCAL_CAP, "push_scalars: %lld > capacity", (long long)count);
    KH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(ctx->hpin, host, count * sizeof(double));
    KH_HIP(hipMemcpyAsync(dev, ctx->hpin, count * sizeof(double), hipMemcpyHostToDevice,
                          ctx->stream));
    return 0;
}

// ---- panel helpers ------------------------------------------------------------------------
template <int C>
static void launch_multidot(kh_ctx ctx, int64_t n, const ColPtrs& cp, const double* w,
                            double* part) {
    hipLaunchKernelGGL((k_multidot<C>), dim3(grid_for(ctx, n)), dim3(BS), 0, ctx->stream, n, cp, w,
                       part, NB_MAX);
}

// partial sums of <V[:, j0+c], w>, c < nc <= MAXC, into part slots 0..nc-1
static int multidot_chunk(kh_ctx ctx, kh_vec V, int64_t j0, int nc, const double* w) {
    ColPtrs cp;
    int done = 0;
    const int64_t n = V->n;
    while (done < nc) {
        int c = nc - done;
        c = c >= 16 ? 16 : c >= 8 ? 8 : c >= 4 ? 4 : c >= 2 ? 2 : 1;
        for (int i = 0; i < c; ++i) cp.c[i] = V->col(j0 + done + i);
        double* part = part_slot(ctx, done);
        switch (c) {
            case 16: launch_multidot<16>(ctx, n, cp, w, part); break;
            case 8: launch_multidot<8>(ctx, n, cp, w, part); break;
            case 4: launch_multidot<4>(ctx, n, cp, w, part); break;
            case 2: launch_multidot<2>(ctx, n, cp, w, part); break;
            default: launch_multidot<1>(ctx, n, cp, w, part); break;
        }
        done += c;
    }
    KH_HIP(hipGetLastError());
    return 0;
}

template <int C, int TAIL, int BETA>
static void launch_multiaxpy(kh_ctx ctx, int64_t n, const ColPtrs& cp, const double* coef,
                             double sign, double beta, double* w, const double* dg, double* mw,
                             double* part) {
    hipLaunchKernelGGL((k_multiaxpy<C, TAIL, BETA>), dim3(grid_for(ctx, n)), dim3(BS), 0,
                       ctx->stream, n, cp, coef, sign, beta, w, dg, mw, part);
}

template <int TAIL, int BETA>
static void dispatch_multiaxpy(kh_ctx ctx, int c, int64_t n, const ColPtrs& cp, const double* coef,
                               double sign, double beta, double* w, const double* dg, double* mw,
                               double* part) {
    switch (c) {
        case 16: launch_multiaxpy<16, TAIL, BETA>(ctx, n, cp, coef, sign, beta, w, dg, mw, part); break;
        case 8: launch_multiaxpy<8, TAIL, BETA>(ctx, n, cp, coef, sign, beta, w, dg, mw, part); break;
        case 4: launch_multiaxpy<4, TAIL, BETA>(ctx, n, cp, coef, sign, beta, w, dg, mw, part); break;
        case 2: launch_multiaxpy<2, TAIL, BETA>(ctx, n, cp, coef, sign, beta, w, dg, mw, part); break;
        default: launch_multiaxpy<1, TAIL, BETA>(ctx, n, cp, coef, sign, beta, w, dg, mw, part); break;
    }
}

// w = beta*w - sign * sum_c coef_dev[c] * X[:, j0+c]   over nc columns (any nc >= 1), in order.
// tail (T_NONE/T_NRM/T_NRM_DIAG) is fused into the last chunk and lands in SLOT_NRM.
static int multiaxpy_cols(kh_ctx ctx, kh_vec X, int64_t j0, int64_t nc, const double* coef_dev,
                          double sign, double beta, double* w, int tail, const double* dg,
                          double* mw) {
    const int64_t n = X->n;
    int64_t done = 0;
    ColPtrs cp;
    while (done < nc) {
        int64_t left = nc - done;
        int c = left >= 16 ? 16 : left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
        for (int i = 0; i < c; ++i) cp.c[i] = X->col(j0 + done + i);
        const bool last = (done + c == nc);
        const int t = last ? tail : T_NONE;
        const int b = (done > 0) ? 1 : (beta == 1.0 ? 1 : (beta == 0.0 ? 0 : 2));
        double* part = part_slot(ctx, SLOT_NRM);
        const double* cf = coef_dev + done;
#define KH_MA(T, B) dispatch_multiaxpy<T, B>(ctx, c, n, cp, cf, sign, beta, w, dg, mw, part)
        if (t == T_NONE) {
            if (b == 1) KH_MA(T_NONE, 1); else if (b == 0) KH_MA(T_NONE, 0); else KH_MA(T_NONE, 2);
        } else if (t == T_NRM) {
            if (b == 1) KH_MA(T_NRM, 1); else if (b == 0) KH_MA(T_NRM, 0); else KH_MA(T_NRM, 2);
        } else {
            if (b == 1) KH_MA(T_NRM_DIAG, 1); else if (b == 0) KH_MA(T_NRM_DIAG, 0); else KH_MA(T_NRM_DIAG, 2);
        }
#undef KH_MA
        done += c;
    }
    KH_HIP(hipGetLastError());
    return 0;
}

// ---- operator application -------------------------------------------------------------------
// which row blocks of the operator a launch covers: all of them, or - for a shard whose halo is still on its
// way - the blocks [lo, hi) of interior rows / the blocks outside (boundary rows); `off` = first partial-sum slot
struct SpmvRange {
    int grid = -1;                 // -1: every block
    int blk_lo = 0x7fffffff, blk_skip = 0, part_off = 0;
    static SpmvRange interior(int lo, int hi) { SpmvRange r; r.grid = hi - lo; r.blk_lo = 0; r.blk_skip = lo; r.part_off = 0; return r; }
    static SpmvRange boundary(int lo, int hi, int nblk) {
        SpmvRange r; r.grid = lo + (nblk - hi); r.blk_lo = lo; r.blk_skip = hi - lo; r.part_off = hi - lo; return r;
    }
};

template <int EPI, int ITEMS>
static void launch_spmv_items(kh_ctx ctx, kh_mat A, const double* x, double* y, const double* aux, const SpmvRange& rg) {
    const size_t lds = (size_t)A->tile * sizeof(double);
    const int grid = rg.grid < 0 ? A->nblk : rg.grid;
    if (grid == 0) return;
    if (ctx->spmv_win && A->win_cap > 0 && A->blkwin != nullptr && A->nrecv_prev + A->nrecv_next == 0) {
        // (no ghost columns: a window is a run of x itself)
        // tile 4096 + a 4096-entry window is 64 KB (+ the 64 B of static LDS): above what a kernel gets without asking
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv_stream<EPI, ITEMS, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4096 + 4096) * sizeof(double)));
            (void)hipGetLastError();
            attr_done = true;
        }
        hipLaunchKernelGGL((k_spmv_stream<EPI, ITEMS, true>), dim3(grid), dim3(BS), lds + (size_t)A->win_cap * sizeof(double),
                           ctx->stream, A->indptr, A->indices, A->data, A->rowblk, A->nblk, A->tile, A->n_cols, x, A->ghost, y,
                           aux, A->part, rg.blk_lo, rg.blk_skip, rg.part_off, A->blkwin, A->win_cap);
        ctx->n_spmv_win += 1;
        return;
    }
    hipLaunchKernelGGL((k_spmv_stream<EPI, ITEMS>), dim3(grid), dim3(BS), lds, ctx->stream, A->indptr,
                       A->indices, A->data, A->rowblk, A->nblk, A->tile,
                       A->n_cols - A->nrecv_prev - A->nrecv_next, x, A->ghost, y, aux, A->part, rg.blk_lo, rg.blk_skip,
                       rg.part_off);
}

template <int EPI, int ND, int RPT, bool HALO>
static void launch_dia_nd(kh_ctx ctx, kh_mat A, const DiaOffs& o, const double* x, double* y, const double* aux,
                          const SpmvRange& rg) {
    const int grid = rg.grid < 0 ? A->dia_nblk : rg.grid;
    if (grid == 0) return;
    if constexpr (HALO) {
        if (A->xh_on) {
            // the halo inside this launch (xr_dev.h): every block of the slab, one epoch per SpMV of this operator
            XhArgs xh;
            xh.mine = A->xh_box;
            xh.prev = A->xh_prev != nullptr ? A->xh_prev : A->xh_box;
            xh.next = A->xh_next != nullptr ? A->xh_next : A->xh_box;
            xh.my_ng = A->nrecv_prev + A->nrecv_next;
            xh.prev_ng = A->xh_prev_ng;
            xh.next_ng = A->xh_next_ng;
            xh.prev_off = A->xh_prev_off;
            xh.nsend_prev = A->xh_prev != nullptr ? (int)A->nsend_prev : 0;
            xh.nsend_next = A->xh_next != nullptr ? (int)A->nsend_next : 0;
            xh.ilo = A->dia_b0 < A->dia_b1 ? A->dia_b0 : 0;
            xh.ihi = A->dia_b0 < A->dia_b1 ? A->dia_b1 : 0;
            xh.epoch = A->xh_epoch++;
            xh.timeout_ticks = (long long)(ctx->xr_timeout_ms > 0 ? ctx->xr_timeout_ms : 60000) * 100000ll;
            void* dp = nullptr;
            (void)hipHostGetDevicePointer(&dp, ctx->xr_err_pin, 0);
            xh.err = static_cast<int*>(dp);
            hipLaunchKernelGGL((k_spmv_dia<EPI, ND, RPT, true, true>), dim3(grid), dim3(BS), 0, ctx->stream, o, A->dia,
                               A->dia_ld, A->n_rows, A->dia_nblk, x, A->ghost, (int)A->nrecv_prev, (int)A->nrecv_next, y,
                               aux, A->part, rg.blk_lo, rg.blk_skip, rg.part_off, xh);
            return;
        }
    }
    hipLaunchKernelGGL((k_spmv_dia<EPI, ND, RPT, HALO>), dim3(grid), dim3(BS), 0, ctx->stream, o, A->dia,
                       A->dia_ld, A->n_rows, A->dia_nblk, x, A->ghost, (int)A->nrecv_prev, (int)A->nrecv_next, y,
                       aux, A->part, rg.blk_lo, rg.blk_skip, rg.part_off, XhArgs());
}

template <int EPI>
static void launch_dia(kh_ctx ctx, kh_mat A, const double* x, double* y, const double* aux, const SpmvRange& rg) {
    DiaOffs o;
    o.nd = A->dia_nd;
    for (int d = 0; d < KH_DIA_MAX; ++d) o.off[d] = d < A->dia_nd ? A->dia_off[d] : 0;
    // rows per workgroup = 2 * BS * RPT (fixed at upload: dia_ld covers whole workgroups)
#define KH_DIA_RPT(R, H)                                                            \
    switch (A->dia_nd) {                                                            \
        case 3: launch_dia_nd<EPI, 3, R, H>(ctx, A, o, x, y, aux, rg); break;           \
        case 5: launch_dia_nd<EPI, 5, R, H>(ctx, A, o, x, y, aux, rg); break;           \
        case 7: launch_dia_nd<EPI, 7, R, H>(ctx, A, o, x, y, aux, rg); break;           \
        case 9: launch_dia_nd<EPI, 9, R, H>(ctx, A, o, x, y, aux, rg); break;           \
        default: launch_dia_nd<EPI, 0, R, H>(ctx, A, o, x, y, aux, rg); break;          \
    }
    if (A->nrecv_prev + A->nrecv_next > 0) { KH_DIA_RPT(4, true) }      // shard with ghost rows: dia_rpt == 4
    else if (A->dia_rpt == 4) { KH_DIA_RPT(4, false) }
    else if (A->dia_rpt == 2) { KH_DIA_RPT(2, false) }
    else { KH_DIA_RPT(1, false) }
#undef KH_DIA_RPT
}

// the banded kernel serves when the operator has a diagonal-major copy (and nobody switched it off)
static inline bool use_dia(kh_ctx ctx, kh_mat A, const double* y) {
    return A->dia != nullptr && ctx->spmv_dia && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

template <int EPI>
static void launch_spmv(kh_ctx ctx, kh_mat A, const double* x, double* y, const double* aux,
                        const SpmvRange& rg = SpmvRange()) {
    if (use_dia(ctx, A, y)) {
        launch_dia<EPI>(ctx, A, x, y, aux, rg);
        return;
    }
    switch (A->tile / BS) {      // tile is one of 1024 / 2048 / 4096 (kh_ctx_tune)
        case 4: launch_spmv_items<EPI, 4>(ctx, A, x, y, aux, rg); break;
        case 16: launch_spmv_items<EPI, 16>(ctx, A, x, y, aux, rg); break;
        default: launch_spmv_items<EPI, 8>(ctx, A, x, y, aux, rg); break;
    }
}

// y = A x with a dense A.  Rows per wave (kernels.h): four from 16 K rows on - 6.91 against 6.73 TB/s at n = 32768, 1255
// against 1308 us per CG iteration (profiles/r05_gemv_ab.log) - one below (the grid of a small matrix does not fill the chip
// with fewer waves: measured slower at n = 8192 and 3001).
static void gemv_dense(kh_ctx ctx, kh_mat A, const double* x, double* y) {
    const int rows = ctx->gemv_rows > 0 ? ctx->gemv_rows : (A->n_rows >= 16 * 1024 ? 4 : 1);
    const int rows_per_wg = (BS / 64) * rows;
    const int grid = (int)((A->n_rows + rows_per_wg - 1) / rows_per_wg);
    if (rows == 4)
        hipLaunchKernelGGL(k_gemv_dense<4>, dim3(grid), dim3(BS), 0, ctx->stream, A->n_rows, A->n_cols, A->a, A->lda, x, y);
    else if (rows == 2)
        hipLaunchKernelGGL(k_gemv_dense<2>, dim3(grid), dim3(BS), 0, ctx->stream, A->n_rows, A->n_cols, A->a, A->lda, x, y);
    else
        hipLaunchKernelGGL(k_gemv_dense<1>, dim3(grid), dim3(BS), 0, ctx->stream, A->n_rows, A->n_cols, A->a, A->lda, x, y);
}

// y = A x for one column; epi/aux select the fused epilogue of the CSR kernel (its partial sums
// land in A->part and are reduced into scal_out with `rmode` of k_reduce_partials).
int apply_one(kh_ctx ctx, kh_mat A, const double* x, double* y, int epi, const double* aux,
                     double* scal_out, int rmode) {
    if (A->kind == KH_MAT_CSR) {
        const bool halo = kh_multi(ctx) && (A->nrecv_prev + A->nrecv_next + A->nsend_prev + A->nsend_next) > 0;
        if (A->nblk == 0) {
            if (halo) KH_TRY(comm_halo_exchange(ctx, A, x, ctx->stream));
            return 0;
        }
        const bool dia = use_dia(ctx, A, y);
        if (A->xh_on && halo) {
            // the halo travels inside the banded kernel's own launch (kh_mat_xh_*): ONE launch, no RCCL kernel, no second
            // stream.  Every rank switched this on together - a shape the banded kernel does not serve is an error here, not a
            // fallback to the RCCL exchange its neighbours are not taking part in.
            if (!dia)
                return fail(KH_ERR_COMM, "sharded SpMV: the halo is exchanged inside the banded kernel (xh) but this call cannot take "
                                         "it (KRYPY_AMD_SPMV_DIA=0, or an output column that is not 16-byte aligned)");
            if (A->xh_epoch > 0xfff00000u)
                return fail(KH_ERR_COMM, "xh: the epoch counter of this operator's halo exchange is exhausted; upload it again");
            if (epi == EPI_NONE) launch_spmv<EPI_NONE>(ctx, A, x, y, nullptr, SpmvRange());
            if (epi == EPI_DOT) launch_spmv<EPI_DOT>(ctx, A, x, y, aux, SpmvRange());
            if (epi == EPI_RES) launch_spmv<EPI_RES>(ctx, A, x, y, aux, SpmvRange());
            KH_HIP(hipGetLastError());
            ctx->n_halo_xh += 1;
            if (epi != EPI_NONE) {
                hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, A->part, A->dia_nblk, 0, scal_out, rmode);
                KH_HIP(hipGetLastError());
            }
            return 0;
        }
        // a shard's rows that touch no ghost column (the bulk of a slab) do not wait for the halo: the exchange
        // runs on the communication stream while they are multiplied, the boundary rows follow it
        const int lo = dia ? A->dia_b0 : A->csr_b0, hi = dia ? A->dia_b1 : A->csr_b1;
        const int nblk = dia ? A->dia_nblk : A->nblk;
        const bool split = kh_multi(ctx) && ctx->spmv_split && ctx->comm_stream != nullptr && lo < hi &&
                           (halo || ctx->force_multi) && (lo > 0 || hi < nblk);
#define KH_SPMV(RG)                                                            \
    do {                                                                       \
        if (epi == EPI_NONE) launch_spmv<EPI_NONE>(ctx, A, x, y, nullptr, RG); \
        if (epi == EPI_DOT) launch_spmv<EPI_DOT>(ctx, A, x, y, aux, RG);       \
        if (epi == EPI_RES) launch_spmv<EPI_RES>(ctx, A, x, y, aux, RG);       \
    } while (0)
        if (split) {
            KH_HIP(hipEventRecord(ctx->ev_x, ctx->stream));                   // x is complete
            // the interior launch is ENQUEUED first: issuing the grouped send / recv costs the host ~10 us, and in the
            // other order the interior rows start that much later (profiles/r03_halo_overlap.md: the RCCL kernel was
            // over 3 us after the interior launch began - nothing overlapped)
            KH_SPMV(SpmvRange::interior(lo, hi));
            KH_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_x, 0));
            if (halo) KH_TRY(comm_halo_exchange(ctx, A, x, ctx->comm_stream));
            KH_HIP(hipEventRecord(ctx->ev_halo, ctx->comm_stream));
            KH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));
            KH_SPMV(SpmvRange::boundary(lo, hi, nblk));
            ctx->n_spmv_split += 1;
        } else {
            if (halo) KH_TRY(comm_halo_exchange(ctx, A, x, ctx->stream));
            KH_SPMV(SpmvRange());
        }
#undef KH_SPMV
        KH_HIP(hipGetLastError());
        if (epi != EPI_NONE) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, A->part,
                               dia ? A->dia_nblk : A->nblk, 0, scal_out, rmode);
            KH_HIP(hipGetLastError());
        }
        return 0;
    }
    KH_ARG(epi == EPI_NONE, "fused epilogues exist for CSR operators only");
    if (A->kind == KH_MAT_DENSE) {
        gemv_dense(ctx, A, x, y);
    } else {
        const int grid = (int)std::min<int64_t>((A->n_rows + BS - 1) / BS, ctx->nb * 2);
        hipLaunchKernelGGL(k_diag_apply, dim3(std::max(grid, 1)), dim3(BS), 0, ctx->stream,
                           A->n_rows, A->diag, x, y);
    }
    KH_HIP(hipGetLastError());
    return 0;
}

// ---- register-resident MGS chain (chain.h) -------------------------------------------------
// Plain launch (a cooperative launch goes through a separate hardware queue and costs ~1 ms of
// cross-queue synchronisation per Arnoldi step when interleaved with ordinary kernels).  Residency
// is what matters for the in-kernel grid reduction, and it is identical for plain and cooperative
// launches: it is checked here against the occupancy of this instantiation, and every spin in the
// kernel is bounded.
template <int R2, bool MASKED, bool CPLX = false, int FND = 0, int WL = 0>
static hipError_t launch_chain(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = (size_t)WL * CH_BS * sizeof(double2);      // rows of w that live in LDS (long vectors)
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mgs_chain<R2, MASKED, CPLX, FND, WL>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mgs_chain<R2, MASKED, CPLX, FND, WL>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_mgs_chain<R2, MASKED, CPLX, FND, WL>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// the variant that parks the head of every column in LDS between its dot and its update (B == V)
template <int R2, bool MASKED, bool CPLX = false, int FND = 0>
static hipError_t launch_chain_lds(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ChainShapeLds<R2, CPLX>::LDS_BYTES;
    if (blocks_per_cu < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mgs_chain_lds<R2, MASKED, CPLX, FND>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mgs_chain_lds<R2, MASKED, CPLX, FND>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_mgs_chain_lds<R2, MASKED, CPLX, FND>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// 48 rows per lane with a third of every column kept on the chip between its two uses (chain_long.h)
template <int FND>
static hipError_t launch_chain_long(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ChainShapeLong::LDS_BYTES;
    if (blocks_per_cu < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mgs_chain_long<FND, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mgs_chain_long<FND, false>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_mgs_chain_long<FND, false>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// the variant that keeps HBM busy through the update phase (chain.h: k_mgs_chain_pf)
template <int R2, bool MASKED, bool CPLX = false, int FND = 0>
static hipError_t launch_chain_pf(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ChainShapePf<R2>::LDS_BYTES;
    if (blocks_per_cu < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mgs_chain_pf<R2, MASKED, CPLX, FND>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mgs_chain_pf<R2, MASKED, CPLX, FND>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_mgs_chain_pf<R2, MASKED, CPLX, FND>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// short vectors: all working workgroups on one XCD (chain.h, ONEX).  8 G + 8 workgroups are launched; the G that
// work must be co-resident on the 1/8 of the CUs one XCD has.
template <int R2, bool MASKED, bool CPLX, bool PF>
static hipError_t launch_chain_onex(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = PF ? ChainShapePf<R2>::LDS_BYTES : 0;
    auto kern = PF ? k_mgs_chain_pf<R2, MASKED, CPLX, 0, true> : k_mgs_chain<R2, MASKED, CPLX, 0, 0, true>;
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * (ctx->ncu / 8) < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(8 * G + 8), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// short vectors without a preconditioner: whole columns in a register ring, requested several links ahead
// (chain.h, k_mgs_chain_small); ONEX or spread over the chip
template <int R2, bool MASKED, int FND, bool ONEX>
static hipError_t launch_chain_small(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
#ifndef KH_SMALL_LA4
#define KH_SMALL_LA4 3
#define KH_SMALL_LA8 2
#endif
    constexpr int LA = (R2 == 4) ? KH_SMALL_LA4 : KH_SMALL_LA8;          // columns requested ahead
    auto kern = k_mgs_chain_small<R2, LA, MASKED, FND, ONEX>;
    if (blocks_per_cu < 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, ONEX ? CH_BS + 64 : CH_BS, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * (ONEX ? ctx->ncu / 8 : ctx->ncu) < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(ONEX ? 8 * G + 8 : G), dim3(ONEX ? CH_BS + 64 : CH_BS), 0, ctx->stream, a);   // ONEX: + the communication wave
    return hipGetLastError();
}

// ... with the banded operator in the prologue (no SpMV launch in front of the step)
template <int R2, int FND, bool PF>
static hipError_t launch_chain_onex_fused(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = PF ? ChainShapePf<R2>::LDS_BYTES : 0;
    auto kern = PF ? k_mgs_chain_pf<R2, false, false, FND, true> : k_mgs_chain<R2, false, false, FND, 0, true>;
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * (ctx->ncu / 8) < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(8 * G + 8), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// one Lanczos step in three passes (lanczos.h)
template <int R2, int FND, bool JAC, bool MR>
static hipError_t launch_lanczos(kh_ctx ctx, int G, ChainArgs& a, const MinresJob& mr) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = (size_t)(LanczosShape<R2>::WL + (JAC ? LanczosShape<R2>::DL : 0)) * CH_BS * sizeof(double2);
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lanczos_fused<R2, FND, JAC, MR>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_lanczos_fused<R2, FND, JAC, MR>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_lanczos_fused<R2, FND, JAC, MR>), dim3(G), dim3(CH_BS), lds, ctx->stream, a, mr);
    return hipGetLastError();
}

// rows-per-workgroup (= template R2) and grid of the chain kernel for vectors of length n
bool chain_geometry(kh_ctx ctx, int64_t n, int* r2_out, int* g_out, bool onex) {
    if (n < 2) return false;
    // an odd n is handled as n+1: the extra element is the (always zero) padding behind the column
    const int64_t n2 = (n + 1) >> 1;
    static const int kR2[] = {4, 8, 16, 24, 32, 40, 48, 56};   // 48 / 56: the last 8 / 16 rows of w live in LDS
    // short vectors: as few workgroups as one XCD has CUs, so that all of them can run there (chain.h, ONEX) - 8 rows
    // per lane on 32 workgroups beat 4 rows on 64 spread over the chip, the sum being the whole link
    if (onex) {
        if (!ctx->chain_onex || ctx->ncu % 8 != 0) return false;
        for (int c : {4, 8}) {
            const int64_t g = (n2 + (int64_t)c * CH_BS - 1) / ((int64_t)c * CH_BS);
            if (g <= ctx->ncu / 8 && g <= 32) {
                *r2_out = c;
                *g_out = (int)g;
                return true;
            }
        }
        return false;
    }
    for (int c : kR2) {
        const int64_t g = (n2 + (int64_t)c * CH_BS - 1) / ((int64_t)c * CH_BS);
        if (g <= ctx->ncu && g <= CH_GMAX) {
            *r2_out = c;
            *g_out = (int)g;
            return true;
        }
    }
    return false;   // w fits neither registers nor registers + LDS (N > 14.68 M on 256 CUs): stream it (k_gs_link)
}

// leading dimension of a block of n-vectors: large vectors are padded to whole chain chunks so
// that the predicate-free kernel applies (<= 0.4 % extra memory at N = 10^7)
static int64_t padded_ld(kh_ctx ctx, int64_t n) {
    int64_t ld = ((n + 31) / 32) * 32;
    int r2 = 0, g = 0;
    // (from 4096 rows on: below that a vector is a fraction of one workgroup's chunk)
    if (n >= (1 << 12) && chain_geometry(ctx, n, &r2, &g)) ld = (int64_t)g * r2 * CH_BS * 2;
    if (n >= (1 << 12) && chain_geometry(ctx, n, &r2, &g, true)) ld = std::max(ld, (int64_t)g * r2 * CH_BS * 2);   // (one-XCD shape)
    int cw = 0;
    if (n >= (1 << 12) && chain_blk2_shape(ctx, n, &r2, &g, &cw, ctx->blk2_one >= 1))      // (chain_blk2.h: 4 ... 11 rows, 448 or 512 lanes with rows)
        ld = std::max(ld, (int64_t)g * r2 * (cw ? CH_BS - 64 : CH_BS) * 2);
    return ld == 0 ? 32 : ld;
}

#define KH_BLK_MIN_LINKS 8     // Gram-Schmidt links from which the blocked kernel (chain_blk.h) takes a step of short vectors
// true when the step goes to the blocked kernel (which has no operator prologue yet)
static inline bool blk_takes_step(kh_ctx ctx, const ChainArgs& a, int r2) {
    return ctx->chain_blk && r2 == 4 && !a.presub && a.sweeps == 1 && a.col0 == 0 && a.ncol >= KH_BLK_MIN_LINKS &&
           a.ncol + 2 <= KH_BLK_TABCOLS && !kh_multi(ctx);
}

// returns 1 if the chain was launched, 0 if this step is not eligible (caller uses the link
// kernels), negative on error
// the epoch counter of the grid-wide sums nears its wrap: everything that carries tags is zeroed, the count starts over
int chain_epoch_check(kh_ctx ctx) {
    if (ctx->chain_epoch <= 0xfff00000u) return 0;
    KH_HIP(hipStreamSynchronize(ctx->stream));
    KH_HIP(hipMemset(ctx->chain_gran, 0, sizeof(unsigned long long) * 4 * CH_GMAX));
    KH_HIP(hipMemset(ctx->chain_xcc, 0, sizeof(unsigned long long) * 128 + sizeof(unsigned) * 16));
    if (ctx->blk_gran != nullptr) {
        // (the blocked kernels' granules only: chain_blk_reset leaves the Gram table of the running Arnoldi sequence
        // alone when the buffer exists - ADVICE r04: a zeroed table behind a still matching (blk_V, blk_next) key
        // silently dropped the corrections.  The key is withdrawn as well: the next blocked step rebuilds its rows.)
        KH_HIP(chain_blk_reset(ctx));
        KH_HIP(hipStreamSynchronize(ctx->stream));
        ctx->blk_next = -1;
    }
    ctx->chain_epoch = 1;
    ctx->n_epoch_wraps += 1;
    return 0;
}

int dot_panel_raw(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* w, double* out_dev) {
    return ::dot_panel_dev(ctx, V, j0, ncols, w, out_dev, 0);
}

int try_chain(kh_ctx ctx, kh_vec V, kh_vec B, const double* w, int64_t wld, const double* dg,
              kh_vec P, int64_t k, int64_t start, int sweeps, bool presub, double h_km1,
              const double* h_km1_dev, double* hdev, int slot, bool cplx, double* hpin,
              int hcount, kh_mat Afuse, const double* xk, const MinresJob* mr) {
    // Afuse: compute w = Afuse * xk in the kernel's prologue instead of reading w (banded operators;
    // returns 0 without launching anything when that variant does not apply - the caller then runs the
    // SpMV and calls again without Afuse)
    // cplx: V, B, w are (re, im) views of complex vectors (zpath.h); hdev holds (re, im) pairs
    if (!ctx->chain_enabled || kh_multi(ctx)) return 0;
    if (cplx && (dg != nullptr || P != nullptr)) return 0;
    const int64_t n = V->n;
    if (n == ctx->chain_refused_n) return 0;
    const int64_t n2 = (n + 1) >> 1;
    int r2 = 0, G = 0;
    if (!chain_geometry(ctx, n, &r2, &G)) return 0;
    // short vectors with at least a few links in the launch: as few workgroups as one XCD has CUs, all of them
    // there (chain.h, ONEX) - the sum is the whole link.  (One or two links - Lanczos, the first Arnoldi steps - do
    // not pay for the larger grid: MINRES + Jacobi at N = 10^5 32,400 vs 27,400 it/s.)
    bool want_onex = false;
    if ((k - start + 1) * sweeps >= 3 && (ctx->chain_debug == 0) && ctx->onex_ticket != nullptr) {
        int r2x = 0, Gx = 0;
        // (the column-ring kernel with 8 rows per lane - 1.3e5 < N <= 2.6e5 - is faster spread over the chip with 4:
        // a one-XCD link moves the whole column through one XCD's port; 7,240 vs 6,770 it/s at N = 2.5e5)
        const bool ring_kernel = ctx->chain_small && B == V && dg == nullptr && !cplx;
        // (... and the blocked kernel - four times fewer sums - is bound by the one XCD's memory port from
        // ctx->blk_onex_maxn rows on: it then spreads over the chip as well)
        const bool blk_long = ctx->chain_blk && ring_kernel && !presub && sweeps == 1 && start == 0 &&
                              k + 1 >= KH_BLK_MIN_LINKS && r2 == 4 && n > ctx->blk_onex_maxn;
        if (chain_geometry(ctx, n, &r2x, &Gx, true) && !(ring_kernel && r2x == 8 && r2 == 4) && !blk_long) {
            r2 = r2x;
            G = Gx;
            want_onex = true;
        }
    }
    if (cplx && 4 * G > 2 * CH_GMAX) return 0;      // grid_sum2: four granules per workgroup
    if ((n & 1) && (V->ld <= n || B->ld <= n || wld <= n || (P && P->ld <= n))) return 0;
    // 1.05 M ... 2.5 M rows (8 / 16 rows per lane in the general geometry, 5 ... 11 here), no preconditioner, a long chain: the
    // eight-wave blocked kernel (chain_blk2.h) - one grid-wide sum per four columns, columns read once.  It loads w: with the
    // operator to be fused (Afuse) nothing is launched here, the caller runs the SpMV and comes back without it.
    {
        int r2b = 0, Gb = 0;
        if (ctx->chain_blk2 && ctx->chain_blk && ctx->chain_small && (r2 == 8 || (r2 == 16 && ctx->blk2_one >= 2)) && !want_onex && B == V && dg == nullptr && !cplx && !presub &&
            sweeps == 1 && start == 0 && k + 1 >= KH_BLK_MIN_LINKS && (ctx->chain_debug == 0) && n != ctx->blk2_refused_n &&
            chain_blk2_shape(ctx, n, &r2b, &Gb, nullptr, ctx->blk2_one >= 2) && r2b > 4) {
            if (Afuse != nullptr) return 0;
            const int rc = chain_blk2_step(ctx, V, w, wld, k, hdev, slot, hpin, hcount, false);
            if (rc != 0) {
                if (rc == 1) ctx->n_chain_lds += 1;      // (the column is used twice from the chip: counted with that family)
                return rc;
            }
        }
    }
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    // predicate-free kernel iff every block involved is padded to G whole chunks
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = V->ld >= need_ld && B->ld >= need_ld && (P == nullptr || P->ld >= need_ld) &&
                        wld >= need_ld;
    KH_TRY(chain_epoch_check(ctx));
    ChainArgs a;
    a.n2 = n2;
    a.chunk2 = chunk2;
    a.V = V->d;
    a.B = B->d;
    a.ld = V->ld;
    a.col0 = start;
    a.ncol = (int)(k - start + 1);
    a.sweeps = sweeps;
    a.w_in = w;
    a.dg = dg;
    a.vnext = V->col(k + 1);
    a.pnext = P ? P->col(k + 1) : nullptr;
    a.hdev = hdev;
    a.hnext = cplx ? 2 * (k + 1) : k + 1;
    a.gran = ctx->chain_gran;
    a.xcc_res = ctx->chain_xcc;
    a.xcc_leader = reinterpret_cast<unsigned*>(ctx->chain_xcc + 128);
    a.epoch0 = ctx->chain_epoch;
    a.err = ctx->chain_err;
    a.debug = ctx->chain_debug;
    if (ctx->chain_fault) a.debug = 4;     // kh_ctx_set("chain_fault", 1): the next launch behaves like a timed-out one
    a.presub = presub ? 1 : 0;
    a.h_km1 = h_km1;
    a.h_km1_dev = h_km1_dev;
    a.bprev = presub ? B->col(k - 1) : nullptr;
    a.hpin = hpin;
    a.hcount = hcount;
    a.errpin = ctx->chain_err_pin[slot];
    // completion tag (kh_internal.h): whichever chain-family kernel this call ends up launching writes it
    a.donepin = (hpin != nullptr && ctx->tag_wait) ? ctx->done_pin[slot] : nullptr;
    a.done_tag = 0;
    if (a.donepin != nullptr) {
        ctx->done_counter = (ctx->done_counter == 0x7fffffff) ? 1 : ctx->done_counter + 1;
        a.done_tag = ctx->done_counter;
        ctx->done_seq[slot] = a.done_tag;
    }
#ifdef KH_CHAIN_TRACE
    a.trace = ctx->chain_trace;
#endif
    hipError_t e;
    const int lds_env = ctx->chain_lds;
    // (the complex instantiation with 40 rows per lane spills 36 registers with the LDS traffic on top
    // and still beats the plain kernel, 769 vs 677 it/s at N = 5*10^6; 32 rows fit without spills since
    // both parts of the coefficient share one grid reduction)
    static thread_local bool lds_failed = false;   // the LDS variant could not be launched once: plain kernel from then on
    // (an unpadded block of a long vector - not what kh_vec_alloc produces - takes the plain kernel: the
    // masked LDS instantiations with 32 / 40 rows spill)
    bool use_lds = lds_env && !lds_failed && B == V && dg == nullptr && (a.debug == 0 || a.debug == 4) &&
                   (padded || r2 <= 24) && r2 <= 40;     // (48 / 56 rows: LDS holds a part of w itself)
    if (r2 > 40 && cplx) return 0;
    // 48 rows per lane, no preconditioner, padded blocks: a third of every column stays on the chip between its two uses
    static thread_local bool long_failed = false;
    bool use_long = ctx->chain_long && ctx->chain_lds && !long_failed && r2 == 48 && B == V && dg == nullptr && !cplx && padded && (a.debug == 0 || a.debug == 4);
    // fused operator: the padded real kernels with 16 ... 40 rows per lane (N > 2.1 M) have that prologue
    bool fused = false;
    if (Afuse != nullptr) {
        // (a step with one Gram-Schmidt link has the three-pass kernel of lanczos.h for every shape up to 40 rows)
        const bool lz_shape = ctx->lanczos_fused && a.ncol == 1 && a.sweeps == 1 && (dg == nullptr || P != nullptr) &&
                              (r2 == 4 || r2 == 8);
        const bool small_shape = ctx->chain_small && r2 <= 8 && B == V && dg == nullptr;
        if (cplx) {
            // complex banded operator (zpath.h: zdia, leading dimension and rows counted in complex entries): the padded
            // chain kernels with 16 ... 40 rows per lane, spread over the chip
            fused = ctx->chain_spmv && padded && ((a.debug & 3) == 0) && r2 >= 16 && r2 <= 40 && !want_onex && xk != nullptr &&
                    Afuse->kind == KH_MAT_ZCSR && Afuse->zdia != nullptr && (Afuse->dia_nd == 5 || Afuse->dia_nd == 7) &&
                    2 * Afuse->n_rows == n && Afuse->nrecv_prev + Afuse->nrecv_next == 0 && 2 * Afuse->zdia_ld >= need_ld;
        } else {
            fused = ctx->chain_spmv && padded && ((a.debug & 3) == 0) && ((r2 >= 16 && r2 <= 48) || want_onex || lz_shape || small_shape) && xk != nullptr &&
                    Afuse->kind == KH_MAT_CSR && Afuse->dia != nullptr && (Afuse->dia_nd == 5 || Afuse->dia_nd == 7) &&
                    Afuse->n_rows == n && Afuse->nrecv_prev + Afuse->nrecv_next == 0 && Afuse->dia_ld >= need_ld;
        }
        if (!fused) return 0;
        a.dia = cplx ? Afuse->zdia : Afuse->dia;
        a.dia_ld = cplx ? Afuse->zdia_ld : Afuse->dia_ld;
        a.xk = xk;
        a.n_last = cplx ? Afuse->n_rows - 1 : n - 1;
        a.offs.nd = Afuse->dia_nd;
        for (int d = 0; d < KH_DIA_MAX; ++d) a.offs.off[d] = d < Afuse->dia_nd ? Afuse->dia_off[d] : 0;
    } else {
        a.dia = nullptr;
        a.dia_ld = 0;
        a.xk = nullptr;
        a.n_last = n - 1;
        a.offs.nd = 0;
    }
#define KH_CHAIN_PLAIN(R)                                                                             \
    (cplx ? (padded ? launch_chain<R, false, true>(ctx, G, a) : launch_chain<R, true, true>(ctx, G, a)) \
          : (padded ? launch_chain<R, false>(ctx, G, a) : launch_chain<R, true>(ctx, G, a)))
#define KH_CHAIN_LDS(R)                                                                                       \
    (cplx ? (padded ? launch_chain_lds<R, false, true>(ctx, G, a) : launch_chain_lds<R, true, true>(ctx, G, a)) \
          : (padded ? launch_chain_lds<R, false>(ctx, G, a) : launch_chain_lds<R, true>(ctx, G, a)))
#define KH_CHAIN_PF(R)                                                                                      \
    (cplx ? (padded ? launch_chain_pf<R, false, true>(ctx, G, a) : launch_chain_pf<R, true, true>(ctx, G, a)) \
          : (padded ? launch_chain_pf<R, false>(ctx, G, a) : launch_chain_pf<R, true>(ctx, G, a)))
    // k_mgs_chain_pf wins where the whole column stays on chip (<= 24 rows per lane: 6.9 vs 7.2 us per link at
    // N = 4*10^6); with 32 / 40 rows its prefetches sit in the CU's memory queue in front of the reduction's polls
    // and cost what they save (17.4 vs 16.3 us per link at N = 10^7)
    const bool use_pf = use_lds && ctx->chain_pf && (r2 <= 24 || (ctx->chain_pf == 2 && r2 <= 40));   // (2: measurement)
#define KH_CHAIN(R) (use_lds ? (use_pf ? KH_CHAIN_PF(R) : KH_CHAIN_LDS(R)) : KH_CHAIN_PLAIN(R))
    // a step with ONE Gram-Schmidt link (Lanczos / MINRES, the first Arnoldi step): three passes instead of six
    if (fused && !cplx && r2 <= 40 && ctx->lanczos_fused && a.ncol == 1 && a.sweeps == 1 && (dg == nullptr || P != nullptr)) {
        if (!presub) {                 // no previous column: subtract 0 * (some valid column)
            a.bprev = B->col(k);
            a.h_km1 = 0.0;
            a.h_km1_dev = nullptr;
        }
        MinresJob job;
        if (mr != nullptr) job = *mr;
        else {
            job.on = 0;
            job.v = job.w1 = w;           // (never dereferenced)
            job.w0 = job.yk = const_cast<double*>(w);
            job.r0 = job.r1 = job.y0 = 0.0;
            job.r2 = 1.0;
        }
#define KH_LZ(R, D)                                                                                             \
    (dg != nullptr ? (job.on ? launch_lanczos<R, D, true, true>(ctx, G, a, job) : launch_lanczos<R, D, true, false>(ctx, G, a, job)) \
                   : (job.on ? launch_lanczos<R, D, false, true>(ctx, G, a, job) : launch_lanczos<R, D, false, false>(ctx, G, a, job)))
        if (r2 == 40) e = (a.offs.nd == 5) ? KH_LZ(40, 5) : KH_LZ(40, 7);
        else if (r2 == 32) e = (a.offs.nd == 5) ? KH_LZ(32, 5) : KH_LZ(32, 7);
        else if (r2 == 24) e = (a.offs.nd == 5) ? KH_LZ(24, 5) : KH_LZ(24, 7);
        else if (r2 == 16) e = (a.offs.nd == 5) ? KH_LZ(16, 5) : KH_LZ(16, 7);
        else if (r2 == 8) e = (a.offs.nd == 5) ? KH_LZ(8, 5) : KH_LZ(8, 7);
        else e = (a.offs.nd == 5) ? KH_LZ(4, 5) : KH_LZ(4, 7);
#undef KH_LZ
        if (e == hipSuccess) {
            if (a.debug == 4) ctx->chain_fault = 0;
            ctx->n_chain += 1;
            ctx->n_chain_fused += 1;
            ctx->n_lanczos_fused += 1;
            ctx->chain_epoch += 2u;          // the coefficient's and the norm's grid-wide sums
            if (hpin == nullptr)
                KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost,
                                      ctx->stream));
            ctx->mr_taken = job.on ? 1 : 0;     // (the MINRES job - if any - went along)
            { ctx->wait_tag[slot] = a.donepin != nullptr; return 1; }
        }
        (void)hipGetLastError();             // e.g. the dynamic LDS was refused: the general chain kernel below
        if (!presub) a.bprev = nullptr;
    }
    // short vectors (4 ... 32 workgroups of 4 / 8 rows per lane): a link is its grid-wide sum - all working
    // workgroups on ONE XCD, where the sum is an L2 round trip (chain.h, ONEX)
    // short vectors, no preconditioner, real data: the column-ring kernel (one read per column, several links of look-ahead)
    const bool blk_debug = (a.debug >= 1 && a.debug <= 3) && !fused && padded && blk_takes_step(ctx, a, r2);   // measurement modes of the blocked kernel
    if (ctx->chain_small && r2 <= 8 && B == V && dg == nullptr && !cplx && (a.debug == 0 || a.debug == 4 || blk_debug) &&
        !(fused && ctx->lanczos_fused && a.ncol == 1 && a.sweeps == 1)) {
        if (want_onex) {
            const unsigned slot_ = (unsigned)(ctx->n_chain_onex & 255);
            a.onex_G = G;
            a.onex_target = 0u;
            a.onex_ticket = ctx->onex_ticket + slot_;
            a.onex_clear = ctx->onex_ticket + ((slot_ + 128u) & 255u);
        }
        // a long chain: the blocked form - one grid-wide sum per FOUR columns (chain_blk.h)
        if (blk_takes_step(ctx, a, r2) && padded && chain_blk_shape_ok(r2, G, a, fused ? a.offs.nd : 0) && ctx->blk_refused_n != n) {
            int nsums = 0;
            // the Gram table: valid when this is the next step of the sequence that owns it; otherwise (a sequence's
            // first blocked step, a block grown / recycled / written by another entry point since) its rows are
            // rebuilt from the basis - one panel product per column, once
            if (!(ctx->blk_V == V && ctx->blk_next == k && ctx->blk_kind == 1)) {
                double* gt = chain_blk_table(ctx);
                if (gt == nullptr) return fail(KH_ERR_NOMEM, "chain_blk: no memory for the Gram table");
                for (int64_t j = 1; j <= k; ++j) {
                    const int64_t b0 = (j / KH_BLK_BC) * KH_BLK_BC;
                    if (b0 >= KH_BLK_BC)      // the block before column j's (the kernel takes its dots one block ahead)
                        KH_TRY(::dot_panel_dev(ctx, V, b0 - KH_BLK_BC, KH_BLK_BC, V->col(j), gt + j * KH_BLK_TW, 0));
                    if (j > b0) KH_TRY(::dot_panel_dev(ctx, V, b0, j - b0, V->col(j), gt + j * KH_BLK_TW + KH_BLK_BC, 0));
                }
                ctx->blk_V = V;
                ctx->blk_next = k;
                ctx->blk_kind = 1;
                ctx->n_blk_rebuild += 1;
            }
            e = chain_blk_launch(ctx, r2, G, want_onex, padded, fused ? a.offs.nd : 0, a, V, &nsums);
            if (e == hipSuccess) {
                ctx->blk_kind = 1;
                if (a.debug == 4) ctx->chain_fault = 0;
                ctx->n_chain += 1;
                ctx->n_chain_small += 1;
                ctx->n_chain_blk += 1;
                ctx->n_chain_onex += want_onex ? 1 : 0;
                ctx->n_chain_fused += fused ? 1 : 0;
                ctx->n_chain_lds += 1;
                ctx->chain_epoch += (unsigned)nsums;
                if (hpin == nullptr)
                    KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost,
                                          ctx->stream));
                { ctx->wait_tag[slot] = a.donepin != nullptr; return 1; }
            }
            (void)hipGetLastError();
            // refused (occupancy: other kernels resident, fewer compute units ...): remembered for vectors of this length, so
            // that the steps to come do not rebuild the table - up to 2 k panel products - just to be refused again
            // (kh_ctx_set("chain_blk", 1) forgets it)
            ctx->blk_refused_n = n;
        }
#define KH_SM(R, X)                                                                                          \
    (fused ? (a.offs.nd == 5 ? launch_chain_small<R, false, 5, X>(ctx, G, a) : launch_chain_small<R, false, 7, X>(ctx, G, a)) \
           : (padded ? launch_chain_small<R, false, 0, X>(ctx, G, a) : launch_chain_small<R, true, 0, X>(ctx, G, a)))
        if (want_onex) e = (r2 == 4) ? KH_SM(4, true) : KH_SM(8, true);
        else e = (r2 == 4) ? KH_SM(4, false) : KH_SM(8, false);
#undef KH_SM
        if (e == hipSuccess) {
            if (a.debug == 4) ctx->chain_fault = 0;
            ctx->n_chain += 1;
            ctx->n_chain_small += 1;
            ctx->n_chain_onex += want_onex ? 1 : 0;
            ctx->n_chain_fused += fused ? 1 : 0;
            ctx->n_chain_lds += 1;          // (the column is used twice from the chip: counted with the LDS / ring family)
            ctx->chain_epoch += (unsigned)(a.ncol * a.sweeps + 1);
            if (hpin == nullptr)
                KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost,
                                      ctx->stream));
            { ctx->wait_tag[slot] = a.donepin != nullptr; return 1; }
        }
        (void)hipGetLastError();
    }
    if (want_onex) {
        const unsigned slot_ = (unsigned)(ctx->n_chain_onex & 255);
        a.onex_G = G;
        a.onex_target = 0u;
        a.onex_ticket = ctx->onex_ticket + slot_;
        a.onex_clear = ctx->onex_ticket + ((slot_ + 128u) & 255u);
        const bool pf_ = (B == V && dg == nullptr && ctx->chain_pf != 0 && ctx->chain_lds != 0);
#define KH_OX(R)                                                                                                    \
    (pf_ ? (cplx ? (padded ? launch_chain_onex<R, false, true, true>(ctx, G, a) : launch_chain_onex<R, true, true, true>(ctx, G, a))     \
                 : (padded ? launch_chain_onex<R, false, false, true>(ctx, G, a) : launch_chain_onex<R, true, false, true>(ctx, G, a)))  \
         : (cplx ? (padded ? launch_chain_onex<R, false, true, false>(ctx, G, a) : launch_chain_onex<R, true, true, false>(ctx, G, a))   \
                 : (padded ? launch_chain_onex<R, false, false, false>(ctx, G, a) : launch_chain_onex<R, true, false, false>(ctx, G, a))))
#define KH_OXF(R) (pf_ ? (a.offs.nd == 5 ? launch_chain_onex_fused<R, 5, true>(ctx, G, a) : launch_chain_onex_fused<R, 7, true>(ctx, G, a)) \
                      : (a.offs.nd == 5 ? launch_chain_onex_fused<R, 5, false>(ctx, G, a) : launch_chain_onex_fused<R, 7, false>(ctx, G, a)))
        if (fused) e = (r2 == 4) ? KH_OXF(4) : KH_OXF(8);
        else e = (r2 == 4) ? KH_OX(4) : KH_OX(8);
#undef KH_OXF
#undef KH_OX
        if (e == hipSuccess) {
            if (a.debug == 4) ctx->chain_fault = 0;
            ctx->n_chain += 1;
            ctx->n_chain_onex += 1;
            ctx->n_chain_fused += fused ? 1 : 0;
            ctx->n_chain_lds += pf_ ? 1 : 0;
            ctx->n_chain_pf += pf_ ? 1 : 0;
            ctx->chain_epoch += (unsigned)(a.ncol * a.sweeps + 1);
            if (hpin == nullptr)
                KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost,
                                      ctx->stream));
            { ctx->wait_tag[slot] = a.donepin != nullptr; return 1; }
        }
        (void)hipGetLastError();
    }
    a.onex_G = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (fused) {
#define KH_FUSED(R, D)                                                                                   \
    (use_lds ? (use_pf ? launch_chain_pf<R, false, false, D>(ctx, G, a) : launch_chain_lds<R, false, false, D>(ctx, G, a)) \
             : launch_chain<R, false, false, D>(ctx, G, a))
#define KH_FUSED_Z(R, D) (use_lds ? launch_chain_lds<R, false, true, D>(ctx, G, a) : launch_chain<R, false, true, D>(ctx, G, a))
            if (r2 < 16) return 0;       // (4 / 8 rows: only the Lanczos and the one-XCD kernels have the prologue)
            if (cplx) {
                if (r2 == 40) e = (a.offs.nd == 5) ? KH_FUSED_Z(40, 5) : KH_FUSED_Z(40, 7);
                else if (r2 == 32) e = (a.offs.nd == 5) ? KH_FUSED_Z(32, 5) : KH_FUSED_Z(32, 7);
                else if (r2 == 24) e = (a.offs.nd == 5) ? KH_FUSED_Z(24, 5) : KH_FUSED_Z(24, 7);
                else e = (a.offs.nd == 5) ? KH_FUSED_Z(16, 5) : KH_FUSED_Z(16, 7);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    return 0;
                }
                break;
            }
            if (r2 == 48 && use_long) {
                e = (a.offs.nd == 5) ? launch_chain_long<5>(ctx, G, a) : launch_chain_long<7>(ctx, G, a);
                if (e != hipSuccess) {       // (e.g. the 128 KB of dynamic LDS were refused: the kernel with both reads from memory)
                    (void)hipGetLastError();
                    long_failed = true;
                    use_long = false;
                    e = (a.offs.nd == 5) ? launch_chain<48, false, false, 5, 8>(ctx, G, a) : launch_chain<48, false, false, 7, 8>(ctx, G, a);
                }
            } else if (r2 == 48) e = (a.offs.nd == 5) ? launch_chain<48, false, false, 5, 8>(ctx, G, a) : launch_chain<48, false, false, 7, 8>(ctx, G, a);
            else if (r2 == 40) e = (a.offs.nd == 5) ? KH_FUSED(40, 5) : KH_FUSED(40, 7);
            else if (r2 == 32) e = (a.offs.nd == 5) ? KH_FUSED(32, 5) : KH_FUSED(32, 7);
            else if (r2 == 24) e = (a.offs.nd == 5) ? KH_FUSED(24, 5) : KH_FUSED(24, 7);
            else e = (a.offs.nd == 5) ? KH_FUSED(16, 5) : KH_FUSED(16, 7);
#undef KH_FUSED
#undef KH_FUSED_Z
            if (e != hipSuccess) {       // the caller falls back to SpMV + the ordinary chain
                (void)hipGetLastError();
                return 0;
            }
            break;
        }
        if (r2 == 4) e = KH_CHAIN(4);
        else if (r2 == 8) e = KH_CHAIN(8);
        else if (r2 == 16) e = KH_CHAIN(16);
        else if (r2 == 24) e = KH_CHAIN(24);
        else if (r2 == 32) e = KH_CHAIN(32);
        else if (r2 == 40) e = KH_CHAIN(40);
        else if (r2 == 48 && use_long) {
            e = launch_chain_long<0>(ctx, G, a);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                long_failed = true;
                use_long = false;
                e = launch_chain<48, false, false, 0, 8>(ctx, G, a);
            }
        }
        else if (r2 == 48) e = padded ? launch_chain<48, false, false, 0, 8>(ctx, G, a) : launch_chain<48, true, false, 0, 8>(ctx, G, a);
        else e = padded ? launch_chain<56, false, false, 0, 16>(ctx, G, a) : launch_chain<56, true, false, 0, 16>(ctx, G, a);
        if (e == hipSuccess || !use_lds) break;
        (void)hipGetLastError();     // e.g. the 120 KB of dynamic LDS were refused: fall back to the plain kernel
        lds_failed = true;
        use_lds = false;
    }
#undef KH_CHAIN
#undef KH_CHAIN_PF
#undef KH_CHAIN_LDS
#undef KH_CHAIN_PLAIN
    if (e != hipSuccess) {
        // e.g. hipErrorCooperativeLaunchTooLarge: not all workgroups can be co-resident.  A property of this shape
        // on this device, not of the context: vectors of this length take the per-column kernels from now on
        (void)hipGetLastError();
        ctx->chain_refused_n = n;
        return 0;
    }
    if (a.debug == 4) ctx->chain_fault = 0;
    ctx->n_chain += 1;
    ctx->n_chain_lds += (use_lds || use_long) ? 1 : 0;
    ctx->n_chain_long += use_long ? 1 : 0;
    ctx->n_chain_pf += use_pf ? 1 : 0;
    ctx->n_chain_fused += fused ? 1 : 0;
    ctx->chain_epoch += (unsigned)(a.ncol * a.sweeps + 1);   // one grid reduction per link (complex: both parts in it) + the norm
    if (hpin == nullptr)      // (otherwise workgroup 0 has written the error word to the pinned slot itself)
        KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost,
                              ctx->stream));
    { ctx->wait_tag[slot] = a.donepin != nullptr; return 1; }
}

// ---- register-resident panel Gram-Schmidt (k_cgs_dots / k_cgs_update, chain.h) -----------------
constexpr int CGS_MAXCOL = 256;
constexpr int CGS_PSTRIDE = CH_GMAX * (CH_BS / 64);   // wave partials per column

template <int R2, bool MASKED, int WL = 0, bool CPLX = false>
static hipError_t launch_cgs(kh_ctx ctx, int G, CgsArgs& a, bool update) {
    constexpr size_t lds = (size_t)WL * CH_BS * sizeof(double2);
    if (lds > 0) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgs_update<R2, MASKED, WL, CPLX>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgs_dots<R2, MASKED, true, WL, CPLX>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgs_dots<R2, MASKED, false, WL, CPLX>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
    }
    if (update)
        hipLaunchKernelGGL((k_cgs_update<R2, MASKED, WL, CPLX>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    else if (a.nt_cols)
        hipLaunchKernelGGL((k_cgs_dots<R2, MASKED, true, WL, CPLX>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    else
        hipLaunchKernelGGL((k_cgs_dots<R2, MASKED, false, WL, CPLX>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// One Arnoldi step's panel sweeps with w register-resident.  Returns 1 when done (the norm's wave
// partials are in SLOT_NRM.., *nrm_count of them), 0 when not eligible, negative on error.
// cplx: V, B, w are complex blocks (real views of even length), hdev / coef hold (re, im) pairs and `start` counts
// complex entries of the H column; no Jacobi tail then (dg must be NULL).
int try_cgs_reg(kh_ctx ctx, kh_vec V, kh_vec B, double* w, int64_t wld, const double* dg, double* mw,
                int64_t start, int64_t ncol, int sweeps, bool multi, double* hdev, double* coef,
                int* nrm_count, bool cplx) {
    if (!ctx->chain_enabled || ncol > CGS_MAXCOL) return 0;
    if (cplx && (dg != nullptr || (V->n & 1))) return 0;
    const int cw = cplx ? 2 : 1;           // doubles per coefficient
    const int64_t n = V->n;
    int r2 = 0, G = 0;
    if (!chain_geometry(ctx, n, &r2, &G)) return 0;
    if ((n & 1) && (V->ld <= n || B->ld <= n || wld <= n)) return 0;
    const int64_t n2 = (n + 1) >> 1;
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = V->ld >= need_ld && B->ld >= need_ld && wld >= need_ld;
    if (ctx->cgs_part == nullptr)
        KH_HIP(hipMalloc(&ctx->cgs_part, sizeof(double) * (size_t)2 * CGS_MAXCOL * CGS_PSTRIDE));   // (re, im rows when complex)
    const int nwave = G * (CH_BS / 64);
    CgsArgs a;
    a.n2 = n2;
    a.chunk2 = chunk2;
    a.ld = V->ld;
    a.col0 = start;
    a.ncol = (int)ncol;
    a.w = w;
    a.pstride = CGS_PSTRIDE;
    a.dg = nullptr;
    a.mw = nullptr;
    static const int rev_env = [] {
        const char* e = getenv("KRYPY_AMD_CGS_REVERSE");
        return e ? atoi(e) : 1;
    }();
    a.reverse = rev_env;
    // the update pass re-reads what the dots pass read last from the Infinity Cache (256 MB); when the
    // panel is many times larger than that, streaming it non-temporally is worth more
    static const double nt_gb = [] {
        const char* e = getenv("KRYPY_AMD_CGS_NT_GB");
        return e ? atof(e) : 0.75;       // GB; measured: N = 10^7 668 -> 703 it/s, N/2 1315 -> 1441, N/4 2504 -> 2576, N/8 equal
    }();
    a.nt_cols = ((double)ncol * (double)n * 8.0 > nt_gb * 1e9) ? 1 : 0;
#define KH_CGS_L(R, WL, UPD)                                                                                      \
    (cplx ? (padded ? launch_cgs<R, false, WL, true>(ctx, G, a, UPD) : launch_cgs<R, true, WL, true>(ctx, G, a, UPD)) \
          : (padded ? launch_cgs<R, false, WL>(ctx, G, a, UPD) : launch_cgs<R, true, WL>(ctx, G, a, UPD)))
#define KH_CGS(R, UPD) KH_CGS_L(R, 0, UPD)
#define KH_CGS_ANY(UPD)                                                                        \
    (r2 == 4 ? KH_CGS(4, UPD) : r2 == 8 ? KH_CGS(8, UPD) : r2 == 16 ? KH_CGS(16, UPD)            \
     : r2 == 24 ? KH_CGS(24, UPD) : r2 == 32 ? KH_CGS(32, UPD) : r2 == 40 ? KH_CGS(40, UPD)     \
     : r2 == 48 ? KH_CGS_L(48, 8, UPD) : KH_CGS_L(56, 16, UPD))
    for (int s = 0; s < sweeps; ++s) {
        a.Vb = V->d;
        a.coef = nullptr;
        a.part = ctx->cgs_part;
        KH_HIP(KH_CGS_ANY(false));
        // one sweep: the coefficients ARE the H entries - reduce (and all-reduce) straight into the H
        // column, which the caller then need not clear, and skip the accumulate launch
        if (sweeps == 1) coef = hdev + cw * start;
        // (N ranks with the xr transport: the reduction of the wave partials and the sum across the ranks in ONE launch)
        KH_TRY(reduce_partials_allreduce(ctx, ctx->cgs_part, nwave, CGS_PSTRIDE, coef, (int)(cw * ncol), multi));
        if (sweeps > 1)
            hipLaunchKernelGGL(k_waxpby, dim3(1), dim3(BS), 0, ctx->stream, cw * ncol, hdev + cw * start, 1.0,
                               hdev + cw * start, 1.0, coef);
        a.Vb = B->d;
        a.ld = B->ld;
        a.coef = coef;
        a.part = part_slot(ctx, SLOT_NRM);
        a.dg = (s == sweeps - 1) ? dg : nullptr;
        a.mw = mw;
        KH_HIP(KH_CGS_ANY(true));
        a.ld = V->ld;
    }
#undef KH_CGS_ANY
#undef KH_CGS_L
#undef KH_CGS
    *nrm_count = nwave;
    ctx->n_cgs_reg += 1;
    return 1;
}

// One pass of the register-resident panel kernels over columns 0 .. ncol-1 of block X with w in registers: the dots pass
// (wave partials of <x_j, w> into ctx->cgs_part, *nwave of them per column) or the update pass (w -= sum_j coef[j] x_j).
// What try_cgs_reg does for an Arnoldi step, for callers with their own coefficients in between - the deflation projector
// on N ranks.  Returns 1 when launched, 0 when this shape is not served.
static int cgs_panel_pass(kh_ctx ctx, kh_vec X, int ncol, double* w, int64_t wld, const double* coef, bool update, int* nwave) {
    if (!ctx->chain_enabled || ncol < 1 || ncol > CGS_MAXCOL) return 0;
    const int64_t n = X->n;
    int r2 = 0, G = 0;
    // (measured, d = 16, two sweeps: 12.5 M rows - 48 rows per lane - 1075 against 1111 us for the chunked kernels; 8 M rows -
    // 32 rows per lane - 764 against 737: the long shapes only)
    if (!chain_geometry(ctx, n, &r2, &G) || r2 < 40) return 0;
    if ((n & 1) && (X->ld <= n || wld <= n)) return 0;
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = X->ld >= need_ld && wld >= need_ld;
    if (ctx->cgs_part == nullptr)
        KH_HIP(hipMalloc(&ctx->cgs_part, sizeof(double) * (size_t)2 * CGS_MAXCOL * CGS_PSTRIDE));
    CgsArgs a;
    a.n2 = (n + 1) >> 1;
    a.chunk2 = chunk2;
    a.Vb = X->d;
    a.ld = X->ld;
    a.col0 = 0;
    a.ncol = ncol;
    a.w = w;
    a.coef = coef;
    a.part = update ? part_slot(ctx, SLOT_NRM) : ctx->cgs_part;
    a.pstride = CGS_PSTRIDE;
    a.dg = nullptr;
    a.mw = nullptr;
    a.reverse = 0;
    a.nt_cols = ((double)ncol * (double)n * 8.0 > 0.75e9) ? 1 : 0;
    a.x2 = nullptr;
    a.part2 = nullptr;
#define KH_PP_L(R, WL) (padded ? launch_cgs<R, false, WL>(ctx, G, a, update) : launch_cgs<R, true, WL>(ctx, G, a, update))
    const hipError_t e = r2 == 40 ? KH_PP_L(40, 0) : r2 == 48 ? KH_PP_L(48, 8) : KH_PP_L(56, 16);
#undef KH_PP_L
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    *nwave = G * (CH_BS / 64);
    return 1;
}

// ---- reference-order Gram-Schmidt with one reduction per step (chain.h: k_cgs_dots<..., X2>, k_lowsync_solve) -----
template <int R2, bool MASKED>
static hipError_t launch_dots_x2(kh_ctx ctx, int G, CgsArgs& a) {
    constexpr size_t lds = (size_t)(R2 < 16 ? R2 : 16) * CH_BS * sizeof(double2);      // (the rows of x beyond sixteen: registers)
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgs_dots<R2, MASKED, false, 0, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_cgs_dots<R2, MASKED, false, 0, false, true>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

static __global__ void k_ls_put_column(double* gt, int j, const double* vals) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < j) gt[(size_t)m * LS_MAXCOL + j] = vals[m];
}

// One Arnoldi step's reference-order Gram-Schmidt on N ranks (or a 1-rank communicator in forced mode) with ONE
// all-reduce for the k + 1 coefficients: dots pass with w register-resident (all columns against the un-updated w, and
// against v_k for the new column of the Gram table), reduce, all-reduce of 2 k + 1 values, forward substitution on the
// device, update pass.  Returns 1 when done (norm partials in SLOT_NRM, *nrm_count of them), 0 when not eligible.
// The Gram table (ctx->ls_tab) belongs to ONE Arnoldi sequence, like the blocked kernel's: (ls_V, ls_next) name the
// basis block and the step that finds columns 0 .. k-1 of it valid; any other step rebuilds them from the basis first.
// The LONGEST slab of the run for a step on local vectors of length n: what the operator of the step was told
// (kh_mat_set_rows_max: keyed on the operator, ADVICE r05 - two sharded operators of equal local length but different longest
// slabs used to share one entry of the context's table, and the last announcement won), else the context's table by local
// length (kh_ctx_set "lowsync_rows": steps without an operator), else 0 = unknown.
static int64_t longest_slab(kh_ctx ctx, kh_mat A, int64_t n) {
    if (ctx->nranks <= 1) return n;
    if (A != nullptr && A->rows_max > 0) return A->rows_max;
    int64_t nmax = 0;
    for (int i = 0; i < 4; ++i)
        if (ctx->ls_rows_local[i] == n) nmax = ctx->ls_rows_max[i];
    return nmax;
}

static bool lowsync_eligible(kh_ctx ctx, kh_vec V, int64_t wld, int64_t k, kh_mat A = nullptr) {
    if (!ctx->mgs_lowsync || !ctx->chain_configured || k + 1 > LS_MAXCOL) return false;
    const int64_t n = V->n;
    int r2 = 0, G = 0;
    // N ranks: this choice changes the PATTERN of all-reduces (one of 2 k + 1 values against k + 1 of one value), so every
    // rank must make it alike whatever the length of its own slab: it is made for the longest slab of the run, which the
    // host layer has announced for vectors of this local length (krypy_amd/dist.py: kh_ctx_set "lowsync_rows"); without
    // that announcement the per-link path - the same on every rank - is taken
    if (ctx->nranks > 1) {
        const int64_t nmax = longest_slab(ctx, A, n);
        int r2m = 0, Gm = 0;
        if (nmax < n || !chain_geometry(ctx, nmax, &r2m, &Gm) || r2m > 24) return false;
    }
    if (!chain_geometry(ctx, n, &r2, &G) || r2 > 24) return false;      // (the second right-hand side: 16 rows of 8 KB in LDS, 8 more in registers;
                                                                        //  at 32 rows per lane the dots kernel spills 102 registers)
    if ((n & 1) && (V->ld <= n || wld <= n)) return false;
    return true;
}

static int try_lowsync_mgs(kh_ctx ctx, kh_vec V, double* w, int64_t wld, int64_t k, bool multi, double* hdev, double* coef,
                           int* nrm_count, kh_mat A = nullptr) {
    if (!lowsync_eligible(ctx, V, wld, k, A)) return 0;
    const int64_t n = V->n;
    int r2 = 0, G = 0;
    chain_geometry(ctx, n, &r2, &G);
    const int64_t n2 = (n + 1) >> 1;
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = V->ld >= need_ld && wld >= need_ld;
    if (ctx->cgs_part == nullptr)
        KH_HIP(hipMalloc(&ctx->cgs_part, sizeof(double) * (size_t)2 * CGS_MAXCOL * CGS_PSTRIDE));
    if (ctx->ls_tab == nullptr) {
        KH_HIP(hipMalloc(&ctx->ls_tab, sizeof(double) * (size_t)LS_MAXCOL * LS_MAXCOL));
        KH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lowsync_solve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(double) * ((size_t)LS_MAXCOL * LS_MAXCOL + 2))));
    }
    const int ncol = (int)(k + 1);
    const int nwave = G * (CH_BS / 64);
    if (!(ctx->ls_V == V && ctx->ls_next == k)) {
        // columns 1 .. k-1 of the table from the basis (a sequence that did not start here: a grown / recycled block, a
        // step re-run after a timeout): one panel product per column, once
        for (int64_t j = 1; j < k; ++j) {
            KH_TRY(::dot_panel_dev(ctx, V, 0, j, V->col(j), coef, 0));
            if (multi) KH_TRY(comm_allreduce_dev(ctx, coef, j));
            hipLaunchKernelGGL(k_ls_put_column, dim3((unsigned)((j + 127) / 128)), dim3(128), 0, ctx->stream, ctx->ls_tab, (int)j, coef);
        }
        KH_HIP(hipGetLastError());
        ctx->ls_V = V;
        ctx->n_ls_rebuild += 1;
    }
    ctx->ls_next = -1;
    CgsArgs a;
    a.n2 = n2;
    a.chunk2 = chunk2;
    a.Vb = V->d;
    a.ld = V->ld;
    a.col0 = 0;
    a.ncol = ncol;
    a.w = w;
    a.coef = nullptr;
    a.part = ctx->cgs_part;
    a.pstride = CGS_PSTRIDE;
    a.dg = nullptr;
    a.mw = nullptr;
    a.reverse = 1;
    a.nt_cols = 0;
    a.x2 = V->col(k);
    a.part2 = ctx->cgs_part + (size_t)ncol * CGS_PSTRIDE;
    hipError_t e;
#define KH_DX2(R) (padded ? launch_dots_x2<R, false>(ctx, G, a) : launch_dots_x2<R, true>(ctx, G, a))
    e = (r2 == 4) ? KH_DX2(4) : (r2 == 8 ? KH_DX2(8) : (r2 == 16 ? KH_DX2(16) : KH_DX2(24)));
#undef KH_DX2
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // On N ranks this form has a different PATTERN of all-reduces than the per-link path (one of 2 k + 1 values against
        // k + 1 of one value): a rank that declined here on its own while its peers went ahead would mismatch the collectives
        // (a hang, or sums of unrelated numbers).  Every decision up to this point (lowsync_eligible: switches, the longest
        // slab of the run, k; the table rebuild: (ls_V, ls_next), which follow the call sequence every rank makes alike) is
        // the same on every rank - a launch failure is not, so with a communicator it is an error, never a fallback.
        if (multi)
            return fail(KH_ERR_HIP, "one-reduction Gram-Schmidt: the dots kernel could not be launched (%s); no rank-local "
                                    "fallback on a communicator (the peers would wait in a different all-reduce)",
                        hipGetErrorString(e));
        return 0;
    }
    // [c_0 .. c_k | g_0 .. g_{k-1}]: one reduction launch, ONE all-reduce
    double* cg = ctx->scal + SC_LS;
    KH_TRY(reduce_partials_allreduce(ctx, ctx->cgs_part, nwave, CGS_PSTRIDE, cg, 2 * ncol - 1, multi));
    hipLaunchKernelGGL(k_lowsync_solve, dim3(1), dim3(LS_MAXCOL), sizeof(double) * ((size_t)k * (k + 1) + 2), ctx->stream, (int)k, cg,
                       ctx->ls_tab, coef, hdev);
    KH_HIP(hipGetLastError());
    a.coef = coef;
    a.part = part_slot(ctx, SLOT_NRM);
#define KH_UPD(R) (padded ? launch_cgs<R, false, 0>(ctx, G, a, true) : launch_cgs<R, true, 0>(ctx, G, a, true))
    KH_HIP((r2 == 4) ? KH_UPD(4) : (r2 == 8 ? KH_UPD(8) : (r2 == 16 ? KH_UPD(16) : KH_UPD(24))));
#undef KH_UPD
    *nrm_count = nwave;
    ctx->ls_V = V;
    ctx->ls_next = k + 1;
    ctx->n_lowsync += 1;
    return 1;
}

int grid_lin(kh_ctx ctx, int64_t n) {
    int64_t need = (n + BS - 1) / BS;
    if (need < 1) need = 1;
    return (int)std::min<int64_t>(need, (int64_t)ctx->nb * 2);
}

}  // namespace kh

using namespace kh;

// =============================================================================================
// C ABI
// =============================================================================================
// complex operators (zpath.h, included at the end of this file)
static int zapply_cols(kh_ctx ctx, kh_mat A, kh_vec X, int64_t xcol, kh_vec Y, int64_t ycol, int64_t ncols);

extern "C" {

const char* kh_last_error(void) { return g_err.c_str(); }

int kh_version(void) { return 100; }

int kh_device_count(int* count) {
    KH_ARG(count != nullptr, "kh_device_count: NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(KH_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return 0;
}

int kh_ctx_create(int device, kh_ctx* out) {
    KH_ARG(out != nullptr, "kh_ctx_create: NULL out");
    int ndev = 0;
    KH_HIP(hipGetDeviceCount(&ndev));
    KH_ARG(device >= 0 && device < ndev, "kh_ctx_create: device %d not in [0,%d)", device, ndev);
    KH_HIP(hipSetDevice(device));
    kh_ctx ctx = new kh_ctx_s();
    ctx->device = device;
    hipDeviceProp_t prop;
    KH_HIP(hipGetDeviceProperties(&prop, device));
    ctx->ncu = prop.multiProcessorCount;
    ctx->nb = std::min(NB_MAX, std::max(64, ctx->ncu * 4));
    KH_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    KH_HIP(hipMalloc(&ctx->part, sizeof(double) * (size_t)(MAXC + 4) * NB_MAX));
    KH_HIP(hipMalloc(&ctx->scal, sizeof(double) * SCAL_CAP));
    KH_HIP(hipMemset(ctx->scal, 0, sizeof(double) * SCAL_CAP));
    KH_HIP(hipHostMalloc(&ctx->hpin, sizeof(double) * SCAL_CAP, hipHostMallocDefault));
    KH_HIP(hipEventCreate(&ctx->ev0));
    KH_HIP(hipEventCreate(&ctx->ev1));
    KH_HIP(hipMalloc(&ctx->chain_gran, sizeof(unsigned long long) * 4 * CH_GMAX));
    KH_HIP(hipMemset(ctx->chain_gran, 0, sizeof(unsigned long long) * 4 * CH_GMAX));
    // XCD-leader hand-off of the chain kernel's grid-wide sums: [16][2][4] result granules + [16] election stamps
    KH_HIP(hipMalloc(&ctx->chain_xcc, sizeof(unsigned long long) * 128 + sizeof(unsigned) * 16));
    KH_HIP(hipMalloc(&ctx->onex_ticket, sizeof(unsigned) * 256));
    KH_HIP(hipMemset(ctx->onex_ticket, 0, sizeof(unsigned) * 256));
    KH_HIP(hipMemset(ctx->chain_xcc, 0, sizeof(unsigned long long) * 128 + sizeof(unsigned) * 16));
    KH_HIP(hipMalloc(&ctx->chain_err, sizeof(int)));
    KH_HIP(hipMemset(ctx->chain_err, 0, sizeof(int)));
    for (int s = 0; s < KH_NSLOT; ++s) {
        KH_HIP(hipHostMalloc(&ctx->chain_err_pin[s], sizeof(int), hipHostMallocDefault));
        *ctx->chain_err_pin[s] = 0;
        KH_HIP(hipHostMalloc(&ctx->done_pin[s], sizeof(int), hipHostMallocCoherent));          // (fine-grained: the tag is visible when it is written)
        *ctx->done_pin[s] = 0;
    }
    {
        const char* e = getenv("KRYPY_AMD_MGS_CHAIN");
        ctx->chain_enabled = (e == nullptr) ? 1 : atoi(e);
        if (ctx->ncu > CH_GMAX) ctx->chain_enabled = 0;
        ctx->chain_configured = ctx->chain_enabled;
        e = getenv("KRYPY_AMD_CHAIN_SPMV");
        ctx->chain_spmv = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_LDS");
        ctx->chain_lds = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_PF");
        ctx->chain_pf = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_TAG_WAIT");
        ctx->tag_wait = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_SMALL");
        ctx->chain_small = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_BLK");
        ctx->chain_blk = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_BLK2");
        ctx->chain_blk2 = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_LONG");
        ctx->chain_long = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_GRAM_MFMA");
        ctx->gram_mfma = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_CHAIN_XR");
        ctx->chain_xr = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_BLK2_CW");
        ctx->blk2_cw = (e == nullptr) ? 1 : atoi(e);
        ctx->blk2_cw_maxrows = ctx->blk2_cw == 2 ? 6 : 7;
        e = getenv("KRYPY_AMD_BLK2_ONE");
        ctx->blk2_one = (e == nullptr) ? 2 : atoi(e);
        e = getenv("KRYPY_AMD_SPMV_WIN");
        ctx->spmv_win = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_PROJ_REG");
        ctx->proj_reg = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_PROJ_PANEL");
        ctx->proj_panel = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_MGS_LOWSYNC");
        ctx->mgs_lowsync = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_BLK_ONEX_MAXN");
        if (e != nullptr) ctx->blk_onex_maxn = atoll(e);
        e = getenv("KRYPY_AMD_BLK_NX");
        if (e != nullptr) ctx->blk_nx = atoi(e) > 0 ? 8 : 0;
        e = getenv("KRYPY_AMD_CHAIN_ONEX");
        ctx->chain_onex = (e == nullptr) ? 1 : atoi(e);
        if (ctx->chain_onex) {
            // the one-XCD launches rely on workgroups being dealt round-robin over the XCDs (8 G + 8 launched, G + 1
            // land on each): checked once here - a device that deals differently (a partition mode, masked CUs)
            // simply keeps the spread launches
            hipLaunchKernelGGL(k_onex_probe, dim3(8 * 32 + 8), dim3(64), 0, ctx->stream, ctx->onex_ticket);
            unsigned cnt[16];
            KH_HIP(hipMemcpyAsync(cnt, ctx->onex_ticket, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
            KH_HIP(hipStreamSynchronize(ctx->stream));
            KH_HIP(hipMemset(ctx->onex_ticket, 0, sizeof(unsigned) * 256));
            if (cnt[0] < 33u) ctx->chain_onex = 0;
        }
        e = getenv("KRYPY_AMD_LANCZOS_FUSED");
        ctx->lanczos_fused = (e == nullptr) ? 1 : atoi(e);
        e = getenv("KRYPY_AMD_ROCTX");
        ctx->roctx = (e == nullptr) ? 0 : atoi(e);
    }
    *out = ctx;
    return 0;
}

int kh_ctx_destroy(kh_ctx ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    kh_comm_destroy(ctx);
    if (ctx->xr_err_pin != nullptr) (void)hipHostFree(ctx->xr_err_pin);
    ctx->xr_err_pin = nullptr;
    for (int s = 0; s < KH_NSLOT; ++s) {
        if (ctx->hslot_dev[s]) (void)hipFree(ctx->hslot_dev[s]);
        if (ctx->hslot_pin[s]) (void)hipHostFree(ctx->hslot_pin[s]);
        if (ctx->hev[s]) (void)hipEventDestroy(ctx->hev[s]);
    }
    (void)hipFree(ctx->cgs_part);
    (void)hipFree(ctx->chain_gran);
    (void)hipFree(ctx->chain_xcc);
    (void)hipFree(ctx->onex_ticket);
    chain_blk_free(ctx);
    proj_reg_free(ctx);
    if (ctx->ls_tab != nullptr) (void)hipFree(ctx->ls_tab);
    (void)hipFree(ctx->chain_err);
    for (int s = 0; s < KH_NSLOT; ++s) {
        if (ctx->chain_err_pin[s]) (void)hipHostFree(ctx->chain_err_pin[s]);
        if (ctx->done_pin[s]) (void)hipHostFree(ctx->done_pin[s]);
    }
    (void)hipFree(ctx->part);
    if (ctx->gram_part) (void)hipFree(ctx->gram_part);
    (void)hipFree(ctx->scal);
    (void)hipHostFree(ctx->hpin);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int kh_ctx_sync(kh_ctx ctx) {
    KH_ARG(ctx != nullptr, "kh_ctx_sync: NULL ctx");
    KH_HIP(hipStreamSynchronize(ctx->stream));
    return xr_check(ctx);
}

int kh_ctx_info(kh_ctx ctx, int64_t info[4]) {
    KH_ARG(ctx != nullptr && info != nullptr, "kh_ctx_info: NULL");
    size_t fr = 0, tot = 0;
    KH_HIP(hipSetDevice(ctx->device));
    KH_HIP(hipMemGetInfo(&fr, &tot));
    info[0] = ctx->ncu;
    info[1] = (int64_t)tot;
    info[2] = (int64_t)fr;
    info[3] = ctx->nb;
    return 0;
}

int kh_ctx_counters(kh_ctx ctx, int64_t out[4]) {
    KH_ARG(ctx != nullptr && out != nullptr, "kh_ctx_counters: NULL");
    out[0] = ctx->n_chain;
    out[1] = ctx->n_chain_lds;
    out[2] = ctx->n_chain_fused;
    out[3] = ctx->n_cgs_reg;
    return 0;
}

int kh_ctx_tune(kh_ctx ctx, int reduce_blocks, int spmv_tile) {
    KH_ARG(ctx != nullptr, "kh_ctx_tune: NULL ctx");
    if (reduce_blocks > 0) {
        KH_ARG(reduce_blocks <= NB_MAX, "reduce_blocks %d > %d", reduce_blocks, NB_MAX);
        ctx->nb = reduce_blocks;
    }
    if (spmv_tile > 0) {
        KH_ARG(spmv_tile == 1024 || spmv_tile == 2048 || spmv_tile == 4096,
               "spmv_tile %d must be 1024, 2048 or 4096", spmv_tile);
        ctx->spmv_tile = spmv_tile;
    }
    return 0;
}

int kh_ctx_set(kh_ctx ctx, const char* key, int64_t value) {
    KH_ARG(ctx != nullptr && key != nullptr, "kh_ctx_set: NULL");
    if (!strcmp(key, "spmv_dia")) ctx->spmv_dia = value != 0;
    else if (!strcmp(key, "spmv_win")) ctx->spmv_win = value != 0;
    else if (!strcmp(key, "chain")) {
        ctx->chain_enabled = ctx->chain_configured = (value != 0 && ctx->ncu <= CH_GMAX);
        ctx->chain_recoveries = 0;          // (an explicit setting starts the count again)
        ctx->chain_clean_steps = 0;
        ctx->chain_refused_n = -1;
    }
    else if (!strcmp(key, "chain_lds")) ctx->chain_lds = value != 0;
    else if (!strcmp(key, "chain_pf")) ctx->chain_pf = (int)value;
    else if (!strcmp(key, "chain_spmv")) ctx->chain_spmv = value != 0;
    else if (!strcmp(key, "chain_fault")) ctx->chain_fault = value != 0;
    else if (!strcmp(key, "spmv_split")) ctx->spmv_split = value != 0;
    else if (!strcmp(key, "halo_loopback")) ctx->halo_loopback = value != 0;
    else if (!strcmp(key, "lanczos_fused")) ctx->lanczos_fused = value != 0;
    else if (!strcmp(key, "chain_onex")) ctx->chain_onex = value != 0;
    else if (!strcmp(key, "chain_small")) ctx->chain_small = value != 0;
    else if (!strcmp(key, "chain_blk")) { ctx->chain_blk = value != 0; ctx->blk_refused_n = -1; }
    else if (!strcmp(key, "chain_blk2")) { ctx->chain_blk2 = value != 0; ctx->blk2_refused_n = -1; }
    else if (!strcmp(key, "chain_blk2_one")) { ctx->blk2_one = (int)value; ctx->blk2_refused_n = -1; }
    else if (!strcmp(key, "chain_xr")) ctx->chain_xr = value != 0;
    else if (!strcmp(key, "chain_long")) ctx->chain_long = value != 0;
    else if (!strcmp(key, "gram_mfma")) ctx->gram_mfma = value != 0;
    else if (!strcmp(key, "chain_xr_cus")) ctx->chain_xr_cus = (int)value;      // tests: shapes for this many compute units (0: all)
    else if (!strcmp(key, "gemv_rows")) ctx->gemv_rows = (int)value;       // rows per wave of the dense GEMV (0: by size; 1 / 2 / 4)
    else if (!strcmp(key, "chain_blk2_cw")) {       // 1: a communication wave, 4 ... 7 rows; 2: the same up to 6 rows; 0: 512 lanes with rows
        ctx->blk2_cw = (int)value;
        ctx->blk2_cw_maxrows = value == 2 ? 6 : 7;
        ctx->blk2_refused_n = -1;
    }
    else if (!strcmp(key, "mgs_lowsync")) ctx->mgs_lowsync = value != 0;
    else if (!strcmp(key, "lowsync_rows")) {        // (local slab length << 32) | longest slab of the run
        const int64_t loc = value >> 32, mx = value & 0xffffffffll;
        int slot = ctx->ls_rows_n % 4;
        for (int i = 0; i < 4; ++i)
            if (ctx->ls_rows_local[i] == loc) slot = i;
        if (ctx->ls_rows_local[slot] != loc) ctx->ls_rows_n += 1;
        ctx->ls_rows_local[slot] = loc;
        ctx->ls_rows_max[slot] = mx;
    }
    else if (!strcmp(key, "xr")) {
        // the cross-rank sums through IPC mailboxes (xr.hip).  EVERY rank must make the same setting: the host layer
        // (krypy_amd/dist.py: enable_xr) switches it on after all ranks have attached
        if (value != 0) {
            KH_ARG(ctx->xr_nranks > 0, "kh_ctx_set(\"xr\", 1): kh_xr_attach first");
            if (ctx->comm == nullptr) {          // no RCCL communicator: the attach defines the ranks
                ctx->rank = ctx->xr_rank;
                ctx->nranks = ctx->xr_nranks;
                ctx->xr_own_comm = 1;
            }
            ctx->xr_on = 1;
        } else {
            ctx->xr_on = 0;
            if (ctx->xr_own_comm) {
                ctx->rank = 0;
                ctx->nranks = 1;
                ctx->xr_own_comm = 0;
            }
        }
    }
    else if (!strcmp(key, "xr_timeout_ms")) ctx->xr_timeout_ms = value;
    else if (!strcmp(key, "proj_reg")) ctx->proj_reg = value != 0;
    else if (!strcmp(key, "proj_fault")) ctx->proj_fault = value != 0;
    else if (!strcmp(key, "proj_panel")) ctx->proj_panel = value != 0;
    else if (!strcmp(key, "blk_onex_maxn")) ctx->blk_onex_maxn = value;
    else if (!strcmp(key, "blk_nx")) ctx->blk_nx = value > 0 ? 8 : 0;
    else if (!strcmp(key, "tag_wait")) ctx->tag_wait = value != 0;
    else if (!strcmp(key, "chain_epoch")) ctx->chain_epoch = (unsigned)value;      // tests: bring the epoch counter of the grid-wide sums near its wrap
    else if (!strcmp(key, "chain_debug")) ctx->chain_debug = (int)value;    // measurement: phases switched off (garbage results)
    else return fail(KH_ERR_ARG, "kh_ctx_set: unknown key '%s'", key);
    return 0;
}

int kh_ctx_get(kh_ctx ctx, const char* key, int64_t* value) {
    KH_ARG(ctx != nullptr && key != nullptr && value != nullptr, "kh_ctx_get: NULL");
    if (!strcmp(key, "spmv_dia")) *value = ctx->spmv_dia;
    else if (!strcmp(key, "spmv_win")) *value = ctx->spmv_win;
    else if (!strcmp(key, "n_spmv_win")) *value = ctx->n_spmv_win;
    else if (!strcmp(key, "chain")) *value = ctx->chain_enabled;
    else if (!strcmp(key, "chain_recoveries")) *value = ctx->chain_recoveries;
    else if (!strcmp(key, "n_chain_rearmed")) *value = ctx->n_chain_rearmed;
    else if (!strcmp(key, "chain_lds")) *value = ctx->chain_lds;
    else if (!strcmp(key, "chain_pf")) *value = ctx->chain_pf;
    else if (!strcmp(key, "n_chain_pf")) *value = ctx->n_chain_pf;
    else if (!strcmp(key, "chain_spmv")) *value = ctx->chain_spmv;
    else if (!strcmp(key, "n_spmm")) *value = ctx->n_spmm;
    else if (!strcmp(key, "spmv_split")) *value = ctx->spmv_split;
    else if (!strcmp(key, "n_spmv_split")) *value = ctx->n_spmv_split;
    else if (!strcmp(key, "halo_loopback")) *value = ctx->halo_loopback;
    else if (!strcmp(key, "lanczos_fused")) *value = ctx->lanczos_fused;
    else if (!strcmp(key, "n_lanczos_fused")) *value = ctx->n_lanczos_fused;
    else if (!strcmp(key, "n_minres_rides")) *value = ctx->n_minres_rides;
    else if (!strcmp(key, "chain_onex")) *value = ctx->chain_onex;
    else if (!strcmp(key, "n_chain_onex")) *value = ctx->n_chain_onex;
    else if (!strcmp(key, "chain_small")) *value = ctx->chain_small;
    else if (!strcmp(key, "chain_blk")) *value = ctx->chain_blk;
    else if (!strcmp(key, "n_chain_blk")) *value = ctx->n_chain_blk;
    else if (!strcmp(key, "chain_blk2")) *value = ctx->chain_blk2;
    else if (!strcmp(key, "chain_blk2_cw")) *value = ctx->blk2_cw;
    else if (!strcmp(key, "gemv_rows")) *value = ctx->gemv_rows;
    else if (!strcmp(key, "chain_blk2_one")) *value = ctx->blk2_one;
    else if (!strcmp(key, "n_chain_blk2")) *value = ctx->n_chain_blk2;
    else if (!strcmp(key, "chain_xr")) *value = ctx->chain_xr;
    else if (!strcmp(key, "chain_long")) *value = ctx->chain_long;
    else if (!strcmp(key, "n_chain_long")) *value = ctx->n_chain_long;
    else if (!strcmp(key, "gram_mfma")) *value = ctx->gram_mfma;
    else if (!strcmp(key, "n_gram_mfma")) *value = ctx->n_gram_mfma;
    else if (!strcmp(key, "n_panel_gemm")) *value = ctx->n_panel_gemm;
    else if (!strcmp(key, "n_zspmv_dia")) *value = ctx->n_zspmv_dia;
    else if (!strcmp(key, "n_chain_xr")) *value = ctx->n_chain_xr;
    else if (!strcmp(key, "n_blk_rebuild")) *value = ctx->n_blk_rebuild;
    else if (!strcmp(key, "n_blk_rowless")) *value = ctx->n_blk_rowless;
    else if (!strcmp(key, "blk_nx")) *value = ctx->blk_nx;
    else if (!strcmp(key, "tag_wait")) *value = ctx->tag_wait;
    else if (!strcmp(key, "n_tag_waits")) *value = ctx->n_tag_waits;
    else if (!strcmp(key, "n_chain_small")) *value = ctx->n_chain_small;
    else if (!strcmp(key, "n_cycle_steps")) *value = ctx->n_cycle_steps;
    else if (!strcmp(key, "n_minres_cycle_steps")) *value = ctx->n_minres_cycle_steps;
    else if (!strcmp(key, "n_cg_cycle_steps")) *value = ctx->n_cg_cycle_steps;
    else if (!strcmp(key, "mgs_lowsync")) *value = ctx->mgs_lowsync;
    else if (!strcmp(key, "proj_reg")) *value = ctx->proj_reg;
    else if (!strcmp(key, "xr")) *value = ctx->xr_on;
    else if (!strcmp(key, "n_xr")) *value = ctx->n_xr;
    else if (!strcmp(key, "n_xr_fused")) *value = ctx->n_xr_fused;
    else if (!strcmp(key, "n_proj_reg")) *value = ctx->n_proj_reg;
    else if (!strcmp(key, "n_proj_recovered")) *value = ctx->n_proj_recovered;
    else if (!strcmp(key, "proj_reg_why")) *value = ctx->proj_reg_why;
    else if (!strcmp(key, "proj_panel")) *value = ctx->proj_panel;
    else if (!strcmp(key, "n_proj_panel")) *value = ctx->n_proj_panel;
    else if (!strcmp(key, "n_lowsync")) *value = ctx->n_lowsync;
    else if (!strcmp(key, "n_ls_rebuild")) *value = ctx->n_ls_rebuild;
    else if (!strcmp(key, "n_halo_exchange")) *value = ctx->n_halo_exchange;
    else if (!strcmp(key, "n_halo_xh")) *value = ctx->n_halo_xh;
    else if (!strcmp(key, "n_allreduce")) *value = ctx->n_allreduce;
    else if (!strcmp(key, "n_chain_recovered")) *value = ctx->n_chain_recovered;
    else if (!strcmp(key, "chain_epoch")) *value = ctx->chain_epoch;
    else if (!strcmp(key, "n_epoch_wraps")) *value = ctx->n_epoch_wraps;
    else return fail(KH_ERR_ARG, "kh_ctx_get: unknown key '%s'", key);
    return 0;
}

int kh_timer_start(kh_ctx ctx) {
    KH_ARG(ctx != nullptr, "kh_timer_start: NULL ctx");
    KH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int kh_timer_stop(kh_ctx ctx, double* elapsed_ms) {
    KH_ARG(ctx != nullptr && elapsed_ms != nullptr, "kh_timer_stop: NULL");
    KH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    KH_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    KH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return 0;
}

// ---- vectors ----------------------------------------------------------------------------------
int kh_vec_alloc(kh_ctx ctx, int64_t n, int64_t ncols, kh_vec* out) {
    KH_ARG(ctx != nullptr && out != nullptr, "kh_vec_alloc: NULL");
    KH_ARG(n >= 0 && ncols >= 0, "kh_vec_alloc: negative shape");
    kh_vec v = new kh_vec_s();
    v->ctx = ctx;
    v->n = n;
    v->ncols = ncols;
    v->ld = padded_ld(ctx, n);
    v->d = nullptr;
    // CH_SLACK zeroed doubles behind the last column: the register-resident chain kernel reads
    // whole CH_BS-strided rows without clamping (chain.h)
    const size_t bytes = sizeof(double) * ((size_t)v->ld * (size_t)std::max<int64_t>(ncols, 1) + CH_SLACK);
    hipError_t e = hipMalloc(&v->d, bytes);
    if (e != hipSuccess) {
        delete v;
        return fail(KH_ERR_NOMEM, "kh_vec_alloc: hipMalloc(%zu bytes) for a (%lld x %lld) block: %s",
                    bytes, (long long)n, (long long)ncols, hipGetErrorString(e));
    }
    KH_HIP(hipMemsetAsync(v->d, 0, bytes, ctx->stream));
    *out = v;
    return 0;
}

// a freed handle must not be dereferenced by the recovery of a chain timeout (kh_arnoldi_step_end re-runs the step
// from the slot's record): forget every record that names it
static void forget_steps(kh_ctx ctx, const void* handle) {
    if (ctx->blk_V == handle) { ctx->blk_V = nullptr; ctx->blk_next = -1; }
    if (ctx->ls_V == handle) { ctx->ls_V = nullptr; ctx->ls_next = -1; }
    if (ctx->mr_pending.on && (ctx->mr_pending.V == handle || ctx->mr_pending.W == handle || ctx->mr_pending.YK == handle))
        ctx->mr_pending.on = 0;          // (kh_vec_free has run the update already; a matrix / projector is never named)
    for (int s = 0; s < KH_NSLOT; ++s) {
        kh_step_s& st = ctx->step[s];
        if (st.kind != 0 && (st.A == handle || st.Md == handle || st.proj == handle || st.V == handle ||
                             st.P == handle || st.W == handle))
            st = kh_step_s();
    }
}


int kh_vec_free(kh_vec v) {
    if (!v) return 0;
    // A deferred MINRES update that names this block runs NOW: the basis of an unwindowed MINRES grows by moving to a
    // new block (utils.Arnoldi._grow) while the previous iteration's update still points at the old one - dropping the
    // job would leave W and yk one recurrence step short.
    {
        const auto& j = v->ctx->mr_pending;
        if (j.on && (j.V == v || j.W == v || j.YK == v)) (void)kh_minres_flush(v->ctx);
    }
    (void)hipStreamSynchronize(v->ctx->stream);
    forget_steps(v->ctx, v);
    (void)hipFree(v->d);
    delete v;
    return 0;
}

int kh_vec_shape(kh_vec v, int64_t* n, int64_t* ncols, int64_t* ld) {
    KH_ARG(v != nullptr, "kh_vec_shape: NULL");
    if (n) *n = v->n;
    if (ncols) *ncols = v->ncols;
    if (ld) *ld = v->ld;
    return 0;
}

int kh_vec_upload(kh_vec v, int64_t col0, int64_t ncols, const double* host, int64_t host_ld) {
    KH_TRY(check_vec(v, col0, ncols, "kh_vec_upload"));
    chain_blk_touch(v->ctx, v);
    if (ncols == 0 || v->n == 0) return 0;
    KH_ARG(host != nullptr && host_ld >= v->n, "kh_vec_upload: bad host buffer");
    KH_HIP(hipMemcpy2DAsync(v->col(col0), v->ld * sizeof(double), host, host_ld * sizeof(double),
                            v->n * sizeof(double), ncols, hipMemcpyHostToDevice, v->ctx->stream));
    KH_HIP(hipStreamSynchronize(v->ctx->stream));
    return 0;
}

int kh_vec_download(kh_vec v, int64_t col0, int64_t ncols, double* host, int64_t host_ld) {
    KH_TRY(check_vec(v, col0, ncols, "kh_vec_download"));
    if (ncols == 0 || v->n == 0) return 0;
    KH_ARG(host != nullptr && host_ld >= v->n, "kh_vec_download: bad host buffer");
    KH_HIP(hipMemcpy2DAsync(host, host_ld * sizeof(double), v->col(col0), v->ld * sizeof(double),
                            v->n * sizeof(double), ncols, hipMemcpyDeviceToHost, v->ctx->stream));
    KH_HIP(hipStreamSynchronize(v->ctx->stream));
    return 0;
}

int kh_vec_zero(kh_vec v, int64_t col0, int64_t ncols) {
    KH_TRY(check_vec(v, col0, ncols, "kh_vec_zero"));
    chain_blk_touch(v->ctx, v);
    if (ncols == 0) return 0;
    KH_HIP(hipMemsetAsync(v->col(col0), 0, sizeof(double) * v->ld * ncols, v->ctx->stream));
    return 0;
}

int kh_vec_copy(kh_vec dst, int64_t dcol, kh_vec src, int64_t scol, int64_t ncols) {
    KH_TRY(check_vec(dst, dcol, ncols, "kh_vec_copy(dst)"));
    chain_blk_touch(dst->ctx, dst);
    KH_TRY(check_vec(src, scol, ncols, "kh_vec_copy(src)"));
    KH_ARG(dst->n == src->n, "kh_vec_copy: length mismatch %lld vs %lld", (long long)dst->n,
           (long long)src->n);
    if (ncols == 0 || (dst == src && dcol == scol)) return 0;
    if (dst == src && dcol < scol + ncols && scol < dcol + ncols) {
        // overlapping ranges of one block (a sliding window moving to the front): column by column, in
        // the order that never overwrites a column before it has been read
        const bool fwd = dcol < scol;
        for (int64_t i = 0; i < ncols; ++i) {
            const int64_t c = fwd ? i : ncols - 1 - i;
            KH_HIP(hipMemcpyAsync(dst->col(dcol + c), src->col(scol + c), sizeof(double) * src->ld,
                                  hipMemcpyDeviceToDevice, dst->ctx->stream));
        }
        return 0;
    }
    KH_ARG(dst->ld == src->ld || ncols == 1, "kh_vec_copy: blocks with different leading dimensions");
    KH_HIP(hipMemcpyAsync(dst->col(dcol), src->col(scol), sizeof(double) * src->ld * ncols,
                          hipMemcpyDeviceToDevice, dst->ctx->stream));
    return 0;
}

static int check_range(kh_vec v, int64_t col, int64_t i0, int64_t count, const char* what) {
    KH_TRY(check_vec(v, col, 1, what));
    KH_ARG(i0 >= 0 && count >= 0 && i0 + count <= v->n, "%s: entries [%lld, %lld) out of range (n=%lld)", what,
           (long long)i0, (long long)(i0 + count), (long long)v->n);
    return 0;
}

int kh_vec_get(kh_vec v, int64_t col, int64_t i0, int64_t count, double* out) {
    KH_TRY(check_range(v, col, i0, count, "kh_vec_get"));
    if (count == 0) return 0;
    KH_ARG(out != nullptr, "kh_vec_get: NULL");
    KH_HIP(hipMemcpyAsync(out, v->col(col) + i0, sizeof(double) * count, hipMemcpyDeviceToHost, v->ctx->stream));
    KH_HIP(hipStreamSynchronize(v->ctx->stream));
    return 0;
}

int kh_vec_set(kh_vec v, int64_t col, int64_t i0, int64_t count, const double* in) {
    KH_TRY(check_range(v, col, i0, count, "kh_vec_set"));
    chain_blk_touch(v->ctx, v);
    if (count == 0) return 0;
    KH_ARG(in != nullptr, "kh_vec_set: NULL");
    KH_HIP(hipMemcpyAsync(v->col(col) + i0, in, sizeof(double) * count, hipMemcpyHostToDevice, v->ctx->stream));
    KH_HIP(hipStreamSynchronize(v->ctx->stream));
    return 0;
}

int kh_vec_zero_range(kh_vec v, int64_t col, int64_t i0, int64_t count) {
    KH_TRY(check_range(v, col, i0, count, "kh_vec_zero_range"));
    chain_blk_touch(v->ctx, v);
    if (count == 0) return 0;
    KH_HIP(hipMemsetAsync(v->col(col) + i0, 0, sizeof(double) * count, v->ctx->stream));
    return 0;
}

// ---- operators --------------------------------------------------------------------------------
static int build_rowblocks(const int32_t* indptr, int64_t n_rows, int tile,
                           std::vector<int32_t>& blk) {
    // greedy: consecutive rows while their nnz fit the LDS tile (and <= 4 rows per lane);
    // a row longer than the tile becomes a block of its own
    blk.clear();
    blk.push_back(0);
    const int max_rows = BS * 4;
    int64_t r = 0;
    while (r < n_rows) {
        int64_t r_end = r;
        int64_t acc = 0;
        while (r_end < n_rows && (r_end - r) < max_rows) {
            const int64_t nz = (int64_t)indptr[r_end + 1] - indptr[r_end];
            if (acc + nz > tile) break;
            acc += nz;
            ++r_end;
        }
        if (r_end == r) r_end = r + 1;  // long row
        blk.push_back((int32_t)r_end);
        r = r_end;
    }
    return 0;
}

// Banded structure (k_spmv_dia): square, columns strictly ascending within each row, no stored
// zeros, at most KH_DIA_MAX distinct diagonals, and those at least 70 % full (8 B per slot against
// 12 B per CSR entry).  One pass over the host arrays; gives up at the first violation.
static bool detect_dia(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* indptr,
                       const int32_t* indices, const double* data, std::vector<int>& offs,
                       int64_t nprev = 0, int64_t nnext = 0, int cw = 1) {
    // cw = 2: complex data, (re, im) pairs - a stored zero is an entry whose two parts are zero
    // nprev / nnext: ghost columns of a block-row shard, taken as the rows before / after the slab
    offs.clear();
    if (n_rows + nprev + nnext != n_cols || n_rows < 2 || nnz == 0 || data == nullptr) return false;
    if (nnz > (int64_t)KH_DIA_MAX * n_rows) return false;
    int tab[KH_DIA_MAX];
    int nd = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        int prev = 0;
        for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const int c = dia_virtual_col(indices[p], (int)n_rows, (int)nprev);
            if ((p > indptr[r] && c <= prev) || (data[p * cw] == 0.0 && (cw == 1 || data[p * cw + 1] == 0.0))) return false;
            prev = c;
            const int off = c - (int)r;
            int d = 0;
            while (d < nd && tab[d] != off) ++d;
            if (d == nd) {
                if (nd == KH_DIA_MAX) return false;
                tab[nd++] = off;
            } else if (d > 0) {            // move-to-front keeps the common offsets cheap to find
                std::swap(tab[d], tab[d - 1]);
            }
        }
    }
    if ((double)nnz < 0.7 * (double)nd * (double)n_rows) return false;
    offs.assign(tab, tab + nd);
    std::sort(offs.begin(), offs.end());
    return true;
}

static int build_dia(kh_ctx ctx, kh_mat A, const std::vector<int>& offs, bool halo = false) {
    static int mode = -2;     // KRYPY_AMD_SPMV_DIA: 0 = off, 1/2/4 = row pairs per lane (default 4)
    if (mode == -2) {
        const char* e = getenv("KRYPY_AMD_SPMV_DIA");
        mode = e ? atoi(e) : 4;
        if (mode != 0 && mode != 1 && mode != 2 && mode != 4) mode = 4;
    }
    if (mode == 0) return 0;
    const int rpt = halo ? 4 : mode;                 // the ghost-row variant exists for 4 row pairs per lane
    const int64_t rows_per_wg = 2 * (int64_t)BS * rpt;
    const int64_t nblk = (A->n_rows + rows_per_wg - 1) / rows_per_wg;
    // leading dimension: whole workgroups, and at least the padded length of a device vector so that
    // the chain kernel's fused prologue (chain.h) can read every diagonal without a predicate
    const int64_t ld = ((std::max(nblk * rows_per_wg, padded_ld(ctx, A->n_rows)) + rows_per_wg - 1) / rows_per_wg) *
                       rows_per_wg;
    const size_t bytes = sizeof(double) * (size_t)ld * offs.size();
    double* dia = nullptr;
    if (hipMalloc(&dia, bytes) != hipSuccess) {     // no room for the copy: the CSR kernel serves
        (void)hipGetLastError();
        return 0;
    }
    KH_HIP(hipMemsetAsync(dia, 0, bytes, ctx->stream));
    DiaOffs o;
    o.nd = (int)offs.size();
    for (int d = 0; d < KH_DIA_MAX; ++d) o.off[d] = d < o.nd ? offs[d] : 0;
    hipLaunchKernelGGL(k_dia_fill, dim3((unsigned)((A->n_rows + BS - 1) / BS)), dim3(BS), 0, ctx->stream,
                       A->indptr, A->indices, A->data, A->n_rows, (int)A->nrecv_prev, o, dia, ld);
    KH_HIP(hipGetLastError());
    KH_HIP(hipStreamSynchronize(ctx->stream));
    A->dia = dia;
    A->dia_ld = ld;
    A->dia_nd = o.nd;
    A->dia_nblk = (int)nblk;
    A->dia_rpt = rpt;
    for (int d = 0; d < o.nd; ++d) A->dia_off[d] = offs[d];
    return 0;
}

extern "C++" {
namespace kh {
// kh_mat_set_halo: the ghost columns are known now - look for the banded structure of the shard
int dia_rebuild_for_halo(kh_ctx ctx, kh_mat A) {
    const int64_t ng = A->nrecv_prev + A->nrecv_next;
    // interior / boundary row blocks of both SpMV kernels (apply_one overlaps the halo exchange with the interior)
    A->dia_b0 = A->csr_b0 = 0;
    A->dia_b1 = A->dia_nblk;
    A->csr_b1 = A->nblk;
    if (ng == 0) {                               // square operator: what kh_csr_upload found stands
        if (ctx->force_multi && ctx->nranks == 1) {      // tests / bench --force-sharded: a one-block "boundary" at
            A->dia_b0 = std::min(1, A->dia_nblk);         // both ends exercises the split launches on one GPU
            A->dia_b1 = std::max(A->dia_nblk - 1, A->dia_b0);
            A->csr_b0 = std::min(1, A->nblk);
            A->csr_b1 = std::max(A->nblk - 1, A->csr_b0);
        }
        return 0;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(A->dia);
    A->dia = nullptr;
    A->dia_nd = 0;
    A->dia_nblk = 0;
    if (A->nnz == 0) return 0;
    std::vector<int32_t> indptr(A->n_rows + 1), indices(A->nnz);
    std::vector<double> data(A->nnz);
    KH_HIP(hipMemcpy(indptr.data(), A->indptr, sizeof(int32_t) * indptr.size(), hipMemcpyDeviceToHost));
    KH_HIP(hipMemcpy(indices.data(), A->indices, sizeof(int32_t) * indices.size(), hipMemcpyDeviceToHost));
    KH_HIP(hipMemcpy(data.data(), A->data, sizeof(double) * data.size(), hipMemcpyDeviceToHost));
    // rows [0, bnd_lo) read ghosts of the previous slab, rows [bnd_hi, n) ghosts of the next one
    int64_t bnd_lo = 0, bnd_hi = A->n_rows;
    const int64_t nloc = A->n_rows, gprev = nloc + A->nrecv_prev;
    for (int64_t r = 0; r < nloc; ++r)
        for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const int64_t c = indices[p];
            if (c >= gprev) bnd_hi = std::min(bnd_hi, r);
            else if (c >= nloc) bnd_lo = std::max(bnd_lo, r + 1);
        }
    {
        std::vector<int32_t> blk;
        build_rowblocks(indptr.data(), A->n_rows, A->tile, blk);       // (the table kh_csr_upload built)
        int b0 = 0, b1 = (int)blk.size() - 1;
        while (b0 < b1 && blk[b0] < bnd_lo) ++b0;                       // first block that starts at / behind bnd_lo
        while (b1 > b0 && blk[b1] > bnd_hi) --b1;                       // blocks [b0, b1) end at / before bnd_hi
        A->csr_b0 = b0;
        A->csr_b1 = std::max(b1, b0);
    }
    std::vector<int> offs;
    if (!detect_dia(A->n_rows, A->n_cols, A->nnz, indptr.data(), indices.data(), data.data(), offs,
                    A->nrecv_prev, A->nrecv_next))
        return 0;
    const int64_t need = (A->n_rows + 2 * BS - 1) / (2 * BS);
    if (need > std::max(A->nblk, 1)) {           // partial sums of the fused epilogues: one per workgroup
        (void)hipFree(A->part);
        A->part = nullptr;
        KH_HIP(hipMalloc(&A->part, sizeof(double) * need));
    }
    KH_TRY(build_dia(ctx, A, offs, true));
    if (A->dia != nullptr) {
        const int64_t rows_per_wg = 2 * (int64_t)BS * A->dia_rpt;
        A->dia_b0 = (int)std::min<int64_t>((bnd_lo + rows_per_wg - 1) / rows_per_wg, A->dia_nblk);
        A->dia_b1 = (int)std::max<int64_t>(bnd_hi / rows_per_wg, A->dia_b0);
    }
    return 0;
}
}  // namespace kh
}  // extern "C++"

int kh_mat_set_rows_max(kh_mat A, int64_t rows_max) {
    KH_ARG(A != nullptr && rows_max >= 0, "kh_mat_set_rows_max: NULL / negative");
    A->rows_max = rows_max;
    return 0;
}

int kh_mat_set_ghost(kh_mat A, const double* values, int64_t count) {
    KH_ARG(A && (values || count == 0), "kh_mat_set_ghost: NULL");
    KH_ARG((A->kind == KH_MAT_CSR && count == A->nrecv_prev + A->nrecv_next) ||
               (A->kind == KH_MAT_ZCSR && count == 2 * (A->nrecv_prev + A->nrecv_next)),
           "kh_mat_set_ghost: %lld doubles for %lld ghost columns", (long long)count,
           (long long)(A->nrecv_prev + A->nrecv_next));
    if (count == 0) return 0;
    KH_HIP(hipMemcpyAsync(A->ghost, values, sizeof(double) * count, hipMemcpyHostToDevice, A->ctx->stream));
    KH_HIP(hipStreamSynchronize(A->ctx->stream));
    return 0;
}

int kh_mat_get_ghost(kh_mat A, double* values, int64_t count) {
    KH_ARG(A && (values || count == 0), "kh_mat_get_ghost: NULL");
    KH_ARG((A->kind == KH_MAT_CSR && count == A->nrecv_prev + A->nrecv_next) ||
               (A->kind == KH_MAT_ZCSR && count == 2 * (A->nrecv_prev + A->nrecv_next)),
           "kh_mat_get_ghost: %lld doubles for %lld ghost columns", (long long)count,
           (long long)(A->nrecv_prev + A->nrecv_next));
    if (count == 0) return 0;
    // (the exchange may have run on the communication stream: the compute stream waited for it before the boundary
    // rows were multiplied, so the compute stream is the one to wait for)
    KH_HIP(hipStreamSynchronize(A->ctx->stream));
    KH_HIP(hipMemcpy(values, A->ghost, sizeof(double) * count, hipMemcpyDeviceToHost));
    return 0;
}

int kh_csr_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* indptr,
                  const int32_t* indices, const double* data, kh_mat* out) {
    KH_ARG(ctx && out && indptr, "kh_csr_upload: NULL argument");
    KH_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "kh_csr_upload: negative size");
    KH_ARG(n_rows < 2147483647LL && n_cols < 2147483647LL && nnz < 2147483647LL,
           "kh_csr_upload: int32 CSR limits exceeded");
    KH_ARG(nnz == 0 || (indices != nullptr && data != nullptr), "kh_csr_upload: NULL indices / data with nnz > 0");
    KH_ARG(indptr[0] == 0 && indptr[n_rows] == nnz, "kh_csr_upload: indptr[0]=%d indptr[n]=%d nnz=%lld",
           indptr[0], indptr[n_rows], (long long)nnz);
    for (int64_t r = 0; r < n_rows; ++r)
        KH_ARG(indptr[r] <= indptr[r + 1], "kh_csr_upload: indptr decreases at row %lld", (long long)r);
    for (int64_t i = 0; i < nnz; ++i)
        KH_ARG(indices[i] >= 0 && indices[i] < n_cols, "kh_csr_upload: column index %d out of range at %lld",
               indices[i], (long long)i);
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_CSR;
    A->n_rows = n_rows;
    A->n_cols = n_cols;
    A->nnz = nnz;
    A->tile = ctx->spmv_tile;
    std::vector<int32_t> blk;
    build_rowblocks(indptr, n_rows, A->tile, blk);
    A->nblk = (int)blk.size() - 1;
    // any failure from here on releases what has been allocated so far (kh_mat_free takes a partial handle)
    auto body = [&]() -> int {
        KH_HIP(hipMalloc(&A->indptr, sizeof(int32_t) * (n_rows + 1)));
        KH_HIP(hipMalloc(&A->indices, sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
        KH_HIP(hipMalloc(&A->data, sizeof(double) * std::max<int64_t>(nnz, 1)));
        KH_HIP(hipMalloc(&A->rowblk, sizeof(int32_t) * blk.size()));
        std::vector<int> offs;
        const bool banded = detect_dia(n_rows, n_cols, nnz, indptr, indices, data, offs);
        const int64_t npart = std::max<int64_t>(std::max(A->nblk, 1), banded ? (n_rows + 2 * BS - 1) / (2 * BS) : 1   /* RPT >= 1 */);
        KH_HIP(hipMalloc(&A->part, sizeof(double) * npart));
        KH_HIP(hipMemcpy(A->indptr, indptr, sizeof(int32_t) * (n_rows + 1), hipMemcpyHostToDevice));
        if (nnz > 0) {
            KH_HIP(hipMemcpy(A->indices, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
            KH_HIP(hipMemcpy(A->data, data, sizeof(double) * nnz, hipMemcpyHostToDevice));
        }
        KH_HIP(hipMemcpy(A->rowblk, blk.data(), sizeof(int32_t) * blk.size(), hipMemcpyHostToDevice));
        if (banded) KH_TRY(build_dia(ctx, A, offs));
        // the columns every row block touches (k_spmv_stream<.., WIN>): where nine blocks in ten fit an LDS window of at most
        // SPMV_WIN_CAP entries of x, the operator's launches carry a window as wide as the widest of those
        if (A->nblk > 0 && nnz > 0) {
            constexpr int SPMV_WIN_CAP = 4096;              // 32 KB beside the tile of products (16 ... 32 KB)
            std::vector<int32_t> win(2 * (size_t)A->nblk);
            int64_t fit = 0;
            int widest = 0;
            for (int b = 0; b < A->nblk; ++b) {
                const int32_t z0 = indptr[blk[b]], z1 = indptr[blk[b + 1]];
                int32_t lo = 0, hi = -1;
                if (z1 > z0) {
                    lo = hi = indices[z0];
                    for (int32_t z = z0 + 1; z < z1; ++z) {
                        lo = std::min(lo, indices[z]);
                        hi = std::max(hi, indices[z]);
                    }
                }
                win[2 * (size_t)b] = lo;
                win[2 * (size_t)b + 1] = hi - lo + 1;
                if (z1 - z0 <= A->tile && hi - lo + 1 <= SPMV_WIN_CAP) {
                    fit += 1;
                    widest = std::max(widest, hi - lo + 1);
                }
            }
            if (fit * 10 >= (int64_t)A->nblk * 9 && widest > 0) {
                KH_HIP(hipMalloc(&A->blkwin, sizeof(int32_t) * win.size()));
                KH_HIP(hipMemcpy(A->blkwin, win.data(), sizeof(int32_t) * win.size(), hipMemcpyHostToDevice));
                A->win_cap = widest;
            }
        }
        return 0;
    };
    const int rc = body();
    if (rc != 0) {
        const std::string keep = g_err;
        kh_mat_free(A);
        g_err = keep;
        return rc;
    }
    *out = A;
    return 0;
}

int kh_dense_upload(kh_ctx ctx, int64_t n_rows, int64_t n_cols, const double* a, int64_t lda,
                    kh_mat* out) {
    KH_ARG(ctx && out && a, "kh_dense_upload: NULL argument");
    KH_ARG(n_rows >= 0 && n_cols >= 0 && lda >= n_cols, "kh_dense_upload: bad shape");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_DENSE;
    A->n_rows = n_rows;
    A->n_cols = n_cols;
    A->lda = ((n_cols + 1) / 2) * 2;  // even leading dimension: rows stay 16-byte aligned
    const size_t bytes = sizeof(double) * (size_t)A->lda * (size_t)std::max<int64_t>(n_rows, 1);
    hipError_t e = hipMalloc(&A->a, bytes);
    if (e != hipSuccess) {
        delete A;
        return fail(KH_ERR_NOMEM, "kh_dense_upload: hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    KH_HIP(hipMemcpy2D(A->a, A->lda * sizeof(double), a, lda * sizeof(double),
                       n_cols * sizeof(double), n_rows, hipMemcpyHostToDevice));
    *out = A;
    return 0;
}

// A dense operator from a block that is already on the device: row i of the operator = column col0 + i of X, scaled, plus
// beta on the diagonal - A = alpha X[:, col0 : col0 + nrows]^T + beta I.  What a symmetric product formed on the device
// (Y = G G^T through kh_apply's panel path: the columns of Y ARE the rows of the row-major operator) needs to become the
// operator of a solve without a trip over the host ("build on device").
static __global__ void k_dense_from_block(int64_t n_rows, int64_t n_cols, const double* __restrict__ x, int64_t ldx,
                                          double alpha, double beta, double* __restrict__ a, int64_t lda) {
    const int64_t i = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += (int64_t)gridDim.x * blockDim.x) {
        const double v = alpha * x[i * ldx + j];
        a[i * lda + j] = (i == j) ? v + beta : v;
    }
}

int kh_dense_from_block(kh_ctx ctx, kh_vec X, int64_t col0, int64_t n_rows, double alpha, double beta, kh_mat* out) {
    KH_ARG(ctx && X && out, "kh_dense_from_block: NULL argument");
    KH_TRY(check_vec(X, col0, n_rows, "kh_dense_from_block(X)"));
    KH_ARG(n_rows >= 1 && n_rows <= 65535, "kh_dense_from_block: 1 ... 65535 rows");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_DENSE;
    A->n_rows = n_rows;
    A->n_cols = X->n;
    A->lda = ((X->n + 1) / 2) * 2;
    const size_t bytes = sizeof(double) * (size_t)A->lda * (size_t)n_rows;
    hipError_t e = hipMalloc(&A->a, bytes);
    if (e != hipSuccess) {
        delete A;
        return fail(KH_ERR_NOMEM, "kh_dense_from_block: hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    if (A->lda > A->n_cols) KH_HIP(hipMemsetAsync(A->a, 0, bytes, ctx->stream));
    const unsigned gx = (unsigned)std::min<int64_t>((X->n + BS - 1) / BS, 64);
    hipLaunchKernelGGL(k_dense_from_block, dim3(gx, (unsigned)n_rows), dim3(BS), 0, ctx->stream, n_rows, X->n, X->col(col0), X->ld,
                       alpha, beta, A->a, A->lda);
    hipError_t e2 = hipGetLastError();
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess) {
        (void)hipFree(A->a);
        delete A;
        return fail(KH_ERR_HIP, "kh_dense_from_block: %s", hipGetErrorString(e2));
    }
    *out = A;
    return 0;
}

int kh_diag_upload(kh_ctx ctx, int64_t n, const double* d, kh_mat* out) {
    KH_ARG(ctx && out && (d || n == 0), "kh_diag_upload: NULL argument");
    KH_HIP(hipSetDevice(ctx->device));
    kh_mat A = new kh_mat_s();
    A->ctx = ctx;
    A->kind = KH_MAT_DIAG;
    A->n_rows = A->n_cols = n;
    const size_t dbytes = sizeof(double) * (((n + 31) / 32) * 32 + CH_SLACK);
    hipError_t e = hipMalloc(&A->diag, dbytes);
    if (e == hipSuccess) e = hipMemset(A->diag, 0, dbytes);
    if (e == hipSuccess && n > 0) e = hipMemcpy(A->diag, d, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(A->diag);
        delete A;
        return fail(e == hipErrorOutOfMemory ? KH_ERR_NOMEM : KH_ERR_HIP, "kh_diag_upload(%lld): %s", (long long)n,
                    hipGetErrorString(e));
    }
    *out = A;
    return 0;
}

int kh_mat_free(kh_mat A) {
    if (!A) return 0;
    (void)hipStreamSynchronize(A->ctx->stream);
    forget_steps(A->ctx, A);
    (void)hipFree(A->indptr);
    (void)hipFree(A->indices);
    (void)hipFree(A->data);
    (void)hipFree(A->rowblk);
    (void)hipFree(A->blkwin);
    (void)hipFree(A->part);
    (void)hipFree(A->dia);
    (void)hipFree(A->zdia);
    (void)hipFree(A->a);
    (void)hipFree(A->diag);
    (void)hipFree(A->ghost);
    (void)hipFree(A->ghost_panel);
    xh_free(A);
    delete A;
    return 0;
}

int kh_mat_diagonals(kh_mat A) { return (A != nullptr && A->dia != nullptr) ? A->dia_nd : 0; }

int kh_apply(kh_ctx ctx, kh_mat A, kh_vec X, int64_t xcol, kh_vec Y, int64_t ycol, int64_t ncols) {
    KH_ARG(ctx && A, "kh_apply: NULL handle");
    if (Y != nullptr) chain_blk_touch(ctx, Y);
    KH_TRY(check_vec(X, xcol, ncols, "kh_apply(X)"));
    KH_TRY(check_vec(Y, ycol, ncols, "kh_apply(Y)"));
    if (A->kind >= KH_MAT_ZCSR) return zapply_cols(ctx, A, X, xcol, Y, ycol, ncols);
    const int64_t xneed = (A->kind == KH_MAT_CSR) ? A->n_cols - A->nrecv_prev - A->nrecv_next : A->n_cols;
    KH_ARG(X->n == xneed && Y->n == A->n_rows, "kh_apply: dimension mismatch (A %lldx%lld, x %lld, y %lld)",
           (long long)A->n_rows, (long long)A->n_cols, (long long)X->n, (long long)Y->n);
    KH_ARG(!(X == Y && xcol < ycol + ncols && ycol < xcol + ncols),
           "kh_apply: input and output column ranges overlap (in-place application is not supported)");
    if (A->kind == KH_MAT_DENSE && ncols >= 2 && A->n_rows >= 1024 && X != Y) {
        // a panel: stream A once per 16 columns on the FP64 matrix cores (k_gemm_dense_mfma)
        constexpr int RT = 2;
        const int grid = (int)((A->n_rows + (BS / 64) * 16 * RT - 1) / ((BS / 64) * 16 * RT));
        for (int64_t c = 0; c < ncols; c += 16) {
            const int nc = (int)std::min<int64_t>(16, ncols - c);
            hipLaunchKernelGGL((k_gemm_dense_mfma<RT>), dim3(grid), dim3(BS), 0, ctx->stream, A->n_rows, A->n_cols,
                               A->a, A->lda, X->col(xcol + c), X->ld, nc, Y->col(ycol + c), Y->ld);
        }
        KH_HIP(hipGetLastError());
        return 0;
    }
    const bool shard = (A->kind == KH_MAT_CSR) && (A->nrecv_prev + A->nrecv_next + A->nsend_prev + A->nsend_next) > 0;
    if (A->kind == KH_MAT_CSR && ncols >= 2 && A->nblk > 0) {
        // a panel: the matrix is streamed once for all columns.  A block-row shard first exchanges the halo of ALL
        // columns in one group (one RCCL kernel instead of ncols) into a ghost panel, then takes the CSR kernel with
        // ghost columns; on one rank without neighbours the ghost entries are the ones kh_mat_set_ghost wrote,
        // repeated per column by the caller's choice of set_ghost - so only a communicator's shard goes this way
        const int64_t ng = A->nrecv_prev + A->nrecv_next;
        const bool exchange = shard && kh_multi(ctx) && ctx->comm != nullptr && (ctx->nranks > 1 || ctx->halo_loopback);
        if (shard && !exchange) goto column_loop;
        if (shard) {
            if (A->ghost_panel_cols < ncols) {
                KH_HIP(hipStreamSynchronize(ctx->stream));
                (void)hipFree(A->ghost_panel);
                A->ghost_panel = nullptr;
                A->ghost_panel_cols = 0;
                KH_HIP(hipMalloc(&A->ghost_panel, sizeof(double) * std::max<int64_t>(ng, 1) * ncols));
                A->ghost_panel_cols = ncols;
            }
            KH_TRY(comm_halo_exchange_panel(ctx, A, X->col(xcol), X->ld, ncols, ctx->stream));
        }
        if (!shard && use_dia(ctx, A, Y->col(ycol)) && (Y->ld & 1) == 0) {
            constexpr int DC = 8;
            DiaOffs o;
            o.nd = A->dia_nd;
            for (int d = 0; d < KH_DIA_MAX; ++d) o.off[d] = d < A->dia_nd ? A->dia_off[d] : 0;
            const unsigned grid = (unsigned)((A->n_rows + 2 * BS - 1) / (2 * BS));
            for (int64_t c = 0; c < ncols; c += DC) {
                const int nc = (int)std::min<int64_t>(DC, ncols - c);
                hipLaunchKernelGGL((k_spmm_dia<DC>), dim3(grid), dim3(BS), 0, ctx->stream, o, A->dia, A->dia_ld,
                                   A->n_rows, X->col(xcol + c), X->ld, Y->col(ycol + c), Y->ld, nc);
            }
            KH_HIP(hipGetLastError());
            ctx->n_spmm += 1;
            return 0;
        }
        constexpr int DC = 4;
        const size_t lds = (size_t)A->tile * sizeof(double) * DC;
        static bool attr_done = false;
        if (!attr_done) {
            // 32-64 KB of dynamic LDS: above the 48 KB a kernel gets without asking
            KH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_stream<4, DC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BS * 8 * DC));
            KH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_stream<8, DC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 8 * BS * 8 * DC));
            KH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_stream<16, DC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 16 * BS * 8 * DC));
            attr_done = true;
        }
#define KH_SPMM(I)                                                                                             \
    hipLaunchKernelGGL((k_spmm_stream<I, DC>), dim3(A->nblk), dim3(BS), lds, ctx->stream, A->indptr, A->indices, \
                       A->data, A->rowblk, A->nblk, A->tile, X->col(xcol), X->ld, Y->col(ycol), Y->ld, (int)ncols, \
                       shard ? (int64_t)A->n_rows : ((int64_t)1 << 62), A->ghost_panel, ng)
        switch (A->tile / BS) {
            case 4: KH_SPMM(4); break;
            case 16: KH_SPMM(16); break;
            default: KH_SPMM(8); break;
        }
#undef KH_SPMM
        KH_HIP(hipGetLastError());
        ctx->n_spmm += 1;
        return 0;
    }
column_loop:
    for (int64_t c = 0; c < ncols; ++c)
        KH_TRY(apply_one(ctx, A, X->col(xcol + c), Y->col(ycol + c), EPI_NONE, nullptr, nullptr, 0));
    return 0;
}

// ---- inner products, norms, updates -----------------------------------------------------------
static int dot_panel_dev(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* w,
                         double* out_dev, int rmode) {
    // out_dev[j] (rmode) = <V[:, j0+j], w>, all-reduced over ranks
    int64_t done = 0;
    while (done < ncols) {
        const int nc = (int)std::min<int64_t>(MAXC, ncols - done);
        KH_TRY(multidot_chunk(ctx, V, j0 + done, nc, w));
        hipLaunchKernelGGL(k_reduce_partials, dim3(nc), dim3(BS), 0, ctx->stream, ctx->part,
                           grid_for(ctx, V->n), NB_MAX, out_dev + done, rmode);
        KH_HIP(hipGetLastError());
        done += nc;
    }
    return 0;
}

// ---- deflation projector on the device ---------------------------------------------------------
// z = complement projection of z (in place); ya_dev (d doubles on the device) gets <Y, z_in>
static int proj_apply_dev(kh_ctx ctx, kh_proj p, double* z, double* ya_dev, int64_t zld = 0) {
    const int d = (int)p->d;
    // long vectors on one GPU: both sweeps in ONE launch with z in registers (proj_reg.h)
    if (zld > 0) {
        int r2 = 0, G = 0;
        if (chain_geometry(ctx, p->W->n, &r2, &G)) {
            const int rc = proj_reg_apply(ctx, p, z, zld, r2, G, ya_dev);
            if (rc != 0) return rc < 0 ? rc : 0;
        }
    }
    for (int it = 0; it < p->iterations; ++it) {
        // long vectors (N ranks, or one GPU without the one-launch form): the passes over W and V through the
        // register-resident panel kernels - the same launches and bytes as the chunked kernels, at their streaming rate
        int nwave = 0;
        const bool panel = zld > 0 && ctx->proj_panel && p->W->ld == p->V->ld;
        const int rc_d = panel ? cgs_panel_pass(ctx, p->W, d, z, zld, nullptr, false, &nwave) : 0;
        if (rc_d < 0) return rc_d;
        if (rc_d == 1) {
            KH_TRY(reduce_partials_allreduce(ctx, ctx->cgs_part, nwave, CGS_PSTRIDE, p->c0, d, kh_multi(ctx)));
        } else {
            KH_TRY(dot_panel_dev(ctx, p->W, 0, d, z, p->c0, 0));
            if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, p->c0, d));
        }
        if (it == 0 && ya_dev != nullptr)
            hipLaunchKernelGGL(k_small_matvec, dim3(1), dim3(BS), 0, ctx->stream, d, p->WRH, p->c0, ya_dev);
        hipLaunchKernelGGL(k_small_matvec, dim3(1), dim3(BS), 0, ctx->stream, d, p->T, p->c0, p->c1);
        KH_HIP(hipGetLastError());
        const int rc_u = panel ? cgs_panel_pass(ctx, p->V, d, z, zld, p->c1, true, &nwave) : 0;
        if (rc_u < 0) return rc_u;
        if (rc_u == 0) KH_TRY(multiaxpy_cols(ctx, p->V, 0, d, p->c1, 1.0, 1.0, z, T_NONE, nullptr, nullptr));
        if (rc_u == 1) ctx->n_proj_panel += 1;
    }
    return 0;
}


int kh_dot_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, kh_vec W, int64_t wcol,
                 double* out) {
    KH_ARG(ctx && out, "kh_dot_panel: NULL");
    KH_TRY(check_vec(V, j0, ncols, "kh_dot_panel(V)"));
    KH_TRY(check_vec(W, wcol, 1, "kh_dot_panel(W)"));
    KH_ARG(V->n == W->n, "kh_dot_panel: length mismatch");
    KH_ARG(ncols <= 1024, "kh_dot_panel: at most 1024 columns per call");
    if (ncols == 0) return 0;
    double* dev = ctx->scal + SC_COEF;
    KH_TRY(dot_panel_dev(ctx, V, j0, ncols, W->col(wcol), dev, 0));
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, dev, ncols));
    return fetch_scalars(ctx, dev, ncols, out);
}

int kh_gemm_tn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t nx, kh_vec Y, int64_t y0, int64_t ny,
               double* out) {
    KH_ARG(ctx && out, "kh_gemm_tn: NULL");
    KH_TRY(check_vec(X, x0, nx, "kh_gemm_tn(X)"));
    KH_TRY(check_vec(Y, y0, ny, "kh_gemm_tn(Y)"));
    KH_ARG(X->n == Y->n, "kh_gemm_tn: length mismatch");
    KH_ARG(nx <= 1024, "kh_gemm_tn: at most 1024 rows");
    if (ctx->gram_mfma && ny >= 2 && nx >= 1 && (X->ld & 1) == 0 && (Y->ld & 1) == 0) {
        // both blocks read once per 16 x 16 tile of the product (k_gram_mfma); on N ranks one all-reduce per tile - every rank
        // decides alike (shapes and the switch only)
        if (ctx->gram_part == nullptr) KH_HIP(hipMalloc(&ctx->gram_part, sizeof(double) * 256 * (size_t)KH_GRAM_NB));
        const int G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(KH_GRAM_NB, ctx->ncu * 4), (X->n + 255) / 256));
        double* dev = ctx->scal + SC_COEF;
        double tile[256];
        for (int64_t j0 = 0; j0 < ny; j0 += 16) {
            const int tj = (int)std::min<int64_t>(16, ny - j0);
            for (int64_t i0 = 0; i0 < nx; i0 += 16) {
                const int ti = (int)std::min<int64_t>(16, nx - i0);
                hipLaunchKernelGGL(k_gram_mfma, dim3(G), dim3(BS), 0, ctx->stream, X->n, X->col(x0 + i0), X->ld, ti,
                                   Y->col(y0 + j0), Y->ld, tj, ctx->gram_part, KH_GRAM_NB);
                KH_HIP(hipGetLastError());
                hipLaunchKernelGGL(k_reduce_partials, dim3(256), dim3(BS), 0, ctx->stream, ctx->gram_part, G, KH_GRAM_NB, dev, 0);
                KH_HIP(hipGetLastError());
                if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, dev, 256));
                KH_TRY(fetch_scalars(ctx, dev, 256, tile));
                for (int i = 0; i < ti; ++i)
                    for (int j = 0; j < tj; ++j) out[(i0 + i) * ny + (j0 + j)] = tile[i * 16 + j];
                ctx->n_gram_mfma += 1;
            }
        }
        return 0;
    }
    std::vector<double> colbuf((size_t)std::max<int64_t>(nx, 1));
    for (int64_t j = 0; j < ny; ++j) {
        KH_TRY(kh_dot_panel(ctx, X, x0, nx, Y, y0 + j, colbuf.data()));
        for (int64_t i = 0; i < nx; ++i) out[i * ny + j] = colbuf[i];
    }
    return 0;
}

int kh_axpy_panel(kh_ctx ctx, kh_vec V, int64_t j0, int64_t ncols, const double* h, kh_vec W,
                  int64_t wcol) {
    KH_ARG(ctx && (h || ncols == 0), "kh_axpy_panel: NULL");
    if (W != nullptr) chain_blk_touch(ctx, W);
    KH_TRY(check_vec(V, j0, ncols, "kh_axpy_panel(V)"));
    KH_TRY(check_vec(W, wcol, 1, "kh_axpy_panel(W)"));
    KH_ARG(V->n == W->n, "kh_axpy_panel: length mismatch");
    KH_ARG(ncols <= 1024, "kh_axpy_panel: at most 1024 columns per call");
    if (ncols == 0) return 0;
    double* dev = ctx->scal + SC_COEF;
    KH_TRY(push_scalars(ctx, h, ncols, dev));
    return multiaxpy_cols(ctx, V, j0, ncols, dev, 1.0, 1.0, W->col(wcol), T_NONE, nullptr, nullptr);
}

int kh_gemm_nn(kh_ctx ctx, kh_vec X, int64_t x0, int64_t k, const double* C, int64_t nc,
               double alpha, double beta, kh_vec Y, int64_t y0) {
    KH_ARG(ctx && (C || k == 0 || nc == 0), "kh_gemm_nn: NULL");
    RoctxScope range_(ctx, "kh_gemm_nn k=%lld nc=%lld", (long long)k, (long long)nc);
    KH_TRY(check_vec(X, x0, k, "kh_gemm_nn(X)"));
    KH_TRY(check_vec(Y, y0, nc, "kh_gemm_nn(Y)"));
    KH_ARG(X->n == Y->n, "kh_gemm_nn: length mismatch");
    KH_ARG(X != Y, "kh_gemm_nn: X and Y must be different blocks");
    chain_blk_touch(ctx, Y);
    if (ctx->gram_mfma && nc >= 2 && nc <= 16 && k >= 1 && (X->ld & 1) == 0 && (Y->ld & 1) == 0) {
        // the block read once for all nc output columns (k_panel_gemm_mfma), in passes of 64 columns of X (the coefficients
        // of a pass: 64 x nc <= 1024 staged doubles, 8 KB of LDS)
        constexpr int64_t KP = 64;
        double* devc = ctx->scal + SC_COEF;
        std::vector<double> cpass((size_t)(KP * nc));
        const int G = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->ncu * 4, (X->n + 255) / 256));
        for (int64_t i0 = 0; i0 < k; i0 += KP) {
            const int kk = (int)std::min(KP, k - i0);
            for (int i = 0; i < kk; ++i)
                for (int64_t c = 0; c < nc; ++c) cpass[(size_t)(i * nc + c)] = alpha * C[(i0 + i) * nc + c];
            KH_TRY(push_scalars(ctx, cpass.data(), (int64_t)kk * nc, devc));
            const size_t lds = sizeof(double) * 16 * (size_t)((kk + 3) & ~3);
            const double b = i0 == 0 ? beta : 1.0;
            if (b == 0.0)
                hipLaunchKernelGGL((k_panel_gemm_mfma<2, 0>), dim3(G), dim3(BS), lds, ctx->stream, X->n, X->col(x0 + i0), X->ld, kk,
                                   devc, (int)nc, 0.0, Y->col(y0), Y->ld);
            else if (b == 1.0)
                hipLaunchKernelGGL((k_panel_gemm_mfma<2, 1>), dim3(G), dim3(BS), lds, ctx->stream, X->n, X->col(x0 + i0), X->ld, kk,
                                   devc, (int)nc, 1.0, Y->col(y0), Y->ld);
            else
                hipLaunchKernelGGL((k_panel_gemm_mfma<2, 2>), dim3(G), dim3(BS), lds, ctx->stream, X->n, X->col(x0 + i0), X->ld, kk,
                                   devc, (int)nc, b, Y->col(y0), Y->ld);
            KH_HIP(hipGetLastError());
            ctx->n_panel_gemm += 1;
        }
        return 0;
    }
    constexpr int64_t KMAX = 1024;      // coefficients staged per pass (SC_COEF region of the device scalars)
    std::vector<double> coef((size_t)std::max<int64_t>(std::min(k, KMAX), 1));
    double* dev = ctx->scal + SC_COEF;
    for (int64_t c = 0; c < nc; ++c) {
        if (k == 0) {
            if (beta == 0.0) KH_TRY(kh_vec_zero(Y, y0 + c, 1));
            else if (beta != 1.0) KH_TRY(kh_waxpby(ctx, Y, y0 + c, beta, Y, y0 + c, 0.0, Y, y0 + c));
            continue;
        }
        // a basis longer than KMAX columns (GMRES with maxiter > 1024): passes of KMAX columns, left to right
        for (int64_t i0 = 0; i0 < k; i0 += KMAX) {
            const int64_t kk = std::min(KMAX, k - i0);
            for (int64_t i = 0; i < kk; ++i) coef[i] = alpha * C[(i0 + i) * nc + c];
            KH_TRY(push_scalars(ctx, coef.data(), kk, dev));
            // y = beta*y - (-1) * sum coef_i x_i  : exact sign flip, additions left to right
            KH_TRY(multiaxpy_cols(ctx, X, x0 + i0, kk, dev, -1.0, i0 == 0 ? beta : 1.0, Y->col(y0 + c), T_NONE,
                                  nullptr, nullptr));
        }
    }
    return 0;
}

int kh_nrm2(kh_ctx ctx, kh_vec W, int64_t wcol, double* out) {
    KH_ARG(ctx && out, "kh_nrm2: NULL");
    KH_TRY(check_vec(W, wcol, 1, "kh_nrm2"));
    const int64_t n = W->n;
    double* part = part_slot(ctx, SLOT_NRM);
    hipLaunchKernelGGL((k_gs_link<A_NONE, T_NRM>), dim3(grid_for(ctx, n)), dim3(BS), 0, ctx->stream,
                       n, nullptr, nullptr, W->col(wcol), nullptr, nullptr, nullptr, 0, nullptr, 0.0,
                       part, nullptr);
    KH_HIP(hipGetLastError());
    double* dev = ctx->scal + SC_TMP;
    const int mode = kh_multi(ctx) ? 0 : 2;
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, part, grid_for(ctx, n),
                       NB_MAX, dev, mode);
    KH_HIP(hipGetLastError());
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, dev, 1));
    KH_TRY(fetch_scalars(ctx, dev, 1, out));
    if (kh_multi(ctx)) *out = sqrt(fabs(*out));
    return 0;
}

int kh_waxpby(kh_ctx ctx, kh_vec Z, int64_t zcol, double alpha, kh_vec X, int64_t xcol, double beta,
              kh_vec Y, int64_t ycol) {
    KH_ARG(ctx, "kh_waxpby: NULL ctx");
    if (Z != nullptr) chain_blk_touch(ctx, Z);
    KH_TRY(check_vec(Z, zcol, 1, "kh_waxpby(Z)"));
    KH_TRY(check_vec(X, xcol, 1, "kh_waxpby(X)"));
    KH_TRY(check_vec(Y, ycol, 1, "kh_waxpby(Y)"));
    KH_ARG(Z->n == X->n && Z->n == Y->n, "kh_waxpby: length mismatch");
    const int64_t n = Z->n;
    hipLaunchKernelGGL(k_waxpby, dim3(grid_lin(ctx, n)), dim3(BS), 0, ctx->stream, n, Z->col(zcol),
                       alpha, X->col(xcol), beta, Y->col(ycol));
    KH_HIP(hipGetLastError());
    return 0;
}

int kh_vdiv(kh_ctx ctx, kh_vec Z, int64_t zcol, kh_vec X, int64_t xcol, double s) {
    KH_ARG(ctx, "kh_vdiv: NULL ctx");
    if (Z != nullptr) chain_blk_touch(ctx, Z);
    KH_TRY(check_vec(Z, zcol, 1, "kh_vdiv(Z)"));
    KH_TRY(check_vec(X, xcol, 1, "kh_vdiv(X)"));
    KH_ARG(Z->n == X->n, "kh_vdiv: length mismatch");
    hipLaunchKernelGGL(k_vdiv, dim3(grid_lin(ctx, Z->n)), dim3(BS), 0, ctx->stream, Z->n,
                       Z->col(zcol), X->col(xcol), s);
    KH_HIP(hipGetLastError());
    return 0;
}

// Called when an Arnoldi step is begun: bring the chain family back after a recovered timeout (kh_internal.h).
static inline void chain_rearm(kh_ctx ctx, int64_t k) {
    if (ctx->chain_enabled || !ctx->chain_configured || ctx->chain_in_recovery || ctx->chain_recoveries == 0 ||
        ctx->chain_recoveries >= KH_CHAIN_MAX_RECOVERIES)
        return;
    for (int s = 0; s < KH_NSLOT; ++s)      // steps that still wait to be re-run read the per-column path's state
        if (*ctx->chain_err_pin[s] != 0) return;
    if (k == 0 || ++ctx->chain_clean_steps >= KH_CHAIN_REARM_STEPS) {
        ctx->chain_enabled = 1;
        ctx->chain_clean_steps = 0;
        ctx->n_chain_rearmed += 1;
    }
}

// what a timeout does to the context (kh_arnoldi_step_end, kh_zarnoldi_step)
static inline void chain_switch_off(kh_ctx ctx) {
    ctx->chain_enabled = 0;
    ctx->blk_next = -1;              // (the faulted launch may have left a garbage row in the Gram table)
    ctx->chain_clean_steps = 0;
    ctx->chain_recoveries += 1;
    if (ctx->chain_recoveries == KH_CHAIN_MAX_RECOVERIES)
        fprintf(stderr, "krylov_hip: the grid-wide sum of the register-resident Gram-Schmidt kernels timed out %d times in "
                        "this context (a GPU shared with other work?); they stay off, the per-column kernels take over\n",
                KH_CHAIN_MAX_RECOVERIES);
}

// A step begun while its predecessor (same basis, step k-1) is marked "timed out, to be re-run" would
// read a garbage column: it is not launched at all, only marked the same way, and kh_arnoldi_step_end runs
// it when its turn comes (the predecessor has been recovered by then: the host fetches the steps in order).
static bool step_poisoned(kh_ctx ctx, int slot) {
    const kh_step_s& st = ctx->step[slot];
    for (int s2 = 0; s2 < KH_NSLOT; ++s2) {
        const kh_step_s& o = ctx->step[s2];
        if (s2 != slot && o.kind != 0 && o.V == st.V && o.k == st.k - 1 && *ctx->chain_err_pin[s2] != 0) {
            *ctx->chain_err_pin[slot] = *ctx->chain_err_pin[s2];      // (3: "to be re-run" behind a projector timeout - not the chain's fault)
            return true;
        }
    }
    return false;
}

// ---- fused hot path -----------------------------------------------------------------------------
#define KH_LINK(ASRC, TAIL, P, VN, DG, MW, PIN, SIN, AARG, POUT, HS)                                  \
    hipLaunchKernelGGL((k_gs_link<ASRC, TAIL>), dim3(grid), dim3(BS), 0, ctx->stream, n, P, VN, w, DG, \
                       MW, PIN, grid, SIN, AARG, POUT, HS)

int kh_arnoldi_step_begin(kh_ctx ctx, kh_mat A, kh_proj proj, kh_mat Md, kh_vec V, kh_vec P, kh_vec W,
                          int64_t wcol, int64_t k, int64_t start, int sweeps, int gs_mode, double h_km1,
                          int slot) {
    KH_ARG(ctx && V && W, "kh_arnoldi_step: NULL argument");
    KH_ARG(slot >= 0 && slot < KH_NSLOT, "kh_arnoldi_step: slot %d not in [0,%d)", slot, KH_NSLOT);
    RoctxScope range_(ctx, "kh_arnoldi_step_begin k=%lld start=%lld", (long long)k, (long long)start);
    KH_ARG(k >= 0 && k + 1 < V->ncols, "kh_arnoldi_step: k=%lld needs %lld basis columns, have %lld",
           (long long)k, (long long)(k + 2), (long long)V->ncols);
    KH_ARG(start >= 0 && start <= k, "kh_arnoldi_step: start=%lld not in [0,k]", (long long)start);
    KH_ARG(sweeps >= 1 && sweeps <= 4, "kh_arnoldi_step: sweeps=%d", sweeps);
    // sized for the whole basis up front: never reallocated while steps of this basis are in flight
    const int64_t pd = proj ? proj->d : 0;
    KH_ARG(proj == nullptr || (proj->W->n == V->n && A != nullptr), "kh_arnoldi_step: projector needs A and length N");
    KH_TRY(ensure_hcap(ctx, std::max<int64_t>(k + 2, V->ncols + 1) + pd));
    // Md: the Jacobi preconditioner (diagonal; V = Md P), or - same recurrence with the roles of the blocks swapped,
    // utils.py:184-193 - the SPD matrix B of a non-Euclidean inner product (diagonal, CSR or dense; V = B P holds
    // B times the basis, P the basis itself): dots against V, updates with P, norm sqrt(<w, Md w>)
    KH_ARG(Md == nullptr || Md->kind == KH_MAT_DIAG || Md->kind == KH_MAT_CSR || Md->kind == KH_MAT_DENSE,
           "kh_arnoldi_step: Md must be a real diagonal, CSR or dense operator");
    KH_ARG(Md == nullptr || Md->kind == KH_MAT_DIAG || (Md->n_rows == V->n && Md->n_cols == V->n && proj == nullptr),
           "kh_arnoldi_step: a matrix Md must be N x N (and takes no projector)");
    KH_ARG((Md == nullptr) == (P == nullptr), "kh_arnoldi_step: P and Md go together");
    KH_ARG(P == nullptr || (P->n == V->n && P->ncols >= V->ncols), "kh_arnoldi_step: P shape");
    KH_TRY(check_vec(W, wcol, Md ? 2 : 1, "kh_arnoldi_step(W)"));
    KH_ARG(W->n == V->n, "kh_arnoldi_step: W length");
    const int64_t n = V->n;
    ctx->wait_tag[slot] = false;
    chain_rearm(ctx, k);
    {
        kh_step_s& st = ctx->step[slot];
        st.kind = 1;
        st.A = A; st.proj = proj; st.Md = Md; st.V = V; st.P = P; st.W = W;
        st.wcol = wcol; st.k = k; st.start = start; st.sweeps = sweeps; st.gs_mode = gs_mode;
        st.h_km1[0] = h_km1;
        if (step_poisoned(ctx, slot)) {     // its predecessor waits to be re-run: so will this one (kh_arnoldi_step_end)
            KH_HIP(hipEventRecord(ctx->hev[slot], ctx->stream));
            return 0;
        }
    }
    kh_vec B = P ? P : V;
    double* w = W->col(wcol);
    double* mw = Md ? W->col(wcol + 1) : nullptr;
    const bool md_mat = Md != nullptr && Md->kind != KH_MAT_DIAG;      // Md w needs an operator application
    const double* dg = (Md && !md_mat) ? Md->diag : nullptr;
    const int grid = grid_for(ctx, n);
    const bool multi = kh_multi(ctx);
    double* hdev = ctx->hslot_dev[slot];
    double* tmp = ctx->scal + SC_TMP;

    const bool presub = (start > 0 && start == k);  // Lanczos three-term recurrence
    // look-ahead Lanczos: h_km1 = NaN means "H[k,k-1] of the step begun just before this one",
    // still on the device in the previous H-column slot
    const double* hk_dev = nullptr;
    if (presub && h_km1 != h_km1) hk_dev = ctx->hslot_dev[(slot + KH_NSLOT - 1) % KH_NSLOT] + k;
    // reference-order MGS: keep w in registers for the whole chain when it fits (chain.h)
    int cr2 = 0, cg = 0;
    const bool want_chain = (gs_mode == KH_GS_MGS && ctx->chain_enabled && !kh_multi(ctx) && !md_mat &&
                             chain_geometry(ctx, n, &cr2, &cg));
    // N ranks, reference order: all coefficients of the step from one pass and ONE all-reduce (try_lowsync_mgs below)
    const bool want_lowsync = (kh_multi(ctx) && gs_mode == KH_GS_MGS && sweeps == 1 && start == 0 && !presub && Md == nullptr &&
                               lowsync_eligible(ctx, V, W->ld, k, A));
    // ... or, with the xr transport on and slabs of up to 6 rows per lane: the blocked kernel with the cross-rank sums INSIDE the
    // launch (chain_blk2.h) - the local basis is read once, no all-reduce call in the step.  Eligibility is decided for the
    // longest slab of the run (like the one-reduction form's), so every rank decides alike; from there on a refusal is an error.
    bool want_blk2 = false;
    if (kh_multi(ctx) && ctx->xr_on && ctx->chain_blk2 && ctx->chain_configured && gs_mode == KH_GS_MGS && sweeps == 1 && start == 0 &&
        !presub && Md == nullptr) {
        const int64_t nmax = longest_slab(ctx, A, n);
        int r2m = 0, gm = 0;
        want_blk2 = nmax >= n && chain_blk2_shape(ctx, nmax, &r2m, &gm, nullptr, ctx->blk2_one >= 1) && k + 3 <= KH_BLK_TABCOLS;
    }
    // ... and beyond the blocked kernel's 2.5 M rows per rank: the register-resident chain kernels (16 ... 56 rows per lane, up
    // to 14.68 M rows) with the cross-rank stage inside every grid-wide sum (chain_xr.hip) - one sum across the ranks per link,
    // the local basis read ONCE where the one-reduction and panel forms read it twice.  Decided for the longest slab, like the
    // other two; it takes the step in front of the one-reduction form (KRYPY_AMD_CHAIN_XR=0: that form / the panel kernels).
    bool want_chain_xr = false;
    if (!want_blk2 && kh_multi(ctx) && ctx->xr_on && ctx->chain_xr && ctx->chain_configured && gs_mode == KH_GS_MGS && sweeps == 1 &&
        start == 0 && !presub && Md == nullptr) {
        const int64_t nmax = longest_slab(ctx, A, n);
        int r2m = 0, gm = 0;
        // (a slab that fills less than half the compute units at 16 rows per lane - under 2.1 M rows - belongs to the blocked
        // kernel; with that switched off it keeps the one-reduction form rather than a chain on a handful of workgroups)
        want_chain_xr = nmax >= n && chain_xr_shape(ctx, nmax, &r2m, &gm) && (ctx->chain_xr_cus > 0 || 2 * gm >= ctx->ncu) &&
                        !((n & 1) && (V->ld <= n || W->ld <= n));
    }
    const bool fuse_dot0 = (A != nullptr && A->kind == KH_MAT_CSR && !presub && gs_mode == KH_GS_MGS &&
                            A->nblk > 0 && !want_chain && proj == nullptr && !want_lowsync && !want_blk2 && !want_chain_xr);
    // the H column accumulates over sweeps, so it starts from zero - except under the chain kernel, whose
    // first sweep assigns (one memset launch and its queue bubble less per step)
    // (nor under the single-sweep register-resident panel kernels, which write the entries directly)
    const bool want_cgs1 = (gs_mode == KH_GS_CGS && sweeps == 1 && ctx->chain_enabled);
    // (... nor under the blocked kernel with the cross-rank sums inside, which assigns every entry as well)
    // (... nor under the chain kernels with the cross-rank stage: first sweep assigns, like on one GPU)
    if (!want_chain && !want_cgs1 && !(want_blk2 && pd == 0) && !want_chain_xr)
        KH_HIP(hipMemsetAsync(hdev, 0, sizeof(double) * (k + 2 + pd), ctx->stream));
    // 1. operator
    bool fused_chain = false;
    if (A != nullptr) {
        KH_ARG(A->n_rows == n, "kh_arnoldi_step: operator rows %lld != %lld", (long long)A->n_rows,
               (long long)n);
        if (want_chain && proj == nullptr && A->kind == KH_MAT_CSR && A->dia != nullptr && ctx->spmv_dia) {
            // banded operator: the chain kernel computes w = A v_k in its prologue (no SpMV launch, w
            // never touches HBM); returns 0 when that instantiation does not apply
            // a deferred MINRES update of an earlier iteration rides along when the step runs as the three-pass
            // Lanczos kernel and the update's blocks have the same padded geometry (lanczos.h)
            MinresJob job;
            job.on = 0;
            {
                const auto& j = ctx->mr_pending;
                if (j.on && j.V->n == n && j.V->ld == V->ld && j.W->ld == V->ld && j.YK->ld == V->ld &&
                    !(j.V == V && j.vcol == k + 1)) {
                    job.on = 1;
                    job.v = j.V->col(j.vcol);
                    job.w0 = j.W->col(j.slot);
                    job.w1 = j.W->col(1 - j.slot);
                    job.yk = j.YK->col(j.ycol);
                    job.r0 = j.r0; job.r1 = j.r1; job.r2 = j.r2; job.y0 = j.y0;
                }
            }
            ctx->mr_taken = 0;
            const int rc = try_chain(ctx, V, B, w, W->ld, dg, P, k, start, sweeps, presub, h_km1, hk_dev, hdev,
                                     slot, false, ctx->hslot_pin[slot], (int)(k + 2 + pd), A, V->col(k),
                                     job.on ? &job : nullptr);
            if (rc < 0) return rc;
            fused_chain = (rc == 1);
            if (ctx->mr_taken) {
                ctx->mr_pending.on = 0;
                ctx->n_minres_rides += 1;
                ctx->mr_taken = 0;
            }
        }
        if (fused_chain) {
        } else if (fuse_dot0)
            KH_TRY(apply_one(ctx, A, V->col(k), w, EPI_DOT, V->col(start), tmp, 0));
        else
            KH_TRY(apply_one(ctx, A, V->col(k), w, EPI_NONE, nullptr, nullptr, 0));
        // deflated solvers: w <- (I - P) w, and <U, A v_k> behind the H column (deflation.py:135-143)
        if (proj != nullptr) KH_TRY(proj_apply_dev(ctx, proj, w, hdev + (k + 2), W->ld));
    }

    double* nrm_part = part_slot(ctx, SLOT_NRM);
    int nrm_count = grid;     // number of partial sums the norm arrives in
    bool chained = fused_chain;
    if (want_chain && !fused_chain) {
        const int rc = try_chain(ctx, V, B, w, W->ld, dg, P, k, start, sweeps, presub, h_km1, hk_dev, hdev, slot,
                                 false, ctx->hslot_pin[slot], (int)(k + 2 + pd));
        if (rc < 0) return rc;
        chained = (rc == 1);
        if (!chained) {   // not eligible after all: clear the column now, nothing has been accumulated yet
            KH_HIP(hipMemsetAsync(hdev, 0, sizeof(double) * (k + 2), ctx->stream));
        }
    }
    if (!chained && want_blk2) {
        const int rc = chain_blk2_step(ctx, V, w, W->ld, k, hdev, slot, ctx->hslot_pin[slot], (int)(k + 2 + pd), true);
        if (rc < 0) return rc;
        chained = (rc == 1);
    }
    if (!chained && want_chain_xr) {
        const int rc = chain_xr_step(ctx, V, w, W->ld, k, hdev, slot, ctx->hslot_pin[slot], (int)(k + 2 + pd));
        if (rc < 0) return rc;
        chained = (rc == 1);
    }
    bool lowsync = false;
    if (!chained && want_lowsync) {
        const int rc = try_lowsync_mgs(ctx, V, w, W->ld, k, multi, hdev, ctx->scal + SC_COEF, &nrm_count, A);
        if (rc < 0) return rc;
        lowsync = (rc == 1);
    }
    if (chained) {
        // the whole Gram-Schmidt chain, the norm and the normalised store ran in one launch
    } else if (lowsync) {
        // all k + 1 coefficients from one pass and one all-reduce (try_lowsync_mgs); the norm follows below
    } else if (gs_mode == KH_GS_MGS) {
        // column visiting order of all sweeps
        const int64_t ncol = k - start + 1;
        const int64_t len = ncol * sweeps;
        auto colof = [&](int64_t t) { return start + (t % ncol); };
        // state: where the coefficient of the pending axpy comes from
        int src = A_NONE;            // A_PART (slot), A_SCAL (tmp)
        double* pin = nullptr;
        // first dot
        if (fuse_dot0) {
            if (multi) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
            src = A_SCAL;
        } else if (presub) {
            double* pout = part_slot(ctx, SLOT_PING);
            if (hk_dev)
                KH_LINK(A_SCAL, T_DOT, B->col(k - 1), V->col(colof(0)), nullptr, nullptr, nullptr, hk_dev,
                        0.0, pout, nullptr);
            else
                KH_LINK(A_ARG, T_DOT, B->col(k - 1), V->col(colof(0)), nullptr, nullptr, nullptr, nullptr,
                        h_km1, pout, nullptr);
            src = A_PART;
            pin = pout;
        } else {
            double* pout = part_slot(ctx, SLOT_PING);
            KH_LINK(A_NONE, T_DOT, nullptr, V->col(colof(0)), nullptr, nullptr, nullptr, nullptr, 0.0,
                    pout, nullptr);
            src = A_PART;
            pin = pout;
        }
        if (multi && src == A_PART) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, pin, grid, 0, tmp, 0);
            KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
            src = A_SCAL;
        }
        for (int64_t t = 1; t <= len; ++t) {
            const int64_t jprev = colof(t - 1);
            const bool last = (t == len);
            double* pout = last ? nrm_part : part_slot(ctx, (t & 1) ? SLOT_PONG : SLOT_PING);
            const double* pcol = B->col(jprev);
            double* hs = hdev + jprev;
            if (!last) {
                const double* vn = V->col(colof(t));
                if (src == A_PART) KH_LINK(A_PART, T_DOT, pcol, vn, nullptr, nullptr, pin, nullptr, 0.0, pout, hs);
                else KH_LINK(A_SCAL, T_DOT, pcol, vn, nullptr, nullptr, nullptr, tmp, 0.0, pout, hs);
            } else if (md_mat) {      // last update only; Md w and <w, Md w> follow below
                if (src == A_PART) KH_LINK(A_PART, T_NONE, pcol, nullptr, nullptr, nullptr, pin, nullptr, 0.0, nullptr, hs);
                else KH_LINK(A_SCAL, T_NONE, pcol, nullptr, nullptr, nullptr, nullptr, tmp, 0.0, nullptr, hs);
            } else if (Md) {
                if (src == A_PART) KH_LINK(A_PART, T_NRM_DIAG, pcol, nullptr, dg, mw, pin, nullptr, 0.0, pout, hs);
                else KH_LINK(A_SCAL, T_NRM_DIAG, pcol, nullptr, dg, mw, nullptr, tmp, 0.0, pout, hs);
            } else {
                if (src == A_PART) KH_LINK(A_PART, T_NRM, pcol, nullptr, nullptr, nullptr, pin, nullptr, 0.0, pout, hs);
                else KH_LINK(A_SCAL, T_NRM, pcol, nullptr, nullptr, nullptr, nullptr, tmp, 0.0, pout, hs);
            }
            KH_HIP(hipGetLastError());
            src = A_PART;
            pin = pout;
            if (multi && !last) {
                hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, pin, grid, 0, tmp, 0);
                KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
                src = A_SCAL;
            }
        }
    } else {
        // panel (classical) Gram-Schmidt: one reduction round per sweep
        KH_ARG(gs_mode == KH_GS_CGS, "kh_arnoldi_step: unknown gs_mode %d", gs_mode);
        const int64_t ncol = k - start + 1;
        KH_ARG(ncol <= 1024, "kh_arnoldi_step: panel mode handles at most 1024 columns");
        double* coef = ctx->scal + SC_COEF;
        if (presub) {
            // w -= h_km1 * B[:, k-1]
            if (hk_dev)
                KH_LINK(A_SCAL, T_NONE, B->col(k - 1), nullptr, nullptr, nullptr, nullptr, hk_dev, 0.0,
                        nullptr, nullptr);
            else
                KH_LINK(A_ARG, T_NONE, B->col(k - 1), nullptr, nullptr, nullptr, nullptr, nullptr, h_km1,
                        nullptr, nullptr);
        }
        const int rc = try_cgs_reg(ctx, V, B, w, W->ld, dg, mw, start, ncol, sweeps, multi, hdev, coef,
                                   &nrm_count);
        if (rc < 0) return rc;
        if (rc == 0 && want_cgs1)     // not eligible after all: the chunked path accumulates, clear the column now
            KH_HIP(hipMemsetAsync(hdev, 0, sizeof(double) * (k + 2), ctx->stream));
        for (int s = 0; rc == 0 && s < sweeps; ++s) {     // chunked panel kernels (w streamed)
            KH_TRY(dot_panel_dev(ctx, V, start, ncol, w, coef, 0));
            if (multi) KH_TRY(comm_allreduce_dev(ctx, coef, ncol));
            // H[j,k] += coef  (device-side accumulate through the same reduce kernel is not
            // needed: one tiny axpy on scalars)
            hipLaunchKernelGGL(k_waxpby, dim3(1), dim3(BS), 0, ctx->stream, ncol, hdev + start, 1.0,
                               hdev + start, 1.0, coef);
            const int tail = (s == sweeps - 1 && !md_mat) ? (Md ? T_NRM_DIAG : T_NRM) : T_NONE;
            KH_TRY(multiaxpy_cols(ctx, B, start, ncol, coef, 1.0, 1.0, w, tail, dg, mw));
        }
    }
    if (md_mat) {
        // non-Euclidean inner product with a matrix B: mw = B w, then the partial sums of <w, B w>
        KH_TRY(apply_one(ctx, Md, w, mw, EPI_NONE, nullptr, nullptr, 0));
        ColPtrs cp;
        cp.c[0] = mw;
        launch_multidot<1>(ctx, n, cp, w, nrm_part);
        KH_HIP(hipGetLastError());
        nrm_count = grid;
    }
    // 3. norm and normalise
    if (!chained) {
        double* hs = hdev + (k + 1);
        double* vn = V->col(k + 1);
        double* pn = P ? P->col(k + 1) : nullptr;
        if (multi) {
            KH_TRY(reduce_partials_allreduce(ctx, nrm_part, nrm_count, 0, tmp + 1, 1, true));
            hipLaunchKernelGGL((k_scale_store<A_SCAL>), dim3(grid), dim3(BS), 0, ctx->stream, n, w, mw, vn,
                               pn, nullptr, 0, tmp + 1, hs, hdev, (int)(k + 2 + pd), ctx->hslot_pin[slot]);
        } else {
            hipLaunchKernelGGL((k_scale_store<A_PART>), dim3(grid), dim3(BS), 0, ctx->stream, n, w, mw, vn,
                               pn, nrm_part, nrm_count, nullptr, hs, hdev, (int)(k + 2 + pd), ctx->hslot_pin[slot]);
        }
        KH_HIP(hipGetLastError());
    }
    // (the last kernel of the step - chain or scale-store - has written the H column to the pinned slot; the chain
    // kernels also write a completion tag there: no event then, kh_arnoldi_step_end polls the tag)
    if (!(chained && ctx->wait_tag[slot])) {
        ctx->wait_tag[slot] = false;
        KH_HIP(hipEventRecord(ctx->hev[slot], ctx->stream));
    }
    return 0;
}

int kh_arnoldi_step_end(kh_ctx ctx, int slot, int64_t count, double* hcol_out) {
    KH_ARG(ctx && hcol_out, "kh_arnoldi_step_end: NULL");
    RoctxScope range_(ctx, "kh_arnoldi_step_end slot=%lld count=%lld", (long long)slot, (long long)count);
    KH_ARG(slot >= 0 && slot < KH_NSLOT && count >= 0 && count <= ctx->hcap,
           "kh_arnoldi_step_end: slot %d / count %lld", slot, (long long)count);
    KH_ARG(ctx->hev[slot] != nullptr, "kh_arnoldi_step_end: no step was begun");
    if (ctx->wait_tag[slot]) {
        // the step's chain kernel writes its tag behind the H column (CH_SIGNAL_DONE).  Should the tag never show up
        // the stream running empty says the same thing later.
        volatile int* tagp = ctx->done_pin[slot];
        const int want = ctx->done_seq[slot];
        for (unsigned spins = 1; *tagp != want; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spins & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) {          // a sticky launch / device error: the tag will never come
                    ctx->wait_tag[slot] = false;
                    KH_HIP(q);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ctx->wait_tag[slot] = false;
        ctx->n_tag_waits += 1;
    } else {
        KH_HIP(hipEventSynchronize(ctx->hev[slot]));
    }
    KH_TRY(xr_check(ctx));
    // the one-launch projector of a deflated step (proj_reg.h) reports a timed-out sum in a word of its own: the step - its w
    // was garbage whatever Gram-Schmidt kernels followed - is re-run below with the four-launch projector, the kernel stays
    // off for this context
    bool proj_timeout = false;
    if (ctx->proj_err_pin != nullptr && *ctx->proj_err_pin != 0 && ctx->step[slot].kind != 0 && ctx->step[slot].proj != nullptr) {
        proj_timeout = true;
        *ctx->proj_err_pin = 0;
        ctx->proj_reg = 0;
        ctx->n_proj_recovered += 1;
        fprintf(stderr, "krylov_hip: the grid-wide sum of the one-launch deflation projector timed out (a GPU shared with other "
                        "work?); it stays off for this context, the four-launch projector takes over\n");
        *ctx->chain_err_pin[slot] = 1;
    }
    if (*ctx->chain_err_pin[slot] != 0 && kh_multi(ctx) && !proj_timeout) {
        // (chain kernels run on a communicator only as the blocked kernel with the cross-rank sums inside: whatever timed
        // out - a workgroup of this launch, a peer rank - re-running the step HERE alone on other kernels would change the
        // pattern of collectives the peers see)
        const int code = *ctx->chain_err_pin[slot];
        // The caller gives this Arnoldi sequence up.  The steps begun behind the failed one (look-ahead) ran with the same error
        // word set and copied it into THEIR slots: wait for them, then clear the device's word and every slot's copy, and forget
        // the Gram tables those launches may have written garbage rows into - whatever the caller does next on this context
        // (bench.py: every rank back to RCCL and the panel form, together) must not find a stale "timed out" in a slot it
        // re-uses (a panel step does not write its slot's word: it would report the aborted look-ahead step's).
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemset(ctx->chain_err, 0, sizeof(int));
        for (int s2 = 0; s2 < KH_NSLOT; ++s2) {
            *ctx->chain_err_pin[s2] = 0;
            ctx->wait_tag[s2] = false;
            if (s2 != slot) ctx->step[s2].kind = 0;
        }
        ctx->blk_next = -1;
        ctx->ls_next = -1;
        return fail(KH_ERR_COMM, "a sum inside the Gram-Schmidt kernel with in-launch cross-rank sums timed out on rank %d of %d (%s); no rank-local recovery on "
                                 "a communicator", ctx->rank, ctx->nranks, code == 2 ? "a peer rank's contribution did not arrive" :
                                 "a workgroup of the launch did not arrive");
    }
    if (*ctx->chain_err_pin[slot] != 0) {
        // The grid-wide reduction of the chain kernel timed out (its workgroups were not co-resident: a shared
        // or partially masked GPU).  Column k+1 of the basis and this H column are garbage, columns 0..k are
        // intact: switch the chain off for this context and run the SAME step again on the per-column kernels.
        // Steps begun after it (look-ahead) consumed the garbage; each of them reports the error in its own
        // slot and is recovered in turn when the host asks for it, in order.
        // (code 3: a step that consumed the garbage a PROJECTOR timeout left behind - re-run like any other, but the chain
        // kernels are not to blame and stay on)
        if (*ctx->chain_err_pin[slot] == 3) proj_timeout = true;
        *ctx->chain_err_pin[slot] = 0;
        if (ctx->chain_enabled && !proj_timeout) chain_switch_off(ctx);      // (look-ahead steps that saw the same timeout do not count again)
        ctx->n_chain_recovered += proj_timeout ? 0 : 1;
        KH_HIP(hipStreamSynchronize(ctx->stream));
        KH_HIP(hipMemset(ctx->chain_err, 0, sizeof(int)));
        if (proj_timeout && ctx->proj_err != nullptr) {
            KH_HIP(hipMemset(ctx->proj_err, 0, sizeof(int)));
            *ctx->proj_err_pin = 0;          // (look-ahead steps in flight copied the set word again)
        }
        for (int s2 = 0; s2 < KH_NSLOT; ++s2)      // later steps in flight saw the same error word
            if (s2 != slot && *ctx->chain_err_pin[s2] == 0 && ctx->step[s2].kind != 0 &&
                ctx->step[s2].V == ctx->step[slot].V && ctx->step[s2].k > ctx->step[slot].k)
                *ctx->chain_err_pin[s2] = proj_timeout ? 3 : 1;
        const kh_step_s st = ctx->step[slot];
        struct InRecovery {
            kh_ctx c;
            explicit InRecovery(kh_ctx c_) : c(c_) { c->chain_in_recovery += 1; }
            ~InRecovery() { c->chain_in_recovery -= 1; }
        } in_recovery_(ctx);
        if (st.kind == 1)
            KH_TRY(kh_arnoldi_step_begin(ctx, st.A, st.proj, st.Md, st.V, st.P, st.W, st.wcol, st.k, st.start, st.sweeps,
                                         st.gs_mode, st.h_km1[0], slot));
        else if (st.kind == 2)
            KH_TRY(kh_zarnoldi_step_begin_md(ctx, st.A, st.proj, st.Md, st.V, st.P, st.W, st.wcol, st.k, st.start,
                                             st.sweeps, st.gs_mode, st.h_km1, slot));
        else
            return fail(KH_ERR_HIP, "grid-wide reduction of the MGS chain kernel timed out and the step cannot be "
                                    "re-run (no record of it)");
        // (after a projector timeout the chain kernels are still on: the re-run step may have ended in a chain launch that
        // signals through its completion tag and records NO event - wait for the stream itself)
        KH_HIP(hipStreamSynchronize(ctx->stream));
        ctx->wait_tag[slot] = false;
    }
    memcpy(hcol_out, ctx->hslot_pin[slot], sizeof(double) * count);
    return 0;
}

int kh_arnoldi_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t wcol,
                    int64_t k, int64_t start, int sweeps, int gs_mode, double h_km1,
                    double* hcol_out) {
    KH_ARG(hcol_out != nullptr, "kh_arnoldi_step: NULL argument");
    KH_TRY(kh_arnoldi_step_begin(ctx, A, nullptr, Md, V, P, W, wcol, k, start, sweeps, gs_mode, h_km1, 0));
    return kh_arnoldi_step_end(ctx, 0, k + 2, hcol_out);
}

int kh_proj_create(kh_ctx ctx, kh_vec W, kh_vec V, int64_t d, const double* T, const double* WRH,
                   int iterations, kh_proj* out) {
    KH_ARG(ctx && W && V && out, "kh_proj_create: NULL");
    KH_ARG(d >= 1 && d <= 1024 && W->ncols >= d && V->ncols >= d && W->n == V->n, "kh_proj_create: shapes");
    KH_ARG(iterations >= 1, "kh_proj_create: iterations < 1");
    kh_proj p = new kh_proj_s();
    p->ctx = ctx;
    p->W = W;
    p->V = V;
    p->d = d;
    p->iterations = iterations;
    auto body = [&]() -> int {
        KH_HIP(hipMalloc(&p->c0, sizeof(double) * 3 * d));
        p->c1 = p->c0 + d;
        p->ya = p->c0 + 2 * d;
        if (T) {
            KH_HIP(hipMalloc(&p->T, sizeof(double) * d * d));
            KH_HIP(hipMemcpy(p->T, T, sizeof(double) * d * d, hipMemcpyHostToDevice));
        }
        if (WRH) {
            KH_HIP(hipMalloc(&p->WRH, sizeof(double) * d * d));
            KH_HIP(hipMemcpy(p->WRH, WRH, sizeof(double) * d * d, hipMemcpyHostToDevice));
        }
        return 0;
    };
    const int rc = body();
    if (rc != 0) {          // nothing half-built stays behind
        kh_proj_free(p);
        return rc;
    }
    *out = p;
    return 0;
}

int kh_proj_free(kh_proj p) {
    if (!p) return 0;
    (void)hipStreamSynchronize(p->ctx->stream);
    forget_steps(p->ctx, p);
    (void)hipFree(p->c0);
    (void)hipFree(p->T);
    (void)hipFree(p->WRH);
    delete p;
    return 0;
}

int kh_proj_apply_complement(kh_ctx ctx, kh_proj p, kh_vec A, int64_t acol, kh_vec Z, int64_t zcol,
                             double* ya_out) {
    KH_ARG(ctx && p, "kh_proj_apply_complement: NULL");
    KH_TRY(check_vec(A, acol, 1, "kh_proj_apply_complement(a)"));
    KH_TRY(check_vec(Z, zcol, 1, "kh_proj_apply_complement(z)"));
    KH_ARG(A->n == p->W->n && Z->n == A->n, "kh_proj_apply_complement: length mismatch");
    if (!(A == Z && acol == zcol))
        KH_HIP(hipMemcpyAsync(Z->col(zcol), A->col(acol), sizeof(double) * A->n, hipMemcpyDeviceToDevice,
                              ctx->stream));
    const int64_t pr0 = ctx->n_proj_reg;
    KH_TRY(proj_apply_dev(ctx, p, Z->col(zcol), ya_out ? p->ya : nullptr, Z->ld));
    if (ctx->n_proj_reg != pr0) {
        // the one-launch form ran: its error word (a timed-out grid-wide sum) is looked at HERE - nobody else would
        KH_HIP(hipStreamSynchronize(ctx->stream));
        if (*ctx->proj_err_pin != 0) {
            *ctx->proj_err_pin = 0;
            KH_HIP(hipMemset(ctx->proj_err, 0, sizeof(int)));
            ctx->proj_reg = 0;
            ctx->n_proj_recovered += 1;
            fprintf(stderr, "krylov_hip: the grid-wide sum of the one-launch deflation projector timed out; it stays off for this "
                            "context, the four-launch projector takes over\n");
            if (A == Z && acol == zcol)
                return fail(KH_ERR_HIP, "kh_proj_apply_complement: the one-launch projector timed out on an in-place call (the "
                                        "input has been overwritten); call again with the input restored");
            KH_HIP(hipMemcpyAsync(Z->col(zcol), A->col(acol), sizeof(double) * A->n, hipMemcpyDeviceToDevice, ctx->stream));
            KH_TRY(proj_apply_dev(ctx, p, Z->col(zcol), ya_out ? p->ya : nullptr, Z->ld));
        }
    }
    if (ya_out) return fetch_scalars(ctx, p->ya, p->d, ya_out);
    return 0;
}

}  // extern "C"

#include "zpath.h"
