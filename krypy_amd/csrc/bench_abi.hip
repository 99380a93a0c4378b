// The measurement ABI of bench.py (split from krylov_hip.hip in round 5): HIP-event timing of single kernels and of the solver's
// own Arnoldi steps on the context's stream, the phase trace of the diagnostic build.  Not part of the solver path.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "kernels.h"
#include "krylov_steps.h"

using namespace kh;

extern "C" {

// Timing harness for bench.py: `reps` back-to-back launches of one hot kernel between two HIP
// events on the context's stream.  The launches rotate through the columns of V exactly like the
// solver does (p = V[:, j], vnext = V[:, j+1]) so cache behaviour matches the real chain.
// An empty launch in front of and behind every kh_bench_kernel loop: a kernel trace / PMC pass tells bench.py's micro-launches from
// the solver's launches of the same kernel by what lies between two markers (tools/summarize_prof.py) - not by "the last N
// launches", and not by a template argument that other launches share.  Outside the timed region (before ev0, behind ev1).
static __global__ void k_bench_marker(int which) { (void)which; }

int kh_bench_kernel(kh_ctx ctx, int which, kh_vec V, kh_vec W, int reps, double* avg_ms) {
    KH_ARG(ctx && V && W && avg_ms, "kh_bench_kernel: NULL");
    KH_ARG(V->ncols >= 17 && W->ncols >= 2 && V->n == W->n, "kh_bench_kernel: need >= 17 basis columns");
    KH_ARG(reps >= 1, "kh_bench_kernel: reps");
    const int64_t n = V->n;
    const int grid = grid_for(ctx, n);
    double* w = W->col(0);
    double* mw = W->col(1);
    double* pa = part_slot(ctx, SLOT_PING);
    double* pb = part_slot(ctx, SLOT_PONG);
    KH_TRY(ensure_hcap(ctx, 1024));
    KH_HIP(hipMemsetAsync(pa, 0, sizeof(double) * NB_MAX * 2, ctx->stream));  // alpha = 0: w unchanged
    KH_HIP(hipMemsetAsync(ctx->scal + SC_COEF, 0, sizeof(double) * MAXC, ctx->stream));
    const double four = 4.0;  // k_scale_store divides by sqrt(4)
    KH_TRY(push_scalars(ctx, &four, 1, ctx->scal + SC_TMP + 8));
    ColPtrs cp;
    for (int i = 0; i < MAXC; ++i) cp.c[i] = V->col(i);
    hipLaunchKernelGGL(k_bench_marker, dim3(1), dim3(64), 0, ctx->stream, which);
    KH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        const int j = r % 16;
        switch (which) {
            case 0:
                hipLaunchKernelGGL((k_gs_link<A_PART, T_DOT>), dim3(grid), dim3(BS), 0, ctx->stream, n,
                                   V->col(j), V->col(j + 1), w, nullptr, nullptr, (r & 1) ? pb : pa, grid,
                                   nullptr, 0.0, (r & 1) ? pa : pb, nullptr);
                break;
            case 1:
                hipLaunchKernelGGL((k_multidot<16>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp, w, ctx->part,
                                   NB_MAX);
                break;
            case 2:
                hipLaunchKernelGGL((k_multiaxpy<16, T_NONE, 1>), dim3(grid), dim3(BS), 0, ctx->stream, n, cp,
                                   ctx->scal + SC_COEF, 1.0, 1.0, w, nullptr, nullptr, nullptr);
                break;
            case 3:
                hipLaunchKernelGGL((k_gs_link<A_PART, T_NRM>), dim3(grid), dim3(BS), 0, ctx->stream, n,
                                   V->col(j), nullptr, w, nullptr, nullptr, pa, grid, nullptr, 0.0,
                                   part_slot(ctx, SLOT_NRM), nullptr);
                break;
            case 4:
                hipLaunchKernelGGL((k_scale_store<A_SCAL>), dim3(grid), dim3(BS), 0, ctx->stream, n, w, nullptr,
                                   mw, nullptr, nullptr, 0, ctx->scal + SC_TMP + 8, nullptr);
                break;
            case 5:
            case 6:
            case 7: {
                // the register-resident chain over 16 columns x 4 sweeps = 64 links per launch
                ctx->chain_debug = which - 5;
                KH_HIP(hipMemsetAsync(ctx->hslot_dev[0], 0, sizeof(double) * 64, ctx->stream));
                const int rc = try_chain(ctx, V, V, w, W->ld, nullptr, nullptr, 15, 0, 4, false, 0.0,
                                         nullptr, ctx->hslot_dev[0], 0);
                ctx->chain_debug = 0;
                if (rc != 1) return fail(KH_ERR_UNSUPPORTED, "kh_bench_kernel: chain kernel not eligible");
                break;
            }
            case 20:
            case 21:
            case 22:
            case 23: {
                // the blocked chain (chain_blk.h) over 64 columns, one sweep, Gram entries from the table (steady state
                // of a sequence; the table's content does not matter for the time); 21 / 22 / 23: without the exchange
                // between workgroups / without the column stream / without both
                KH_ARG(V->ncols >= 66, "kh_bench_kernel: the blocked chain needs 66 basis columns");
                ctx->chain_debug = which - 20;
                ctx->blk_V = V;
                ctx->blk_next = 63;
                const int64_t nb0 = ctx->n_chain_blk;
                const int rc = try_chain(ctx, V, V, w, W->ld, nullptr, nullptr, 63, 0, 1, false, 0.0, nullptr,
                                         ctx->hslot_dev[0], 0);
                ctx->chain_debug = 0;
                ctx->blk_next = -1;
                if (rc != 1 || ctx->n_chain_blk == nb0) return fail(KH_ERR_UNSUPPORTED, "kh_bench_kernel: blocked chain kernel not eligible");
                break;
            }
            case 24: {     // ... and the per-column kernel on the same 64 columns
                const int keep = ctx->chain_blk;
                ctx->chain_blk = 0;
                const int rc = try_chain(ctx, V, V, w, W->ld, nullptr, nullptr, 63, 0, 1, false, 0.0, nullptr,
                                         ctx->hslot_dev[0], 0);
                ctx->chain_blk = keep;
                if (rc != 1) return fail(KH_ERR_UNSUPPORTED, "kh_bench_kernel: chain kernel not eligible");
                break;
            }
            case 8: {
                // register-resident panel GS over 16 columns: k_cgs_dots + reduce + k_cgs_update
                int cnt = 0;
                KH_HIP(hipMemsetAsync(ctx->hslot_dev[0], 0, sizeof(double) * 64, ctx->stream));
                const int rc = try_cgs_reg(ctx, V, V, w, W->ld, nullptr, nullptr, 0, 16, 1, false,
                                           ctx->hslot_dev[0], ctx->scal + SC_COEF, &cnt);
                if (rc != 1) return fail(KH_ERR_UNSUPPORTED, "kh_bench_kernel: cgs kernels not eligible");
                break;
            }
            case 9:     // attainable ceiling: copy of 8 columns (8 N doubles read + 8 N written per launch)
                hipLaunchKernelGGL(k_stream_copy, dim3(ctx->ncu * 8), dim3(BS), 0, ctx->stream, (V->ld * 8) >> 1,
                                   reinterpret_cast<const double2*>(V->col(0)), reinterpret_cast<double2*>(V->col(8)));
                break;
            case 10:    // attainable ceiling: triad a = b + s c on 4-column chunks (2 reads + 1 write)
                hipLaunchKernelGGL(k_stream_triad, dim3(ctx->ncu * 8), dim3(BS), 0, ctx->stream, (V->ld * 4) >> 1,
                                   reinterpret_cast<const double2*>(V->col(0)),
                                   reinterpret_cast<const double2*>(V->col(4)), 0.5,
                                   reinterpret_cast<double2*>(V->col(8)));
                break;
            case 11:    // attainable ceiling: read-only sum of 16 columns (what a dot phase does)
                hipLaunchKernelGGL(k_stream_read, dim3(ctx->ncu * 8), dim3(BS), 0, ctx->stream, (V->ld * 16) >> 1,
                                   reinterpret_cast<const double2*>(V->col(0)), part_slot(ctx, SLOT_PING));
                break;
#define KH_PROBE_COPY(U, NTS)                                                                                  \
    hipLaunchKernelGGL((k_probe_copy<U, NTS>), dim3((unsigned)((((V->ld * 8) >> 1) + U * BS - 1) / (U * BS))),  \
                       dim3(BS), 0, ctx->stream, (V->ld * 8) >> 1, reinterpret_cast<const double2*>(V->col(0)), \
                       reinterpret_cast<double2*>(V->col(8)))
            case 12: KH_PROBE_COPY(1, false); break;
            case 13: KH_PROBE_COPY(4, false); break;
            case 14: KH_PROBE_COPY(8, false); break;
            case 15: KH_PROBE_COPY(4, true); break;
#undef KH_PROBE_COPY
#define KH_PROBE_READ(U)                                                                                       \
    hipLaunchKernelGGL((k_probe_read<U>), dim3((unsigned)((((V->ld * 16) >> 1) + U * BS - 1) / (U * BS))),       \
                       dim3(BS), 0, ctx->stream, (V->ld * 16) >> 1, reinterpret_cast<const double2*>(V->col(0)), \
                       part_slot(ctx, SLOT_PING))
            case 16: KH_PROBE_READ(4); break;
            case 17: KH_PROBE_READ(8); break;
            case 18: KH_PROBE_READ(16); break;
#undef KH_PROBE_READ
            default:
                return fail(KH_ERR_ARG, "kh_bench_kernel: unknown kernel id %d", which);
        }
    }
    KH_HIP(hipGetLastError());
    KH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    hipLaunchKernelGGL(k_bench_marker, dim3(1), dim3(64), 0, ctx->stream, which);
    KH_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    KH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / reps;
    return 0;
}

// The solver's own launch sequence, timed: `reps` times the Arnoldi steps k = 0 .. m-1 on basis V (column 0 = the
// caller's unit vector), begun with one step of look-ahead and fetched in order exactly like kh_gmres_cycle does -
// without the Givens bookkeeping.  avg_step_ms = HIP-event time / (reps * m): with a banded operator and w in
// registers that is the average duration of ONE launch of the fused chain kernel over k+1 = 1 .. m links.
int kh_bench_arnoldi(kh_ctx ctx, kh_mat A, kh_vec V, kh_vec W, int64_t m, int gs_mode, int reps, double* avg_step_ms) {
    KH_ARG(ctx && A && V && W && avg_step_ms, "kh_bench_arnoldi: NULL");
    KH_ARG(m >= 1 && V->ncols >= m + 1 && reps >= 1, "kh_bench_arnoldi: need m + 1 = %lld basis columns, have %lld",
           (long long)(m + 1), (long long)V->ncols);
    std::vector<double> col((size_t)m + 2);
    KH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        int64_t enq = 0;
        for (int64_t k = 0; k < m; ++k) {
            const int64_t last = std::min<int64_t>(k + 1, m - 1);
            while (enq <= last) {
                KH_TRY(kh_arnoldi_step_begin(ctx, A, nullptr, nullptr, V, nullptr, W, 0, enq, 0, 1, gs_mode, 0.0,
                                             (int)(enq % KH_NSLOT)));
                ++enq;
            }
            KH_TRY(kh_arnoldi_step_end(ctx, (int)(k % KH_NSLOT), k + 2, col.data()));
        }
    }
    KH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    KH_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    KH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_step_ms = (double)ms / ((double)reps * (double)m);
    return 0;
}

#ifdef KH_CHAIN_TRACE
// Diagnostic build only (make -C krypy_amd/csrc trace; tools/chain_trace.py): one traced 64-link launch of
// the chain kernel (16 columns x 4 sweeps, like kh_bench_kernel); out gets [G][64][2 waves][8] stamps of
// the constant 100 MHz clock.  Not part of the C ABI of the product.
int kh_chain_trace(kh_ctx ctx, kh_vec V, kh_vec W, unsigned long long* out, int64_t cap, int* g_out) {
    KH_ARG(ctx && V && W && out && g_out, "kh_chain_trace: NULL");
    KH_ARG(V->ncols >= 17 && W->ncols >= 1 && V->n == W->n, "kh_chain_trace: need >= 17 basis columns");
    int r2 = 0, G = 0;
    KH_ARG(chain_geometry(ctx, V->n, &r2, &G), "kh_chain_trace: chain not eligible");
    const size_t words = (size_t)G * 64 * 2 * 8;
    KH_ARG((int64_t)words <= cap, "kh_chain_trace: buffer too small (%zu words)", words);
    KH_TRY(ensure_hcap(ctx, 1024));
    unsigned long long* dev = nullptr;
    KH_HIP(hipMalloc(&dev, words * sizeof(unsigned long long)));
    KH_HIP(hipMemset(dev, 0, words * sizeof(unsigned long long)));
    for (int rep = 0; rep < 4; ++rep) {
        ctx->chain_trace = (rep == 3) ? dev : nullptr;
        const int rc = try_chain(ctx, V, V, W->col(0), W->ld, nullptr, nullptr, 15, 0, 4, false, 0.0, nullptr,
                                 ctx->hslot_dev[0], 0);
        ctx->chain_trace = nullptr;
        if (rc != 1) {
            (void)hipFree(dev);
            return fail(KH_ERR_UNSUPPORTED, "kh_chain_trace: chain kernel not eligible");
        }
    }
    KH_HIP(hipStreamSynchronize(ctx->stream));
    KH_HIP(hipMemcpy(out, dev, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    *g_out = G;
    return 0;
}
#endif

}  // extern "C"
