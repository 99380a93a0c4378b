// The host loops of the solvers in C (split from krylov_hip.hip in round 5: krylov_steps.h has the map): the caller's drotg,
// kh_gmres_cycle, kh_residual, the MINRES recurrences and kh_minres_cycle, the CG step and kh_cg_cycle.
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

// BLAS drotg (reference implementation): c, s with [c s; -s c] [a; b] = [r; 0]
static inline void host_drotg(double a, double b, double* c, double* s) {
    const double roe = std::fabs(a) > std::fabs(b) ? a : b;
    const double scale = std::fabs(a) + std::fabs(b);
    if (scale == 0.0) {
        *c = 1.0;
        *s = 0.0;
        return;
    }
    double r = scale * std::sqrt((a / scale) * (a / scale) + (b / scale) * (b / scale));
    if (roe < 0.0) r = -r;
    *c = a / r;
    *s = b / r;
}

// the rotation of a C host loop: the caller's own BLAS drotg when it has handed one over (kh_ctx_set_rotg: the same bits
// as the per-step loop of the host layer), the reference formula otherwise
static inline void ctx_rotg(kh_ctx ctx, double a, double b, double* c, double* s) {
    if (ctx->rotg != nullptr) {
        double a_ = a, b_ = b;
        ctx->rotg(&a_, &b_, c, s);
    } else {
        host_drotg(a, b, c, s);
    }
}

int kh_ctx_set_rotg(kh_ctx ctx, void (*drotg)(double*, double*, double*, double*)) {
    KH_ARG(ctx != nullptr, "kh_ctx_set_rotg: NULL context");
    ctx->rotg = drotg;
    return 0;
}

int kh_gmres_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t k0, int64_t k_stop,
                   int64_t k_last, int sweeps, int gs_mode, int64_t* enq_io, double tol, double bnorm, double* H, int64_t ldh,
                   double* R, int64_t ldr, double* cs, double* y, double* h2_io, double* resn, int64_t* k_done,
                   int* reason) {
    KH_ARG(ctx && A && V && W && enq_io && H && R && cs && y && h2_io && resn && k_done && reason, "kh_gmres_cycle: NULL");
    KH_ARG(k0 >= 0 && k0 <= k_stop && k_stop <= k_last + 1 && k_last + 2 <= V->ncols, "kh_gmres_cycle: steps [%lld, %lld), last %lld, %lld basis columns",
           (long long)k0, (long long)k_stop, (long long)k_last, (long long)V->ncols);
    KH_ARG(*enq_io >= k0 && *enq_io <= k0 + KH_NSLOT - 1, "kh_gmres_cycle: %lld steps in flight", (long long)(*enq_io - k0));
    KH_ARG(ldh >= k_stop && ldr >= k_stop, "kh_gmres_cycle: leading dimensions");
    RoctxScope range_(ctx, "kh_gmres_cycle");
    int64_t enq = *enq_io;
    double h2 = *h2_io;
    *reason = KH_CYCLE_LIMIT;
    int64_t k = k0;
    std::vector<double> col((size_t)k_stop + 2);
    for (; k < k_stop; ++k) {
        // look-ahead: step k + 1 depends on device data only - it is enqueued before the host waits for step k
        const int64_t last = std::min<int64_t>(k + 1, k_last);
        while (enq <= last) {
            KH_TRY(kh_arnoldi_step_begin(ctx, A, nullptr, Md, V, P, W, 0, enq, 0, sweeps, gs_mode, 0.0, (int)(enq % KH_NSLOT)));
            ++enq;
        }
        KH_TRY(kh_arnoldi_step_end(ctx, (int)(k % KH_NSLOT), k + 2, col.data()));
        const double hn = col[(size_t)k + 1];
        // invariance pre-test of the host layer (utils.py:1035-1039 through the Frobenius norm): when it does not
        // clear the step, the caller decides - the column stays in its slot, nothing of it is recorded here
        double c2 = 0.0;
        for (int64_t i = 0; i <= k + 1; ++i) c2 += col[(size_t)i] * col[(size_t)i];
        const double fro = std::sqrt(h2 + c2);
        if (!(fro > 0.0) || !(hn / fro > 1e-14) || !std::isfinite(fro)) {
            *reason = KH_CYCLE_CHECK;
            break;
        }
        h2 += c2;
        for (int64_t i = 0; i <= k + 1; ++i) H[i * ldh + k] = col[(size_t)i];
        // the new column through the previous rotations, then its own (linsys.py:980-991)
        for (int64_t i = 0; i < k; ++i) {
            const double c = cs[2 * i], s = cs[2 * i + 1];
            const double t0 = col[(size_t)i], t1 = col[(size_t)i + 1];
            col[(size_t)i] = c * t0 + s * t1;
            col[(size_t)i + 1] = -s * t0 + c * t1;
        }
        double c, s;
        ctx_rotg(ctx, col[(size_t)k], col[(size_t)k + 1], &c, &s);
        cs[2 * k] = c;
        cs[2 * k + 1] = s;
        {
            const double t0 = col[(size_t)k], t1 = col[(size_t)k + 1];
            col[(size_t)k] = c * t0 + s * t1;
            col[(size_t)k + 1] = -s * t0 + c * t1;
        }
        for (int64_t i = 0; i <= k + 1; ++i) R[i * ldr + k] = col[(size_t)i];
        {
            const double t0 = y[k], t1 = y[k + 1];
            y[k] = c * t0 + s * t1;
            y[k + 1] = -s * t0 + c * t1;
        }
        resn[k] = std::fabs(y[k + 1]);
        if (!(resn[k] / bnorm > tol)) {      // the caller's own test, linsys.py:476 (also a nan: its loop sees it)
            ++k;
            *reason = KH_CYCLE_TOL;
            break;
        }
    }
    *k_done = k;
    *enq_io = enq;
    *h2_io = h2;
    ctx->n_cycle_steps += k - k0;
    return 0;
}

int kh_residual(kh_ctx ctx, kh_mat A, kh_vec Bv, int64_t bcol, kh_vec X, int64_t xcol, kh_vec R,
                int64_t rcol, double* nrm) {
    KH_ARG(ctx && A && nrm, "kh_residual: NULL");
    RoctxScope range_(ctx, "kh_residual");
    KH_TRY(check_vec(Bv, bcol, 1, "kh_residual(B)"));
    KH_TRY(check_vec(X, xcol, 1, "kh_residual(X)"));
    KH_TRY(check_vec(R, rcol, 1, "kh_residual(R)"));
    KH_ARG(Bv->n == A->n_rows && R->n == A->n_rows, "kh_residual: dimension mismatch");
    KH_ARG(!(R == X && rcol == xcol), "kh_residual: r must not alias x");
    double* tmp = ctx->scal + SC_TMP;
    if (A->kind == KH_MAT_CSR && A->nblk > 0) {
        KH_TRY(apply_one(ctx, A, X->col(xcol), R->col(rcol), EPI_RES, Bv->col(bcol), tmp,
                         kh_multi(ctx) ? 0 : 2));
        if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
        KH_TRY(fetch_scalars(ctx, tmp, 1, nrm));
        if (kh_multi(ctx)) *nrm = sqrt(fabs(*nrm));
        return 0;
    }
    KH_ARG(!(R == Bv && rcol == bcol), "kh_residual: r must not alias b for non-CSR operators");
    KH_TRY(apply_one(ctx, A, X->col(xcol), R->col(rcol), EPI_NONE, nullptr, nullptr, 0));
    KH_TRY(kh_waxpby(ctx, R, rcol, 1.0, Bv, bcol, -1.0, R, rcol));
    return kh_nrm2(ctx, R, rcol, nrm);
}

// run a deferred MINRES recurrence update now, as a launch of its own
static int minres_flush(kh_ctx ctx) {
    if (!ctx->mr_pending.on) return 0;
    auto& j = ctx->mr_pending;
    j.on = 0;
    const int64_t n = j.V->n;
    hipLaunchKernelGGL(k_minres_update, dim3(grid_lin(ctx, n)), dim3(BS), 0, ctx->stream, n, j.V->col(j.vcol),
                       j.W->col(j.slot), j.W->col(1 - j.slot), j.r0, j.r1, j.r2, j.y0, j.YK->col(j.ycol));
    KH_HIP(hipGetLastError());
    return 0;
}

int kh_minres_flush(kh_ctx ctx) {
    KH_ARG(ctx, "kh_minres_flush: NULL ctx");
    return minres_flush(ctx);
}

int kh_minres_update_deferred(kh_ctx ctx, kh_vec V, int64_t k, kh_vec Wk, int slot, double r0, double r1,
                              double r2, double y0, kh_vec YK, int64_t ycol) {
    KH_ARG(ctx, "kh_minres_update_deferred: NULL ctx");
    KH_TRY(check_vec(V, k, 1, "kh_minres_update_deferred(V)"));
    KH_TRY(check_vec(Wk, 0, 2, "kh_minres_update_deferred(W)"));
    KH_TRY(check_vec(YK, ycol, 1, "kh_minres_update_deferred(yk)"));
    KH_ARG(slot == 0 || slot == 1, "kh_minres_update_deferred: slot");
    KH_ARG(V->n == Wk->n && V->n == YK->n, "kh_minres_update_deferred: length mismatch");
    KH_TRY(minres_flush(ctx));          // updates run in the order they were given
    auto& j = ctx->mr_pending;
    j.V = V; j.vcol = k; j.W = Wk; j.slot = slot; j.YK = YK; j.ycol = ycol;
    j.r0 = r0; j.r1 = r1; j.r2 = r2; j.y0 = y0;
    j.on = 1;
    return 0;
}

int kh_minres_update(kh_ctx ctx, kh_vec V, int64_t k, kh_vec Wk, int slot, double r0, double r1,
                     double r2, double y0, kh_vec YK, int64_t ycol) {
    KH_ARG(ctx, "kh_minres_update: NULL ctx");
    KH_TRY(check_vec(V, k, 1, "kh_minres_update(V)"));
    KH_TRY(check_vec(Wk, 0, 2, "kh_minres_update(W)"));
    KH_TRY(check_vec(YK, ycol, 1, "kh_minres_update(yk)"));
    KH_ARG(slot == 0 || slot == 1, "kh_minres_update: slot");
    KH_ARG(V->n == Wk->n && V->n == YK->n, "kh_minres_update: length mismatch");
    KH_TRY(minres_flush(ctx));
    const int64_t n = V->n;
    hipLaunchKernelGGL(k_minres_update, dim3(grid_lin(ctx, n)), dim3(BS), 0, ctx->stream, n, V->col(k),
                       Wk->col(slot), Wk->col(1 - slot), r0, r1, r2, y0, YK->col(ycol));
    KH_HIP(hipGetLastError());
    return 0;
}

// A run of MINRES iterations in one call (krypy/linsys.py:791-853; the header has the contract)
int kh_minres_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec V, kh_vec P, kh_vec W, int64_t k0, int64_t k_stop,
                    int64_t k_last, int64_t base, int64_t* enq_io, double tol, double bnorm, double* H, int64_t ldh,
                    kh_vec Wm, int* wslot_io, kh_vec YK, int64_t ycol, double* st, double* h2_io, double* resn,
                    int64_t* k_done, int* reason) {
    KH_ARG(ctx && A && V && W && enq_io && H && Wm && wslot_io && YK && st && h2_io && resn && k_done && reason,
           "kh_minres_cycle: NULL");
    KH_ARG(base >= 0 && k0 >= base && k0 <= k_stop && k_stop <= k_last + 1 && k_last + 2 - base <= V->ncols,
           "kh_minres_cycle: steps [%lld, %lld), last %lld, window base %lld, %lld basis columns", (long long)k0,
           (long long)k_stop, (long long)k_last, (long long)base, (long long)V->ncols);
    KH_ARG(k0 == 0 || k0 - 1 >= base, "kh_minres_cycle: column k0 - 1 is not in the window");
    KH_ARG(*enq_io >= k0 && *enq_io <= k0 + KH_NSLOT - 1, "kh_minres_cycle: %lld steps in flight", (long long)(*enq_io - k0));
    KH_ARG(ldh >= k_stop, "kh_minres_cycle: leading dimension of H");
    KH_ARG(*wslot_io == 0 || *wslot_io == 1, "kh_minres_cycle: W slot");
    KH_ARG(Wm->ncols >= 2 && Wm->n == V->n && YK->n == V->n, "kh_minres_cycle: W / yk shape");
    RoctxScope range_(ctx, "kh_minres_cycle");
    int64_t enq = *enq_io;
    double h2 = *h2_io;
    int wslot = *wslot_io;
    // st: the two remembered rotations (older first), how many of them exist, the rotated right-hand side
    double g1c = st[0], g1s = st[1], g2c = st[2], g2s = st[3];
    int nrot = (int)st[4];
    double y0 = st[5], y1 = st[6];
    *reason = KH_CYCLE_LIMIT;
    int64_t k = k0;
    for (; k < k_stop; ++k) {
        // look-ahead (utils.Arnoldi._begin): a Lanczos step takes H[e, e-1] from the host when its predecessor has been
        // fetched, from the predecessor's device-side H column (NaN) when it is still in flight
        const int64_t last = std::min<int64_t>(k + 1, k_last);
        while (enq <= last) {
            const int64_t e = enq;
            double h_km1 = 0.0;
            if (e > 0) h_km1 = (e <= k) ? H[e * ldh + (e - 1)] : std::nan("");
            KH_TRY(kh_arnoldi_step_begin(ctx, A, nullptr, Md, V, P, W, 0, e - base, e > 0 ? e - base : 0, 1, KH_GS_MGS, h_km1,
                                         (int)(e % KH_NSLOT)));
            ++enq;
        }
        // the step's column arrives in window coordinates: only its last two entries are this step's
        const int64_t kp = k - base;
        std::vector<double>& colv = ctx->cyc_col;
        if ((int64_t)colv.size() < kp + 2) colv.resize((size_t)(kp + 2) + 64);
        KH_TRY(kh_arnoldi_step_end(ctx, (int)(k % KH_NSLOT), kp + 2, colv.data()));
        const double alpha = colv[(size_t)kp], hn = colv[(size_t)kp + 1];
        const double hkm = (k > 0) ? H[k * ldh + (k - 1)] : 0.0;       // H[k-1, k] = H[k, k-1]  (utils.py:1000-1003)
        // invariance pre-test (utils.py:1035-1039 through the Frobenius norm); a step that does not clear it is not
        // recorded: it stays in its slot and the caller's Arnoldi.advance decides it with the exact 2-norm
        const double c2 = (k > 0 ? hkm * hkm : 0.0) + alpha * alpha + hn * hn;
        const double fro = std::sqrt(h2 + c2);
        if (!(fro > 0.0) || !(hn / fro > 1e-14) || !std::isfinite(fro)) {
            *reason = KH_CYCLE_CHECK;
            break;
        }
        h2 += c2;
        if (k > 0) H[(k - 1) * ldh + k] = hkm;
        H[k * ldh + k] += alpha;
        H[(k + 1) * ldh + k] = hn;
        // QR update of the Lanczos matrix with the two remembered rotations (linsys.py:826-841), the expressions of
        // the host layer's loop term for term
        double R0 = 0.0, R1 = (k > 0) ? hkm : 0.0;
        if (nrot >= 2) {
            const double u = R0, v = R1;
            R0 = g1c * u + g1s * v;
            R1 = -g1s * u + g1c * v;
        }
        double R2 = H[k * ldh + k];
        const double R3 = hn;
        if (nrot >= 1) {
            const double u = R1, v = R2;
            R1 = g2c * u + g2s * v;
            R2 = -g2s * u + g2c * v;
        }
        g1c = g2c; g1s = g2s;
        double c, s;
        ctx_rotg(ctx, R2, R3, &c, &s);
        g2c = c; g2s = s;
        nrot = nrot < 2 ? nrot + 1 : 2;
        R2 = c * R2 + s * R3;
        {
            const double u = y0, v = y1;
            y0 = c * u + s * v;
            y1 = -s * u + c * v;
        }
        // z = (v_k - R0 W0 - R1 W1) / R2;  W <- [W1, z];  yk += y0 z   (linsys.py:844-846), carried by the next launch
        KH_TRY(kh_minres_update_deferred(ctx, V, kp, Wm, wslot, R0, R1, R2, y0, YK, ycol));
        wslot = 1 - wslot;
        y0 = y1;
        y1 = 0.0;
        resn[k] = std::fabs(y0);
        if (!(resn[k] / bnorm > tol)) {      // the caller's own test, linsys.py:476
            ++k;
            *reason = KH_CYCLE_TOL;
            break;
        }
    }
    st[0] = g1c; st[1] = g1s; st[2] = g2c; st[3] = g2s; st[4] = (double)nrot; st[5] = y0; st[6] = y1;
    *wslot_io = wslot;
    *k_done = k;
    *enq_io = enq;
    *h2_io = h2;
    ctx->n_minres_cycle_steps += k - k0;
    return 0;
}

int kh_cg_update(kh_ctx ctx, double alpha, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK,
                 int64_t ycol, kh_vec R, int64_t rcol, kh_mat Md, kh_vec Z, int64_t zcol,
                 double* rho_new) {
    KH_ARG(ctx && rho_new, "kh_cg_update: NULL");
    KH_TRY(check_vec(Pd, pcol, 1, "kh_cg_update(p)"));
    KH_TRY(check_vec(AP, apcol, 1, "kh_cg_update(Ap)"));
    KH_TRY(check_vec(YK, ycol, 1, "kh_cg_update(yk)"));
    KH_TRY(check_vec(R, rcol, 1, "kh_cg_update(r)"));
    KH_ARG(Md == nullptr || Md->kind == KH_MAT_DIAG, "kh_cg_update: Md must be diagonal");
    if (Md) KH_TRY(check_vec(Z, zcol, 1, "kh_cg_update(z)"));
    const int64_t n = R->n;
    KH_ARG(Pd->n == n && AP->n == n && YK->n == n && (!Md || (Md->n_rows == n && Z->n == n)),
           "kh_cg_update: length mismatch");
    double* part = part_slot(ctx, SLOT_NRM);
    const int grid = grid_lin(ctx, n);
    if (Md)
        hipLaunchKernelGGL((k_cg_update<true>), dim3(grid), dim3(BS), 0, ctx->stream, n, alpha,
                           Pd->col(pcol), AP->col(apcol), YK->col(ycol), R->col(rcol), Md->diag,
                           Z->col(zcol), part);
    else
        hipLaunchKernelGGL((k_cg_update<false>), dim3(grid), dim3(BS), 0, ctx->stream, n, alpha,
                           Pd->col(pcol), AP->col(apcol), YK->col(ycol), R->col(rcol), nullptr, nullptr,
                           part);
    KH_HIP(hipGetLastError());
    double* tmp = ctx->scal + SC_TMP;
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, part, grid, 0, tmp, 0);
    KH_HIP(hipGetLastError());
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
    return fetch_scalars(ctx, tmp, 1, rho_new);
}

int kh_cg_step(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK,
               int64_t ycol, kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol, int first, double omega,
               double rho, double* out) {
    KH_ARG(ctx && A && out, "kh_cg_step: NULL");
    RoctxScope range_(ctx, "kh_cg_step");
    KH_TRY(check_vec(Pd, pcol, 1, "kh_cg_step(p)"));
    KH_TRY(check_vec(AP, apcol, 1, "kh_cg_step(Ap)"));
    KH_TRY(check_vec(YK, ycol, 1, "kh_cg_step(yk)"));
    KH_TRY(check_vec(R, rcol, 1, "kh_cg_step(r)"));
    KH_ARG(A->kind <= KH_MAT_DIAG, "kh_cg_step: real operator expected");
    KH_ARG(Md == nullptr || Md->kind == KH_MAT_DIAG, "kh_cg_step: Md must be diagonal");
    if (Md) KH_TRY(check_vec(Z, zcol, 1, "kh_cg_step(z)"));
    const int64_t n = R->n;
    KH_ARG(Pd->n == n && AP->n == n && YK->n == n && A->n_rows == n && (!Md || (Md->n_rows == n && Z->n == n)),
           "kh_cg_step: length mismatch");
    KH_ARG(!(Pd == AP && pcol == apcol), "kh_cg_step: p and Ap must be different columns");
    double* p = Pd->col(pcol);
    double* ap = AP->col(apcol);
    double* r = R->col(rcol);
    double* z = Md ? Z->col(zcol) : r;
    const int grid = grid_lin(ctx, n);
    double* tmp = ctx->scal + SC_TMP;       // tmp[0] = <p, Ap>, tmp[1] = rho_new
    if (!first)                             // p = z + omega p   (linsys.py:627)
        hipLaunchKernelGGL(k_waxpby, dim3(grid), dim3(BS), 0, ctx->stream, n, p, 1.0, z, omega, p);
    if (A->kind == KH_MAT_CSR) {            // Ap = A p with <p, Ap> fused into the SpMV
        KH_TRY(apply_one(ctx, A, p, ap, EPI_DOT, p, tmp, 0));
    } else {
        KH_TRY(apply_one(ctx, A, p, ap, EPI_NONE, nullptr, nullptr, 0));
        KH_TRY(dot_panel_raw(ctx, Pd, pcol, 1, ap, tmp));
    }
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp, 1));
    double* part = part_slot(ctx, SLOT_NRM);
    if (Md)
        hipLaunchKernelGGL((k_cg_update<true>), dim3(grid), dim3(BS), 0, ctx->stream, n, rho, p, ap,
                           YK->col(ycol), r, Md->diag, z, part, tmp);
    else
        hipLaunchKernelGGL((k_cg_update<false>), dim3(grid), dim3(BS), 0, ctx->stream, n, rho, p, ap,
                           YK->col(ycol), r, nullptr, nullptr, part, tmp);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(BS), 0, ctx->stream, part, grid, 0, tmp + 1, 0);
    KH_HIP(hipGetLastError());
    if (kh_multi(ctx)) KH_TRY(comm_allreduce_dev(ctx, tmp + 1, 1));
    KH_TRY(fetch_scalars(ctx, tmp, 2, out));
    out[2] = (double)cg_sanity(out[0], out[1], rho);
    return 0;
}

// A run of CG iterations in one call (krypy/linsys.py:622-690; the header has the contract)
int kh_cg_cycle(kh_ctx ctx, kh_mat A, kh_mat Md, kh_vec Pd, int64_t pcol, kh_vec AP, int64_t apcol, kh_vec YK, int64_t ycol,
                kh_vec R, int64_t rcol, kh_vec Z, int64_t zcol, int64_t k0, int64_t k_stop, double tol, double bnorm,
                double* rhos, double* trace, int64_t* k_done, int* reason) {
    KH_ARG(ctx && A && rhos && trace && k_done && reason, "kh_cg_cycle: NULL");
    KH_ARG(k0 >= 0 && k0 <= k_stop, "kh_cg_cycle: iterations [%lld, %lld)", (long long)k0, (long long)k_stop);
    RoctxScope range_(ctx, "kh_cg_cycle");
    *reason = KH_CYCLE_LIMIT;
    int64_t k = k0;
    for (; k < k_stop; ++k) {
        const double rho = rhos[k];
        const double omega = (k > 0) ? rho / rhos[k - 1] : 0.0;          // p = z + rhos[-1] / rhos[-2] p  (linsys.py:627)
        double out[3];
        KH_TRY(kh_cg_step(ctx, A, Md, Pd, pcol, AP, apcol, YK, ycol, R, rcol, Z, zcol, k == 0, omega, rho, out));
        double* t = trace + 6 * k;
        t[0] = rho; t[1] = out[0]; t[2] = out[0]; t[3] = out[1]; t[4] = out[2]; t[5] = 0.0;
        const int flags = (int)out[2];
        if ((flags & (KH_CG_NONFINITE_PAP | KH_CG_NONFINITE_RHO | KH_CG_STEP_CLAMPED)) && std::isfinite(rho)) {
            *reason = KH_CYCLE_CHECK;            // finite data in, inf / nan out: the caller raises with the trace
            break;
        }
        // ||M Ml r_k|| in the M^-1 norm, then rho AS THE CALLER FORMS IT: `MMlrk_norm ** 2` on a NumPy scalar is libm's
        // pow(x, 2.0), which is not always the correctly rounded x * x - the call goes through a volatile pointer so that
        // the compiler does not replace it by the multiplication
        static double (*volatile libm_pow)(double, double) = &pow;
        const double nrm = std::sqrt(std::fabs(out[1]));
        t[5] = nrm;
        rhos[k + 1] = libm_pow(nrm, 2.0);
        if (!(nrm / bnorm > tol)) {              // the caller's own test, linsys.py:476: that iteration is the caller's to finalise
            ++k;
            *reason = KH_CYCLE_TOL;
            break;
        }
    }
    *k_done = k;
    ctx->n_cg_cycle_steps += k - k0;
    return 0;
}

}  // extern "C"
