// The translation units that krylov_hip.hip was split into (round 5) share these helpers of the Arnoldi step:
//   krylov_hip.hip   context, device blocks, operators (upload, banded copy, SpMV / SpMM launchers), inner products and updates,
//                    the Arnoldi step (chain / panel / one-reduction launchers, kh_arnoldi_step_begin / _end), the projector
//   cycles.hip       the host loops in C: kh_gmres_cycle, kh_residual, the MINRES recurrences and kh_minres_cycle, the CG step
//                    and kh_cg_cycle, the caller's drotg
//   bench_abi.hip    the measurement ABI of bench.py: kh_bench_kernel, kh_bench_arnoldi, kh_chain_trace
//   chain_blk.hip / chain_blk2.hip / proj_reg.hip / xr.hip / comm.hip   one kernel family or transport each
#pragma once
#include "kh_internal.h"
#include "chain.h"
#include "lanczos.h"

#include <cmath>

namespace kh {

void roctx_push(kh_ctx ctx, const char* fmt, long long a, long long b);
void roctx_pop(kh_ctx ctx);
struct RoctxScope {        // one range per C entry point of the hot loop
    kh_ctx ctx;
    RoctxScope(kh_ctx c, const char* fmt, long long a = 0, long long b = 0) : ctx(c) { roctx_push(c, fmt, a, b); }
    ~RoctxScope() { roctx_pop(ctx); }
};

// partial-sum slots inside ctx->part, device scalar layout inside ctx->scal
constexpr int SLOT_PING = MAXC, SLOT_PONG = MAXC + 1, SLOT_NRM = MAXC + 2;
constexpr int SC_TMP = 6144;     // scratch scalars (dot0 of the fused SpMV, norms, ...)
constexpr int SC_COEF = 6400;    // panel coefficients for axpy_panel / gemm_nn (<= 1024)
constexpr int SC_LS = 7424;      // [c | g] of the one-reduction Gram-Schmidt (2 * LS_MAXCOL)

// sanity word of a fused CG step (KH_CG_* bits of the header): the step length never visits the host, so a divisor
// that is not a positive finite number (an operator that is not positive definite - or a fault) is reported with
// the scalars; k_cg_update leaves yk and r untouched when the step length is not finite
static inline int cg_sanity(double d, double rho_new, double rho) {
    int f = 0;
    if (!std::isfinite(d)) f |= KH_CG_NONFINITE_PAP;
    else if (!(d > 0.0)) f |= KH_CG_NONPOSITIVE_PAP;
    // the device clamps a step length that is not finite to "no step" (k_cg_update): a zero (or tiny) divisor under
    // a finite rho is neither what the reference does nor an ordinary indefinite operator - the host is told
    if (std::isfinite(d) && std::isfinite(rho) && !std::isfinite(rho / d)) f |= KH_CG_STEP_CLAMPED;
    if (!std::isfinite(rho_new)) f |= KH_CG_NONFINITE_RHO;
    else if (rho_new < 0.0) f |= KH_CG_NEGATIVE_RHO;
    return f;
}

int grid_for(kh_ctx ctx, int64_t n);
int grid_lin(kh_ctx ctx, int64_t n);
double* part_slot(kh_ctx ctx, int slot);          // partial-sum slots inside ctx->part
int ensure_hcap(kh_ctx ctx, int64_t need);
int check_vec(kh_vec v, int64_t col, int64_t ncols, const char* what);
int fetch_scalars(kh_ctx ctx, const double* dev, int64_t count, double* out);
int push_scalars(kh_ctx ctx, const double* host, int64_t count, double* dev);
// y = A x for one column; epi / aux select the fused epilogue of the CSR kernels
int apply_one(kh_ctx ctx, kh_mat A, const double* x, double* y, int epi, const double* aux, double* scal_out, int rmode);
bool chain_geometry(kh_ctx ctx, int64_t n, int* r2_out, int* g_out, bool onex = false);
int try_chain(kh_ctx ctx, kh_vec V, kh_vec B, const double* w, int64_t wld, const double* dg, kh_vec P, int64_t k, int64_t start,
              int sweeps, bool presub, double h_km1, const double* h_km1_dev, double* hdev, int slot, bool cplx = false,
              double* hpin = nullptr, int hcount = 0, kh_mat Afuse = nullptr, const double* xk = nullptr,
              const MinresJob* mr = nullptr);
int try_cgs_reg(kh_ctx ctx, kh_vec V, kh_vec B, double* w, int64_t wld, const double* dg, double* mw, int64_t start, int64_t ncol,
                int sweeps, bool multi, double* hdev, double* coef, int* nrm_count, bool cplx = false);

}  // namespace kh
