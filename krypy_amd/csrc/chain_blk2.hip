// Launcher of the eight-wave blocked Gram-Schmidt kernel with the cross-rank stage (chain_blk2.h): a translation unit of
// its own.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "kh_internal.h"
#include "chain_blk2.h"

namespace kh {

static constexpr size_t BLK_GRAN_WORDS = (size_t)2 * CH_GMAX * BLK_NVS * 2;
static constexpr size_t BLK_GRAN2_WORDS = (size_t)2 * BLK_NG2 * BLK_NVS * 2;

// rows of 16 B per lane (4, 5 or 6), workgroups WITH rows and whether wave 0 is a communication wave (448 lanes with rows instead of
// 512) for a vector of n doubles: the fewest rows whose grid fits the chip - beside the workgroups without rows when there is room
// for them; with a communication wave where that fits (KRYPY_AMD_BLK2_CW: 1 = where it fits, 0 = never), else without
bool chain_blk2_shape(kh_ctx ctx, int64_t n, int* r2_out, int* g_out, int* cw_out, bool one_slot) {
    if (n < 2 || ctx->ncu > CH_GMAX / 2) return false;
    const int64_t n2 = (n + 1) >> 1;
    const int room = ctx->ncu - (ctx->blk_nx > 0 ? ctx->blk_nx : 0);
    // In order of measured speed: a communication wave and up to 6 rows per lane (no spills), the same with 7 rows (92 B of
    // scratch per lane), then all 512 lanes with rows (6 rows: 320 B of scratch).  Within each: first with room left for the
    // workgroups without rows, then without (6 rows without them beat 7 rows with them: 6,030 against 5,710 it/s at 1.36 M rows).
    // Last, when the caller takes them: 8 ... 11 rows per lane with ONE block in registers (communication wave, no spills).
    const int tiers[4][3] = {{1, 4, 6}, {1, 7, ctx->blk2_cw_maxrows}, {0, 4, 6}, {1, 8, 11}};       // {communication wave, rows from, rows to}
    const int ntier = (one_slot && ctx->blk2_cw) ? 4 : 3;
    for (int t = ctx->blk2_cw ? 0 : 2; t < ntier; ++t) {
        const int cw = tiers[t][0];
        const int nwork = cw ? CH_BS - 64 : CH_BS;
        for (int pass = 0; pass < 2; ++pass) {
            const int cap = pass == 0 ? room : ctx->ncu;
            for (int r2 = tiers[t][1]; r2 <= tiers[t][2]; ++r2) {
                const int64_t g = (n2 + (int64_t)r2 * nwork - 1) / ((int64_t)r2 * nwork);
                if (g >= 1 && g <= cap) {
                    *r2_out = r2;
                    *g_out = (int)g;
                    if (cw_out) *cw_out = cw;
                    return true;
                }
            }
        }
    }
    return false;
}

template <int R2, bool MASKED, bool XR, bool CW, bool ONE = false>
static hipError_t launch_blk2(kh_ctx ctx, int G, ChainArgs& a, BlkBufs bf, const XrDev& xr) {
    static int blocks_per_cu = -1;
    auto kern = k_mgs_chain_blk2<R2, MASKED, XR, CW, ONE>;
    if (blocks_per_cu < 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;      // the sums need every workgroup resident
    bf.nx = 0;
    if (ctx->blk_nx > 0 && (int64_t)blocks_per_cu * ctx->ncu >= G + ctx->blk_nx && G + ctx->blk_nx <= CH_GMAX / 2 && G > BLK_PFLAT * (64 / BLK_BC))
        bf.nx = ctx->blk_nx;
    if (bf.nx > 0) ctx->n_blk_rowless += 1;
    hipLaunchKernelGGL(kern, dim3(G + bf.nx), dim3(CH_BS), 0, ctx->stream, a, bf, xr);
    return hipGetLastError();
}

// One Arnoldi step k of basis block V (columns 0 .. k, reference order, one sweep; w holds A v_k) on the eight-wave blocked
// kernel: coefficients into hdev[0 .. k], the norm into hdev[k + 1], v_{k+1} stored.  multi: the sums cross the ranks through
// the xr mailboxes inside the launch (no all-reduce call in the step).  Returns 1 when launched, 0 when this shape is not
// served (one GPU only: on a communicator the caller has established eligibility for the longest slab of the run and a
// refusal here is an ERROR - a rank-local fallback would change the pattern of collectives its peers see), negative on error.
int chain_blk2_step(kh_ctx ctx, kh_vec V, const double* w, int64_t wld, int64_t k, double* hdev, int slot, double* hpin, int hcount,
                    bool multi) {
    const int64_t n = V->n;
    int r2 = 0, G = 0, cwi = 0;
    if (!ctx->chain_blk2 || !ctx->chain_configured || !chain_blk2_shape(ctx, n, &r2, &G, &cwi, ctx->blk2_one >= (multi ? 1 : 2)) ||
        k + 3 > BLK_TABCOLS ||
        ((n & 1) && (V->ld <= n || wld <= n)) || (!multi && ctx->blk2_refused_n == n)) {
        if (multi) return fail(KH_ERR_COMM, "blocked Gram-Schmidt with in-kernel cross-rank sums: a slab of %lld rows is not served "
                                            "here although the run's longest slab was found eligible", (long long)n);
        return 0;
    }
    if (multi && !ctx->xr_on) return fail(KH_ERR_COMM, "chain_blk2_step: the xr transport is off");
    KH_TRY(chain_epoch_check(ctx));
    const bool cw = cwi != 0;
    const int64_t chunk2 = (int64_t)r2 * (cw ? CH_BS - 64 : CH_BS);
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = V->ld >= need_ld && wld >= need_ld;
    BlkBufs bf;
    bf.gtab = chain_blk_table(ctx);
    if (bf.gtab == nullptr) return fail(KH_ERR_NOMEM, "chain_blk2: no memory for the Gram table");
    bf.gran = ctx->blk_gran;
    bf.gran2 = ctx->blk_gran + BLK_GRAN_WORDS;
    bf.res = ctx->blk_gran + BLK_GRAN_WORDS + BLK_GRAN2_WORDS;
    bf.nx = 0;
    // the Gram table (own-block entries: kind 2): valid when this is the next step of the sequence that owns it, otherwise
    // its rows are rebuilt from the basis.  The decision follows (blk_V, blk_next, blk_kind), i.e. the sequence of calls
    // every rank makes alike; on a communicator the panel products are summed across the ranks like any inner product.
    if (!(ctx->blk_V == V && ctx->blk_next == k && ctx->blk_kind == 2)) {
        for (int64_t j = 1; j <= k; ++j) {
            const int64_t b0 = (j / KH_BLK_BC) * KH_BLK_BC;
            if (j > b0) {
                double* row = bf.gtab + j * KH_BLK_TW + KH_BLK_BC;
                KH_TRY(dot_panel_raw(ctx, V, b0, j - b0, V->col(j), row));
                if (multi) KH_TRY(comm_allreduce_dev(ctx, row, j - b0));
            }
        }
        ctx->blk_V = V;
        ctx->blk_next = k;
        ctx->blk_kind = 2;
        ctx->n_blk_rebuild += 1;
    }
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.n2 = (n + 1) >> 1;
    a.chunk2 = chunk2;
    a.V = V->d;
    a.B = V->d;
    a.ld = V->ld;
    a.col0 = 0;
    a.ncol = (int)(k + 1);
    a.sweeps = 1;
    a.w_in = w;
    a.vnext = V->col(k + 1);
    a.hdev = hdev;
    a.hnext = k + 1;
    a.gran = ctx->chain_gran;
    a.xcc_res = ctx->chain_xcc;
    a.xcc_leader = reinterpret_cast<unsigned*>(ctx->chain_xcc + 128);
    a.epoch0 = ctx->chain_epoch;
    a.err = ctx->chain_err;
    a.debug = ctx->chain_fault ? 4 : 0;
    a.hpin = hpin;
    a.hcount = hcount;
    a.errpin = ctx->chain_err_pin[slot];
    a.donepin = (hpin != nullptr && ctx->tag_wait) ? ctx->done_pin[slot] : nullptr;
    if (a.donepin != nullptr) {
        ctx->done_counter = (ctx->done_counter == 0x7fffffff) ? 1 : ctx->done_counter + 1;
        a.done_tag = ctx->done_counter;
        ctx->done_seq[slot] = a.done_tag;
    }
    a.n_last = n - 1;
    const int nsums = (a.ncol + BLK_BC - 1) / BLK_BC + 1;
    XrDev xr;
    memset(&xr, 0, sizeof(xr));
    if (multi) {
        if (ctx->xr_epoch > 0xfff00000u - (unsigned)nsums)
            return fail(KH_ERR_COMM, "xr: the epoch counter of the cross-rank exchange is exhausted; create a new context");
        for (int r = 0; r < ctx->xr_nranks; ++r) xr.peer[r] = ctx->xr_peer[r];
        xr.rank = ctx->xr_rank;
        xr.nranks = ctx->xr_nranks;
        xr.epoch0 = ctx->xr_epoch;
        xr.timeout_ticks = (long long)(ctx->xr_timeout_ms > 0 ? ctx->xr_timeout_ms : 60000) * 100000ll;
    }
    hipError_t e;
#define KH_B2C(R, C) (multi ? (padded ? launch_blk2<R, false, true, C>(ctx, G, a, bf, xr) : launch_blk2<R, true, true, C>(ctx, G, a, bf, xr)) \
                            : (padded ? launch_blk2<R, false, false, C>(ctx, G, a, bf, xr) : launch_blk2<R, true, false, C>(ctx, G, a, bf, xr)))
#define KH_B2(R) (cw ? KH_B2C(R, true) : KH_B2C(R, false))
#define KH_B2ONE(R) (multi ? (padded ? launch_blk2<R, false, true, true, true>(ctx, G, a, bf, xr) : launch_blk2<R, true, true, true, true>(ctx, G, a, bf, xr)) \
                           : (padded ? launch_blk2<R, false, false, true, true>(ctx, G, a, bf, xr) : launch_blk2<R, true, false, true, true>(ctx, G, a, bf, xr)))
    switch (r2) {
        case 4: e = KH_B2(4); break;
        case 5: e = KH_B2(5); break;
        case 6: e = KH_B2(6); break;
        case 7: e = KH_B2C(7, true); break;
        case 8: e = KH_B2ONE(8); break;
        case 9: e = KH_B2ONE(9); break;
        case 10: e = KH_B2ONE(10); break;
        default: e = KH_B2ONE(11); break;
    }
#undef KH_B2ONE
#undef KH_B2C
#undef KH_B2
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->blk_next = -1;
        if (multi)
            return fail(KH_ERR_HIP, "blocked Gram-Schmidt with in-kernel cross-rank sums: the launch failed (%s); no rank-local "
                                    "fallback on a communicator", hipGetErrorString(e));
        ctx->blk2_refused_n = n;          // (occupancy ...: not tried - nor its table rebuilt - again for vectors of this length)
        return 0;
    }
    if (a.debug == 4) ctx->chain_fault = 0;
    ctx->blk_V = V;
    ctx->blk_next = k + 1;
    ctx->blk_kind = 2;
    ctx->chain_epoch += (unsigned)nsums;
    if (multi) {
        ctx->xr_epoch += (unsigned)nsums;
        ctx->n_xr += nsums;
    }
    ctx->n_chain += 1;
    ctx->n_chain_blk2 += 1;
    if (hpin == nullptr)
        KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ctx->wait_tag[slot] = a.donepin != nullptr;
    return 1;
}

}  // namespace kh
