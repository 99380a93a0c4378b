// The register-resident Gram-Schmidt chain (chain.h) on N ranks: every grid-wide sum of the launch also crosses the RANKS, inside
// the XCD leaders' hand-over (grid_sum<true>: one tagged store per peer mailbox and a rank-ordered gather, xr_dev.h).  A
// reference-order Arnoldi step (utils.py:1012-1034) on a slab of 2.5 M ... 14.68 M rows is then the sharded SpMV + ONE launch
// of ours: no all-reduce call, the local basis read once - where the one-reduction and the panel forms read it twice and the
// blocked kernel (chain_blk2.h) ends at 2.5 M rows.  A translation unit of its own (the instantiations with the cross-rank
// stage are compiled here; krylov_hip.hip holds the one-GPU ones).
//
// Which kernel: 16 ... 40 rows per lane the LDS-parking kernel (k_mgs_chain_lds: padded vectors, or up to 24 rows masked),
// else the plain kernel; 48 / 56 rows (the last 8 / 16 rows of w in LDS: config 5's 12.5 M-row slabs) the long-vector kernel.
// Every rank takes the shape of ITS slab - the number of sums (k + 1 links and the norm) is the same on all of them, and that
// is all the exchange protocol cares about; eligibility is decided by the caller for the LONGEST slab of the run, so that all
// ranks decide alike, and from there on a refusal is an error (a rank-local fallback would change the pattern of collectives
// its peers see).  Results: the one-GPU chain kernels' arithmetic with the partial sums of the ranks added in rank order - the
// same bits on every rank, run to run; against one GPU the sums are taken in another order (1e-10 parity, tests/test_gpu_chain_xr.py).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "kh_internal.h"
#include "chain.h"
#include "chain_long.h"

namespace kh {

// rows per lane (16 ... 56) and workgroups for a slab of n doubles: the fewest rows whose grid fits the compute units the
// shape is chosen for (all of them; tests: kh_ctx_set "chain_xr_cus").  Short slabs take 16 rows on fewer workgroups.
bool chain_xr_shape(kh_ctx ctx, int64_t n, int* r2_out, int* g_out) {
    if (n < 2) return false;
    const int cus = ctx->chain_xr_cus > 0 ? std::min(ctx->chain_xr_cus, ctx->ncu) : ctx->ncu;
    if (cus > CH_GMAX / 2) return false;              // (grid_sum's XCD-leader form: 2 G granules, one per thread)
    const int64_t n2 = (n + 1) >> 1;
    static const int kR2[] = {16, 24, 32, 40, 48, 56};
    for (int c : kR2) {
        const int64_t g = (n2 + (int64_t)c * CH_BS - 1) / ((int64_t)c * CH_BS);
        if (g <= cus) {
            *r2_out = c;
            *g_out = (int)g;
            return true;
        }
    }
    return false;
}

template <int R2, bool MASKED, int WL>
static hipError_t launch_xr_plain(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = (size_t)WL * CH_BS * sizeof(double2);
    auto kern = k_mgs_chain<R2, MASKED, false, 0, WL, false, true>;
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

static hipError_t launch_xr_long(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ChainShapeLong::LDS_BYTES;
    auto kern = k_mgs_chain_long<0, true>;
    if (blocks_per_cu < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

template <int R2, bool MASKED>
static hipError_t launch_xr_lds(kh_ctx ctx, int G, ChainArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ChainShapeLds<R2, false>::LDS_BYTES;
    auto kern = k_mgs_chain_lds<R2, MASKED, false, 0, true>;
    if (blocks_per_cu < 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(kern, dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// One Arnoldi step k of basis block V (columns 0 .. k, reference order, one sweep; w holds this rank's rows of A v_k - after
// the projector of a deflated step) with every sum across the ranks inside the launch: coefficients into hdev[0 .. k], the
// norm into hdev[k + 1], v_{k+1} stored.  Returns 1 when launched, negative on error - never 0: the caller has established
// eligibility for the longest slab of the run, every rank alike.
int chain_xr_step(kh_ctx ctx, kh_vec V, const double* w, int64_t wld, int64_t k, double* hdev, int slot, double* hpin, int hcount) {
    const int64_t n = V->n;
    int r2 = 0, G = 0;
    if (!ctx->xr_on) return fail(KH_ERR_COMM, "chain_xr_step: the xr transport is off");
    if (!chain_xr_shape(ctx, n, &r2, &G))
        return fail(KH_ERR_COMM, "Gram-Schmidt chain with in-kernel cross-rank sums: a slab of %lld rows is not served here although "
                                 "the run's longest slab was found eligible", (long long)n);
    KH_TRY(chain_epoch_check(ctx));
    const int nsums = (int)(k + 2);                       // k + 1 links and the norm
    if (ctx->xr_epoch > 0xfff00000u - (unsigned)nsums)
        return fail(KH_ERR_COMM, "xr: the epoch counter of the cross-rank exchange is exhausted; create a new context");
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    const bool padded = V->ld >= need_ld && wld >= need_ld;
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
    for (int r = 0; r < ctx->xr_nranks; ++r) a.xr.peer[r] = ctx->xr_peer[r];
    a.xr.rank = ctx->xr_rank;
    a.xr.nranks = ctx->xr_nranks;
    a.xr.epoch0 = ctx->xr_epoch;
    a.xr.timeout_ticks = (long long)(ctx->xr_timeout_ms > 0 ? ctx->xr_timeout_ms : 60000) * 100000ll;
    // the LDS-parking kernel where it exists without spilled registers: padded vectors up to 40 rows, masked ones up to 24
    const bool use_lds = ctx->chain_lds && r2 <= 40 && (padded || r2 <= 24);
    hipError_t e;
    const char* which = "";
#define KH_XL(R) (padded ? launch_xr_lds<R, false>(ctx, G, a) : launch_xr_lds<R, true>(ctx, G, a))
#define KH_XP(R, W) (padded ? launch_xr_plain<R, false, W>(ctx, G, a) : launch_xr_plain<R, true, W>(ctx, G, a))
    switch (r2) {
        case 16: e = use_lds ? KH_XL(16) : KH_XP(16, 0); break;
        case 24: e = use_lds ? KH_XL(24) : KH_XP(24, 0); break;
        case 32: e = use_lds ? launch_xr_lds<32, false>(ctx, G, a) : KH_XP(32, 0); break;
        case 40: e = use_lds ? launch_xr_lds<40, false>(ctx, G, a) : KH_XP(40, 0); break;
        case 48:
            e = (ctx->chain_long && ctx->chain_lds && padded) ? launch_xr_long(ctx, G, a) : hipErrorUnknown;
            if (e != hipSuccess) {          // (switched off, an unpadded block, or the 128 KB of dynamic LDS refused: both reads from memory)
                (void)hipGetLastError();
                e = KH_XP(48, 8);
            } else {
                ctx->n_chain_long += 1;
            }
            break;
        default: e = KH_XP(56, 16); break;
    }
#undef KH_XP
#undef KH_XL
    (void)which;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(KH_ERR_HIP, "Gram-Schmidt chain with in-kernel cross-rank sums: the launch failed (%s); no rank-local fallback on a "
                                "communicator", hipGetErrorString(e));
    }
    if (a.debug == 4) ctx->chain_fault = 0;
    ctx->chain_epoch += (unsigned)nsums;
    ctx->xr_epoch += (unsigned)nsums;
    ctx->n_xr += nsums;
    ctx->n_chain += 1;
    ctx->n_chain_xr += 1;
    ctx->n_chain_lds += use_lds ? 1 : 0;
    if (hpin == nullptr)
        KH_HIP(hipMemcpyAsync(ctx->chain_err_pin[slot], ctx->chain_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ctx->wait_tag[slot] = a.donepin != nullptr;
    return 1;
}

}  // namespace kh
