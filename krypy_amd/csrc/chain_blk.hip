// Launcher of the blocked Gram-Schmidt chain kernel (chain_blk.h): a translation unit of its own, so that work on
// this kernel family does not recompile the rest of the library.
#include <hip/hip_runtime.h>

#include "kh_internal.h"
#include "chain_blk.h"

namespace kh {

static constexpr size_t BLK_GRAN_WORDS = (size_t)2 * CH_GMAX * BLK_NVS * 2;
static constexpr size_t BLK_GRAN2_WORDS = (size_t)2 * BLK_NG2 * BLK_NVS * 2;
static constexpr size_t BLK_RES_WORDS = (size_t)16 * 2 * BLK_NVS * 2;
static constexpr size_t BLK_TAB_WORDS = (size_t)BLK_TABCOLS * BLK_TW;   // doubles: BLK_TW entries per column

// granule buffers of the blocked kernel's sums; zero = "no epoch yet" (epochs start at 1)
// (an existing buffer: the granules only - the Gram table behind them belongs to the Arnoldi sequence that is running and
// an epoch wrap in the middle of it must not zero rows that (ctx->blk_V, ctx->blk_next) still vouch for)
hipError_t chain_blk_reset(kh_ctx ctx) {
    size_t words = BLK_GRAN_WORDS + BLK_GRAN2_WORDS + BLK_RES_WORDS;
    if (ctx->blk_gran == nullptr) {
        hipError_t e = hipMalloc(&ctx->blk_gran, sizeof(unsigned long long) * (BLK_GRAN_WORDS + BLK_GRAN2_WORDS + BLK_RES_WORDS + BLK_TAB_WORDS));
        if (e != hipSuccess) {
            ctx->blk_gran = nullptr;
            return e;
        }
        words += BLK_TAB_WORDS;
        ctx->blk_V = nullptr;
        ctx->blk_next = -1;
    }
    return hipMemsetAsync(ctx->blk_gran, 0, sizeof(unsigned long long) * words, ctx->stream);
}

void chain_blk_free(kh_ctx ctx) {
    if (ctx->blk_gran != nullptr) (void)hipFree(ctx->blk_gran);
    ctx->blk_gran = nullptr;
}

template <int R2, bool MASKED, int FND, bool ONEX, int DBG = 0>
static hipError_t launch_blk(kh_ctx ctx, int G, ChainArgs& a, BlkBufs bf) {
    static int blocks_per_cu = -1;
    auto kern = k_mgs_chain_blk<R2, BLK_BC, BLK_NSLOT, MASKED, FND, ONEX, DBG>;
    if (blocks_per_cu < 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, CH_BS + 64, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * (ONEX ? ctx->ncu / 8 : ctx->ncu) < G) return hipErrorCooperativeLaunchTooLarge;
    // workgroups WITHOUT rows in front of the G with rows (chain_blk.h, BlkBufs::nx): when the chip has room for them, and
    // from as many workgroups on as the two-level exchange takes (fewer: the leaders gather everything themselves)
    bf.nx = 0;
    if (!ONEX && DBG == 0 && ctx->blk_nx > 0 && (int64_t)blocks_per_cu * ctx->ncu >= G + ctx->blk_nx && G + ctx->blk_nx <= CH_GMAX / 2)
        bf.nx = ctx->blk_nx;
    if (bf.nx > 0) ctx->n_blk_rowless += 1;
    hipLaunchKernelGGL(kern, dim3(ONEX ? 8 * G + 8 : G + bf.nx), dim3(CH_BS + 64), 0, ctx->stream, a, bf);
    return hipGetLastError();
}

// The Gram table (device): row j = <v_m, v_j> for the BLK_BC columns of the block before j's and the columns m < j of j's block.
double* chain_blk_table(kh_ctx ctx) {
    if (ctx->blk_gran == nullptr && chain_blk_reset(ctx) != hipSuccess) return nullptr;
    return reinterpret_cast<double*>(ctx->blk_gran + BLK_GRAN_WORDS + BLK_GRAN2_WORDS + BLK_RES_WORDS);
}

// can the blocked kernel take this step at all (shape only; the table is the caller's business)?
bool chain_blk_shape_ok(int r2, int G, const ChainArgs& a, int fnd) {
    return r2 == 4 && !a.presub && a.sweeps == 1 && a.col0 == 0 && (fnd == 0 || fnd == 5 || fnd == 7) && G <= CH_GMAX / 2 &&
           a.ncol > BLK_BC * (BLK_NSLOT - 1) && a.ncol + 2 <= BLK_TABCOLS;      // (at least BLK_NSLOT blocks: the kernel assumes it)
}

// One Arnoldi step k = ncol - 1 of basis block V (columns 0 .. k, one sweep) on the blocked kernel.  r2: rows of 16 B per
// lane (4), fnd: diagonals of the operator in the prologue (0: w is loaded).  *nsums = grid-wide sums of the launch (epochs
// it consumes).  hipErrorInvalidValue: no instantiation for this shape.
// The Gram table belongs to ONE Arnoldi sequence at a time: (ctx->blk_V, ctx->blk_next) name the basis block and the
// step whose launch finds rows 0 .. k of the table valid - the caller has checked that, or has rebuilt the rows
// (krylov_hip.hip: blk_table_ready).  The launch leaves row k + 1 behind.
hipError_t chain_blk_launch(kh_ctx ctx, int r2, int G, bool onex, bool padded, int fnd, ChainArgs& a, const void* V,
                            int* nsums) {
    if (!chain_blk_shape_ok(r2, G, a, fnd) || !padded) return hipErrorInvalidValue;      // (padded blocks: every vector from 4096 rows on)
    BlkBufs bf;
    bf.gtab = chain_blk_table(ctx);
    if (bf.gtab == nullptr) return hipErrorOutOfMemory;
    bf.gran = ctx->blk_gran;
    bf.gran2 = ctx->blk_gran + BLK_GRAN_WORDS;
    bf.res = ctx->blk_gran + BLK_GRAN_WORDS + BLK_GRAN2_WORDS;
    const int64_t k = a.ncol - 1;
    *nsums = (a.ncol + BLK_BC - 1) / BLK_BC + 1;
    hipError_t e;
#define KH_BLK(X)                                                 \
    (fnd == 5 ? launch_blk<4, false, 5, X>(ctx, G, a, bf)         \
              : (fnd == 7 ? launch_blk<4, false, 7, X>(ctx, G, a, bf) : launch_blk<4, false, 0, X>(ctx, G, a, bf)))
    if (a.debug >= 1 && a.debug <= 3) {        // measurement (kh_bench_kernel 21 .. 23): padded blocks, spread over the chip
        if (onex || !padded || fnd != 0) return hipErrorInvalidValue;
        e = a.debug == 1 ? launch_blk<4, false, 0, false, 1>(ctx, G, a, bf)
                         : (a.debug == 2 ? launch_blk<4, false, 0, false, 2>(ctx, G, a, bf)
                                         : launch_blk<4, false, 0, false, 3>(ctx, G, a, bf));
    } else {
        e = onex ? KH_BLK(true) : KH_BLK(false);
    }
#undef KH_BLK
    if (e == hipSuccess) {
        ctx->blk_V = V;
        ctx->blk_next = k + 1;
    } else {
        ctx->blk_next = -1;
    }
    return e;
}

}  // namespace kh
