// Blocked reference-order modified Gram-Schmidt for vectors whose COLUMNS fit the register file:
// ONE grid-wide sum per block of BC = 4 basis columns instead of one per column.
//
// The reference's loop (utils.py:1012-1029) is   for j: alpha_j = <v_j, w_j>;  w_{j+1} = w_j - alpha_j v_j .
// For a block of BC consecutive columns v_0 .. v_{BC-1} (link order) and w = the vector as the block finds it,
//
//     c_l      = <v_l, w>                         all BC of them against the NOT YET UPDATED w
//     G_{m,l}  = <v_m, v_l>,  m < l               the block's strict upper Gram matrix
//     alpha_l  = c_l - sum_{m<l} alpha_m G_{m,l}  ( = <v_l, w - sum_{m<l} alpha_m v_m> = the reference's alpha_l )
//     w       -= alpha_0 v_0;  w -= alpha_1 v_1; ...   (the reference's updates in its order)
//
// is the SAME recurrence in exact arithmetic: the coefficient of link l is taken against the vector the reference
// takes it against, only the inner product with the already subtracted part is formed from the Gram entries instead
// of from the updated vector.  Rounding differs by O(eps |alpha_m| |G_{m,l}|) per term, G being the basis'
// orthogonality defect - 1e-16 ... 1e-10 in a GMRES cycle -, i.e. far below the rounding of the dot products
// themselves (the low-synchronisation MGS of Swirydowicz et al. rests on the same identity).  This kernel is therefore
// NOT one of the bit-for-bit ones (it also fuses the update's multiply and subtract); it is held to north_star's 1e-10
// against the per-column kernels and the CPU oracle (tests/test_gpu_blocked.py).
//
// The Gram entries are a property of the basis: row j of a small device table holds <v_m, v_j> for the BC columns m of the block
// BEFORE j's and for the columns m < j of j's own block.  The launch of step k computes row k + 1 while it still holds the last
// two blocks and the final w (2 BC - 1 more values in the sum that carries ||w||^2; G = <v_m, w> / h) and reads the rows of
// columns <= k.  The table belongs to ONE Arnoldi sequence (ctx->blk_V, ctx->blk_next: this basis block, the next step); a step
// that is not the next one of that sequence - another basis, a block recycled / grown / written to through any other entry
// point since (chain_blk_touch) - rebuilds the rows from the basis first (two panel products per column, krylov_hip.hip), so a
// stale table cannot be used.
// (A stateless form - the block's six Gram entries computed from the resident columns in every launch, ten values per
// sum - was built first: its arithmetic, 900 instructions per block and wave, cost what the saved sums gained.)
//
// ONE BLOCK AHEAD.  The c_l of block i + 1 are taken while block i's sums are exchanged, i.e. against a w that block i has not
// updated yet; blk_alphas corrects for block i's coefficients with the entries between the two blocks exactly as it does for
// the earlier columns of the own block (the same identity, BC more terms).  A block is then
//     partial sums of block i -> LDS | barrier A | request block i + 1, its dots | barrier B | w -= block i
// and what runs between the barriers - the stream of the next block and its dots - runs UNDER the exchange.
//
// Shape: the short-vector geometry of k_mgs_chain_small (4 rows of 16 B per lane, 512 working lanes per workgroup), two blocks
// of BC whole columns in registers (the block being subtracted and the block whose dots run), and a NINTH wave per workgroup
// that owns no rows and does all the communication.  The eight working waves leave their wave partials in LDS and wait at two
// LDS-only barriers; the communication wave publishes the workgroup's partials as tagged 8-byte granules (chain.h), gathers
// everybody's - every workgroup itself when all run on one XCD (ONEX, L2 hits), otherwise group by group (the wave of
// workgroup j adds group j's 32 records) and then the XCD leaders the groups, the totals handed on through the XCD's L2 -,
// adds them in an order that depends on the workgroup numbers only (every workgroup: the same bits, run after run), forms
// the block's coefficients (blk_alphas) and leaves them in LDS.
//
// What bounds it, and the two things that made the exchange and the stream overlap (tools/blk_prof.py, blk_bench.py;
// profiles/r04_blk_*.log).  A request of the communication wave waits in the compute unit's memory queue behind whatever the
// working waves have requested.  (1) Requests issued BEFORE the sum kept the publication of the partial sums waiting for the
// whole stream - stream and exchange took turns (first form of this round: stream 4.0 + exchange 3.2 + arithmetic 2.0 us per
// block at N = 10^6).  The working waves now request the next block a moment AFTER barrier A (BLK_REQ_SLEEP), behind the
// publication.  (2) The gathers still queued behind the stream on the compute units that do them.  Spread over the chip, the
// launch therefore has EIGHT workgroups WITHOUT rows in front of the ones with rows (BlkBufs::nx): they come to the leader
// election first and they are the workgroups 0 .. 7 that add up the groups, so both levels of the gather run on compute
// units that carry no stream; a workgroup with rows only publishes (early) and polls its XCD's L2 for the totals (late,
// behind its stream - which the dots need anyway).  N = 10^6: 7.5k -> 8.1k it/s of GMRES(100), 2.5 * 10^5: 11.4k -> 12.0k,
// 10^5: 12.7k -> 13.9k (now spread over the chip: ctx->blk_onex_maxn), 10^4: 16.9k -> 17.7k; either measure alone gains
// nothing (7.1k-7.6k at 10^6).
#pragma once
#include "chain.h"

namespace kh {

#ifndef KH_BLK_BC_CFG
#define KH_BLK_BC_CFG 4                       // (kh_internal.h has the same default: the Gram table's row length)
#endif
#ifndef KH_BLK_NSLOT_CFG
#define KH_BLK_NSLOT_CFG 2
#endif
constexpr int BLK_BC = KH_BLK_BC_CFG;         // basis columns per block
constexpr int BLK_NSLOT = KH_BLK_NSLOT_CFG;   // blocks of columns in registers
constexpr int BLK_NVMAX = 8;                  // values per sum (the block's BC coefficients; the norm + BC - 1 table entries)
constexpr int BLK_NVS = 16;                   // granule PAIRS reserved per workgroup and parity (256 B records)
constexpr int BLK_TABCOLS = KH_BLK_TABCOLS;   // basis columns the Gram table has rows for (BLK_BC entries each); the eligibility tests
                                              // of krylov_hip.hip use the same constant (a literal 4096 stood there)
#ifndef BLK_GS_CFG
#define BLK_GS_CFG 32
#endif
constexpr int BLK_GS = BLK_GS_CFG;            // workgroups per group of the two-level exchange (32: the groups of a full chip are
                                              // added up by the first EIGHT workgroups - the ones without rows, see BlkBufs::nx)
constexpr int BLK_NG2 = CH_GMAX / BLK_GS;     // groups at most

#ifndef BLK_REQ_SLEEP
#define BLK_REQ_SLEEP 8                       // s_sleep units (64 clocks) between barrier A and the working waves' requests of the
#endif                                        // next block: the publication of the partial sums goes out first (spread over the chip)
#ifndef BLK_REQ_SLEEP_ONEX
#define BLK_REQ_SLEEP_ONEX 16                 // ... on one XCD
#endif
constexpr int BLK_TW = 2 * BLK_BC;            // entries per row of the Gram table: BC of the previous block, < BC of the own
template <int BC>
struct BlkShape {
    static constexpr int NG = BC * (BC - 1) / 2;      // strict upper Gram entries of a block
    static constexpr int NX = BC * BC;                // entries between a block and the block before it
    static constexpr int NT = BC + NG + NX;           // what the working waves read from LDS per block: coefficients + entries
    static_assert(2 * BC <= BLK_NVMAX && NT <= 64, "one communication wave handles a block's values");
};

struct BlkBufs {
    unsigned long long* gran;     // [2 parities][CH_GMAX][BLK_NVS][2] granules: the workgroups' partial sums
    unsigned long long* gran2;    // [2 parities][BLK_NG2][BLK_NVS][2] granules: the groups' partial sums (write-through)
    unsigned long long* res;      // [16 XCDs][2 parities][BLK_NVS][2] the totals, handed on inside an XCD through its L2
    double* gtab;                 // [BLK_TABCOLS][BLK_TW] Gram table of the current Arnoldi sequence: row j = <v_m, v_j> for the BC
                                  // columns m of the block before j's (entries 0 .. BC-1) and the columns m < j of j's block (BC ..)
    int nx;                       // the first nx workgroups own NO rows: their compute units carry no column stream, so the sums
                                  // they gather (the groups', the XCD leaders') do not queue behind one (0: everybody has rows)
};

// ---- the communication wave's side of a sum -------------------------------------------------------------------
// Gather the NV values of `count` records (BLK_NVS granule pairs each, tagged with this epoch) starting at `rec` and add
// them in a fixed order: lane = q * NV + v takes value v of the records p * NGRP + q, p = 0, 1, ... (ascending), the NGRP
// lane groups meet in LDS and lane v adds them in ascending q.  All loads of a sweep are issued back to back; a sweep
// that finds a tag of an older epoch is repeated as a whole (guide, Guideline 16 R2).  count <= PMAX * NGRP.
template <int NV, int PMAX>
__device__ __forceinline__ double blk_gather(const unsigned long long* rec, unsigned epoch, int count, int* err,
                                             double* smq, bool l2) {
    constexpr int NGRP = 64 / NV;
    const int lane = threadIdx.x & 63;
    const int q = lane / NV, v = lane - q * NV;
    const bool mine = lane < NGRP * NV;
    const unsigned long long* e0 = rec + ((unsigned)q * BLK_NVS + (unsigned)v) * 2u;
    unsigned lo[PMAX], hi[PMAX];
    unsigned spins = 0;
    while (true) {
        bool ok = true;
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            lo[p] = 0u;
            hi[p] = 0u;
            if (mine && p * NGRP + q < count) {
                const unsigned long long* e = e0 + (unsigned)(p * NGRP * BLK_NVS * 2);
                const unsigned long long x0 = ld_agent(e), x1 = ld_agent(e + 1);
                lo[p] = (unsigned)x0;
                hi[p] = (unsigned)x1;
                ok = ok && (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
            }
        }
        if (__all(ok)) break;
        if ((++spins & 255u) == 0) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > (1u << 20)) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (!l2) __builtin_amdgcn_s_sleep(1);
    }
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        const double d = __longlong_as_double((long long)(((unsigned long long)hi[p] << 32) | lo[p]));
        acc += (mine && p * NGRP + q < count) ? d : 0.0;
    }
    if constexpr (NV == 1) {
        return wave_sum_dpp(acc);
    } else {
        if (mine) smq[q * NV + v] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its own LDS writes are in order)
        double t = 0.0;
        if (lane < NV) {
            t = smq[lane];
#pragma unroll
            for (int qq = 1; qq < NGRP; ++qq) t += smq[qq * NV + lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return t;               // lanes 0 .. NV-1: the total of value `lane`
    }
}

// granule pair {epoch | low word}, {epoch | high word} of one double
__device__ __forceinline__ void blk_put(unsigned long long* e, unsigned epoch, double x, bool plain) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned long long tag = (unsigned long long)epoch << 32;
    if (plain) {                                       // plain stores stay in this XCD's L2
        volatile unsigned long long* ev = e;
        ev[0] = tag | (bits & 0xffffffffull);
        ev[1] = tag | (bits >> 32);
    } else {                                           // write-through: visible on every XCD
        st_agent(e, tag | (bits & 0xffffffffull));
        st_agent(e + 1, tag | (bits >> 32));
    }
}

// LDS of the sums: wave partials [NV][8], totals [NV], lane groups of the gather [64]
struct BlkSm {
    double part[BLK_NVMAX * (CH_BS / 64)];
    double tot[BlkShape<BLK_BC>::NT];
    double q[64];
    double al[BLK_BC];            // the block's coefficients, formed by the communication wave for everybody
};

// One grid-wide sum of NV values, communication wave.  Returns (lanes 0 .. NV-1) the totals, which are also in
// sm.tot for the working waves once the second barrier has been passed.
//   ONEX        every workgroup runs on the one XCD: partials published with plain stores, EVERY communication wave
//               gathers all G <= 32 records from the L2
//   otherwise   partials published write-through; up to FLAT workgroups: the XCD leaders gather all records over the
//               fabric; more: the communication wave of workgroup j first adds the records of group j (BLK_GS
//               workgroups) and publishes the group's sums, the leaders gather the <= 16 group records - two short
//               sweeps (<= 3 loads of 16 B per lane each) instead of one of 40 per lane.  The leader leaves the totals
//               in its XCD's L2 (plain stores), everybody else polls them there.
// The order of the additions is a function of the workgroup numbers alone: every workgroup gets the same bits, run
// after run, wherever the workgroups were placed.
template <int NV, bool ONEX>
__device__ __forceinline__ double blk_sum_comm(unsigned epoch, const BlkBufs& bf, int G, int bid, int rid, int* err, BlkSm& sm,
                                               const GridRole role, bool skip = false, int nv_all = NV,
                                               double extra = 0.0, bool barrier_b = true) {
    // bid: this workgroup's number (which group it adds up); rid: the record its partial sums go to - the workgroups with rows
    // first, in their order, the ones without rows (zeros) behind them: the additions meet the same values in the same
    // order whether the launch has workgroups without rows or not (and as on one XCD)
    // (lanes NV .. nv_all-1 put `extra` - the block's Gram entries from the table - behind the totals)
    constexpr int NW = CH_BS / 64;
    constexpr int NGRP = 64 / NV;
#ifndef BLK_PFLAT
#define BLK_PFLAT 6
#endif
    constexpr int PFLAT = (NV == 1) ? (CH_GMAX / 2 + 63) / 64 : BLK_PFLAT;        // passes of a flat sweep
    constexpr int P2 = (BLK_GS + NGRP - 1) / NGRP;                       // passes over one group / over the groups
    const int lane = threadIdx.x & 63;
    ch_lds_barrier();                                  // A: the working waves' partials are in LDS
    unsigned long long* slot = bf.gran + (size_t)(epoch & 1u) * ((size_t)CH_GMAX * BLK_NVS * 2);
    if (lane < NV && !skip) {
        double s = sm.part[lane * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += sm.part[lane * NW + i];
        blk_put(slot + ((size_t)rid * BLK_NVS + lane) * 2, epoch, s, ONEX);
    }
    double t;
    if (skip) {                                        // measurement: no exchange, the workgroup's own partial sums
        t = 0.0;
        if (lane < NV) {
            t = sm.part[lane * NW];
#pragma unroll
            for (int i = 1; i < NW; ++i) t += sm.part[lane * NW + i];
        }
    } else if constexpr (ONEX) {
        t = blk_gather<NV, PFLAT>(slot, epoch, G, err, sm.q, true);
    } else {
        const bool flat = G <= PFLAT * NGRP;
        unsigned long long* slot2 = bf.gran2 + (size_t)(epoch & 1u) * ((size_t)BLK_NG2 * BLK_NVS * 2);
        const int ng2 = (G + BLK_GS - 1) / BLK_GS;
        if (!flat && bid < ng2) {                      // this workgroup adds up group `bid`
            const int cnt = (G - bid * BLK_GS) < BLK_GS ? (G - bid * BLK_GS) : BLK_GS;
            const double g = blk_gather<NV, P2>(slot + (size_t)bid * BLK_GS * BLK_NVS * 2, epoch, cnt, err, sm.q, false);
            if (lane < NV) blk_put(slot2 + ((size_t)bid * BLK_NVS + lane) * 2, epoch, g, false);
        }
        unsigned long long* res = bf.res + ((size_t)role.xcc * 2 + (epoch & 1u)) * (BLK_NVS * 2);
        if (role.leader) {
            t = flat ? blk_gather<NV, PFLAT>(slot, epoch, G, err, sm.q, false)
                     : blk_gather<NV, P2>(slot2, epoch, ng2, err, sm.q, false);
            if (lane < NV) blk_put(res + 2 * lane, epoch, t, true);
        } else {
            unsigned long long x0 = 0, x1 = 0;
            unsigned spins = 0;
            while (true) {
                bool ok = true;
                if (lane < NV) {
                    x0 = ld_agent(res + 2 * lane);
                    x1 = ld_agent(res + 2 * lane + 1);
                    ok = (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
                }
                if (__all(ok)) break;
                if ((++spins & 1023u) == 0) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1u << 22)) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            t = __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
        }
    }
    if (lane < NV) sm.tot[lane] = t;
    else if (lane < nv_all) sm.tot[lane] = extra;
    if (barrier_b) ch_lds_barrier();                   // B: the totals are in LDS
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the caller reads them first: one wave, its LDS writes are in order)
    return t;
}

// ... and the working waves' side: wave partials into LDS, two LDS-only barriers, the totals are in sm.tot
template <int NV>
__device__ __forceinline__ void blk_sum_work(const double (&val)[NV], BlkSm& sm) {
    constexpr int NW = CH_BS / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const double ws = wave_sum_dpp(val[v]);
        if (lane == 0) sm.part[v * NW + wid] = ws;
    }
    ch_lds_barrier();          // A
    ch_lds_barrier();          // B
}

// alpha_l = c_l - sum_{m<l} alpha_m G_{m,l} (fixed order; tot = [c_0 .. c_{BC-1}, G_01, G_02, G_12, G_03, ...]:
// the Gram entries of column l sit behind those of column l - 1)
// ONE BLOCK AHEAD (round 4, second form): the c_l of a block are taken against the w that the block BEFORE it has not yet
// updated either - the dots of block i + 1 run while block i's sums are exchanged -, so the coefficients of the previous block
// (aprev) are corrected for as well:  alpha_l = c_l - sum_{m in previous block} aprev_m X_{m,l} - sum_{m<l} alpha_m G_{m,l};
// tot = [c_0 .. c_{BC-1} | G_01, G_02, G_12, ... | X_{0,0} .. X_{BC-1,0}, X_{0,1} ...] (X_{m,l}: entry m of column l's table row).
// The order of the subtractions is that of the reference's loop: earlier columns first.
template <int BC>
__device__ __forceinline__ void blk_alphas(const double* tot, int nvalid, const double (&aprev)[BC], double (&alpha)[BC]) {
    constexpr int NG = BlkShape<BC>::NG;
    int gi = BC;
#pragma unroll
    for (int l = 0; l < BC; ++l) {
        double a = tot[l];
#pragma unroll
        for (int m = 0; m < BC; ++m) {
            const double pr = aprev[m] * tot[BC + NG + l * BC + m];
            a = a - pr;
        }
#pragma unroll
        for (int m = 0; m < l; ++m) {
            const double pr = alpha[m] * tot[gi + m];
            a = a - pr;
        }
        gi += l;
        alpha[l] = (l < nvalid) ? a : 0.0;
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void blk_unroll(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        blk_unroll<N, I + 1>(f);
    }
}

template <int R2, int BC, int NSLOT, bool MASKED, int FND, bool ONEX, int DBG = 0>
__global__ __launch_bounds__(CH_BS + 64) void k_mgs_chain_blk(ChainArgs a, BlkBufs bf) {
    // DBG: the measurement instantiations (tools/blk_prof.py): 1 = no exchange between workgroups, 2 = no column stream,
    // 3 = both; the product's kernels (DBG = 0) know a.debug = 4 only (tests: a faked timeout)
    constexpr bool dbg_noex = (DBG & 1) != 0;
    constexpr bool dbg_nost = (DBG & 2) != 0;
    static_assert(FND == 0 || !MASKED, "the fused operator exists for padded blocks");
    static_assert(BC == BLK_BC, "the Gram table has BLK_BC entries per column");
    constexpr int NT = BlkShape<BC>::NT;
    __shared__ BlkSm sm;
    __shared__ int slead;
    const int tid = threadIdx.x;
    int G = gridDim.x;
    int bid = blockIdx.x;
    GridRole role;
    int nx = 0;
    if constexpr (ONEX) {
        if (!onex_enter(a, &slead, bid, G, role)) return;
    } else {
        nx = bf.nx;
        // (the workgroups with rows come to the leader election late: the first arrival of an XCD leads it, and the
        // leaders should be workgroups without a column stream.  Who leads changes nothing in the sums' bits.)
        if (nx > 0 && bid >= nx) __builtin_amdgcn_s_sleep(48);
        role = grid_role(a.xcc_leader, a.epoch0, &slead);
    }
    const bool rowless = bid < nx;
    const int rid = rowless ? G - nx + bid : bid - nx;      // its record in the exchange (blk_sum_comm)
    unsigned epoch = a.epoch0;
    const int total = a.ncol;                               // links (one sweep, columns 0 .. ncol-1: blocks are aligned)
    const int nblk = (total + BC - 1) / BC;
    // (chain_blk_shape_ok: at least NSLOT blocks.  Told to the compiler: for the path around the steady-state loop it
    // parks the whole ring in scratch memory in front of the loop.)
    __builtin_assume(total > BC * (NSLOT - 1));
    __builtin_assume(nblk >= NSLOT);
    if (tid >= CH_BS) {
        // ---- the communication wave: the sums, the H entries and the Gram table; no rows ----
        const int lane = tid - CH_BS;
        const bool writer = bid == 0 && lane == 0;
        // lane BC + e holds Gram entry e of a block: (l, m) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2) ...
        int gl = 1, gm = 0;
        {
            int e = lane - BC;
            for (int l = 1; l < BC; ++l)
                if (e >= 0 && e < l) { gl = l; gm = e; e = -1; } else if (e >= 0) e -= l;
        }
        double aprev[BC];
#pragma unroll
        for (int l = 0; l < BC; ++l) aprev[l] = 0.0;
        constexpr int NG_ = BlkShape<BC>::NG;
        for (int ib = 0; ib < nblk; ++ib) {
            double gval = 0.0;                              // requested before the sum: it arrives under it
            if (lane >= BC && lane < BC + NG_ && ib * BC + gl < total)
                gval = __hip_atomic_load(bf.gtab + (size_t)(ib * BC + gl) * BLK_TW + BC + gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane >= BC + NG_ && lane < NT && ib > 0) {      // X_{m,l}: lane BC + NG + l * BC + m
                const int x = lane - (BC + NG_), l = x / BC, m = x - l * BC;
                if (ib * BC + l < total)
                    gval = __hip_atomic_load(bf.gtab + (size_t)(ib * BC + l) * BLK_TW + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            (void)blk_sum_comm<BC, ONEX>(epoch++, bf, G, bid, rid, a.err, sm, role, dbg_noex, NT, gval, false);
            {
                double alpha[BC];
                const int nvalid = total - ib * BC;
                blk_alphas<BC>(sm.tot, nvalid, aprev, alpha);
#pragma unroll
                for (int l = 0; l < BC; ++l) {
                    const int j = ib * BC + l;
                    const double al = (a.debug == 4) ? alpha[l] * 0.5 : alpha[l];      // ... a faked timeout leaves garbage behind
                    if (writer && j < total) a.hdev[a.col0 + j] = al;                  // (what the working waves subtract)
                    if (lane == 0) sm.al[l] = al;
                    aprev[l] = alpha[l];
                }
            }
            ch_lds_barrier();                                     // B: the block's coefficients are in LDS
        }
        // the norm, and <v_m, w> for the columns of the new column's row of the table: lanes 1 .. BC the block before the new
        // column's, lanes BC + 1 .. 2 BC - 1 the earlier columns of its own block
        const double tf = blk_sum_comm<2 * BC, ONEX>(epoch++, bf, G, bid, rid, a.err, sm, role, dbg_noex);
        const double h = sqrt(fabs(sm.tot[0]));
        if (writer) a.hdev[a.hnext] = h;
        {
            const int pnew = total % BC, bnew = total / BC;      // the new column: position in its block, its block
            const int e = lane - 1;                              // table entry this lane's value belongs to
            const bool ok = (e >= 0 && e < BC) ? (bnew >= 1) : (e >= BC && e < BC + pnew);
            if (bid == 0 && ok && e < BLK_TW) bf.gtab[(size_t)total * BLK_TW + e] = tf / h;
        }
        if (bid == 0 && a.hpin != nullptr) {
            __threadfence();                                      // the H entries, for the working waves' copy below
            __syncthreads();
            if (a.donepin != nullptr) __syncthreads();            // (... and the barrier of CH_SIGNAL_DONE)
        }
        return;
    }
    // bid == 0 hands the H entries to the host when the chain is done (after the communication wave has written them)
    auto finish = [&]() __attribute__((always_inline)) {
        if (bid == 0 && a.hpin != nullptr) {
            __syncthreads();          // the H entries were written by the communication wave of this workgroup
            for (int i = tid; i < a.hcount; i += CH_BS)
                a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            CH_SIGNAL_DONE(a);
        }
    };
    if (rowless) {
        // no rows: partial sums of zero, and the barriers of the nblk + 1 sums
        for (int i = tid; i < BLK_NVMAX * (CH_BS / 64); i += CH_BS) sm.part[i] = 0.0;
        for (int s_ = 0; s_ <= nblk; ++s_) {
            ch_lds_barrier();
            ch_lds_barrier();
        }
        finish();
        return;
    }
    const int64_t first = (int64_t)(bid - nx) * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    // (the column's address is a wave-uniform base + this lane's 32-bit byte offset)
#define CH_COL(j) (reinterpret_cast<const char*>(a.V + (a.col0 + ((j) < total ? (j) : total - 1)) * a.ld))
#define CH_ROW(c, r) (*reinterpret_cast<const double2*>((c) + (size_t)(r) * (CH_BS * sizeof(double2)) + boff))
    const unsigned boff = (unsigned)first * (unsigned)sizeof(double2);
    double2 ring[NSLOT][BC][R2];
    double2 w[R2];
    // w first, then the first NSLOT blocks.  The ORDER of the requests matters to the compiler's wait counts (vector-memory
    // results return in order, a wait names how many of the newest requests may still be outstanding): with w requested
    // last every use of w in the loop would wait for everything requested before it - all column blocks in flight.
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { w[r] = make_double2(s0, s1); });
        // w is COMPUTED before the first column is requested: the fence below orders memory operations only, and the
        // compiler otherwise sinks the operator's arithmetic behind the columns' loads - the operator's 160 registers of
        // loaded values live next to the ring's 128
#pragma unroll
        for (int r = 0; r < R2; ++r) asm volatile("" : "+v"(w[r].x), "+v"(w[r].y) : : "memory");
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);
            w[r].x = CH_OK(r) ? v.x : 0.0;
            w[r].y = CH_OK(r) ? v.y : 0.0;
        }
    }
    CH_ISSUE_FENCE();
    // block 0 (block 1 is requested inside block 0's sum, like every other block; the measurement form without the column
    // stream fills both slots here)
#pragma unroll
    for (int s = 0; s < (dbg_nost ? NSLOT : 1); ++s) {
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            const char* __restrict__ c = CH_COL(s * BC + l);
#pragma unroll
            for (int r = 0; r < R2; ++r) ring[s][l][r] = CH_ROW(c, r);
        }
    }
    CH_ISSUE_FENCE();
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    static_assert(NSLOT == 2, "one block ahead: the block being updated and the block whose dots run under its sums");
    // the coefficients of the block in slot S against w as it is now
    auto dots = [&](const int i, auto slot_c, double (&val)[BC]) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        const int nvalid = total - i * BC;
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                double2 v = ring[S][l][r];
                if (MASKED && !CH_OK(r)) v = make_double2(0.0, 0.0);
                ring[S][l][r] = v;
                acc0 = fma(v.x, w[r].x, acc0);
                acc1 = fma(v.y, w[r].y, acc1);
            }
            val[l] = (l < nvalid) ? acc0 + acc1 : 0.0;
        }
    };
    double val[BC];
    dots(0, std::integral_constant<int, 0>{}, val);
    // one block: its sums go out (val: taken BEFORE the previous block's update, see blk_alphas), the NEXT block's dots run
    // while they are exchanged, then the update and the request of the block NSLOT blocks ahead into the freed slot S
    auto block = [&](const int i, auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        const int nvalid = total - i * BC;           // links of this block (>= BC except in the last one)
        {
            constexpr int NW = CH_BS / 64;
            const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
            for (int v = 0; v < BC; ++v) {
                const double ws = wave_sum_dpp(val[v]);
                if (lane == 0) sm.part[v * NW + wid] = ws;
            }
        }
        ch_lds_barrier();          // A: the partial sums are in LDS, the communication wave takes over
        // The NEXT block is requested now, into the slot the previous block's update has freed - BEHIND the communication
        // wave's publication of the partial sums in the compute unit's memory queue, not in front of it (a request issued
        // before the sum keeps the publication waiting for the whole stream: the sum and the stream took turns).  At the end
        // of the chain: the block before the last once more (the new column's row of the Gram table wants both).
        if constexpr (!dbg_nost) {
            if constexpr ((ONEX ? BLK_REQ_SLEEP_ONEX : BLK_REQ_SLEEP) > 0)
                __builtin_amdgcn_s_sleep(ONEX ? BLK_REQ_SLEEP_ONEX : BLK_REQ_SLEEP);
            const int nxt = (i + 1 < nblk) ? i + 1 : i - 1;
#pragma unroll
            for (int l = 0; l < BC; ++l) {
                const char* __restrict__ c = CH_COL(nxt * BC + l);
#pragma unroll
                for (int r = 0; r < R2; ++r) ring[1 - S][l][r] = CH_ROW(c, r);
            }
        }
        CH_ISSUE_FENCE();
        double valn[BC];
        dots(i + 1, std::integral_constant<int, 1 - S>{}, valn);      // (behind the last block: the re-requested block, discarded)
        // (the barriers order memory operations only: without this the compiler moves the dots behind barrier B)
#pragma unroll
        for (int l = 0; l < BC; ++l) asm volatile("" : "+v"(valn[l]) : : "memory");
        ch_lds_barrier();          // B: the block's coefficients (formed by the communication wave, blk_alphas) are in LDS
        double alpha[BC];
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            alpha[l] = sm.al[l];
            val[l] = valn[l];
        }
        (void)nvalid;
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            const double al = alpha[l];
#pragma unroll
            for (int r = 0; r < R2; ++r) {          // (one rounding per entry: this kernel is not the bit-for-bit one)
                w[r].x = CH_OK(r) ? fma(-al, ring[S][l][r].x, w[r].x) : 0.0;
                w[r].y = CH_OK(r) ? fma(-al, ring[S][l][r].y, w[r].y) : 0.0;
            }
        }
        // w is COMPUTED here: otherwise the next requests into this slot are scheduled in front of the update and the
        // slot's old values go to scratch memory
#pragma unroll
        for (int r = 0; r < R2; ++r) asm volatile("" : "+v"(w[r].x), "+v"(w[r].y) : : "memory");
    };
    // The steady-state loop has NO conditional inside: a block that may or may not request columns leaves the compiler's
    // wait counts at "everything" on the back edge (s_waitcnt vmcnt(0) in front of every block: no column would ever be
    // in flight across a sum).  The last nblk % NSLOT blocks run behind the loop.
    int ib = 0;
    for (; ib + NSLOT <= nblk; ib += NSLOT) {
        blk_unroll<NSLOT>([&](auto sc) { block(ib + decltype(sc)::value, sc); });
    }
    blk_unroll<NSLOT - 1>([&](auto sc) {
        if (ib + decltype(sc)::value < nblk) block(ib + decltype(sc)::value, sc);
    });
#undef CH_ROW
#undef CH_COL
    double nv[2 * BC];
    {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            acc = fma(w[r].x, w[r].x, acc);
            acc = fma(w[r].y, w[r].y, acc);
        }
        nv[0] = acc;
        // <v_m, w> for the new column's row of the table: the block before its own (BC columns) and the earlier columns of its
        // own block.  Both are among the last two blocks of the chain, which were requested again and sit in the ring: the last
        // block in slot (nblk - 1) % 2.  New column total = bnew * BC + pnew: pnew > 0 - own block = last block, the block before
        // it in the other slot; pnew == 0 - no earlier column of its own block, the block before it = last block.
        const int sl = (nblk - 1) % NSLOT;
        const int pnew = total % BC, bnew = total / BC;
        const int sprev = (pnew > 0) ? 1 - sl : sl;
#pragma unroll
        for (int m = 0; m < 2 * BC - 1; ++m) {
            const int col = (m < BC) ? m : m - BC;                // column within its block
            const int sm_ = (m < BC) ? sprev : sl;                // the slot it sits in
            // (both slots' sums, the wanted one selected afterwards: a select between two elements of the ring becomes an
            // indexed access and sends the whole ring to scratch memory)
            double s0a = 0.0, s0b = 0.0, s1a = 0.0, s1b = 0.0;
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                double2 v0 = ring[0][col][r], v1 = ring[1][col][r];
                if (MASKED && !CH_OK(r)) v0 = v1 = make_double2(0.0, 0.0);
                s0a = fma(v0.x, w[r].x, s0a);
                s0b = fma(v0.y, w[r].y, s0b);
                s1a = fma(v1.x, w[r].x, s1a);
                s1b = fma(v1.y, w[r].y, s1b);
            }
            const bool ok = (m < BC) ? (bnew >= 1) : (m - BC < pnew);
            const double sum = (sm_ == 1) ? s1a + s1b : s0a + s0b;
            nv[1 + m] = ok ? sum : 0.0;
        }
    }
    blk_sum_work<2 * BC>(nv, sm);
    const double h = sqrt(fabs(sm.tot[0]));
    double2* __restrict__ vn2 = reinterpret_cast<double2*>(a.vnext) + first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        if (r * CH_BS < rem) {
            double2 o;
            o.x = w[r].x / h;
            o.y = w[r].y / h;
            st_nt2(vn2 + (int64_t)r * CH_BS, o);
        }
    }
    finish();
#undef CH_OK
}

}  // namespace kh
