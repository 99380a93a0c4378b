// Blocked reference-order Gram-Schmidt for vectors of 4 ... 6 rows of 16 B per lane (up to ~1.5 M rows per GPU), EIGHT waves,
// with the sums crossing the RANKS of a node inside the launch: ONE read of the local basis per Arnoldi step on N ranks.
//
// The recurrence is chain_blk.h's (reference loop: /root/reference/krypy/utils.py:1012-1029): for a block of BC = 4
// consecutive columns, c_l = <v_l, w> against the w the block finds, alpha_l = c_l - sum_{m<l} alpha_m G_{m,l} with G the
// block's strict upper Gram entries from the table the Arnoldi sequence carries, then the four updates in the reference's
// order.  ONE grid-wide sum per block - and, on N ranks, one exchange between the ranks per block.
//
// What is different from k_mgs_chain_blk (chain_blk.h), and why it is a kernel of its own:
//   * No NINTH wave: nine waves leave 168 registers per lane, i.e. two blocks of 4 rows per lane and no more - 1,048,576
//     rows - and one of eight ranks holds 1.25 M rows of the benchmark problem.  Eight waves have 256 registers: two blocks of
//     up to 6 rows (216) + w (24).  Where the shape fits (CW: up to 1.33 M rows) wave 0 of the eight is the communication
//     wave and owns NO rows (448 lanes with rows): the working waves' code then holds nothing of the gathering (239 registers,
//     no scratch at 6 rows - 256 + 320 B of scratch when wave 0 works as well) and wave 0's polls return as soon as the total
//     is there instead of behind its own rows of the next block.  A / B on one box (profiles/r05_blk2_cw.log): 1.25 M rows
//     6,600 -> 6,990 it/s on one GPU, 5,380 -> 6,070 through the forced multi-rank path, 1.32 M rows 5,280 -> 6,740.
//     Beyond that (up to 1.57 M rows) every wave carries rows.
//   * No dots "one block ahead" against the un-updated w (no X entries of the table): the next block is REQUESTED
//     behind the publication of the partial sums - its stream runs under the exchange - but its dots are taken after the
//     update, as the reference takes them.  Wave 0 of every workgroup is the one that publishes, gathers and forms the
//     coefficients (its polls return behind its own requests, which the dots wait for anyway); workgroups WITHOUT rows in
//     front of the grid (BlkBufs::nx) do the group sums and lead the XCDs, as for the other kernel.
//   * The CROSS-RANK stage (XrDev, xr_dev.h): an XCD leader that holds the rank's total of a sum stores it as tagged
//     granules into every peer's IPC-mapped mailbox (system scope: xGMI writes; all leaders of a rank write the same bits to
//     the same slots - idempotent, no election, no extra hop), polls its own mailbox for the N contributions, adds them in
//     rank order and hands the total on through its XCD's L2 like any other total.  With nranks = 1 (loopback) the same
//     code runs against the rank's own mailbox.
// The Gram table of this kernel holds the own-block entries only (row j: <v_m, v_j> for the columns m < j of j's block, at
// [BLK_BC + m]); the launch of step k leaves row k + 1 behind (the last block is still in registers); ctx->blk_kind tells
// the two kernels' tables apart.  Like k_mgs_chain_blk this is NOT a bit-for-bit kernel (fused multiply-subtract, table
// correction): held to 1e-10 against the per-column kernels and the oracle (tests/test_gpu_blk2.py).
#pragma once
#include "chain_blk.h"
#include "xr_dev.h"

namespace kh {

#ifndef BLK2_REQ_SLEEP
#define BLK2_REQ_SLEEP 8          // s_sleep units (64 clocks) between barrier A and the other waves' requests of the next block
#endif

// the cross-rank stage of a sum of NV values: lanes 0 .. NV-1 of the calling wave hold the rank's totals
template <int NV>
__device__ __forceinline__ double blk2_cross_rank(const XrDev& xr, unsigned xepoch, double t, int* err) {
    const int lane = threadIdx.x & 63;
    if (lane < NV) {
        xr_put_all(xr.peer, xr.rank, xr.nranks, xepoch, lane, t);
        int bad = 0;
        t = xr_take_all(xr.peer[xr.rank], xr.nranks, xepoch, lane, xr.timeout_ticks, &bad);
        if (bad) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (2: a peer rank, not a workgroup, is missing)
    }
    return t;
}

// One grid-wide (and cross-rank) sum of NV values in two halves, both called by wave 0 of every workgroup; the other waves
// have left their wave partials in sm.part and wait at the two LDS-only barriers.  blk2_publish: barrier A, the
// workgroup's partial sums go out.  Between the two the caller issues wave 0's own requests of the next block - BEHIND the
// publication.  blk2_collect: group sums, XCD leaders, the cross-rank stage, the hand-over through the XCD's L2; leaves the
// totals in sm.tot[0 .. NV-1] (and `extra` of lanes NV .. nv_all-1 behind them); the caller passes barrier B when it has
// formed what the other waves need.
template <int NV>
__device__ __forceinline__ void blk2_publish(unsigned epoch, const BlkBufs& bf, int rid, BlkSm& sm) {
    constexpr int NW = CH_BS / 64;
    const int lane = threadIdx.x & 63;
    ch_lds_barrier();                                  // A: every wave's partial sums are in LDS
    unsigned long long* slot = bf.gran + (size_t)(epoch & 1u) * ((size_t)CH_GMAX * BLK_NVS * 2);
    if (lane < NV) {
        double s = sm.part[lane * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += sm.part[lane * NW + i];
        blk_put(slot + ((size_t)rid * BLK_NVS + lane) * 2, epoch, s, false);
    }
}

template <int NV, bool XR>
__device__ __forceinline__ double blk2_collect(unsigned epoch, unsigned xepoch, const BlkBufs& bf, const XrDev& xr, int G, int bid, int* err,
                                            BlkSm& sm, unsigned xcc, bool leader, int nv_all, double extra) {
    constexpr int NGRP = 64 / NV;
    constexpr int PFLAT = BLK_PFLAT;
    constexpr int P2 = (BLK_GS + NGRP - 1) / NGRP;
    const int lane = threadIdx.x & 63;
    unsigned long long* slot = bf.gran + (size_t)(epoch & 1u) * ((size_t)CH_GMAX * BLK_NVS * 2);
    double t;
    const bool flat = G <= PFLAT * NGRP;
    unsigned long long* slot2 = bf.gran2 + (size_t)(epoch & 1u) * ((size_t)BLK_NG2 * BLK_NVS * 2);
    const int ng2 = (G + BLK_GS - 1) / BLK_GS;
    if (!flat && bid < ng2) {                          // this workgroup adds up group `bid`
        const int cnt = (G - bid * BLK_GS) < BLK_GS ? (G - bid * BLK_GS) : BLK_GS;
        const double g = blk_gather<NV, P2>(slot + (size_t)bid * BLK_GS * BLK_NVS * 2, epoch, cnt, err, sm.q, false);
        if (lane < NV) blk_put(slot2 + ((size_t)bid * BLK_NVS + lane) * 2, epoch, g, false);
    }
    unsigned long long* res = bf.res + ((size_t)xcc * 2 + (epoch & 1u)) * (BLK_NVS * 2);
    if (leader) {
        t = flat ? blk_gather<NV, PFLAT>(slot, epoch, G, err, sm.q, false) : blk_gather<NV, P2>(slot2, epoch, ng2, err, sm.q, false);
        if constexpr (XR) {
            if (xr.nranks > 0) t = blk2_cross_rank<NV>(xr, xepoch, t, err);
        }
        if (lane < NV) blk_put(res + 2 * lane, epoch, t, true);
    } else {
        unsigned long long x0 = 0, x1 = 0;
        unsigned spins = 0;
        // (a peer rank may be late by far more than a workgroup of this launch ever is: the wait for the XCD's total is
        // bounded by the clock as well when the sum crosses the ranks)
        long long t0 = 0;
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
                bool give_up = spins > (1u << 22);
                if (XR && xr.nranks > 0) {
                    const long long now = (long long)wall_clock64();
                    if (t0 == 0) t0 = now;
                    give_up = now - t0 > xr.timeout_ticks + 200000000ll;      // (the leader's own timeout + 2 s)
                }
                if (give_up) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        t = __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
    }
    if (lane < NV) sm.tot[lane] = t;
    else if (lane < nv_all) sm.tot[lane] = extra;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its own LDS writes are in order)
    return t;
}

// The same as a CALL: at 5 rows per lane on one GPU the working waves' code keeps more of the ring in registers when wave 0's
// gathering is not inlined into it (A / B on one box, GMRES(100) at 1.25 M rows: 6,339-6,469 -> 6,617-6,631 it/s; with the
// cross-rank stage or 6 rows per lane the call's save area costs more than it gives: 5,371-5,404 -> 4,943-5,004 and
// 4,987-5,008 -> 4,617-4,756 - those stay inlined)
template <int NV, bool XR>
__device__ __noinline__ double blk2_collect_call(unsigned epoch, unsigned xepoch, const BlkBufs& bf, const XrDev& xr, int G, int bid, int* err,
                                                 BlkSm& sm, unsigned xcc, bool leader, int nv_all, double extra) {
    return blk2_collect<NV, XR>(epoch, xepoch, bf, xr, G, bid, err, sm, xcc, leader, nv_all, extra);
}

// alpha_l = c_l - sum_{m<l} alpha_m G_{m,l} in the order of the reference's loop; tot = [c_0 .. c_{BC-1} | G_01, G_02, G_12, G_03 ...]
template <int BC>
__device__ __forceinline__ void blk2_alphas(const double* tot, int nvalid, double (&alpha)[BC]) {
    int gi = BC;
#pragma unroll
    for (int l = 0; l < BC; ++l) {
        double a = tot[l];
#pragma unroll
        for (int m = 0; m < l; ++m) {
            const double pr = alpha[m] * tot[gi + m];
            a = a - pr;
        }
        gi += l;
        alpha[l] = (l < nvalid) ? a : 0.0;
    }
}

// CW: wave 0 of every workgroup owns NO rows - it publishes, gathers and forms the coefficients, and a poll of a wave that has
// requested nothing returns as soon as the total is there (vector-memory results return to a wave in order: with rows of its
// own wave 0 sees the total only behind its rows of the next block).  The other seven waves carry the rows: 448 lanes.
// ONE: a single block of BC columns in registers beside w (20 R2 registers instead of 36 R2): slabs of 8 ... 11 rows per lane, up to
// 2.5 M rows - a rank's share of the benchmark problem on FOUR devices.  Nothing of the next block is in flight under a sum
// (its registers hold the block the update still needs): the stream and the exchange take turns, and the basis is still read
// ONCE per step where the panel forms read it twice.
template <int R2, bool MASKED, bool XR, bool CW = false, bool ONE = false>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain_blk2(ChainArgs a, BlkBufs bf, XrDev xr) {
    constexpr int BC = BLK_BC;
    constexpr int NG = BlkShape<BC>::NG;
    constexpr int NW = CH_BS / 64;
    constexpr int NWORK = CW ? CH_BS - 64 : CH_BS;          // lanes with rows
    static_assert(R2 >= 4 && R2 <= (ONE ? 11 : (CW ? 7 : 6)), "two blocks of BC columns and w: 36 R2 registers of the 256 a lane has (one block: 20 R2)");
    constexpr int NSLOT = ONE ? 1 : 2;
    __shared__ BlkSm sm;
    __shared__ int slead;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool wave0 = wid == 0;
    const int G = gridDim.x;
    const int bid = blockIdx.x;
    const int nx = bf.nx;
    // (the workgroups with rows come to the leader election late: the first arrival of an XCD leads it, and the leaders
    // should be workgroups without a column stream.  Who leads changes nothing in the sums' bits.)
    if (nx > 0 && bid >= nx) __builtin_amdgcn_s_sleep(48);
    const GridRole role = grid_role(a.xcc_leader, a.epoch0, &slead);
    const bool rowless = bid < nx;
    const int rid = rowless ? G - nx + bid : bid - nx;      // its record in the exchange: the workgroups with rows first
    unsigned epoch = a.epoch0, xepoch = xr.epoch0;
    const int total = a.ncol;                               // links: columns 0 .. ncol-1, one sweep, blocks aligned
    const int nblk = (total + BC - 1) / BC;
    const bool writer = bid == 0 && tid == 0;
    // lane BC + e of wave 0 holds Gram entry e of a block: (l, m) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
    int gl = 1, gm = 0;
    {
        int e = lane - BC;
        for (int l = 1; l < BC; ++l)
            if (e >= 0 && e < l) { gl = l; gm = e; e = -1; } else if (e >= 0) e -= l;
    }
    // One block's sum in two halves around the requests of the next block (which every wave issues itself, outside any
    // wave-dependent branch).  sum_begin: every thread has left its wave's partial sums of the BC values in sm.part.
    auto sum_begin = [&](const int ib, double& gval) __attribute__((always_inline)) {
        if (wave0) {
            gval = 0.0;                                     // requested before the sum: it arrives under it
            if (lane >= BC && lane < BC + NG && ib * BC + gl < total)
                gval = __hip_atomic_load(bf.gtab + (size_t)(ib * BC + gl) * BLK_TW + BC + gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            blk2_publish<BC>(epoch, bf, rid, sm);
        } else {
            ch_lds_barrier();                               // A
            if constexpr (BLK2_REQ_SLEEP > 0) __builtin_amdgcn_s_sleep(BLK2_REQ_SLEEP);      // the publication goes out first
        }
    };
    auto sum_end = [&](const int ib, const double gval) __attribute__((always_inline)) {
        if (wave0) {
            if constexpr (R2 == 5 && !XR)
                (void)blk2_collect_call<BC, XR>(epoch, xepoch, bf, xr, G, bid, a.err, sm, role.xcc, role.leader, BC + NG, gval);
            else
                (void)blk2_collect<BC, XR>(epoch, xepoch, bf, xr, G, bid, a.err, sm, role.xcc, role.leader, BC + NG, gval);
            double alpha[BC];
            blk2_alphas<BC>(sm.tot, total - ib * BC, alpha);
#pragma unroll
            for (int l = 0; l < BC; ++l) {
                const int j = ib * BC + l;
                const double al = (a.debug == 4) ? alpha[l] * 0.5 : alpha[l];      // ... a faked timeout leaves garbage behind
                if (writer && j < total) a.hdev[a.col0 + j] = al;
                if (lane == 0) sm.al[l] = al;
            }
        }
        ch_lds_barrier();                                   // B: the block's coefficients are in LDS
        ++epoch;
        ++xepoch;
    };
    // the last sum: ||w||^2 and <v_m, w> for the earlier columns of the new column's block
    auto final_sum = [&]() __attribute__((always_inline)) {
        if (wave0) {
            blk2_publish<BC>(epoch, bf, rid, sm);
            const double tf = blk2_collect<BC, XR>(epoch, xepoch, bf, xr, G, bid, a.err, sm, role.xcc, role.leader, BC, 0.0);
            const double h = sqrt(fabs(sm.tot[0]));
            if (writer) a.hdev[a.hnext] = h;
            const int pnew = total % BC;                    // the new column's position in its block
            if (bid == 0 && lane >= 1 && lane <= pnew && lane < BC) bf.gtab[(size_t)total * BLK_TW + BC + (lane - 1)] = tf / h;
            if (bid == 0) __threadfence();                  // the H entries, for the copy to the host below
        } else {
            ch_lds_barrier();                               // A
        }
        ch_lds_barrier();                                   // B
        ++epoch;
        ++xepoch;
    };
    // bid == 0 hands the H entries to the host when the chain is done (wave 0 of this workgroup has written them)
    auto finish = [&]() __attribute__((always_inline)) {
        if (bid == 0 && a.hpin != nullptr) {
            __syncthreads();
            for (int i = tid; i < a.hcount; i += CH_BS)
                a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            CH_SIGNAL_DONE(a);
        }
    };
    if (rowless || (CW && wave0)) {
        // no rows: partial sums of zero, and the barriers of the nblk + 1 sums (a communication wave zeroes its own entries only)
        if (rowless) {
            for (int i = tid; i < BLK_NVMAX * NW; i += CH_BS) sm.part[i] = 0.0;
        } else if (lane < BLK_NVMAX) {
            sm.part[lane * NW] = 0.0;
        }
        for (int ib = 0; ib < nblk; ++ib) {
            double gval;
            sum_begin(ib, gval);
            sum_end(ib, gval);
        }
        final_sum();
        finish();
        return;
    }
    const int64_t first = (int64_t)(bid - nx) * a.chunk2 + (CW ? tid - 64 : tid);
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * NWORK < rem)
#define CH_COL(j) (reinterpret_cast<const char*>(a.V + (a.col0 + ((j) < total ? (j) : total - 1)) * a.ld))
#define CH_ROW(c, r) (*reinterpret_cast<const double2*>((c) + (size_t)(r) * (NWORK * sizeof(double2)) + boff))
    const unsigned boff = (unsigned)first * (unsigned)sizeof(double2);
    double2 ring[NSLOT][BC][R2];
    double2 w[R2];
    {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * NWORK);
            w[r].x = CH_OK(r) ? v.x : 0.0;
            w[r].y = CH_OK(r) ? v.y : 0.0;
        }
    }
    CH_ISSUE_FENCE();
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    // this lane's rows of block `blk` (clamped to the last one) into slot S
    auto request = [&](const int blk, auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        const int bq = blk < nblk ? blk : nblk - 1;
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            const char* __restrict__ c = CH_COL(bq * BC + l);
#pragma unroll
            for (int r = 0; r < R2; ++r) ring[S][l][r] = CH_ROW(c, r);
        }
        CH_ISSUE_FENCE();
    };
    // the wave's partial sums of <v_l, w> for the block in slot S, w as it is now, into sm.part
    auto dots = [&](const int i, auto slot_c) __attribute__((always_inline)) {
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
            const double ws = wave_sum_dpp((l < nvalid) ? acc0 + acc1 : 0.0);
            if (lane == 0) sm.part[l * NW + wid] = ws;
        }
    };
    // w -= the block in slot S with the coefficients wave 0 has left in LDS (the reference's order: column by column)
    auto update = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
#pragma unroll
        for (int l = 0; l < BC; ++l) {
            const double al = sm.al[l];
#pragma unroll
            for (int r = 0; r < R2; ++r) {          // (one rounding per entry: this kernel is not a bit-for-bit one)
                w[r].x = CH_OK(r) ? fma(-al, ring[S][l][r].x, w[r].x) : 0.0;
                w[r].y = CH_OK(r) ? fma(-al, ring[S][l][r].y, w[r].y) : 0.0;
            }
        }
        // w is COMPUTED here: otherwise the requests into this slot that follow are scheduled in front of the update
#pragma unroll
        for (int r = 0; r < R2; ++r) asm volatile("" : "+v"(w[r].x), "+v"(w[r].y) : : "memory");
    };
    // One block: its dots are in LDS -> the sum (the NEXT block is requested between its two halves, behind the publication of
    // the partial sums: its stream runs under the exchange) -> w -= the block -> the next block's dots.  Nothing is in flight
    // while the update and the dots run: 10 us per block of four columns at 1.25 M rows, of which 6.2 are the stream.
    // Two deeper schedules were built and measured at that shape (profiles/r05_blk2.log) and are gone again:
    //   * every wave TWO blocks ahead (block i + 2 requested into the slot the update with block i has freed, while block
    //     i + 1 still arrives): 11.5 us per block - the publication of the next sum queues behind those requests and wave 0's
    //     polls return behind its own rows of block i + 2 (what chain_blk.h found for requests in front of a sum);
    //   * waves 1 .. 7 two blocks ahead, wave 0 with nothing in flight across a sum and its rows of the next block requested
    //     just in time: 14 us - a request that joins a saturated memory system late waits for everything in front of it, and
    //     the whole workgroup waits for wave 0's rows.
    request(0, std::integral_constant<int, 0>{});
    dots(0, std::integral_constant<int, 0>{});
    auto block = [&](const int i, auto slot_c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_c)::value;
        double gval;
        sum_begin(i, gval);
        // (behind the last block the request is clamped to the last block - no conditional around the requests, or the
        // compiler's wait counts at the loop's back edge become "everything")
        if constexpr (!ONE) request(i + 1, std::integral_constant<int, 1 - S>{});
        sum_end(i, gval);
        update(slot_c);
        if constexpr (ONE) request(i + 1, slot_c);              // (the one slot is free now)
        dots(i + 1, std::integral_constant<int, ONE ? 0 : 1 - S>{});      // (behind the last block: discarded)
    };
    if constexpr (ONE) {
        for (int ib = 0; ib < nblk; ++ib) block(ib, std::integral_constant<int, 0>{});
    } else {
        int ib = 0;
        for (; ib + 2 <= nblk; ib += 2) {
            block(ib, std::integral_constant<int, 0>{});
            block(ib + 1, std::integral_constant<int, 1>{});
        }
        if (ib < nblk) block(ib, std::integral_constant<int, 0>{});
    }
#undef CH_ROW
#undef CH_COL
    {
        // ||w||^2, and <v_m, w> for the earlier columns of the new column's block = the LAST block of the chain (when the new
        // column does not start a block of its own), which sits in slot (nblk - 1) % 2; both slots' sums are formed and the
        // wanted one selected (a select between two elements of the ring would send it to scratch memory)
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            acc = fma(w[r].x, w[r].x, acc);
            acc = fma(w[r].y, w[r].y, acc);
        }
        {
            const double ws = wave_sum_dpp(acc);
            if (lane == 0) sm.part[0 * NW + wid] = ws;
        }
        const int sl = ONE ? 0 : ((nblk - 1) & 1);
        const int pnew = total % BC;
#pragma unroll
        for (int m = 0; m < BC - 1; ++m) {
            double s0a = 0.0, s0b = 0.0, s1a = 0.0, s1b = 0.0;
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                double2 v0 = ring[0][m][r], v1 = ring[NSLOT - 1][m][r];
                if (MASKED && !CH_OK(r)) v0 = v1 = make_double2(0.0, 0.0);
                s0a = fma(v0.x, w[r].x, s0a);
                s0b = fma(v0.y, w[r].y, s0b);
                s1a = fma(v1.x, w[r].x, s1a);
                s1b = fma(v1.y, w[r].y, s1b);
            }
            const double sum = (sl == 1) ? s1a + s1b : s0a + s0b;
            const double ws = wave_sum_dpp((m < pnew) ? sum : 0.0);
            if (lane == 0) sm.part[(1 + m) * NW + wid] = ws;
        }
    }
    final_sum();
    const double h = sqrt(fabs(sm.tot[0]));
    double2* __restrict__ vn2 = reinterpret_cast<double2*>(a.vnext) + first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        if (r * NWORK < rem) {
            double2 o;
            o.x = w[r].x / h;
            o.y = w[r].y / h;
            st_nt2(vn2 + (int64_t)r * NWORK, o);
        }
    }
    finish();
#undef CH_OK
}

}  // namespace kh
