// Register-resident modified Gram-Schmidt chain: ONE launch per Arnoldi step.
//
// The reference's MGS (utils.py:1012-1034) is a chain of k+1 *dependent* dot/axpy pairs on the
// same vector w.  Run as k+1 launches (k_gs_link) every link re-streams w through HBM: 32 N
// bytes per basis column, and the kernel saturates the memory system at 0.41 of the algorithmic
// roofline.  On MI355X the aggregate register file (256 CUs x 512 KB = 128 MB) is larger than w
// (80 MB at N = 10^7), so this kernel keeps w in VGPRs for the whole chain:
//
//   load w (the SpMV result) into registers                                   8 N bytes
//   for every basis column j (reference order, `sweeps` passes):
//       partial <v_j, w>           stream v_j once                            8 N
//       grid-wide fixed-order sum  one 8-byte granule pair per workgroup      (latency)
//       w -= alpha * b_j           re-read of the same column (MALL/L2 warm)  8 N
//   ||w||^2 (or <w, D w>) grid-wide,  v_{k+1} = w / h written straight from registers   8 N
//
// i.e. 16 N per column - the SURVEY 8(d) figure - instead of 32 N, no w traffic at all, and the
// summation order is still the reference's.  Workgroups exchange their partial sums inside the
// launch with the data-tagged 8-byte granule protocol of the CDNA4 guide (Guideline 16, R2):
// one relaxed agent-scope atomic store per granule {epoch tag | 32 value bits}, consumers re-read
// the granules with relaxed agent-scope atomic loads until every tag matches - no flag, no fence,
// placement independent.  Granule buffers alternate by epoch parity; a workgroup can only be one
// epoch ahead of the slowest one, so two buffers suffice.  Every spin is bounded: on timeout an
// error word is set and all workgroups fall through; kh_arnoldi_step_end then switches this path off
// for the context and runs the same step again on the per-column kernels (columns 0..k are intact).
//
// Residency: one 512-thread workgroup per CU (<= 256 VGPRs per lane).  The launch is a plain one (a
// cooperative launch costs ~1 ms of cross-queue synchronisation per step next to ordinary kernels and
// checks nothing a plain launch does not have: residency is identical); the launcher checks the grid
// against the occupancy of the instantiation itself, and the bounded spins cover what is left (a GPU
// shared with another process).
#pragma once
#include <type_traits>
#include "kernels.h"

namespace kh {

constexpr int CH_BS = 512;       // threads per workgroup (8 wave64, 2 per SIMD)
constexpr int CH_GMAX = 512;     // max workgroups (granule sweep = 2*G words, one per thread pass)

struct ChainArgs {
    int64_t n2;        // vector length in double2 (n even)
    int64_t chunk2;    // double2 per workgroup
    const double* V;   // dot columns:    V + j*ld
    const double* B;   // update columns: B + j*ld  (P when preconditioned, else V)
    int64_t ld;
    int64_t col0;      // first column
    int ncol;          // columns per sweep
    int sweeps;
    const double* w_in;
    const double* dg;  // Jacobi diagonal or nullptr
    double* vnext;     // V[:, k+1]
    double* pnext;     // P[:, k+1] or nullptr
    double* hdev;      // H column on the device (zeroed by the caller); hdev[j] += alpha
    int64_t hnext;     // index of H[k+1,k] in hdev
    unsigned long long* gran;  // [2][2*G] granules
    unsigned* xcc_leader;      // [16] launch stamp of the last leader election per XCD
    unsigned long long* xcc_res;   // [16][2 parities][4] the grid-wide sum(s), handed on inside an XCD through its L2
    unsigned epoch0;
    int* err;
    int debug;         // measurement only: 1 = skip the grid reduction, 2 = skip the streaming phases; tests: 4 = fake a timeout
    int presub;        // Lanczos: w -= h_km1 * bprev first
    double h_km1;
    const double* h_km1_dev;   // when non-null the coefficient is read from the device (look-ahead)
    const double* bprev;
    // when hpin != nullptr workgroup 0 finally copies the H column (hcount doubles) and the error word
    // straight into pinned host memory: no device-to-host copies behind the launch
    double* hpin;
    int hcount;
    int* errpin;
    // ... and, when donepin != nullptr, finally stores done_tag there (system scope, behind the H column): the host
    // polls that word instead of waiting for an event (CH_SIGNAL_DONE)
    int* donepin;
    int done_tag;
    // fused operator (k_mgs_chain_lds<..., FND > 0>): w = A x computed by the prologue straight into
    // registers from the diagonal-major copy of a banded operator instead of being loaded
    const double* dia;
    int64_t dia_ld;
    const double* xk;
    int64_t n_last;    // n - 1
    DiaOffs offs;
    // short vectors, all working workgroups on ONE XCD (ONEX instantiations): 8 G + 8 workgroups are launched, those
    // that do not run on XCD `onex_target` leave at once, the others draw a ticket and the first onex_G of them work
    int onex_G;
    unsigned onex_target;
    unsigned* onex_ticket;     // zero when the launch begins
    unsigned* onex_clear;      // the ticket word of a launch far in the future: zeroed by this one
    // N ranks (XR instantiations only: nobody else reads it): every grid-wide sum of the launch also crosses the ranks through
    // the IPC-mapped mailboxes of xr.hip - inside the XCD leaders' hand-over (grid_sum<true>)
    XrDev xr;
#ifdef KH_CHAIN_TRACE
    unsigned long long* trace;   // diagnostic build only (make trace): [G][links][2 waves][8] 100 MHz stamps
#endif
};

// Completion tag of a launch, written by workgroup 0 behind its copy of the H column (every thread of the workgroup
// takes this path): each thread's stores to the pinned buffer are performed at system scope before the barrier, the
// tag follows them.
#define CH_SIGNAL_DONE(a_)                                                                               \
    do {                                                                                                 \
        if ((a_).donepin != nullptr) {                                                                   \
            __threadfence_system();                                                                      \
            __syncthreads();                                                                             \
            if (threadIdx.x == 0)                                                                        \
                __hip_atomic_store((a_).donepin, (a_).done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); \
        }                                                                                                \
    } while (0)

// Phase stamps of the diagnostic build (tools/chain_trace.py): wave 0 and wave 7 of every workgroup
// note the constant 100 MHz clock at the phase boundaries of every link.  Compiled out of the product.
#if defined(KH_CHAIN_TRACE) && KH_CHAIN_TRACE == 2
// (mode 2: every wave keeps the SUM of each phase's duration over the links of a launch in scalar registers and
// wave 0 / wave 7 write the eight sums once at the end - no vector registers, no stores inside the link loop: the
// per-link stamps of mode 1 cost the kernel its register allocation and stretch a link from 16 to 40 us)
#define CH_STAMP(a_, t_, total_, i_)                          \
    do {                                                      \
        const unsigned long long n_ = wall_clock64();         \
        if ((i_) != 0) tr_acc[i_] += n_ - tr_acc[8];          \
        tr_acc[8] = n_;                                       \
    } while (0)
#elif defined(KH_CHAIN_TRACE)
#define CH_STAMP(a_, t_, total_, i_)                                                                   \
    do {                                                                                               \
        if ((a_).trace != nullptr && (threadIdx.x == 0 || threadIdx.x == CH_BS - 64))                  \
            (a_).trace[((((size_t)blockIdx.x * (total_)) + (t_)) * 2 + (threadIdx.x ? 1 : 0)) * 8 + (i_)] = \
                wall_clock64();                                                                        \
    } while (0)
#else
#define CH_STAMP(a_, t_, total_, i_) do {} while (0)
#endif

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 64-lane sum with DPP moves (row_shr 1/2/4/8, row_bcast 15/31): the total ends up in lane 63.  About 100
// cycles; __shfl_down compiles to ds_bpermute pairs and costs six LDS round trips.  Fixed tree, deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {      // wave-uniform result
    v += dpp_take<0x111, 0xf>(v);   // row_shr:1
    v += dpp_take<0x112, 0xf>(v);   // row_shr:2
    v += dpp_take<0x114, 0xf>(v);   // row_shr:4
    v += dpp_take<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_take<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// one granule of the current epoch (bounded spin; on timeout the error word is set and everybody falls through)
__device__ __forceinline__ unsigned long long poll_granule(const unsigned long long* p, unsigned epoch, int* err) {
    unsigned long long x = ld_agent(p);
    unsigned spins = 0;
    while ((unsigned)(x >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        x = ld_agent(p);
        if ((++spins & 1023u) == 0) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (spins > (1u << 22)) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    return x;
}

// Which XCD (accelerator die, private L2) this workgroup runs on, and whether it is that XCD's LEADER for this
// launch: the first of the XCD's workgroups to raise the XCD's stamp to the launch's epoch.  Placement
// independent: a workgroup only ever shares an L2 hand-off with workgroups that read the same XCC id.
struct GridRole {
    unsigned xcc;
    bool leader;
};

__device__ __forceinline__ GridRole grid_role(unsigned* xcc_leader, unsigned stamp, int* sflag) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    GridRole r;
    r.xcc = v & 0xfu;
    if (threadIdx.x == 0)
        *sflag = __hip_atomic_fetch_max(xcc_leader + r.xcc, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stamp;
    __syncthreads();
    r.leader = *sflag != 0;
    return r;
}

// Fixed-order sum of one double per workgroup over the whole grid; every thread gets the result.
//   every workgroup   wave sums (DPP) -> LDS -> thread 0 publishes the workgroup's partial as two tagged granules
//                     (write-through stores: visible on every XCD)
//   XCD leaders (8)   every thread polls ONE granule of the sweep over the fabric (2 G <= 512 of them), neighbouring
//                     lanes pair the halves of a partial with one DPP move, each wave adds its 32 partials (DPP tree),
//                     the eight wave sums meet in LDS; thread 0 then stores the total as a tagged granule pair with
//                     PLAIN stores - it stays in the XCD's L2
//   everybody else    polls that pair with one 16-byte L1-bypassing load: an L2 hit on its own XCD, no fabric traffic
// tools/probe/gsum_probe.hip, 245 workgroups: 2.2 us per sum against 3.6 us when all 245 workgroups sweep the
// fabric themselves (and 30 times fewer polls on the fabric while stragglers still stream).  The order of the
// additions is the same in every leader: all workgroups get the same bits.  (More than 256 workgroups: every
// workgroup sweeps, through LDS.)
// XR (N ranks, chain_xr.hip): the sum also crosses the RANKS - an XCD leader that holds this device's total stores it into
// every rank's mailbox (xr_dev.h: tagged system-scope stores; the eight leaders of a rank write the same bits to the same
// slots - idempotent, no election), polls its own mailbox for the N contributions, adds them in rank order and hands THAT
// total on through its XCD's L2: every workgroup of every rank gets the same bits.  `xepoch` tags the exchange.
template <bool XR = false>
__device__ __forceinline__ double grid_sum(double part, unsigned epoch, unsigned long long* gran,
                                           int G, int* err, double* smd, unsigned* smu, const GridRole role,
                                           unsigned long long* xcc_res, unsigned long long* tr = nullptr,
                                           const XrDev* xr = nullptr, unsigned xepoch = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    // 1. workgroup partial (8 waves, fixed order)
    const double ws = wave_sum_dpp(part);
    if (lane == 0) smd[wid] = ws;
    __syncthreads();
#if defined(KH_CHAIN_TRACE) && KH_CHAIN_TRACE == 2
#define GS_STAMP(i_)                                          \
    do {                                                      \
        if (tr != nullptr) {                                  \
            const unsigned long long n_ = wall_clock64();     \
            tr[i_] += n_ - tr[8];                             \
            tr[8] = n_;                                       \
        }                                                     \
    } while (0)
#elif defined(KH_CHAIN_TRACE)
#define GS_STAMP(i_)                                                                                      \
    do {                                                                                                  \
        if (tr != nullptr && (tid == 0 || tid == CH_BS - 64)) tr[(tid ? 8 : 0) + (i_)] = wall_clock64();  \
    } while (0)
#else
#define GS_STAMP(i_) do {} while (0)
#endif
    GS_STAMP(2);
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (tid == 0) {
        double s = smd[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        st_agent(slot + 2 * blockIdx.x, tag | (bits & 0xffffffffull));
        st_agent(slot + 2 * blockIdx.x + 1, tag | (bits >> 32));
    }
    GS_STAMP(3);
    if (2 * G <= CH_BS) {
        unsigned long long* res = xcc_res + ((size_t)role.xcc * 2 + (epoch & 1u)) * 4;
        if (!role.leader) {
            // 2b. the leader of this XCD will put the total into `res`: one 16-byte load per poll, served by the L2
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 ab;
            unsigned spins = 0;
            long long t0 = 0;
            while (true) {
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ab) : "v"(res) : "memory");
                if ((unsigned)(ab.x >> 32) == epoch && (unsigned)(ab.y >> 32) == epoch) break;
                if ((++spins & 4095u) == 0) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    bool give_up = spins > (1u << 24);
                    if constexpr (XR) {
                        // (a peer rank may be late by far more than a workgroup of this launch ever is: by the clock, the
                        // leader's own timeout + 2 s)
                        if (xr->nranks > 0) {
                            const long long now = (long long)wall_clock64();
                            if (t0 == 0) t0 = now;
                            give_up = now - t0 > xr->timeout_ticks + 200000000ll;
                        }
                    }
                    if (give_up) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            return __longlong_as_double((long long)(((ab.y & 0xffffffffull) << 32) | (ab.x & 0xffffffffull)));
        }
        // 2a. leader: one granule per thread, lane 2b holds the low word of workgroup b's partial, lane 2b+1 the high one
        unsigned mine = 0;
        if (tid < 2 * G) {
            unsigned long long x = ld_agent(slot + tid);
            unsigned spins = 0;
            while ((unsigned)(x >> 32) != epoch) {
                x = ld_agent(slot + tid);
                if ((++spins & 1023u) == 0) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1u << 22)) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            mine = (unsigned)x;
        }
        GS_STAMP(4);
        const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);   // from lane - 1
        const unsigned long long bits = ((unsigned long long)mine << 32) | low;
        const double v = ((lane & 1) && tid < 2 * G) ? __longlong_as_double((long long)bits) : 0.0;
        // 3. 32 partials per wave (DPP tree), eight wave sums through LDS
        const double wv = wave_sum_dpp(v);
        if (lane == 0) smd[NW + wid] = wv;
        __syncthreads();
        double s = smd[NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[NW + i];
        if constexpr (XR) {
            if (xr->nranks > 0) {      // 3b. the cross-rank stage (one lane; the workgroup takes the total from LDS)
                if (tid == 0) {
                    xr_put_all(xr->peer, xr->rank, xr->nranks, xepoch, 0, s);
                    int bad = 0;
                    const double t = xr_take_all(xr->peer[xr->rank], xr->nranks, xepoch, 0, xr->timeout_ticks, &bad);
                    if (bad) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (2: a peer rank is missing)
                    smd[2 * NW] = t;
                }
                __syncthreads();
                s = smd[2 * NW];
            }
        }
        if (tid == 0) {      // 4. plain stores: the pair stays in this XCD's L2, where its other workgroups poll it
            const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
            const unsigned long long tag = (unsigned long long)epoch << 32;
            res[0] = tag | (sb & 0xffffffffull);
            res[1] = tag | (sb >> 32);
        }
        return s;
    }
    // 2'. sweep all granules of this epoch through LDS
    for (int g = tid; g < 2 * G; g += CH_BS) smu[g] = (unsigned)poll_granule(slot + g, epoch, err);
    __syncthreads();
    // 3'. fixed-order sum of the G workgroup partials
    double v = 0.0;
    for (int b = tid; b < G; b += CH_BS) {
        const unsigned long long bits = ((unsigned long long)smu[2 * b + 1] << 32) | smu[2 * b];
        v += __longlong_as_double((long long)bits);
    }
    v = wave_sum_dpp(v);
    if (lane == 0) smd[NW + wid] = v;
    __syncthreads();
    double s = smd[NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) s += smd[NW + i];
    __syncthreads();          // smu reuse by the next call
    return s;
}

#undef GS_STAMP

// Two sums in one round (the real and imaginary part of a complex coefficient): four granules per workgroup, one
// publish; the XCD leaders sweep 4 G <= 1024 granules (two per thread: workgroups b and b + 128 of the same kind
// and half land in the same lane), everybody else reads the leader's two result pairs from the L2.  Same protocol
// and the same fixed summation order in every leader as grid_sum.  smd holds 4 * 8 doubles.
__device__ __forceinline__ void grid_sum2(double& p0, double& p1, unsigned epoch, unsigned long long* gran,
                                          int G, int* err, double* smd, unsigned* smu, const GridRole role,
                                          unsigned long long* xcc_res) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    const double a0 = wave_sum_dpp(p0), a1 = wave_sum_dpp(p1);
    if (lane == 0) {
        smd[wid] = a0;
        smd[NW + wid] = a1;
    }
    __syncthreads();
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (tid < 2) {
        double s = smd[tid * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[tid * NW + i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        st_agent(slot + 4 * blockIdx.x + 2 * tid, tag | (bits & 0xffffffffull));
        st_agent(slot + 4 * blockIdx.x + 2 * tid + 1, tag | (bits >> 32));
    }
    unsigned long long* res = xcc_res + ((size_t)role.xcc * 2 + (epoch & 1u)) * 4;
    if (!role.leader) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 ab, cd;
        unsigned spins = 0;
        while (true) {
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(ab), "=&v"(cd)
                         : "v"(res)
                         : "memory");
            if ((unsigned)(ab.x >> 32) == epoch && (unsigned)(ab.y >> 32) == epoch && (unsigned)(cd.x >> 32) == epoch &&
                (unsigned)(cd.y >> 32) == epoch)
                break;
            if ((++spins & 4095u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 24)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        p0 = __longlong_as_double((long long)(((ab.y & 0xffffffffull) << 32) | (ab.x & 0xffffffffull)));
        p1 = __longlong_as_double((long long)(((cd.y & 0xffffffffull) << 32) | (cd.x & 0xffffffffull)));
        return;
    }
    // leader: granule g = 4 b + 2 kind + half; this thread takes g = tid and g = tid + 512
    double d = 0.0;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int g = tid + rr * CH_BS;
        unsigned mine = 0;
        if (g < 4 * G) {
            unsigned long long x = ld_agent(slot + g);
            unsigned spins = 0;
            while ((unsigned)(x >> 32) != epoch) {
                x = ld_agent(slot + g);
                if ((++spins & 1023u) == 0) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (spins > (1u << 22)) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            mine = (unsigned)x;
        }
        const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);   // from lane - 1
        const unsigned long long bits = ((unsigned long long)mine << 32) | low;
        d += ((lane & 1) && g < 4 * G) ? __longlong_as_double((long long)bits) : 0.0;
    }
    const bool kind1 = ((lane >> 1) & 1) != 0;
    const double w0 = wave_sum_dpp(kind1 ? 0.0 : d), w1 = wave_sum_dpp(kind1 ? d : 0.0);
    if (lane == 0) {
        smd[2 * NW + wid] = w0;
        smd[3 * NW + wid] = w1;
    }
    __syncthreads();
    double s0 = smd[2 * NW], s1 = smd[3 * NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) {
        s0 += smd[2 * NW + i];
        s1 += smd[3 * NW + i];
    }
    if (tid == 0) {
        const unsigned long long tag = (unsigned long long)epoch << 32;
        const unsigned long long b0 = (unsigned long long)__double_as_longlong(s0);
        const unsigned long long b1 = (unsigned long long)__double_as_longlong(s1);
        res[0] = tag | (b0 & 0xffffffffull);
        res[1] = tag | (b0 >> 32);
        res[2] = tag | (b1 & 0xffffffffull);
        res[3] = tag | (b1 >> 32);
    }
    p0 = s0;
    p1 = s1;
}

// ------------------------------------------------------------------------------------------
// Short vectors: all working workgroups on ONE XCD.
// With 4 ... 32 workgroups a Gram-Schmidt link IS its grid-wide sum (1.1 - 2.1 us over the fabric; the stream is
// a few hundred bytes per lane), 50 of them per Arnoldi step.  Workgroups of one XCD share its L2, so when all of
// them run there the granules never leave it: publish with plain stores, EVERY workgroup sweeps the 2 G granules
// itself with L1-bypassing loads (L2 hits) - one L2 round trip, no leader, no fabric: 0.81 - 0.84 us per sum for
// 4 ... 32 workgroups (tools/probe/overlap_probe.hip, PROBE_ONEXCD=1; 1.02 - 1.10 with a leader's extra hop).
// Placement: workgroups are dealt round-robin over the XCDs, so of 8 G + 8 launched G + 1 land on each; those on
// the other XCDs leave at once, the first G ticket holders on the target work.  Nothing depends on the deal being
// exact - only on G workgroups reaching the target, and the bounded spins cover the case that they do not (the
// step is then re-run on the per-column kernels like after any other timeout).  Same partials, same order of
// additions as grid_sum: the same bits.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_l2(const unsigned long long* p) {
    unsigned long long x;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
    return x;
}

// how many workgroups of a launch land on each XCD (kh_ctx_create checks the deal once)
static __global__ void k_onex_probe(unsigned* count) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) atomicAdd(count + (v & 0xfu), 1u);
}

// returns false when this workgroup has nothing to do (not on the target XCD, or no ticket left)
__device__ __forceinline__ bool onex_enter(const ChainArgs& a, int* sflag, int& bid, int& G, GridRole& role) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    v &= 0xfu;
    if (v != a.onex_target) return false;
    if (threadIdx.x == 0) *sflag = (int)atomicAdd(a.onex_ticket, 1u);
    __syncthreads();
    bid = *sflag;
    G = a.onex_G;
    if (bid >= G) return false;
    __syncthreads();          // (sflag may be reused)
    role.xcc = v;
    role.leader = false;
    if (bid == 0 && threadIdx.x == 0) *a.onex_clear = 0u;
    return true;
}

// NV values (1: a real coefficient / norm, 2: both parts of a complex coefficient) over the G <= 32 workgroups of
// one XCD; granule g = 2 NV bid + 2 kind + half.  smd holds 4 * 8 doubles.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + barrier, and on this target the fence
// waits for the wave's outstanding VECTOR-MEMORY operations too (s_waitcnt vmcnt(0)): every barrier of a sum would
// drain the columns a wave has requested ahead.  The sums exchange their data through LDS (and through global memory
// only in the one wave that polls), so the LDS counter is all that has to be waited for.
__device__ __forceinline__ void ch_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NV>
__device__ __forceinline__ void onex_sum(double (&p)[NV], unsigned epoch, unsigned long long* gran, int G, int bid,
                                         int* err, double* smd) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double ws = wave_sum_dpp(p[k]);
        if (lane == 0) smd[k * NW + wid] = ws;
    }
    ch_lds_barrier();
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (tid < NV) {
        double s = smd[tid * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[tid * NW + i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        slot[2 * NV * bid + 2 * tid] = tag | (bits & 0xffffffffull);          // plain stores: they stay in this XCD's L2
        slot[2 * NV * bid + 2 * tid + 1] = tag | (bits >> 32);
    }
    unsigned mine = 0;
    const int ng = 2 * NV * G;          // <= 128
    if (tid < ng) {
        unsigned long long x = ld_l2(slot + tid);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
            x = ld_l2(slot + tid);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 22)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        mine = (unsigned)x;
    }
    const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);   // from lane - 1
    const unsigned long long bits = ((unsigned long long)mine << 32) | low;
    const double d = ((lane & 1) && tid < ng) ? __longlong_as_double((long long)bits) : 0.0;
    if (NV == 1) {
        const double wv = wave_sum_dpp(d);
        if (lane == 0) smd[2 * NW + wid] = wv;
    } else {
        const bool kind1 = ((lane >> 1) & 1) != 0;
        const double w0 = wave_sum_dpp(kind1 ? 0.0 : d), w1 = wave_sum_dpp(kind1 ? d : 0.0);
        if (lane == 0) {
            smd[2 * NW + wid] = w0;
            smd[3 * NW + wid] = w1;
        }
    }
    ch_lds_barrier();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = smd[(2 + k) * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[(2 + k) * NW + i];
        p[k] = s;
    }
    ch_lds_barrier();          // smd is reused by the next sum
}

// The same sum with a COMMUNICATION WAVE: a fifth wave of the workgroup that owns no rows and does nothing but the
// exchange.  Vector-memory results return to a wave in order, so a poll issued by a wave that has requested columns
// ahead waits for those columns first (a miss to HBM: ~2 us, twice the sum); in the wave that only polls nothing is
// ever in front of the poll.  The four working waves leave their partial in LDS and wait at two LDS-only barriers.
// Same partials, same order of additions as onex_sum<1>: the same bits.
__device__ __forceinline__ double onex_sum_work(double part, double* smd) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double ws = wave_sum_dpp(part);
    if (lane == 0) smd[wid] = ws;
    ch_lds_barrier();          // partials are in LDS
    ch_lds_barrier();          // the communication wave has left the total there
    return smd[CH_BS / 64];
}

__device__ __forceinline__ double onex_sum_comm(unsigned epoch, unsigned long long* gran, int G, int bid, int* err,
                                                double* smd) {
    constexpr int NW = CH_BS / 64;
    const int lane = threadIdx.x & 63;
    ch_lds_barrier();
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (lane == 0) {
        double s = smd[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += smd[i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        slot[2 * bid] = tag | (bits & 0xffffffffull);
        slot[2 * bid + 1] = tag | (bits >> 32);
    }
    unsigned mine = 0;
    const int ng = 2 * G;          // <= 64
    if (lane < ng) {
        unsigned long long x = ld_l2(slot + lane);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
            x = ld_l2(slot + lane);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 22)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        mine = (unsigned)x;
    }
    const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);   // from lane - 1
    const unsigned long long bits = ((unsigned long long)mine << 32) | low;
    const double d = ((lane & 1) && lane < ng) ? __longlong_as_double((long long)bits) : 0.0;
    double tot = wave_sum_dpp(d);
#pragma unroll
    for (int i = 1; i < NW; ++i) tot += 0.0;          // (onex_sum adds the empty partials of the other waves)
    if (lane == 0) smd[NW] = tot;
    ch_lds_barrier();
    return tot;
}

// Every vector this kernel reads is a kh_vec / diag buffer allocated with CH_SLACK zeroed doubles
// behind its last column, so loads need no clamping: an out-of-range lane reads finite data of a
// neighbouring chunk/column (or zeros) and its w stays exactly 0 (select on registers).
constexpr int64_t CH_SLACK = 2 * 48 * (int64_t)CH_BS;  // doubles: one workgroup chunk (<= 40 rows of CH_BS double2) + margin

// Software pipeline: the rows a thread owns are streamed in batches of PB rows through a two-deep
// register ring.  The loads of batch b+1 are issued before batch b is consumed, and the ring runs
// ACROSS phase boundaries: the first batch of the update phase (b_j) is in flight while the grid
// reduction of <v_j, w> completes, the first batch of the next column's dot phase while the update
// finishes.  The empty asm is a compiler memory barrier that pins the issue order (w already holds
// 4*R2 VGPRs; the ring adds 8*PB).
#define CH_ISSUE_FENCE() asm volatile("" ::: "memory")

template <int R2>
struct ChainShape {
    static constexpr int PB = (R2 == 40) ? 5 : (R2 == 4 ? 2 : 4);   // rows per batch
    static constexpr int NB = R2 / PB;                              // batches per phase (even)
    static_assert(NB * PB == R2 && (NB % 2) == 0, "ring parity must reset every phase");
};

// Two instantiations per R2.  MASKED=false is the fast kernel for large vectors: the blocks were
// allocated with a leading dimension padded to whole workgroup chunks (kh_vec_alloc), the padding
// is zero and stays zero, so the hot loops carry no predicate at all (out-of-range lanes hold
// w = 0 and read p = 0); only the final store is masked.  MASKED=true is the general kernel for
// small or unpadded vectors.  `rem` = number of valid double2 starting at this thread's first.
//
// CPLX: the same kernel for complex (c128) vectors.  A complex N-vector is a real block of 2N doubles
// (re, im interleaved), i.e. exactly one double2 row per element, so geometry, pipeline and padding
// are unchanged; the dot is conj(v).w (two grid reductions per link: re, im), the update the complex
// multiply of NumPy (separate roundings), H entries are (re, im) pairs in hdev.
// Fused operator of the chain kernels (FND > 0 diagonals of a banded operator, krylov_hip.hip builds the
// diagonal-major copy): this lane's rows of w = A x_k, computed straight into the registers that hold w.
#ifndef KH_RIF
#define KH_RIF 3          // rows of the fused operator in flight - 1 (a mask: 1 = two rows, 3 = four)
#endif
// The same for a complex operator (round 4): one double2 row of w is ONE complex row, the diagonal-major copy holds
// (re, im) pairs (zpath.h: k_zdia_fill), x_k is read as double2; products by NumPy's formula and sums from (0, 0) in
// ascending offset order, empty slots skipped - exactly what k_zspmv_stream does entry by entry: the same bits.
template <int R2, int FND, class Put>
__device__ __forceinline__ void chain_apply_banded_z(const ChainArgs& a, int64_t first, Put&& put) {
    const double2* __restrict__ xk = reinterpret_cast<const double2*>(a.xk);
    const double2* __restrict__ dia = reinterpret_cast<const double2*>(a.dia);
    const int64_t last = a.n_last;           // last complex row
    int64_t fb = first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int64_t row = fb + (int64_t)r * CH_BS;
        double2 av[FND], xv[FND];
#pragma unroll
        for (int d = 0; d < FND; ++d) {
            av[d] = ld_nt2(dia + (int64_t)d * a.dia_ld + row);
            int64_t c = row + a.offs.off[d];
            c = c < 0 ? 0 : (c > last ? last : c);
            xv[d] = xk[c];
        }
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int d = 0; d < FND; ++d) {
            const double px = av[d].x * xv[d].x - av[d].y * xv[d].y;
            const double py = av[d].x * xv[d].y + av[d].y * xv[d].x;
            const bool on = (av[d].x != 0.0) || (av[d].y != 0.0);
            sx = on ? sx + px : sx;
            sy = on ? sy + py : sy;
        }
        put(r, sx, sy);
        if ((r & 1) == 1) asm volatile("" : "+v"(fb) : : "memory");      // two rows of loads in flight
    }
}

template <int R2, int FND, int RIF = KH_RIF, class Put>
__device__ __forceinline__ void chain_apply_banded(const ChainArgs& a, int64_t first, Put&& put) {
    // w = A x_k for this lane's rows, exactly as k_spmv_dia computes them (ascending offsets,
    // separate multiply and add, empty slots skipped): the 80 MB of w are never written nor read.
    // The padding rows behind n hold zeros in every diagonal and come out as w = 0.
    const double* __restrict__ xk = a.xk;
    const int64_t last = a.n_last;
    int64_t fb = first;      // passed through an opaque asm every two rows: that is what bounds the
                             // loads in flight (and the live temporaries) of this fully unrolled loop.
                             // Row r sits at a CONSTANT distance from fb.  (Until late in round 3 the index itself was
                             // advanced by CH_BS per row through the asm: the same loads in the same order, but
                             // rocprofv3 FETCH_SIZE of the MINRES iteration at N = 10^7 read 1223 MB where the streams
                             // add up to 1082 - lines of x fetched again for the +-nx neighbours - and 1126 MB in this
                             // form; pass 1 of the Lanczos kernel 108.6 -> 99.4 us.  Walking the rows in the order
                             // c, c + 4, c + 8, ..., which puts the three uses of a line of x into consecutive steps,
                             // changes neither figure any further: 1123 MB, 99.4 us.)
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int64_t i2 = fb + (int64_t)r * CH_BS;
        const int64_t row = 2 * i2;
        double2 av[FND];
        double x0[FND], x1[FND];
#pragma unroll
        for (int d = 0; d < FND; ++d) {
            const int64_t off = a.offs.off[d];
            av[d] = ld_nt2(reinterpret_cast<const double2*>(a.dia + (int64_t)d * a.dia_ld) + i2);
            // branch-free (clamped scalar loads of x: L2 hits): control flow in this fully
            // unrolled prologue sends the register allocator into > 1000 spills
            int64_t c0 = row + off, c1 = row + 1 + off;
            c0 = c0 < 0 ? 0 : (c0 > last ? last : c0);
            c1 = c1 < 0 ? 0 : (c1 > last ? last : c1);
            x0[d] = xk[c0];
            x1[d] = xk[c1];
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int d = 0; d < FND; ++d) {
            const double p0 = av[d].x * x0[d], p1 = av[d].y * x1[d];
            s0 = (av[d].x != 0.0) ? s0 + p0 : s0;
            s1 = (av[d].y != 0.0) ? s1 + p1 : s1;
        }
        put(r, s0, s1);         // (row r of w: a register, or - long shapes - this lane's LDS entry)
        if ((r & RIF) == RIF) asm volatile("" : "+v"(fb) : : "memory");   // four rows of loads in flight (with the running index of rounds 1-2: one 1023 it/s, two 1028-1037, four 1018-1030; with the constant distances: two 1087-1089, four 1093-1095 on one box)
    }
}

// WL > 0 (with FND > 0 the operator's rows beyond the registers are put into the LDS entries directly: instantiated for
// 48 rows and 7 / 5 diagonals - config 5's slab; at 56 rows the unrolled prologue spills 163 registers):
// the vector is longer than the register file can hold (more than 40 rows per lane, N > 10.48 M on
// 256 CUs): the last WL rows of w live in LDS (8 KB per row and workgroup, each lane touches only its own
// entries: no barrier), the first R2 - WL in registers as ever.  R2 = 48 / 56 (WL = 8 / 16) take N to
// 12.58 M / 14.68 M per GPU with the same single launch per Arnoldi step and the same arithmetic.
template <int R2, bool MASKED, bool CPLX = false, int FND = 0, int WL = 0, bool ONEX = false, bool XR = false>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain(ChainArgs a) {
    static_assert(FND == 0 || !MASKED, "the fused operator exists for the padded kernels");
    static_assert(!XR || (!CPLX && !ONEX), "the cross-rank stage exists for real vectors spread over the chip");
    constexpr int RW = R2 - WL;                   // rows of w in registers
    extern __shared__ __attribute__((aligned(16))) double2 wl[];   // [WL][CH_BS]
#define W_GET(r) (((r) < RW) ? w[((r) < RW) ? (r) : 0] : wl[((r) - RW) * CH_BS + tid])
#define W_PUT(r, val)                                   \
    do {                                                \
        if ((r) < RW) w[((r) < RW) ? (r) : 0] = (val);  \
        else wl[((r) - RW) * CH_BS + tid] = (val);      \
    } while (0)
    constexpr int PB = ChainShape<R2>::PB;
    constexpr int NB = ChainShape<R2>::NB;
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int slead;
    const int tid = threadIdx.x;
    int G = gridDim.x;
    int bid = blockIdx.x;
    GridRole role;
    if constexpr (ONEX) {
        if (!onex_enter(a, &slead, bid, G, role)) return;
    } else {
        role = grid_role(a.xcc_leader, a.epoch0, &slead);
    }
    // chunk2 == R2 * CH_BS: thread `tid` owns elements first + r*CH_BS, r < R2
    const int64_t first = (int64_t)bid * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[RW];
    double2 ring[2][PB];
    if constexpr (FND > 0 && CPLX) {
        chain_apply_banded_z<R2, FND>(a, first, [&](int r, double s0, double s1) { W_PUT(r, make_double2(s0, s1)); });
    } else if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { W_PUT(r, make_double2(s0, s1)); });
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = win2[(int64_t)r * CH_BS];
            double2 t;
            t.x = CH_OK(r) ? v.x : 0.0;
            t.y = CH_OK(r) ? v.y : 0.0;
            W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    if (a.presub) {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(a.bprev) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 p = p2[(int64_t)r * CH_BS];
            double2 t = W_GET(r);
            t.x = CH_OK(r) ? t.x - hk * p.x : 0.0;
            t.y = CH_OK(r) ? t.y - hk * p.y : 0.0;
            W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    unsigned epoch = a.epoch0;
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    const int total = a.ncol * a.sweeps;
    // prologue: first batch of the first column
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + a.col0 * a.ld) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = v2[(int64_t)i * CH_BS];
        CH_ISSUE_FENCE();
    }
    for (int t = 0; t < total; ++t) {
        const int64_t j = a.col0 + (t % a.ncol);
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + j * a.ld) + first;
        const double2* __restrict__ b2 = reinterpret_cast<const double2*>(a.B + j * a.ld) + first;
        // ---- dot phase: <v_j, w> ----
        double acc0 = 0.0, acc1 = 0.0;
        if (a.debug != 2)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // issue the next batch: v_j rows of batch b+1, or the first rows of b_j
            const double2* __restrict__ nx = (b + 1 < NB) ? v2 + (int64_t)(b + 1) * PB * CH_BS : b2;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[(b + 1) & 1][i] = (WL > 0 && b + 1 == NB) ? ld_nt2(nx + (int64_t)i * CH_BS) : nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double2 v = ring[b & 1][i];
                if (MASKED && !CH_OK(b * PB + i)) v = make_double2(0.0, 0.0);   // beyond the vector: whatever the block holds there (0 * NaN)
                const double2 wr = W_GET(b * PB + i);
                if (CPLX) {               // conj(v) * w: acc0 = re, acc1 = im
                    acc0 = fma(v.x, wr.x, acc0);
                    acc0 = fma(v.y, wr.y, acc0);
                    acc1 = fma(v.x, wr.y, acc1);
                    acc1 = fma(-v.y, wr.x, acc1);
                } else {
                    acc0 = fma(v.x, wr.x, acc0);
                    acc1 = fma(v.y, wr.y, acc1);
                }
            }
        }
        double alpha, alpha_i = 0.0;
        if (CPLX) {
            alpha = acc0;
            alpha_i = acc1;
            if constexpr (ONEX) {
                double pv[2] = {alpha, alpha_i};
                onex_sum<2>(pv, epoch++, a.gran, G, bid, a.err, smd);
                alpha = pv[0];
                alpha_i = pv[1];
            } else {
                grid_sum2(alpha, alpha_i, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
            }
            if (bid == 0 && tid == 0) {
                // first sweep assigns (the caller does not clear the H column for a chain launch)
                a.hdev[2 * j] = (t < a.ncol ? 0.0 : a.hdev[2 * j]) + alpha;
                a.hdev[2 * j + 1] = (t < a.ncol ? 0.0 : a.hdev[2 * j + 1]) + alpha_i;
            }
        } else {
            if constexpr (ONEX) {
                double pv[1] = {acc0 + acc1};
                onex_sum<1>(pv, epoch++, a.gran, G, bid, a.err, smd);
                alpha = pv[0];
            } else {
                const unsigned e_ = epoch++;
                alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                       : grid_sum<XR>(acc0 + acc1, e_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr,
                                                      a.xr.epoch0 + (e_ - a.epoch0));
            }
            if (bid == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        }
        if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
        // ---- update phase: w -= alpha * b_j ----
        const int64_t jn = a.col0 + ((t + 1) % a.ncol);
        const double2* __restrict__ vn = (t + 1 < total)
            ? reinterpret_cast<const double2*>(a.V + jn * a.ld) + first
            : reinterpret_cast<const double2*>(a.w_in) + first;       // harmless: valid memory
        if (a.debug != 2)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2* __restrict__ nx = (b + 1 < NB) ? b2 + (int64_t)(b + 1) * PB * CH_BS : vn;
            // long shapes (WL > 0: a 100 MB column, both of whose reads come from memory): the second read is the last
            // use - non-temporal, it then streams from the Infinity Cache, where the first (normal) read left it, at
            // 8.5 instead of 7.6 TB/s of requested bytes (tools/probe/reread_probe.hip: 22.6 vs 25.3 us per link)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[(b + 1) & 1][i] = (WL > 0 && b + 1 < NB) ? ld_nt2(nx + (int64_t)i * CH_BS) : nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = ring[b & 1][i];
                const int r = b * PB + i;
                double2 wr = W_GET(r);
                if (CPLX) {               // (alpha + i alpha_i) * (p.x + i p.y), NumPy's formula
                    const double tr = alpha * p.x - alpha_i * p.y;
                    const double ti = alpha * p.y + alpha_i * p.x;
                    wr.x = CH_OK(r) ? wr.x - tr : 0.0;
                    wr.y = CH_OK(r) ? wr.y - ti : 0.0;
                } else {
                    wr.x = CH_OK(r) ? wr.x - alpha * p.x : 0.0;
                    wr.y = CH_OK(r) ? wr.y - alpha * p.y : 0.0;
                }
                W_PUT(r, wr);
            }
        }
    }
    // norm: <w,w> or <w, D w>
    double acc = 0.0;
    if (a.dg != nullptr) {
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 d = d2[(int64_t)r * CH_BS];
            const double2 wr = W_GET(r);
            acc = fma(wr.x, d.x * wr.x, acc);
            acc = fma(wr.y, d.y * wr.y, acc);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 wr = W_GET(r);
            acc = fma(wr.x, wr.x, acc);
            acc = fma(wr.y, wr.y, acc);
        }
    }
    double h2;
    if constexpr (ONEX) {
        double pv[1] = {acc};
        onex_sum<1>(pv, epoch++, a.gran, G, bid, a.err, smd);
        h2 = pv[0];
    } else {
        const unsigned e_ = epoch++;
        h2 = grid_sum<XR>(acc, e_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr, a.xr.epoch0 + (e_ - a.epoch0));
    }
    const double h = sqrt(fabs(h2));
    if (bid == 0 && tid == 0) a.hdev[a.hnext] = h;
    double2* __restrict__ vn2 = reinterpret_cast<double2*>(a.vnext) + first;
    if (a.dg != nullptr) {
        double2* __restrict__ pn2 = reinterpret_cast<double2*>(a.pnext) + first;
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 d = d2[(int64_t)r * CH_BS];
                const double2 wr = W_GET(r);
                double2 o, m;
                o.x = wr.x / h;
                o.y = wr.y / h;
                m.x = (d.x * wr.x) / h;
                m.y = (d.y * wr.y) / h;
                st_nt2(pn2 + (int64_t)r * CH_BS, o);
                st_nt2(vn2 + (int64_t)r * CH_BS, m);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 wr = W_GET(r);
                double2 o;
                o.x = wr.x / h;
                o.y = wr.y / h;
                st_nt2(vn2 + (int64_t)r * CH_BS, o);
            }
        }
    }
    if (bid == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_SIGNAL_DONE(a);
    }
#undef W_PUT
#undef W_GET
#undef CH_OK
}

// ------------------------------------------------------------------------------------------
// k_mgs_chain_lds: the chain kernel with the head of every basis column parked in LDS.
//
// A link reads its column twice - v_j for the dot, then (without a preconditioner) the same v_j for
// the update, after the grid-wide reduction.  The workgroup owns a whole CU (512 threads, 160 KB of
// LDS, ~4 KB used), so the first LB batches of the column (LB*PB rows x 512 lanes x 16 B = up to
// 128 KB) are copied into LDS while the dot phase streams them and the update phase takes them from
// there.  The update phase also walks the batches in REVERSE order, so the last batch of the dot phase
// is still in the register ring and is used again without any load.  At N = 10^7 (R2 = 40, batches
// of 5 rows): 15 rows from LDS + 5 from the ring = half of the second read, 25 % less HBM traffic per
// link; for R2 <= 16 the second read disappears altogether.  Only for B == V (no preconditioner).
// Ring parity: batch NB-1 sits in ring[1]; the NG = NB - LB - 1 batches that still come from memory
// start in ring[0] and NG is even, so the next column's first batch lands in ring[0] again.
// ------------------------------------------------------------------------------------------
// (experiment knob: the first KH_CH_REUSE_SKIP of the batches that come back from memory for the update are
// streamed non-temporally as well, i.e. given up to HBM so that the others fit the 4 MB of L2 per XCD)
#ifndef KH_CH_REUSE_SKIP
#define KH_CH_REUSE_SKIP 0
#endif
// (experiment knobs, round 2: KH_CH_SKIP_LAST = 1 streams batch NB-2 non-temporally too - its second read is issued
// BEFORE the grid-wide reduction and hidden by it, so it may come from further away and leave the L2 to the other
// NG-1 batches; KH_CH_LDS_EARLY = 1 runs the LDS-parked part of the update while the first re-read batches fly)
#ifndef KH_CH_SKIP_LAST
#define KH_CH_SKIP_LAST 0
#endif
#ifndef KH_CH_LDS_EARLY
#define KH_CH_LDS_EARLY 0
#endif
template <int R2, bool CPLX = false>
struct ChainShapeLds {
    // 5 rows per batch for R2 = 40, like the plain kernel (the complex instantiation spills 36 registers
    // with it and is still 14 % faster than the plain kernel; 4-row batches spill 42)
    static constexpr int PB = ChainShape<R2>::PB;
    static constexpr int NB = R2 / PB;
    static constexpr int LB = NB >= 4 ? 3 : 1;                // leading batches kept in LDS
    static constexpr int NG = NB - LB - 1;                    // update batches that still come from memory
    static_assert((NB % 2) == 0 && NG >= 0 && (NG % 2) == 0, "ring parity must reset every phase");
    // R2 = 40 fills the register file to the last VGPR (w alone takes 160 of the 256); the 36 KB of LDS the parked
    // batches leave free take the last WL = 4 rows of w instead (k_mgs_chain's W_GET / W_PUT): 16 registers back,
    // no scratch traffic in the link loop (18 spilled registers with the operator in the prologue otherwise)
    static constexpr int WL = (R2 == 40 && !CPLX) ? 4 : 0;
    static constexpr size_t LDS_BYTES = (size_t)(LB * PB + WL) * CH_BS * sizeof(double2);
};

template <int R2, bool MASKED, bool CPLX = false, int FND = 0, bool XR = false>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain_lds(ChainArgs a) {
    static_assert(FND == 0 || !MASKED, "the fused operator exists for the padded kernels");
    static_assert(!XR || !CPLX, "the cross-rank stage exists for real vectors");
    constexpr int PB = ChainShapeLds<R2, CPLX>::PB;
    constexpr int NB = ChainShapeLds<R2, CPLX>::NB;
    constexpr int LB = ChainShapeLds<R2, CPLX>::LB;
    constexpr int NG = ChainShapeLds<R2, CPLX>::NG;
    constexpr int WL = ChainShapeLds<R2, CPLX>::WL;
    constexpr int RW = R2 - WL;                   // rows of w in registers
    extern __shared__ __attribute__((aligned(16))) double2 vlds[];   // [LB*PB parked rows + WL rows of w][CH_BS]
    double2* const wl = vlds + (size_t)LB * PB * CH_BS;
#define W_GET(r) (((r) < RW) ? w[((r) < RW) ? (r) : 0] : wl[((r) - RW) * CH_BS + tid])
#define W_PUT(r, val)                                   \
    do {                                                \
        if ((r) < RW) w[((r) < RW) ? (r) : 0] = (val);  \
        else wl[((r) - RW) * CH_BS + tid] = (val);      \
    } while (0)
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int slead;
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const GridRole role = grid_role(a.xcc_leader, a.epoch0, &slead);
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    // a batch that will be read again (update phase) is loaded normally so that the Infinity Cache
    // keeps it; everything that is used once (the batches that live on in LDS / the ring, the second
    // read itself, w) is loaded non-temporally and does not evict it
#define CH_LD(ptr, reuse) ((reuse) ? *(ptr) : ld_nt2(ptr))
    double2 w[RW];
    double2 ring[2][PB];
    if constexpr (FND > 0 && CPLX) {
        chain_apply_banded_z<R2, FND>(a, first, [&](int r, double s0, double s1) { W_PUT(r, make_double2(s0, s1)); });
    } else if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { W_PUT(r, make_double2(s0, s1)); });
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);
            double2 t;
            t.x = CH_OK(r) ? v.x : 0.0;
            t.y = CH_OK(r) ? v.y : 0.0;
            W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    if (a.presub) {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(a.bprev) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 p = p2[(int64_t)r * CH_BS];
            double2 t = W_GET(r);
            t.x = CH_OK(r) ? t.x - hk * p.x : 0.0;
            t.y = CH_OK(r) ? t.y - hk * p.y : 0.0;
            W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    unsigned epoch = a.epoch0;
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    const int total = a.ncol * a.sweeps;
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + a.col0 * a.ld) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = CH_LD(v2 + (int64_t)i * CH_BS, false);
        CH_ISSUE_FENCE();
    }
#if defined(KH_CHAIN_TRACE) && KH_CHAIN_TRACE == 2
    unsigned long long tr_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // [8]: the previous stamp
#endif
    for (int t = 0; t < total; ++t) {
        const int64_t j = a.col0 + (t % a.ncol);
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + j * a.ld) + first;
        const int64_t jn = a.col0 + ((t + 1) % a.ncol);
        const double2* __restrict__ vn = (t + 1 < total)
            ? reinterpret_cast<const double2*>(a.V + jn * a.ld) + first
            : reinterpret_cast<const double2*>(a.w_in) + first;       // harmless: valid memory
        // ---- dot phase: <v_j, w>; the first LB batches are parked in LDS on the way ----
        double acc0 = 0.0, acc1 = 0.0;
        CH_STAMP(a, t, total, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // next: v_j batch b+1; after the last one the first update batch that is not in LDS
            // (or, if the whole column is, the next column's first batch)
            const double2* __restrict__ nx = (b + 1 < NB) ? v2 + (int64_t)(b + 1) * PB * CH_BS
                                                            : (NG > 0 ? v2 + (int64_t)(NB - 2) * PB * CH_BS : vn);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[(b + 1) & 1][i] = CH_LD(nx + (int64_t)i * CH_BS, (b + 1 < NB) && (b + 1 >= LB + KH_CH_REUSE_SKIP) && (b + 1 <= NB - 2 - KH_CH_SKIP_LAST));
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double2 v = ring[b & 1][i];
                if (MASKED && !CH_OK(b * PB + i)) v = make_double2(0.0, 0.0);   // beyond the vector: whatever the block holds there
                if (b < LB) vlds[(b * PB + i) * CH_BS + tid] = v;
                const double2 wr = W_GET(b * PB + i);
                if (CPLX) {
                    acc0 = fma(v.x, wr.x, acc0);
                    acc0 = fma(v.y, wr.y, acc0);
                    acc1 = fma(v.x, wr.y, acc1);
                    acc1 = fma(-v.y, wr.x, acc1);
                } else {
                    acc0 = fma(v.x, wr.x, acc0);
                    acc1 = fma(v.y, wr.y, acc1);
                }
            }
        }
        double alpha, alpha_i = 0.0;
        CH_STAMP(a, t, total, 1);
        if (CPLX) {
            alpha = acc0;
            alpha_i = acc1;
            grid_sum2(alpha, alpha_i, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
            if (blockIdx.x == 0 && tid == 0) {
                // first sweep assigns (the caller does not clear the H column for a chain launch)
                a.hdev[2 * j] = (t < a.ncol ? 0.0 : a.hdev[2 * j]) + alpha;
                a.hdev[2 * j + 1] = (t < a.ncol ? 0.0 : a.hdev[2 * j + 1]) + alpha_i;
            }
        } else {
#if defined(KH_CHAIN_TRACE) && KH_CHAIN_TRACE == 2
            alpha = grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res, tr_acc);
#elif defined(KH_CHAIN_TRACE)
            alpha = grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res,
                             a.trace ? a.trace + (((size_t)blockIdx.x * total) + t) * 16 : nullptr);
#else
            const unsigned e_ = epoch++;
            alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                   : grid_sum<XR>(acc0 + acc1, e_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr,
                                                  a.xr.epoch0 + (e_ - a.epoch0));
#endif
            if (blockIdx.x == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        }
        if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
        CH_STAMP(a, t, total, 5);
        // ---- update phase: w -= alpha * v_j, batches in reverse order ----
#define CH_UPD(r, p)                                              \
    do {                                                          \
        double2 wr_ = W_GET(r);                                   \
        if (CPLX) {                                               \
            const double tr = alpha * (p).x - alpha_i * (p).y;    \
            const double ti = alpha * (p).y + alpha_i * (p).x;    \
            wr_.x = CH_OK(r) ? wr_.x - tr : 0.0;                  \
            wr_.y = CH_OK(r) ? wr_.y - ti : 0.0;                  \
        } else {                                                  \
            wr_.x = CH_OK(r) ? wr_.x - alpha * (p).x : 0.0;       \
            wr_.y = CH_OK(r) ? wr_.y - alpha * (p).y : 0.0;       \
        }                                                         \
        W_PUT(r, wr_);                                            \
    } while (0)
        // (a) the last batch of the dot phase is still in ring[1]
#pragma unroll
        for (int i = 0; i < PB; ++i) CH_UPD((NB - 1) * PB + i, ring[1][i]);
        // (b) batches NB-2 ... LB from memory through the ring
        constexpr bool EARLY = (KH_CH_LDS_EARLY != 0) && NG >= 2;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int b = NB - 2 - g;
            const double2* __restrict__ nx = (g + 1 < NG) ? v2 + (int64_t)(b - 1) * PB * CH_BS : vn;
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[(g + 1) & 1][i] = CH_LD(nx + (int64_t)i * CH_BS, false);
            CH_ISSUE_FENCE();
            if (EARLY && g == 1) {
                // two re-read batches are in flight (this iteration's load and the previous one's): walk the
                // LDS-parked head of the column while they are on their way
#pragma unroll
                for (int bb = LB - 1; bb >= 0; --bb) {
#pragma unroll
                    for (int i = 0; i < PB; ++i) {
                        const double2 p = vlds[(bb * PB + i) * CH_BS + tid];
                        CH_UPD(bb * PB + i, p);
                    }
                    CH_ISSUE_FENCE();
                }
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) CH_UPD(b * PB + i, ring[g & 1][i]);
        }
        CH_STAMP(a, t, total, 6);
        // (c) the head of the column from LDS (own entries: no barrier needed), one batch of reads at
        //     a time (hoisted all together they would spill)
        if (!EARLY) {
#pragma unroll
            for (int b = LB - 1; b >= 0; --b) {
                CH_ISSUE_FENCE();
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const double2 p = vlds[(b * PB + i) * CH_BS + tid];
                    CH_UPD(b * PB + i, p);
                }
            }
        }
#undef CH_UPD
        CH_STAMP(a, t, total, 7);
    }
#if defined(KH_CHAIN_TRACE) && KH_CHAIN_TRACE == 2
    if (a.trace != nullptr && (tid == 0 || tid == CH_BS - 64)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.trace[((size_t)blockIdx.x * 2 + (tid ? 1 : 0)) * 8 + i] = tr_acc[i];
    }
#endif
    // norm: <w,w> or <w, D w>
    double acc = 0.0;
    if (a.dg != nullptr) {
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 d = d2[(int64_t)r * CH_BS];
            const double2 wr = W_GET(r);
            acc = fma(wr.x, d.x * wr.x, acc);
            acc = fma(wr.y, d.y * wr.y, acc);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 wr = W_GET(r);
            acc = fma(wr.x, wr.x, acc);
            acc = fma(wr.y, wr.y, acc);
        }
    }
    const unsigned en_ = epoch++;
    const double h2 = grid_sum<XR>(acc, en_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr, a.xr.epoch0 + (en_ - a.epoch0));
    const double h = sqrt(fabs(h2));
    if (blockIdx.x == 0 && tid == 0) a.hdev[a.hnext] = h;
    double2* __restrict__ vn2 = reinterpret_cast<double2*>(a.vnext) + first;
    if (a.dg != nullptr) {
        double2* __restrict__ pn2 = reinterpret_cast<double2*>(a.pnext) + first;
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 d = d2[(int64_t)r * CH_BS];
                const double2 wr = W_GET(r);
                double2 o, m;
                o.x = wr.x / h;
                o.y = wr.y / h;
                m.x = (d.x * wr.x) / h;
                m.y = (d.y * wr.y) / h;
                st_nt2(pn2 + (int64_t)r * CH_BS, o);
                st_nt2(vn2 + (int64_t)r * CH_BS, m);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 wr = W_GET(r);
                double2 o;
                o.x = wr.x / h;
                o.y = wr.y / h;
                st_nt2(vn2 + (int64_t)r * CH_BS, o);
            }
        }
    }
    if (blockIdx.x == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_SIGNAL_DONE(a);
    }
#undef W_PUT
#undef W_GET
#undef CH_LD
#undef CH_OK
}

// ------------------------------------------------------------------------------------------
// k_mgs_chain_pf: the LDS chain kernel with the memory system kept busy THROUGH the update phase.
//
// In k_mgs_chain_lds HBM idles from the end of a column's dot phase to the start of the next one: during
// the grid-wide reduction (nothing to fetch into) and during the update, which is served by LDS, the ring
// and L2 (phase stamps of the diagnostic build, tools/chain_trace.py: about 6 of 17.4 us per link at
// N = 10^7).  Here
//   * the dot phase issues batch b+2 into the ring slot batch b has just left, so it ends with the LAST TWO
//     batches still in registers (nothing is re-fetched into the ring for the update);
//   * the update starts with the LB batches parked in LDS, which empties those slots; the NG batches that
//     are on chip nowhere are then fetched again (L2 hits for the most part: they were streamed just before
//     the ring's two) straight INTO THE EMPTIED LDS SLOTS by LDS-DMA (global_load_lds_dwordx4, inline asm:
//     no VGPR destination, and hipcc's own s_waitcnt bookkeeping for the ring stays exact because the DMAs
//     are always the OLDEST vector-memory operations in flight);
//   * the two ring slots are consumed next and each is refilled at once with the NEXT column's first two
//     batches - 2*PB rows x 8 KB of HBM reads in flight while the DMA'd batches are waited for (one counted
//     s_waitcnt vmcnt(2*PB)) and consumed, and while the next dot phase starts.
// On chip at the reduction: LB batches in LDS + 2 in the ring (R2 = 40: 25 of 40 rows instead of 20), the
// second read shrinks to NG = NB - 2 - LB batches (15 rows instead of 20).  Arithmetic and the order of
// every floating-point operation are those of k_mgs_chain (each row is updated exactly once per link, the
// dot sums the batches in ascending order): bit-identical results.  Only for B == V (no preconditioner).
// ------------------------------------------------------------------------------------------
template <int R2>
struct ChainShapePf {
    static constexpr int PB = ChainShape<R2>::PB;
    static constexpr int NB = R2 / PB;
    static constexpr int LB = (NB - 2 < 3) ? NB - 2 : 3;      // leading batches parked in LDS
    static constexpr int NG = NB - 2 - LB;                    // batches fetched again (into the emptied LDS slots)
    static_assert((NB % 2) == 0 && NB >= 2 && LB >= 0 && NG >= 0 && NG <= LB, "shape");
    // what is left of the CU's 160 KB of LDS takes the first SP rows of the NEXT column while the grid-wide
    // reduction is in flight (the only stretch of a link with nothing else to fetch into)
    static constexpr int ROW_BYTES = CH_BS * (int)sizeof(double2);
    static constexpr int STATIC_LDS = 2 * (CH_BS / 64) * 8 + 2 * CH_GMAX * 4 + 256;      // smd + smu + slack
    static constexpr int SPARE = (160 * 1024 - STATIC_LDS - LB * PB * ROW_BYTES) / ROW_BYTES;
    // (KH_PF_SP_CAP, experiment: cap the rows prefetched in front of the reduction's polls; 0 = none)
#ifndef KH_PF_SP_CAP
#define KH_PF_SP_CAP 64
#endif
    static constexpr int SP0 = SPARE < PB ? SPARE : PB;
    static constexpr int SP = SP0 < KH_PF_SP_CAP ? SP0 : KH_PF_SP_CAP;
    static_assert(SP0 >= 1, "at least one spare row");
    static constexpr size_t LDS_BYTES = (size_t)(LB * PB + SP) * ROW_BYTES;
};

typedef __attribute__((address_space(3))) double2 ch_lds_double2;

// one wave-row (64 lanes x 16 B) global -> LDS at the wave-uniform byte address lds_dst (+ lane * 16)
__device__ __forceinline__ void ch_dma16(const double2* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int R2, bool MASKED, bool CPLX = false, int FND = 0, bool ONEX = false>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain_pf(ChainArgs a) {
    static_assert(FND == 0 || (!MASKED && !CPLX), "the fused operator exists for the padded real kernel");
    constexpr int PB = ChainShapePf<R2>::PB;
    constexpr int NB = ChainShapePf<R2>::NB;
    constexpr int LB = ChainShapePf<R2>::LB;
    constexpr int NG = ChainShapePf<R2>::NG;
    constexpr int SP = ChainShapePf<R2>::SP;
    extern __shared__ __attribute__((aligned(16))) double2 vlds[];   // [LB*PB parked rows + SP prefetched rows][CH_BS]
    double2* const vsp = vlds + (size_t)LB * PB * CH_BS;
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int slead;
    const int tid = threadIdx.x;
    int G = gridDim.x;
    int bid = blockIdx.x;
    GridRole role;
    if constexpr (ONEX) {
        if (!onex_enter(a, &slead, bid, G, role)) return;
    } else {
        role = grid_role(a.xcc_leader, a.epoch0, &slead);
    }
    const int64_t first = (int64_t)bid * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
    // LDS byte address of this wave's 64 lanes in row 0 (the DMA destination is wave-uniform + lane * 16)
    const unsigned lds_wave = (unsigned)(size_t)((ch_lds_double2*)vlds) +
                              (unsigned)__builtin_amdgcn_readfirstlane((tid >> 6) * 64 * (int)sizeof(double2));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
#define CH_LD(ptr, reuse) ((reuse) ? *(ptr) : ld_nt2(ptr))
    double2 w[R2];
    double2 ring[2][PB];
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { w[r] = make_double2(s0, s1); });
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);
            w[r].x = CH_OK(r) ? v.x : 0.0;
            w[r].y = CH_OK(r) ? v.y : 0.0;
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    if (a.presub) {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(a.bprev) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 p = p2[(int64_t)r * CH_BS];
            w[r].x = CH_OK(r) ? w[r].x - hk * p.x : 0.0;
            w[r].y = CH_OK(r) ? w[r].y - hk * p.y : 0.0;
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    unsigned epoch = a.epoch0;
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    const int total = a.ncol * a.sweeps;
    // a batch that comes back from memory for the update is loaded normally (L2 keeps it), everything that is
    // used once from memory - the batches that live on in LDS / the ring - non-temporally
#define CH_REUSE(b) ((b) >= LB && (b) < LB + NG)
    {   // the first column's first two batches (the first SP rows by LDS-DMA, as in every later link)
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + a.col0 * a.ld) + first;
#pragma unroll
        for (int i = 0; i < SP; ++i)
            ch_dma16(v2 + (int64_t)i * CH_BS, lds_wave + (unsigned)(((LB * PB + i) * CH_BS) * sizeof(double2)));
#pragma unroll
        for (int i = SP; i < PB; ++i) ring[0][i] = CH_LD(v2 + (int64_t)i * CH_BS, CH_REUSE(0));
        CH_ISSUE_FENCE();
        if (NB > 1) {
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[1][i] = CH_LD(v2 + (int64_t)(PB + i) * CH_BS, CH_REUSE(1));
            CH_ISSUE_FENCE();
        }
    }
    for (int t = 0; t < total; ++t) {
        const int64_t j = a.col0 + (t % a.ncol);
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + j * a.ld) + first;
        const int64_t jn = a.col0 + ((t + 1) % a.ncol);
        const double2* __restrict__ vn = (t + 1 < total)
            ? reinterpret_cast<const double2*>(a.V + jn * a.ld) + first
            : reinterpret_cast<const double2*>(a.w_in) + first;       // harmless: valid memory
        // ---- dot phase: <v_j, w>; batch b sits in ring[b & 1], batch b+2 follows it into the same slot ----
        double acc0 = 0.0, acc1 = 0.0;
        // everything older than the two ring batches in flight has landed: the prefetched rows among it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PB - SP) : "memory");
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double2 v = (b == 0 && i < SP) ? vsp[i * CH_BS + tid] : ring[b & 1][i];
                if (MASKED && !CH_OK(b * PB + i)) v = make_double2(0.0, 0.0);   // beyond the vector: whatever the block holds there
                if (NB == 2 && b == 0 && i < SP) ring[0][i] = v;      // (two-batch shapes update from the ring)
                if (b < LB) vlds[(b * PB + i) * CH_BS + tid] = v;
                if (CPLX) {
                    acc0 = fma(v.x, w[b * PB + i].x, acc0);
                    acc0 = fma(v.y, w[b * PB + i].y, acc0);
                    acc1 = fma(v.x, w[b * PB + i].y, acc1);
                    acc1 = fma(-v.y, w[b * PB + i].x, acc1);
                } else {
                    acc0 = fma(v.x, w[b * PB + i].x, acc0);
                    acc1 = fma(v.y, w[b * PB + i].y, acc1);
                }
            }
            if (b + 2 < NB) {
                CH_ISSUE_FENCE();
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    ring[b & 1][i] = CH_LD(v2 + (int64_t)((b + 2) * PB + i) * CH_BS, CH_REUSE(b + 2));
                CH_ISSUE_FENCE();
            }
        }
        // the next column's first SP rows: on their way while the coefficient is summed over the grid
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < SP; ++i)
            ch_dma16(vn + (int64_t)i * CH_BS, lds_wave + (unsigned)(((LB * PB + i) * CH_BS) * sizeof(double2)));
        double alpha, alpha_i = 0.0;
        if (CPLX) {
            alpha = acc0;
            alpha_i = acc1;
            if constexpr (ONEX) {
                double pv[2] = {alpha, alpha_i};
                onex_sum<2>(pv, epoch++, a.gran, G, bid, a.err, smd);
                alpha = pv[0];
                alpha_i = pv[1];
            } else {
                grid_sum2(alpha, alpha_i, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
            }
            if (bid == 0 && tid == 0) {
                a.hdev[2 * j] = (t < a.ncol ? 0.0 : a.hdev[2 * j]) + alpha;
                a.hdev[2 * j + 1] = (t < a.ncol ? 0.0 : a.hdev[2 * j + 1]) + alpha_i;
            }
        } else {
            if constexpr (ONEX) {
                double pv[1] = {acc0 + acc1};
                onex_sum<1>(pv, epoch++, a.gran, G, bid, a.err, smd);
                alpha = pv[0];
            } else {
                alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                       : grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
            }
            if (bid == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        }
        if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
        // ---- update phase: w -= alpha * v_j ----
#define CH_UPD(r, p)                                              \
    do {                                                          \
        if (CPLX) {                                               \
            const double tr = alpha * (p).x - alpha_i * (p).y;    \
            const double ti = alpha * (p).y + alpha_i * (p).x;    \
            w[r].x = CH_OK(r) ? w[r].x - tr : 0.0;                \
            w[r].y = CH_OK(r) ? w[r].y - ti : 0.0;                \
        } else {                                                  \
            w[r].x = CH_OK(r) ? w[r].x - alpha * (p).x : 0.0;     \
            w[r].y = CH_OK(r) ? w[r].y - alpha * (p).y : 0.0;     \
        }                                                         \
    } while (0)
        // (a) the head of the column from LDS (own entries: no barrier needed), one batch of reads at a time
#pragma unroll
        for (int b = 0; b < LB; ++b) {
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = vlds[(b * PB + i) * CH_BS + tid];
                CH_UPD(b * PB + i, p);
            }
        }
        // (b) the batches that are nowhere on chip come back into the emptied slots by LDS-DMA
        if constexpr (NG > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slots have been read
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    ch_dma16(v2 + (int64_t)((LB + g) * PB + i) * CH_BS,
                             lds_wave + (unsigned)(((g * PB + i) * CH_BS) * sizeof(double2)));
        }
        // (c) the two batches the ring still holds; each slot goes straight to the next column
        if (NB > 1) {
#pragma unroll
            for (int i = 0; i < PB; ++i) CH_UPD((NB - 2) * PB + i, ring[0][i]);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = SP; i < PB; ++i) ring[0][i] = CH_LD(vn + (int64_t)i * CH_BS, CH_REUSE(0));
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) CH_UPD((NB - 1) * PB + i, ring[1][i]);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[1][i] = CH_LD(vn + (int64_t)(PB + i) * CH_BS, CH_REUSE(1));
            CH_ISSUE_FENCE();
        }
        // (d) the DMA'd batches: everything older than the 2*PB ring loads just issued has landed
        if constexpr (NG > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PB - SP) : "memory");
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                CH_ISSUE_FENCE();
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const double2 p = vlds[(g * PB + i) * CH_BS + tid];
                    CH_UPD((LB + g) * PB + i, p);
                }
            }
        }
#undef CH_UPD
    }
    // norm: <w,w> or <w, D w>
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        acc = fma(w[r].x, w[r].x, acc);
        acc = fma(w[r].y, w[r].y, acc);
    }
    double h2;
    if constexpr (ONEX) {
        double pv[1] = {acc};
        onex_sum<1>(pv, epoch++, a.gran, G, bid, a.err, smd);
        h2 = pv[0];
    } else {
        h2 = grid_sum(acc, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
    }
    const double h = sqrt(fabs(h2));
    if (bid == 0 && tid == 0) a.hdev[a.hnext] = h;
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
    if (bid == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_SIGNAL_DONE(a);
    }
#undef CH_REUSE
#undef CH_LD
#undef CH_OK
}

// ------------------------------------------------------------------------------------------
// k_mgs_chain_small: the chain for SHORT vectors (4 / 8 rows per lane), where a link is latency, not bandwidth.
//
// With w in 16 or 32 registers there is room for whole COLUMNS: a ring of LA + 1 of them.  Column t + LA + 1 is
// requested when column t has been used, i.e. LA links (LA grid-wide sums) before it is needed - the memory latency
// the one-column look-ahead of k_mgs_chain_pf leaves exposed (a link took 1.4 - 1.9 us around a 0.84 us sum) is
// hidden - and the update takes the column from the same registers the dot used: every column is read once.
// Same accumulators, same row order, same sums as the other chain kernels: the same bits.  B == V only (no
// preconditioner), real data; ONEX as above, FND > 0: the banded operator in the prologue.
// ------------------------------------------------------------------------------------------
template <int R2, int LA, bool MASKED, int FND = 0, bool ONEX = false>
__global__ __launch_bounds__(ONEX ? CH_BS + 64 : CH_BS) void k_mgs_chain_small(ChainArgs a) {
    static_assert(FND == 0 || !MASKED, "the fused operator exists for padded blocks");
    constexpr int NS = LA + 1;
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int slead;
    const int tid = threadIdx.x;
    int G = gridDim.x;
    int bid = blockIdx.x;
    GridRole role;
    if constexpr (ONEX) {
        if (!onex_enter(a, &slead, bid, G, role)) return;
    } else {
        role = grid_role(a.xcc_leader, a.epoch0, &slead);
    }
    unsigned epoch = a.epoch0;
    const int total = a.ncol * a.sweeps;
    if constexpr (ONEX) {
        if (tid >= CH_BS) {          // the communication wave: the sums and the H entries, no rows
            const bool writer = bid == 0 && tid == CH_BS;
            for (int t = 0; t < total; ++t) {
                const double alpha = onex_sum_comm(epoch++, a.gran, G, bid, a.err, smd);
                const int64_t j = a.col0 + (t % a.ncol);
                if (writer) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
            }
            const double h2 = onex_sum_comm(epoch++, a.gran, G, bid, a.err, smd);
            if (writer) a.hdev[a.hnext] = sqrt(fabs(h2));
            if (bid == 0 && a.hpin != nullptr) {
                __syncthreads();          // (the copy at the end, by the working waves)
                if (a.donepin != nullptr) __syncthreads();          // (... and the barrier of CH_SIGNAL_DONE)
            }
            return;
        }
    }
    const int64_t first = (int64_t)bid * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    // the first columns are requested before anything else: they arrive under the operator / the load of w
    double2 ring[NS][R2];
    // column of link t (beyond the last link: the last column again - a harmless reload instead of a branch)
#define CH_COL(t) (reinterpret_cast<const double2*>(a.V + (a.col0 + (((t) < total ? (t) : total - 1) % a.ncol)) * a.ld) + first)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double2* __restrict__ c = CH_COL(s);
#pragma unroll
        for (int r = 0; r < R2; ++r) ring[s][r] = c[(int64_t)r * CH_BS];
    }
    CH_ISSUE_FENCE();
    double2 w[R2];
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { w[r] = make_double2(s0, s1); });
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);
            w[r].x = CH_OK(r) ? v.x : 0.0;
            w[r].y = CH_OK(r) ? v.y : 0.0;
        }
    }
    if (a.presub) {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(a.bprev) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 p = p2[(int64_t)r * CH_BS];
            w[r].x = CH_OK(r) ? w[r].x - hk * p.x : 0.0;
            w[r].y = CH_OK(r) ? w[r].y - hk * p.y : 0.0;
        }
    }
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    for (int tb = 0; tb < total; tb += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = tb + s;
            if (t < total) {
                const int64_t j = a.col0 + (t % a.ncol);
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    double2 v = ring[s][r];
                    if (MASKED && !CH_OK(r)) v = make_double2(0.0, 0.0);
                    ring[s][r] = v;
                    acc0 = fma(v.x, w[r].x, acc0);
                    acc1 = fma(v.y, w[r].y, acc1);
                }
                double alpha;
                if constexpr (ONEX) {
                    alpha = onex_sum_work(acc0 + acc1, smd);
                } else {
                    alpha = grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
                    if (bid == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
                }
                if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    w[r].x = CH_OK(r) ? w[r].x - alpha * ring[s][r].x : 0.0;
                    w[r].y = CH_OK(r) ? w[r].y - alpha * ring[s][r].y : 0.0;
                }
                // the slot is free: the column LA + 1 links ahead
                const double2* __restrict__ c = CH_COL(t + NS);
#pragma unroll
                for (int r = 0; r < R2; ++r) ring[s][r] = c[(int64_t)r * CH_BS];
                CH_ISSUE_FENCE();
            }
        }
    }
#undef CH_COL
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        acc = fma(w[r].x, w[r].x, acc);
        acc = fma(w[r].y, w[r].y, acc);
    }
    double h2;
    if constexpr (ONEX) {
        h2 = onex_sum_work(acc, smd);
    } else {
        h2 = grid_sum(acc, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
    }
    const double h = sqrt(fabs(h2));
    if (!ONEX && bid == 0 && tid == 0) a.hdev[a.hnext] = h;
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
    if (bid == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_SIGNAL_DONE(a);
    }
#undef CH_OK
}

// ------------------------------------------------------------------------------------------
// Register-resident PANEL (classical) Gram-Schmidt: two launches per sweep, any number of ranks.
//
//   k_cgs_dots    w in registers; streams ALL columns once; per column one wave64 partial per
//                 wave (no workgroup barrier, no grid synchronisation at all) -> part[j][G*8]
//   (k_reduce_partials over the G*8 wave partials of every column -> h; ncclAllReduce on N GPUs)
//   k_cgs_update  w in registers; streams all columns again: w -= h_j b_j (multiply-then-subtract,
//                 left to right, like the reference's update), then writes w back together with
//                 the wave partials of <w,w> (or <w, D w>, storing D w)
//
// Traffic per sweep: 16 N per column + 32 N, against 16 N per column + 24 N per 16-column chunk
// for the chunked panel kernels, and 2 launches instead of 2*ceil((k+1)/16): this is the form
// that scales over xGMI (one all-reduce per sweep) and keeps launch count flat when the shards
// get short.
// ------------------------------------------------------------------------------------------
// Batches of the panel kernels: no phase structure to respect here, so short vectors (R2 <= 16, i.e.
// the N/4 and N/8 shards of the bench problem) stream 8-row batches through the ring instead of the
// chain's 4: more bytes in flight per workgroup where there are few workgroups (153 for 256 CUs at
// N/8).  Measured (cgs GMRES(100), tools/shard_sweep.sh): N/8 4778 -> 5100 it/s, N/4 2647 -> 2680;
// 12-row batches at R2 = 24 (N/2) lost 2.5 %, 16-row batches at R2 = 16 gained nothing.  NB may be odd:
// the ring parity then alternates from column to column and the column loop is unrolled by two.
template <int R2>
struct CgsShape {
    static constexpr int PB = R2 <= 8 ? R2 : (R2 == 16 ? 8 : ChainShape<R2>::PB);
    static constexpr int NB = R2 / PB;
    static_assert(NB * PB == R2, "batches must tile the rows");
};

struct CgsArgs {
    int64_t n2, chunk2;
    const double* Vb;      // column base: Vb + j*ld  (V for the dots, B for the update)
    int64_t ld;
    int64_t col0;
    int ncol;
    double* w;             // in (dots) / in+out (update)
    const double* coef;    // update: h[0..ncol)
    double* part;          // dots: [ncol][pstride] wave partials; update: [pstride] norm partials
    int pstride;
    const double* dg;      // update: Jacobi diagonal or nullptr
    double* mw;            // update: D w
    int reverse;           // update: walk the columns last to first (see k_cgs_update)
    int nt_cols;           // dots: non-temporal column loads (a panel far larger than the Infinity Cache)
    const double* x2;      // dots, X2 instantiations: a second right-hand side (v_k: the new row of the Gram table) ...
    double* part2;         // ... and where its wave partials go: [ncol][pstride]
};

// WL > 0: as in k_mgs_chain, the last WL rows of w live in LDS (shards / vectors beyond 10.48 M rows per GPU)
#define CGS_W_GET(r) (((r) < RW) ? w[((r) < RW) ? (r) : 0] : wl[((r) - RW) * CH_BS + tid])
#define CGS_W_PUT(r, val)                               \
    do {                                                \
        if ((r) < RW) w[((r) < RW) ? (r) : 0] = (val);  \
        else wl[((r) - RW) * CH_BS + tid] = (val);      \
    } while (0)

// CPLX: every double2 is one complex number: <v, w> = conj(v) w as two sums per column (partials of column t in
// rows 2t (re) and 2t+1 (im) of `part`, so the reduction leaves (re, im) pairs), the update multiplies by a complex
// coefficient (NumPy's product: (ac - bd, ad + bc)); the norm is that of the real view either way.
// X2 (round 4; real data, shards up to R2 = 24 rows per lane): the same pass also forms <v_j, x> for a second vector x whose
// rows sit in LDS behind the rows of w (sixteen of them; the other eight, at 24 rows per lane, in registers; at 32 rows the
// kernel spills 102 registers and is not instantiated) - x = v_k, the newest basis vector: its inner products with the older columns are the new
// row of the Gram table the one-reduction form of reference-order Gram-Schmidt corrects its coefficients with
// (krylov_hip.hip: try_lowsync_mgs).  One more read of one column per step, no second pass over the basis.
template <int R2, bool MASKED, bool NTC, int WL = 0, bool CPLX = false, bool X2 = false>
__global__ __launch_bounds__(CH_BS) void k_cgs_dots(CgsArgs a) {
    static_assert(!X2 || (!CPLX && WL == 0 && R2 <= 24), "second right-hand side: real data, all of w in registers, x fits LDS + registers");
    constexpr int XR = (X2 && R2 > 16) ? R2 - 16 : 0;      // rows of x in registers (24 rows per lane: LDS takes sixteen)
    constexpr int RW = R2 - WL;
    extern __shared__ __attribute__((aligned(16))) double2 wl[];   // [WL][CH_BS] rows of w, then (X2) [R2][CH_BS] rows of x
    constexpr int PB = CgsShape<R2>::PB;
    constexpr int NB = CgsShape<R2>::NB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[RW];
    double2 ring[2][PB];
    {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);   // (the columns stay normal loads: the update pass re-reads them)
            double2 t;
            t.x = CH_OK(r) ? v.x : 0.0;
            t.y = CH_OK(r) ? v.y : 0.0;
            CGS_W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    double2* const xl = wl + (size_t)WL * CH_BS;      // (X2) this lane's rows of x: own entries only, no barrier
    double2 xr[XR > 0 ? XR : 1];
#define CGS_X_GET(r) (((r) < XR) ? xr[((r) < XR) ? (r) : 0] : xl[((r) - XR) * CH_BS + tid])
    if constexpr (X2) {
        const double2* __restrict__ x2 = reinterpret_cast<const double2*>(a.x2) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = x2[(int64_t)r * CH_BS];
            double2 t;
            t.x = CH_OK(r) ? v.x : 0.0;
            t.y = CH_OK(r) ? v.y : 0.0;
            if (r < XR) xr[(r < XR) ? r : 0] = t;
            else xl[(r - XR) * CH_BS + tid] = t;
        }
    }
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.Vb + a.col0 * a.ld) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = NTC ? ld_nt2(v2 + (int64_t)i * CH_BS) : v2[(int64_t)i * CH_BS];
        CH_ISSUE_FENCE();
    }
    const int slot = blockIdx.x * (CH_BS / 64) + wid;
    // one column, the ring slot its first batch sits in given as a compile-time constant
    auto column = [&](auto par, int t) {
        constexpr int P0 = decltype(par)::value;
        const double2* __restrict__ v2 =
            reinterpret_cast<const double2*>(a.Vb + (a.col0 + t) * a.ld) + first;
        const double2* __restrict__ vn =
            reinterpret_cast<const double2*>(a.Vb + (a.col0 + (t + 1 < a.ncol ? t + 1 : t)) * a.ld) + first;
        double acc0 = 0.0, acc1 = 0.0, aci0 = 0.0, aci1 = 0.0;
        double acx0 = 0.0, acx1 = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2* __restrict__ nx = (b + 1 < NB) ? v2 + (int64_t)(b + 1) * PB * CH_BS : vn;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[(P0 + b + 1) & 1][i] = NTC ? ld_nt2(nx + (int64_t)i * CH_BS) : nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double2 v = ring[(P0 + b) & 1][i];
                if (MASKED && !CH_OK(b * PB + i)) v = make_double2(0.0, 0.0);   // beyond the vector: whatever the block holds there
                const double2 wr = CGS_W_GET(b * PB + i);
                acc0 = fma(v.x, wr.x, acc0);
                acc1 = fma(v.y, wr.y, acc1);
                if (CPLX) {
                    aci0 = fma(v.x, wr.y, aci0);
                    aci1 = fma(v.y, wr.x, aci1);
                }
                if constexpr (X2) {
                    const double2 xv = CGS_X_GET(b * PB + i);
                    acx0 = fma(v.x, xv.x, acx0);
                    acx1 = fma(v.y, xv.y, acx1);
                }
            }
        }
        if constexpr (X2) {
            const double sx = wave_sum(acx0 + acx1);
            if (lane == 0) a.part2[(int64_t)t * a.pstride + slot] = sx;
        }
        const double s = wave_sum(acc0 + acc1);
        if (CPLX) {
            const double si = wave_sum(aci0 - aci1);
            if (lane == 0) {
                a.part[(int64_t)(2 * t) * a.pstride + slot] = s;
                a.part[(int64_t)(2 * t + 1) * a.pstride + slot] = si;
            }
        } else if (lane == 0) {
            a.part[(int64_t)t * a.pstride + slot] = s;
        }
    };
    if constexpr ((NB & 1) == 0) {
        for (int t = 0; t < a.ncol; ++t) column(std::integral_constant<int, 0>{}, t);
    } else {                                    // odd NB: two columns bring the ring parity back to 0
        int t = 0;
        for (; t + 1 < a.ncol; t += 2) {
            column(std::integral_constant<int, 0>{}, t);
            column(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < a.ncol) column(std::integral_constant<int, 0>{}, t);
    }
#undef CGS_X_GET
#undef CH_OK
}

template <int R2, bool MASKED, int WL = 0, bool CPLX = false>
__global__ __launch_bounds__(CH_BS) void k_cgs_update(CgsArgs a) {
    constexpr int RW = R2 - WL;
    extern __shared__ __attribute__((aligned(16))) double2 wl[];   // [WL][CH_BS]
    constexpr int PB = CgsShape<R2>::PB;
    constexpr int NB = CgsShape<R2>::NB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[RW];
    double2 ring[2][PB];
    double2* __restrict__ w2 = reinterpret_cast<double2*>(a.w) + first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const double2 v = ld_nt2(w2 + (int64_t)r * CH_BS);
        double2 t;
        t.x = CH_OK(r) ? v.x : 0.0;
        t.y = CH_OK(r) ? v.y : 0.0;
        CGS_W_PUT(r, t);
        if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
    }
    // reverse: the dots pass has just streamed the columns first to last, so the LAST ones are what
    // the 256 MB Infinity Cache still holds - walking them last to first turns the head of this
    // pass into cache hits (it matters when the local basis is a few hundred MB: sharded runs)
    const int tfirst = a.reverse ? a.ncol - 1 : 0, tstep = a.reverse ? -1 : 1;
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.Vb + (a.col0 + tfirst) * a.ld) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = ld_nt2(v2 + (int64_t)i * CH_BS);
        CH_ISSUE_FENCE();
    }
    auto column = [&](auto par, int it) {
        constexpr int P0 = decltype(par)::value;
        const int t = tfirst + it * tstep;
        const int tn = (it + 1 < a.ncol) ? t + tstep : t;
        const double h = CPLX ? a.coef[2 * t] : a.coef[t];
        const double hi = CPLX ? a.coef[2 * t + 1] : 0.0;
        const double2* __restrict__ b2 =
            reinterpret_cast<const double2*>(a.Vb + (a.col0 + t) * a.ld) + first;
        const double2* __restrict__ bn =
            reinterpret_cast<const double2*>(a.Vb + (a.col0 + tn) * a.ld) + first;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2* __restrict__ nx = (b + 1 < NB) ? b2 + (int64_t)(b + 1) * PB * CH_BS : bn;
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[(P0 + b + 1) & 1][i] = ld_nt2(nx + (int64_t)i * CH_BS);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = ring[(P0 + b) & 1][i];
                const int r = b * PB + i;
                double2 wr = CGS_W_GET(r);
                if (CPLX) {
                    const double tr = h * p.x - hi * p.y;
                    const double ti = h * p.y + hi * p.x;
                    wr.x = CH_OK(r) ? wr.x - tr : 0.0;
                    wr.y = CH_OK(r) ? wr.y - ti : 0.0;
                } else {
                    wr.x = CH_OK(r) ? wr.x - h * p.x : 0.0;
                    wr.y = CH_OK(r) ? wr.y - h * p.y : 0.0;
                }
                CGS_W_PUT(r, wr);
            }
        }
    };
    if constexpr ((NB & 1) == 0) {
        for (int it = 0; it < a.ncol; ++it) column(std::integral_constant<int, 0>{}, it);
    } else {
        int it = 0;
        for (; it + 1 < a.ncol; it += 2) {
            column(std::integral_constant<int, 0>{}, it);
            column(std::integral_constant<int, 1>{}, it + 1);
        }
        if (it < a.ncol) column(std::integral_constant<int, 0>{}, it);
    }
    // write w back with the wave partials of its (M-)norm
    double acc = 0.0;
    if (a.dg != nullptr) {
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
        double2* __restrict__ m2 = reinterpret_cast<double2*>(a.mw) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 d = d2[(int64_t)r * CH_BS];
                const double2 wr = CGS_W_GET(r);
                double2 m;
                m.x = d.x * wr.x;
                m.y = d.y * wr.y;
                acc = fma(wr.x, m.x, acc);
                acc = fma(wr.y, m.y, acc);
                st_nt2(m2 + (int64_t)r * CH_BS, m);
                st_nt2(w2 + (int64_t)r * CH_BS, wr);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                const double2 wr = CGS_W_GET(r);
                acc = fma(wr.x, wr.x, acc);
                acc = fma(wr.y, wr.y, acc);
                st_nt2(w2 + (int64_t)r * CH_BS, wr);
            }
        }
    }
    const double s = wave_sum(acc);
    if (lane == 0) a.part[blockIdx.x * (CH_BS / 64) + wid] = s;
#undef CH_OK
}
#undef CGS_W_PUT
#undef CGS_W_GET

// ---- reference-order Gram-Schmidt with ONE reduction per step (round 4; N > 1 ranks) -------------------------------
// The reference's loop (utils.py:1012-1029):  alpha_j = <v_j, w_j>,  w_{j+1} = w_j - alpha_j v_j.  Since
// w_j = w - sum_{m<j} alpha_m v_m,
//     alpha_j = c_j - sum_{m<j} alpha_m G_{m,j},    c = V^T w (against the NOT YET UPDATED w),  G_{m,j} = <v_m, v_j>,
// i.e. (I + U^T) alpha = c with U the strict upper triangle of the basis' Gram matrix - the SAME coefficients in exact
// arithmetic, with all k+1 inner products in ONE pass (one all-reduce on N ranks instead of k+1).  In floating point it
// differs from the loop by O(eps |alpha| |G|), G being the basis' orthogonality defect (the blocked kernel of
// chain_blk.h rests on the same identity, block by block; Swirydowicz et al.'s low-synchronisation MGS on its
// compact-WY form).  It is NOT classical Gram-Schmidt: the correction keeps the coefficients those of the sequential
// loop however far the basis has drifted from orthogonality.
// k_lowsync_solve: one workgroup.  cg = the reduced sums of the step's dots pass, [c_0 .. c_k | g_0 .. g_{k-1}] with
// g_m = <v_m, v_k> (the new column k of the table; gt[m * LDG + j] = G_{m,j}, m < j).  Forward substitution in the
// order of the reference's loop (alpha_m times G_{m,j} subtracted from c_j for m = 0, 1, ...: the order of its updates);
// the triangle of the table is staged in LDS.  Writes alpha to coef (the update pass' coefficients) and to the H column.
constexpr int LS_MAXCOL = 128;      // basis columns the one-reduction form handles (a GMRES(100) cycle; more: the per-column path)
static __global__ __launch_bounds__(LS_MAXCOL) void k_lowsync_solve(int k, const double* __restrict__ cg, double* __restrict__ gt,
                                                                  double* __restrict__ coef, double* __restrict__ hcol) {
    extern __shared__ double ls_sm[];                 // [k rows m][k + 1 columns j] + 2 broadcast slots
    const int tid = threadIdx.x;
    const int ld = k + 1;
    double* bc = ls_sm + (size_t)k * ld;
    if (tid < k) gt[(size_t)tid * LS_MAXCOL + k] = cg[k + 1 + tid];            // column k of the table: <v_m, v_k>
    // (the whole k x (k + 1) rectangle in one flat loop of independent loads: the entries at and below the diagonal are
    // never read; a row-by-row loop over the triangle waited for memory k times)
#pragma unroll 4
    for (int e = tid; e < k * ld; e += LS_MAXCOL) {
        const int m = e / ld, j = e - m * ld;
        ls_sm[e] = (j == k) ? cg[k + 1 + m] : gt[(size_t)m * LS_MAXCOL + j];
    }
    double c = (tid <= k) ? cg[tid] : 0.0;
    __syncthreads();
    for (int m = 0; m < k; ++m) {
        if (tid == m) bc[m & 1] = c;
        __syncthreads();
        if (tid > m && tid <= k) {
            const double pr = bc[m & 1] * ls_sm[(size_t)m * ld + tid];
            c = c - pr;
        }
    }
    if (tid <= k) {
        coef[tid] = c;
        hcol[tid] = c;
    }
}

}  // namespace kh
