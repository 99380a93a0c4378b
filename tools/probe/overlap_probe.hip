// Probe (not product): can the next column's first batch be fetched WHILE the grid-wide sum of a Gram-Schmidt link
// is in flight?  A link of the chain kernel is  stream (dot) -> sum -> update -> stream ...; HBM idles through the sum
// and the lead-in of the next stream.  Round 2 measured that a prefetch issued in front of the sum's polls costs
// what it hides, because a wave's vector loads return in order: a poll queued behind the prefetch waits for it.
// This probe separates the mechanisms:
//   poll = 0   vector polls (global_load sc1 + vmcnt(0)), the library's grid_sum
//   poll = 1   non-leaders poll the XCD's result pair with SCALAR loads (s_load_dwordx4 glc, lgkmcnt): not ordered
//              with the wave's vector loads; leaders sweep the fabric granules with vector loads as before
//   poll = 2   as 1, and the leaders' sweep runs BEFORE they issue their prefetch (they need no result poll)
//   poll = 3   as 1, and the leaders sweep with scalar loads too (only valid if s_load glc sees other XCDs' stores:
//              the probe checks the sums)
// Each iteration: stream `rows - pf` rows of a fresh column (consumed at once: the dot phase), issue `pf` rows of the
// NEXT column into registers, run the sum, consume the prefetched rows.  Perfect overlap: time(pf) = time(0) - pf rows.
// Also: plain bandwidth modes for re-reads that can only come from the Infinity Cache (MALL).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "chain.h"

using namespace kh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 sload4_glc(const unsigned long long* p) {
    u32x4 v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// publish this workgroup's partial (as grid_sum does) - returns after the store is issued
__device__ __forceinline__ void gs_publish(double part, unsigned epoch, unsigned long long* gran, double* smd) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    const double ws = wave_sum_dpp(part);
    if (lane == 0) smd[wid] = ws;
    __syncthreads();
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
}

// leader: sweep with vector loads (one granule per thread), total to the XCD's result pair
__device__ __forceinline__ double gs_leader_vec(unsigned epoch, unsigned long long* gran, int G, double* smd,
                                                unsigned long long* res, int* err) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    unsigned mine = 0;
    if (tid < 2 * G) {
        unsigned long long x = ld_agent(slot + tid);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
            x = ld_agent(slot + tid);
            if (++spins > (1u << 22)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        mine = (unsigned)x;
    }
    const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);
    const unsigned long long bits = ((unsigned long long)mine << 32) | low;
    const double v = ((lane & 1) && tid < 2 * G) ? __longlong_as_double((long long)bits) : 0.0;
    const double wv = wave_sum_dpp(v);
    if (lane == 0) smd[NW + wid] = wv;
    __syncthreads();
    double s = smd[NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) s += smd[NW + i];
    if (tid == 0) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        res[0] = tag | (sb & 0xffffffffull);
        res[1] = tag | (sb >> 32);
    }
    return s;
}

// leader: sweep with scalar loads: wave w takes granules [64 w, 64 w + 64) = workgroups [32 w, 32 w + 32), two
// workgroups (32 bytes) per s_load_dwordx8... kept simple: x4 loads, one workgroup's granule pair each
__device__ __forceinline__ double gs_leader_scalar(unsigned epoch, unsigned long long* gran, int G, double* smd,
                                                   unsigned long long* res, int* err) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    const unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    double s = 0.0;
    const int b0 = __builtin_amdgcn_readfirstlane(wid * 32);
    for (int i = 0; i < 32; ++i) {
        const int b = b0 + i;
        if (b >= G) break;
        unsigned spins = 0;
        u32x4 g;
        while (true) {
            g = sload4_glc(slot + 2 * b);
            if (g.y == epoch && g.w == epoch) break;
            if (++spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        s += __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
    }
    if (lane == 0) smd[NW + wid] = s;
    __syncthreads();
    double t = smd[NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) t += smd[NW + i];
    if (tid == 0) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(t);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        res[0] = tag | (sb & 0xffffffffull);
        res[1] = tag | (sb >> 32);
    }
    return t;
}

__device__ __forceinline__ double gs_wait_vec(unsigned epoch, const unsigned long long* res, int* err) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 ab;
    unsigned spins = 0;
    while (true) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ab) : "v"(res) : "memory");
        if ((unsigned)(ab.x >> 32) == epoch && (unsigned)(ab.y >> 32) == epoch) break;
        if (++spins > (1u << 22)) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return __longlong_as_double((long long)(((ab.y & 0xffffffffull) << 32) | (ab.x & 0xffffffffull)));
}

__device__ __forceinline__ double gs_wait_scalar(unsigned epoch, const unsigned long long* res, int* err) {
    u32x4 g;
    unsigned spins = 0;
    while (true) {
        g = sload4_glc(res);
        if (g.y == epoch && g.w == epoch) break;
        if (++spins > (1u << 22)) { if ((threadIdx.x & 63) == 0) __hip_atomic_store(err, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
}

// ROWS rows per lane and link in total, PF of them prefetched across the sum
template <int ROWS, int PF, int POLL>
__global__ __launch_bounds__(CH_BS) void k_overlap(const double2* cols, int64_t ld2, int ncols, int links,
                                                   unsigned long long* gran, unsigned* xcc_leader, unsigned long long* xcc_res,
                                                   int* err, unsigned epoch0, unsigned stamp, double* out) {
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ int slead;
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const GridRole role = grid_role(xcc_leader, stamp, &slead);
    const int64_t first = (int64_t)blockIdx.x * ROWS * CH_BS + tid;
    unsigned epoch = epoch0;
    double acc = 0.0, total = 0.0;
    double2 pf[PF > 0 ? PF : 1];
    if (PF > 0) {
#pragma unroll
        for (int i = 0; i < PF; ++i) pf[i] = ld_nt2(cols + first + (int64_t)i * CH_BS);
    }
    for (int t = 0; t < links; ++t) {
        const double2* c = cols + (int64_t)(t % ncols) * ld2 + first;
        const double2* cn = cols + (int64_t)((t + 1) % ncols) * ld2 + first;
        // dot phase: the prefetched rows first, then the rest of the column
#pragma unroll
        for (int i = 0; i < PF; ++i) acc = fma(pf[i].x, pf[i].y, acc);
#pragma unroll
        for (int r0 = PF; r0 < ROWS; r0 += 5) {
            double2 v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) if (r0 + i < ROWS) v[i] = ld_nt2(c + (int64_t)(r0 + i) * CH_BS);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < 5; ++i) if (r0 + i < ROWS) acc = fma(v[i].x, v[i].y, acc);
        }
        unsigned long long* res = xcc_res + ((size_t)role.xcc * 2 + (epoch & 1u)) * 4;
        double s;
        if (POLL == 2 && role.leader) {
            gs_publish(acc, epoch, gran, smd);
            s = gs_leader_vec(epoch, gran, G, smd, res, err);
            if (PF > 0) {
#pragma unroll
                for (int i = 0; i < PF; ++i) pf[i] = ld_nt2(cn + (int64_t)i * CH_BS);
            }
        } else {
            if (PF > 0) {      // the next column's first rows: in flight while the sum runs
#pragma unroll
                for (int i = 0; i < PF; ++i) pf[i] = ld_nt2(cn + (int64_t)i * CH_BS);
                CH_ISSUE_FENCE();
            }
            gs_publish(acc, epoch, gran, smd);
            if (role.leader) s = (POLL == 3) ? gs_leader_scalar(epoch, gran, G, smd, res, err) : gs_leader_vec(epoch, gran, G, smd, res, err);
            else s = (POLL == 0) ? gs_wait_vec(epoch, res, err) : gs_wait_scalar(epoch, res, err);
        }
        ++epoch;
        total += s;
        acc = s * 1e-300;
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) acc = fma(pf[i].x, pf[i].y, acc);
    if (tid == 0) out[blockIdx.x] = total + acc;
}

// plain bandwidth: every link reads `fresh` rows of a new column (nt) and `again` rows of the column `lag` links back
template <int ROWS>
__global__ __launch_bounds__(CH_BS) void k_bw(const double2* cols, int64_t ld2, int ncols, int links, int fresh, int again,
                                              int lag, int nt_again, double* out) {
    const int tid = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * ROWS * CH_BS + tid;
    double acc = 0.0;
    for (int t = 0; t < links; ++t) {
        const double2* c = cols + (int64_t)(t % ncols) * ld2 + first;
        const double2* p = cols + (int64_t)((t + ncols - lag) % ncols) * ld2 + first;
        for (int r0 = 0; r0 < fresh; r0 += 5) {
            double2 v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) v[i] = ld_nt2(c + (int64_t)(r0 + i) * CH_BS);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < 5; ++i) acc = fma(v[i].x, v[i].y, acc);
        }
        for (int r0 = 0; r0 < again; r0 += 5) {
            double2 v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) v[i] = nt_again ? ld_nt2(p + (int64_t)(r0 + i) * CH_BS) : p[(int64_t)(r0 + i) * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < 5; ++i) acc = fma(v[i].x, v[i].y, acc);
        }
    }
    if (tid == 0) out[blockIdx.x] = acc;
}

// All working workgroups on ONE XCD (short vectors: the sum is the whole link).  8 G + 8 workgroups are launched; a
// workgroup that does not run on XCD `target` leaves at once, the others draw a ticket and the first G of them work.
// MODE 0: the library's protocol unchanged (granules with agent-scope stores, the one leader sweeps with agent-scope
// loads, the others poll its result pair); MODE 1: granules published with PLAIN stores (they stay in this XCD's L2),
// the leader sweeps with L1-bypassing loads (sc1: an L2 hit here).
template <int MODE>
__global__ __launch_bounds__(CH_BS) void k_onexcd(int Gw, int links, unsigned long long* gran, unsigned* ticket,
                                                  unsigned long long* xcc_res, int* err, unsigned epoch0, unsigned target,
                                                  double* out) {
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ int svb;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (xcc != target) return;
    if (tid == 0) svb = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int vb = svb;
    if (vb >= Gw) return;
    unsigned epoch = epoch0;
    double total = 0.0;
    unsigned long long* res = xcc_res + ((size_t)target * 2) * 4;
    for (int t = 0; t < links; ++t) {
        const double part = (double)((vb * CH_BS + tid) % 1000 + t % 7) * 1e-3;
        const double ws = wave_sum_dpp(part);
        if (lane == 0) smd[wid] = ws;
        __syncthreads();
        unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
        unsigned long long* rs = res + (epoch & 1u) * 4;
        if (tid == 0) {
            double s = smd[0];
#pragma unroll
            for (int i = 1; i < NW; ++i) s += smd[i];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
            const unsigned long long tag = (unsigned long long)epoch << 32;
            if (MODE == 0) {
                st_agent(slot + 2 * vb, tag | (bits & 0xffffffffull));
                st_agent(slot + 2 * vb + 1, tag | (bits >> 32));
            } else {
                slot[2 * vb] = tag | (bits & 0xffffffffull);
                slot[2 * vb + 1] = tag | (bits >> 32);
            }
        }
        double s;
        if (MODE == 2 || vb == 0) {
            unsigned mine = 0;
            if (tid < 2 * Gw) {
                unsigned long long x;
                unsigned spins = 0;
                while (true) {
                    if (MODE == 0) x = ld_agent(slot + tid);
                    else asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(slot + tid) : "memory");
                    if ((unsigned)(x >> 32) == epoch) break;
                    if (++spins > (1u << 22)) { __hip_atomic_store(err, 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                mine = (unsigned)x;
            }
            const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);
            const unsigned long long bits = ((unsigned long long)mine << 32) | low;
            const double v = ((lane & 1) && tid < 2 * Gw) ? __longlong_as_double((long long)bits) : 0.0;
            const double wv = wave_sum_dpp(v);
            if (lane == 0) smd[NW + wid] = wv;
            __syncthreads();
            s = smd[NW];
#pragma unroll
            for (int i = 1; i < NW; ++i) s += smd[NW + i];
            if (MODE != 2 && tid == 0) {
                const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
                const unsigned long long tag = (unsigned long long)epoch << 32;
                rs[0] = tag | (sb & 0xffffffffull);
                rs[1] = tag | (sb >> 32);
            }
            if (MODE == 2) __syncthreads();      // smd is reused by the next link's wave sums
        } else {
            s = gs_wait_vec(epoch, rs, err);
        }
        ++epoch;
        total += s;
    }
    if (tid == 0) out[vb] = total;
}

template <int MODE>
static void run_onexcd(int Gw, int links, unsigned long long* gran, unsigned* ticket, unsigned long long* xcc_res, int* err,
                       unsigned& epoch, double* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(ticket, 0, sizeof(unsigned), 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_onexcd<MODE>), dim3(8 * Gw + 8), dim3(CH_BS), 0, 0, Gw, links, gran, ticket, xcc_res, err, epoch,
                           0u, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        epoch += links;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<double> h(Gw);
    int herr = 0;
    unsigned tk = 0;
    CK(hipMemcpy(h.data(), out, sizeof(double) * Gw, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tk, ticket, sizeof(unsigned), hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 1; i < Gw; ++i) if (h[i] != h[0]) same = 0;
    printf("one XCD mode=%d G=%2d (%u workgroups landed on XCD 0 of %d launched): %.3f us per sum  (agree: %s, err=%d)\n", MODE, Gw,
           tk, 8 * Gw + 8, best * 1e3 / links, same ? "yes" : "NO", herr);
    if (herr) CK(hipMemset(err, 0, sizeof(int)));
    fflush(stdout);
}

struct Dev {
    double2* cols; int64_t ld2; int ncols;
    unsigned long long* gran; unsigned* xcc_leader; unsigned long long* xcc_res; int* err; double* out;
    unsigned epoch = 1, stamp = 0;
};

template <int ROWS, int PF, int POLL>
static void run_overlap(Dev& d, int G, int links) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        ++d.stamp;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_overlap<ROWS, PF, POLL>), dim3(G), dim3(CH_BS), 0, 0, d.cols, d.ld2, d.ncols, links, d.gran,
                           d.xcc_leader, d.xcc_res, d.err, d.epoch, d.stamp, d.out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        d.epoch += links;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<double> h(G);
    int herr = 0;
    CK(hipMemcpy(h.data(), d.out, sizeof(double) * G, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, d.err, sizeof(int), hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 1; i < G; ++i) if (fabs(h[i] - h[0]) > 1e-9 * fabs(h[0])) same = 0;
    printf("overlap rows=%2d pf=%d poll=%d G=%d: %.3f us per link  (agree: %s, err=%d)\n", ROWS, PF, POLL, G,
           best * 1e3 / links, same ? "yes" : "NO", herr);
    if (herr) CK(hipMemset(d.err, 0, sizeof(int)));
    fflush(stdout);
}

static void run_bw(Dev& d, int G, int links, int fresh, int again, int lag, int nt_again) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_bw<40>), dim3(G), dim3(CH_BS), 0, 0, d.cols, d.ld2, d.ncols, links, fresh, again, lag, nt_again, d.out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)G * CH_BS * 16.0 * (fresh + again);
    printf("bw fresh=%2d again=%2d lag=%d nt_again=%d ncols=%d: %.3f us per link, %.2f TB/s requested\n", fresh, again, lag,
           nt_again, d.ncols, best * 1e3 / links, bytes / (best * 1e-3 / links) * 1e-12);
    fflush(stdout);
}

int main() {
    Dev d;
    const int G = 245;
    d.ld2 = (int64_t)G * 40 * CH_BS;            // double2 per column: 80.3 MB
    d.ncols = 64;
    CK(hipMalloc(&d.cols, sizeof(double2) * d.ld2 * d.ncols + (1 << 20)));
    CK(hipMemset(d.cols, 0, sizeof(double2) * d.ld2 * d.ncols + (1 << 20)));
    CK(hipMalloc(&d.gran, sizeof(unsigned long long) * 4 * CH_GMAX)); CK(hipMemset(d.gran, 0, sizeof(unsigned long long) * 4 * CH_GMAX));
    CK(hipMalloc(&d.xcc_leader, sizeof(unsigned) * 16)); CK(hipMemset(d.xcc_leader, 0, sizeof(unsigned) * 16));
    CK(hipMalloc(&d.xcc_res, sizeof(unsigned long long) * 16 * 8)); CK(hipMemset(d.xcc_res, 0, sizeof(unsigned long long) * 16 * 8));
    CK(hipMalloc(&d.err, sizeof(int))); CK(hipMemset(d.err, 0, sizeof(int)));
    CK(hipMalloc(&d.out, sizeof(double) * 512));
    if (getenv("PROBE_ONEXCD")) {       // sums of short vectors: all workgroups on one XCD vs spread over eight
        unsigned* ticket; CK(hipMalloc(&ticket, sizeof(unsigned)));
        for (int Gw : {4, 13, 25, 32}) {
            run_onexcd<0>(Gw, 4000, d.gran, ticket, d.xcc_res, d.err, d.epoch, d.out);
            run_onexcd<1>(Gw, 4000, d.gran, ticket, d.xcc_res, d.err, d.epoch, d.out);
            run_onexcd<2>(Gw, 4000, d.gran, ticket, d.xcc_res, d.err, d.epoch, d.out);
            run_overlap<0, 0, 0>(d, Gw, 4000);       // the library's sum, workgroups on all XCDs
        }
        return 0;
    }
    const int links = 512;
    // 1. bandwidth: fresh column only; the same column again and again (Infinity Cache); fresh + lagged re-reads
    run_bw(d, G, links, 40, 0, 1, 0);
    d.ncols = 1; run_bw(d, G, links, 40, 0, 1, 0); run_bw(d, G, links, 0, 40, 0, 0); run_bw(d, G, links, 0, 40, 0, 1);
    d.ncols = 2; run_bw(d, G, links, 40, 0, 1, 0);
    d.ncols = 3; run_bw(d, G, links, 40, 0, 1, 0);
    d.ncols = 64;
    for (int nt = 0; nt < 2; ++nt) {
        run_bw(d, G, links, 40, 20, 0, nt);    // second read of the same column right behind the first (today's pattern, no on-chip reuse)
        run_bw(d, G, links, 40, 20, 1, nt);    // half of the PREVIOUS column again
        run_bw(d, G, links, 40, 40, 1, nt);    // all of it
        run_bw(d, G, links, 40, 10, 1, nt);
        run_bw(d, G, links, 40, 10, 2, nt);
    }
    // 2. overlap of a prefetch with the grid-wide sum
    run_overlap<40, 0, 0>(d, G, links);
    run_overlap<40, 5, 0>(d, G, links);
    run_overlap<40, 0, 1>(d, G, links);
    run_overlap<40, 5, 1>(d, G, links);
    run_overlap<40, 5, 2>(d, G, links);
    run_overlap<40, 10, 1>(d, G, links);
    run_overlap<40, 10, 2>(d, G, links);
    run_overlap<40, 0, 3>(d, G, links);
    run_overlap<40, 5, 3>(d, G, links);
    run_overlap<40, 10, 3>(d, G, links);
    // the sum alone
    run_overlap<5, 0, 0>(d, G, links);
    run_overlap<5, 0, 1>(d, G, links);
    run_overlap<5, 0, 3>(d, G, links);
    return 0;
}
