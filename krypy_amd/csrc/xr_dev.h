// Device side of the xr mailboxes (xr.hip): the slot of a granule pair and the cross-rank stage of a sum, shared by the
// stand-alone exchange kernels (xr.hip) and by the kernels that exchange inside a launch (chain_blk2.h).
#pragma once
#include <hip/hip_runtime.h>

#include "kh_internal.h"

namespace kh {

// [parity][sender][value][lo, hi]
__device__ __forceinline__ size_t xr_slot(unsigned epoch, int sender, int v) {
    return (((size_t)(epoch & 1u) * XR_MAXRANKS + (size_t)sender) * XR_MAXV + (size_t)v) * 2;
}

struct XrDev {                 // what a kernel needs to exchange sums with the peers; nranks == 0: no cross-rank stage
    unsigned long long* peer[XR_MAXRANKS];
    int rank, nranks;
    unsigned epoch0;           // xr epoch of the launch's first exchange (the launch consumes one per grid-wide sum)
    long long timeout_ticks;   // of the 100 MHz wall clock
};

// my value v of exchange `epoch` into every rank's mailbox (mine included)
__device__ __forceinline__ void xr_put_all(unsigned long long* const* peer, int rank, int nranks, unsigned epoch, int v, double x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned long long tag = (unsigned long long)epoch << 32;
    const unsigned long long lo = tag | (bits & 0xffffffffull), hi = tag | (bits >> 32);
    const size_t s = xr_slot(epoch, rank, v);
    for (int r = 0; r < nranks; ++r) {
        const int q = (rank + 1 + r) % nranks;      // start with the next rank: N ranks' writes do not all land on rank 0's link first
        __hip_atomic_store(peer[q] + s, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(peer[q] + s + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the sum over the ranks of value v of exchange `epoch`, contributions added in rank order (the same bits on every rank).
// *timed_out = 1 + the rank that did not arrive when the wait exceeds timeout_ticks (the result is NaN then).
__device__ __forceinline__ double xr_take_all(const unsigned long long* box, int nranks, unsigned epoch, int v, long long timeout_ticks,
                                              int* timed_out) {
    double total = 0.0;
    long long t0 = 0;
    for (int r = 0; r < nranks; ++r) {
        const unsigned long long* e = box + xr_slot(epoch, r, v);
        unsigned long long x0, x1;
        unsigned spins = 0;
        while (true) {
            x0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            x1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch) break;
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > timeout_ticks) {
                    *timed_out = 1 + r;
                    return __longlong_as_double(0x7ff8000000000000ll);      // NaN: whatever consumes it shows
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        const double d = __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
        total = (r == 0) ? d : total + d;
    }
    return total;
}

// ---- xh: the halo of a block-row shard through the same kind of granules (kernels.h: k_spmv_dia<..., XH>) ----
struct XhArgs {
    unsigned long long* mine;      // my ghost granules [2 parities][my_ng][lo, hi]
    unsigned long long* prev;      // the previous rank's (my first nsend_prev rows go to [prev_off + i])
    unsigned long long* next;      // the next rank's (my last nsend_next rows go to [i])
    long long my_ng, prev_ng, next_ng, prev_off;
    int nsend_prev, nsend_next;
    int ilo, ihi;                  // the blocks [ilo, ihi) of the slab whose rows touch no ghost column
    unsigned epoch;
    long long timeout_ticks;
    int* err;                      // mapped pinned host word (the xr transport's)
};

__device__ __forceinline__ void xh_put(unsigned long long* box, long long ng, long long idx, unsigned epoch, double x) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned long long tag = (unsigned long long)epoch << 32;
    unsigned long long* e = box + ((size_t)(epoch & 1u) * (size_t)ng + (size_t)idx) * 2;
    __hip_atomic_store(e, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(e + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ghost entry idx of this exchange: polled until both granules carry the epoch (a neighbour that never pushes: NaN and the
// error word after the timeout)
__device__ __forceinline__ double xh_take(const XhArgs& a, long long idx) {
    const unsigned long long* e = a.mine + ((size_t)(a.epoch & 1u) * (size_t)a.my_ng + (size_t)idx) * 2;
    unsigned long long x0, x1;
    unsigned spins = 0;
    long long t0 = 0;
    while (true) {
        x0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        x1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(x0 >> 32) == a.epoch && (unsigned)(x1 >> 32) == a.epoch) break;
        if ((++spins & 255u) == 0) {
            const long long now = (long long)wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > a.timeout_ticks) {
                __hip_atomic_store(a.err, 100, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return __longlong_as_double(0x7ff8000000000000ll);
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
}

// K ghost entries of this exchange at once (idx[k] < 0: not wanted): every poll of the lane in flight together, repeated
// for the entries whose granules do not carry the epoch yet
template <int K>
__device__ __forceinline__ void xh_take_n(const XhArgs& a, const long long (&idx)[K], double (&out)[K]) {
    const unsigned long long* box = a.mine + (size_t)(a.epoch & 1u) * (size_t)a.my_ng * 2;
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) pend |= (idx[k] >= 0) ? (1u << k) : 0u;
    unsigned spins = 0;
    long long t0 = 0;
    while (pend != 0) {
        unsigned long long lo[K], hi[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long* e = box + (size_t)(idx[k] >= 0 ? idx[k] : 0) * 2;     // (an entry not wanted: any valid address)
            lo[k] = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            hi[k] = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (((pend >> k) & 1u) && (unsigned)(lo[k] >> 32) == a.epoch && (unsigned)(hi[k] >> 32) == a.epoch) {
                out[k] = __longlong_as_double((long long)(((hi[k] & 0xffffffffull) << 32) | (lo[k] & 0xffffffffull)));
                pend &= ~(1u << k);
            }
        }
        if (pend != 0) {
            if ((++spins & 255u) == 0) {
                const long long now = (long long)wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > a.timeout_ticks) {
                    __hip_atomic_store(a.err, 100, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if ((pend >> k) & 1u) out[k] = __longlong_as_double(0x7ff8000000000000ll);
                    return;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

}  // namespace kh
