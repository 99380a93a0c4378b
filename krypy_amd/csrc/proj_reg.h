// The deflation projector with the vector in registers (round 4).
//
// Reference: utils.Projection.apply_complement (utils.py:604-627) as the deflated solvers call it
// (deflation.py:127-143): per sweep  c = W^T z,  c' = T c,  z -= V c',  `iterations` (two) sweeps, and <Y, z_in> = WRH c
// of the first sweep handed back.  As four launches per sweep (k_multidot<16>, reduction, the small matrix-vector
// products, k_multiaxpy<16>) z travels to memory and back three times per sweep - 48 N of the 560 N bytes an
// application moves - and a deflated Arnoldi step spends nine launches before its Gram-Schmidt chain starts.
//
// Here ONE launch: the workgroups of the chain-kernel geometry (one per compute unit, 512 lanes, R2 rows of 16 bytes per
// lane, the last WL rows in LDS for vectors beyond the register file) load z once, stream the d <= 16 columns of W through
// the two-deep register ring of the panel kernels (k_cgs_dots), add up all d coefficients in ONE grid-wide sum of d values,
// form c' = T c in every workgroup (d x d: 256 multiply-adds), stream the columns of V (k_cgs_update's loop), again for the
// second sweep, and store z.  2 x 32 columns + z in, z out: 528 N bytes.
//
// The sum (grid_sum16): the protocol of chain.h's grid_sum with d values per workgroup - every workgroup publishes its d
// partial sums as tagged granule pairs (write-through), the eight XCD leaders gather them over the fabric (wave i takes
// the values i and i + 8: each lane the workgroups lane, lane + 64, ... in ascending order, then the DPP tree: the order
// of the additions depends on the workgroup numbers alone, every workgroup gets the same bits) and leave the d totals in
// their XCD's L2 with plain stores, where wave 0 of every other workgroup polls them.  Bounded spins, the chain kernels'
// error word: a timeout makes the Arnoldi step that follows report it, and the step is re-run without this kernel.
#pragma once
#include "chain.h"

namespace kh {

constexpr int PR_NV = 16;                      // coefficients per sum (columns of the deflation basis at most)

struct ProjRegArgs {
    int64_t n2, chunk2;
    const double* Wb;        // d columns, leading dimension ld
    const double* Vb;        // d columns, leading dimension ld
    int64_t ld;
    int d, iterations;
    double* z;               // in / out
    const double* T;         // d x d row-major or nullptr (identity)
    const double* WRH;       // d x d row-major or nullptr (identity)
    double* ya;              // d doubles or nullptr: <Y, z_in> = WRH c of the first sweep
    unsigned long long* gran;     // [2 parities][CH_GMAX][PR_NV][2] granules
    unsigned long long* res;      // [16 XCDs][2 parities][PR_NV][2]
    unsigned* xcc_leader;         // [16] stamps of the leader election
    unsigned epoch0;
    int* err;
};

// d <= PR_NV sums over the grid at once; in: sm_part[v * 8 + wave] wave partials (already in LDS, a barrier behind them);
// out: sm_tot[v], valid for every thread after the call.
__device__ __forceinline__ void grid_sum16(int d, unsigned epoch, const ProjRegArgs& a, int G, const GridRole role,
                                           const double* sm_part, double* sm_tot) {
    constexpr int NW = CH_BS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned long long* slot = a.gran + (size_t)(epoch & 1u) * ((size_t)CH_GMAX * PR_NV * 2);
    if (tid < d) {
        double s = sm_part[tid * NW];
#pragma unroll
        for (int i = 1; i < NW; ++i) s += sm_part[tid * NW + i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        unsigned long long* e = slot + ((size_t)blockIdx.x * PR_NV + tid) * 2;
        st_agent(e, tag | (bits & 0xffffffffull));
        st_agent(e + 1, tag | (bits >> 32));
    }
    unsigned long long* res = a.res + ((size_t)role.xcc * 2 + (epoch & 1u)) * (PR_NV * 2);
    if (role.leader) {
        // wave `wid` gathers the values wid and wid + 8; lane l the workgroups l, l + 64, l + 128, l + 192
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int v = wid + half * NW;
            if (v < d) {                                   // (wave-uniform)
                unsigned lo[4], hi[4];
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int b = lane + 64 * p;
                        lo[p] = hi[p] = 0u;
                        if (b < G) {
                            const unsigned long long* e = slot + ((size_t)b * PR_NV + v) * 2;
                            const unsigned long long x0 = ld_agent(e), x1 = ld_agent(e + 1);
                            lo[p] = (unsigned)x0;
                            hi[p] = (unsigned)x1;
                            ok = ok && (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
                        }
                    }
                    if (__all(ok)) break;
                    if ((++spins & 255u) == 0) {
                        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if (spins > (1u << 20)) {
                            __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const double x = __longlong_as_double((long long)(((unsigned long long)hi[p] << 32) | lo[p]));
                    acc += (lane + 64 * p < G) ? x : 0.0;
                }
                const double t = wave_sum_dpp(acc);
                if (lane == 0) {
                    sm_tot[v] = t;
                    const unsigned long long sb = (unsigned long long)__double_as_longlong(t);
                    const unsigned long long tag = (unsigned long long)epoch << 32;
                    volatile unsigned long long* rv = res + 2 * v;       // plain stores: they stay in this XCD's L2
                    rv[0] = tag | (sb & 0xffffffffull);
                    rv[1] = tag | (sb >> 32);
                }
            }
        }
    } else if (wid == 0) {
        // the leader of this XCD leaves the totals in `res`: lane g < 2 d polls granule g (an L2 hit on this XCD)
        unsigned long long x = 0;
        unsigned spins = 0;
        while (true) {
            bool ok = true;
            if (lane < 2 * d) {
                x = ld_l2(res + lane);
                ok = (unsigned)(x >> 32) == epoch;
            }
            if (__all(ok)) break;
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 22)) {
                    __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        const unsigned mine = (unsigned)x;
        const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);   // from lane - 1
        if ((lane & 1) && lane < 2 * d)
            sm_tot[lane >> 1] = __longlong_as_double((long long)(((unsigned long long)mine << 32) | low));
    }
    __syncthreads();
}

template <int R2>
struct ProjRegShape {
    static constexpr int WL = R2 == 48 ? 8 : (R2 == 56 ? 16 : 0);      // rows of z in LDS (k_mgs_chain's long shapes)
    static constexpr int PB = ChainShape<R2>::PB;
    static constexpr int NB = R2 / PB;
    static_assert((NB % 2) == 0, "ring parity resets every column");
    static constexpr size_t LDS_BYTES = (size_t)WL * CH_BS * sizeof(double2);
};

template <int R2>
__global__ __launch_bounds__(CH_BS) void k_proj_reg(ProjRegArgs a) {
    constexpr int WL = ProjRegShape<R2>::WL;
    constexpr int RW = R2 - WL;
    constexpr int PB = ProjRegShape<R2>::PB;
    constexpr int NB = ProjRegShape<R2>::NB;
    constexpr int NW = CH_BS / 64;
    extern __shared__ __attribute__((aligned(16))) double2 wl[];   // [WL][CH_BS]
#define PR_GET(r) (((r) < RW) ? w[((r) < RW) ? (r) : 0] : wl[((r) - RW) * CH_BS + tid])
#define PR_PUT(r, val)                                  \
    do {                                                \
        if ((r) < RW) w[((r) < RW) ? (r) : 0] = (val);  \
        else wl[((r) - RW) * CH_BS + tid] = (val);      \
    } while (0)
    __shared__ double sm_part[PR_NV * NW];
    __shared__ double sm_tot[PR_NV];
    __shared__ double sm_cp[PR_NV];
    __shared__ int slead;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = gridDim.x;
    const GridRole role = grid_role(a.xcc_leader, a.epoch0, &slead);
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int d = a.d;
    double2 w[RW];
    double2 ring[2][PB];
    double2* __restrict__ z2 = reinterpret_cast<double2*>(a.z) + first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        PR_PUT(r, ld_nt2(z2 + (int64_t)r * CH_BS));           // (padded blocks: the rows behind the vector are zero and stay zero)
        if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
    }
    unsigned epoch = a.epoch0;
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.Wb) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = ld_nt2(v2 + (int64_t)i * CH_BS);
        CH_ISSUE_FENCE();
    }
    for (int it = 0; it < a.iterations; ++it) {
        // ---- c = W^T z: the columns of W through the ring (the first batch is in ring[0]), wave partials to LDS ----
        for (int t = 0; t < d; ++t) {
            const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.Wb + (int64_t)t * a.ld) + first;
            const double2* __restrict__ vn = (t + 1 < d) ? reinterpret_cast<const double2*>(a.Wb + (int64_t)(t + 1) * a.ld) + first
                                                          : reinterpret_cast<const double2*>(a.Vb) + first;     // (the update's first batch)
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double2* __restrict__ nx = (b + 1 < NB) ? v2 + (int64_t)(b + 1) * PB * CH_BS : vn;
#pragma unroll
                for (int i = 0; i < PB; ++i) ring[(b + 1) & 1][i] = ld_nt2(nx + (int64_t)i * CH_BS);      // (32 columns of 8 N bytes each, twice: nothing of it stays in any cache)
                CH_ISSUE_FENCE();
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const double2 v = ring[b & 1][i];
                    const double2 wr = PR_GET(b * PB + i);
                    acc0 = fma(v.x, wr.x, acc0);
                    acc1 = fma(v.y, wr.y, acc1);
                }
            }
            const double s = wave_sum_dpp(acc0 + acc1);
            if (lane == 0) sm_part[t * NW + wid] = s;
        }
        __syncthreads();
        grid_sum16(d, epoch++, a, G, role, sm_part, sm_tot);
        // ---- c' = T c (every workgroup: the same bits), <Y, z_in> = WRH c of the first sweep ----
        if (tid < d) {
            double s = 0.0;
            if (a.T != nullptr) {
                for (int j = 0; j < d; ++j) s += a.T[tid * d + j] * sm_tot[j];
            } else {
                s = sm_tot[tid];
            }
            sm_cp[tid] = s;
        } else if (it == 0 && a.ya != nullptr && blockIdx.x == 0 && tid >= 64 && tid < 64 + d) {
            const int t = tid - 64;
            double s = 0.0;
            if (a.WRH != nullptr) {
                for (int j = 0; j < d; ++j) s += a.WRH[t * d + j] * sm_tot[j];
            } else {
                s = sm_tot[t];
            }
            a.ya[t] = s;
        }
        __syncthreads();
        // ---- z -= V c': the first batch of V's first column is in ring[0] ----
        for (int t = 0; t < d; ++t) {
            const double h = sm_cp[t];
            const double2* __restrict__ b2 = reinterpret_cast<const double2*>(a.Vb + (int64_t)t * a.ld) + first;
            const double2* __restrict__ bn = (t + 1 < d) ? reinterpret_cast<const double2*>(a.Vb + (int64_t)(t + 1) * a.ld) + first
                                                          : reinterpret_cast<const double2*>(a.Wb) + first;     // (the next sweep's first batch)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const double2* __restrict__ nx = (b + 1 < NB) ? b2 + (int64_t)(b + 1) * PB * CH_BS : bn;
#pragma unroll
                for (int i = 0; i < PB; ++i) ring[(b + 1) & 1][i] = ld_nt2(nx + (int64_t)i * CH_BS);      // (32 columns of 8 N bytes each, twice: nothing of it stays in any cache)
                CH_ISSUE_FENCE();
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const double2 p = ring[b & 1][i];
                    const int r = b * PB + i;
                    double2 wr = PR_GET(r);
                    wr.x = wr.x - h * p.x;
                    wr.y = wr.y - h * p.y;
                    PR_PUT(r, wr);
                }
            }
        }
        __syncthreads();          // (sm_part / sm_cp are rewritten by the next sweep)
    }
#pragma unroll
    for (int r = 0; r < R2; ++r) st_nt2(z2 + (int64_t)r * CH_BS, PR_GET(r));
#undef PR_PUT
#undef PR_GET
}

}  // namespace kh
