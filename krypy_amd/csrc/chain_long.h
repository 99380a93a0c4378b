// k_mgs_chain_long: the register-resident Gram-Schmidt chain for vectors BEYOND the register file (48 double2 rows per lane:
// 10.49 M ... 12.58 M rows per GPU - config 5's 12.5 M-row slabs) with a third of every basis column kept on the chip between
// its dot and its update.
//
// k_mgs_chain<48, ..., WL = 8> (chain.h) holds the last 8 rows of w in LDS (64 KB) and reads every column twice from memory:
// rocprofv3 FETCH_SIZE says 7.33 GB per launch where the columns add up to 5.05 GB (profiles/r05_config5s.md: 1.44 x), at 0.81
// of the fabric's peak - the kernel is bound by the bytes it asks for.  The other 92 KB of the compute unit's LDS sat idle.
// Here (reference recurrence utils.py:1012-1029, the same arithmetic in the same order as k_mgs_chain: every row updated once
// per link, the dot summed over the batches in ascending order - the same bits):
//
//   * 12 batches of 4 rows per column.  The dot phase parks batches 0 and 1 (8 rows, 64 KB) in LDS as they stream by, and
//     issues batch b + 2 into the ring slot batch b has just left - so it ends with batches 10 and 11 STILL IN THE RING;
//   * after the grid-wide sum the update consumes 10, 11 (ring), issuing the re-reads of batches 9 and 8 into the freed slots,
//     then the two parked batches from LDS while those fly, then 9 ... 2 through the ring (8 batches from memory, non-temporal:
//     their last use), refilling the last two slots with the NEXT column's batches 0 and 1.  16 of 48 rows never leave the chip
//     between the two uses: the second read shrinks from 48 to 32 rows per lane;
//   * consuming 10 before 11 makes the eight re-read batches land alternately in slot 0, 1, ..., so that the next column's
//     batch b sits in ring[b & 1] again: one loop body, no parity variable (the reversed walk the round-4 notes left unbuilt
//     needed an odd number of ring steps per phase);
//   * batches 2 ... 9 are first read with NORMAL loads (they come back: L2 / Infinity Cache may keep them), everything used
//     once is non-temporal.
// Same interface and grid as k_mgs_chain (ChainArgs, one 512-thread workgroup per compute unit); FND > 0: w = A v_k in the
// prologue (banded operator); XR: the cross-rank stage inside every grid-wide sum (chain_xr.hip).  Padded vectors only (the
// masked 48-row kernels spill 170 registers: unpadded vectors keep k_mgs_chain).  KRYPY_AMD_CHAIN_LONG=0: k_mgs_chain<48>.
#pragma once
#include "chain.h"

namespace kh {

// Buffer addressing: every stream of this kernel is "row r of this workgroup's chunk of one column" = base (uniform: scalar
// registers) + r * 8 KB (a constant: the instruction's scalar offset) + lane * 16 B (ONE vector register for all streams).
// With flat / global loads every row needs a 64-bit address pair per lane, computed ahead of the load: at 48 rows the first
// form of this kernel spilled 52 registers (and the round-4 attempts 28 ... 111); with buffer_load_dwordx4 ... offen the
// address arithmetic is scalar and the spills are gone.  Stores through the same descriptors are range-checked by the
// hardware (num_records = the valid bytes of the chunk): no predicate on the final store.
typedef unsigned int chl_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chl_rsrc(const double* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, bytes, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ double2 chl_ld(__amdgpu_buffer_rsrc_t r, int voff, int row) {
    const chl_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, row * (int)(CH_BS * sizeof(double2)), NT ? 2 : 0);
    double2 d;
    __builtin_memcpy(&d, &x, sizeof(d));
    return d;
}
__device__ __forceinline__ void chl_st_nt(__amdgpu_buffer_rsrc_t r, int voff, int row, double2 d) {
    chl_u32x4 x;
    __builtin_memcpy(&x, &d, sizeof(d));
    __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, row * (int)(CH_BS * sizeof(double2)), 2);
}

struct ChainShapeLong {
    static constexpr int R2 = 48;
    static constexpr int WL = 8;                  // rows of w in LDS
    static constexpr int RW = R2 - WL;            // rows of w in registers
    static constexpr int PB = 4;                  // rows per batch
    static constexpr int NB = R2 / PB;            // 12
    static constexpr int LB = 2;                  // leading batches parked in LDS
    static constexpr int NG = NB - LB - 2;        // 8 batches come back from memory
    static_assert(NG % 2 == 0, "the re-read batches must fill the two ring slots alternately and end where they began");
    static constexpr size_t LDS_BYTES = (size_t)(LB * PB + WL) * CH_BS * sizeof(double2);      // 128 KB
};

template <int FND = 0, bool XR = false>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain_long(ChainArgs a) {
    constexpr int R2 = ChainShapeLong::R2, WL = ChainShapeLong::WL, RW = ChainShapeLong::RW;
    constexpr int PB = ChainShapeLong::PB, NB = ChainShapeLong::NB, LB = ChainShapeLong::LB;
    // LDS: [WL rows of w | LB * PB parked rows][CH_BS].  A ds instruction's immediate offset spans 64 KB: with ONE base register
    // per 64 KB region (this lane's entry of row 0 of w, of parked row 0) every access is base + constant.  (The first build had
    // the parked rows in front: the eight rows of w behind the 64 KB mark each got an address register of their own.)
    extern __shared__ __attribute__((aligned(16))) double2 vlds[];
    const int tid = threadIdx.x;
    double2* const wl = vlds + tid;
    double2* park = vlds + (size_t)WL * CH_BS + tid;
    asm volatile("" : "+v"(park));                 // (a base of its own: not "wl + 64 KB + ..." re-derived per row)
#define W_GET(r) (((r) < RW) ? w[((r) < RW) ? (r) : 0] : wl[((r) - RW) * CH_BS])
#define W_PUT(r, val)                                   \
    do {                                                \
        if ((r) < RW) w[((r) < RW) ? (r) : 0] = (val);  \
        else wl[((r) - RW) * CH_BS] = (val);            \
    } while (0)
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int slead;
    const int G = gridDim.x;
    const GridRole role = grid_role(a.xcc_leader, a.epoch0, &slead);
    const int64_t wg0 = (int64_t)blockIdx.x * a.chunk2;          // this workgroup's first double2 (uniform)
    const int64_t first = wg0 + tid;
    const int64_t leftwg = a.n2 - wg0;
    const int valid_bytes = (int)(leftwg < 0 ? 0 : (leftwg > a.chunk2 ? a.chunk2 : leftwg)) * (int)sizeof(double2);
    constexpr int CHUNK_BYTES = R2 * CH_BS * (int)sizeof(double2);
    const int voff = tid * (int)sizeof(double2);
    double2 w[RW];
    double2 ring[2][PB];
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, [&](int r, double s0, double s1) { W_PUT(r, make_double2(s0, s1)); });
    } else {
        const __amdgpu_buffer_rsrc_t wb = chl_rsrc(a.w_in + 2 * wg0, CHUNK_BYTES);
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            W_PUT(r, chl_ld<true>(wb, voff, r));
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    if (a.presub) {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const __amdgpu_buffer_rsrc_t pb = chl_rsrc(a.bprev + 2 * wg0, CHUNK_BYTES);
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 p = chl_ld<false>(pb, voff, r);
            double2 t = W_GET(r);
            t.x = t.x - hk * p.x;
            t.y = t.y - hk * p.y;
            W_PUT(r, t);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    }
    unsigned epoch = a.epoch0;
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    const int total = a.ncol * a.sweeps;
    // prologue: batches 0 and 1 of the first column (both parked: used once from memory)
    {
        const __amdgpu_buffer_rsrc_t v2 = chl_rsrc(a.V + a.col0 * a.ld + 2 * wg0, CHUNK_BYTES);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[s][i] = chl_ld<true>(v2, voff, s * PB + i);
            CH_ISSUE_FENCE();
        }
    }
    for (int t = 0; t < total; ++t) {
        const int64_t j = a.col0 + (t % a.ncol);
        const __amdgpu_buffer_rsrc_t v2 = chl_rsrc(a.V + j * a.ld + 2 * wg0, CHUNK_BYTES);
        const int64_t jn = a.col0 + ((t + 1) % a.ncol);
        // (behind the last link: the same column again - valid memory, never used)
        const __amdgpu_buffer_rsrc_t vn = chl_rsrc(a.V + (t + 1 < total ? jn : j) * a.ld + 2 * wg0, CHUNK_BYTES);
        // ---- dot phase: <v_j, w>, batch b from ring[b & 1]; batch b + 2 goes into the slot b has just left ----
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 v = ring[b & 1][i];
                if (b < LB) park[(b * PB + i) * CH_BS] = v;
                const double2 wr = W_GET(b * PB + i);
                acc0 = fma(v.x, wr.x, acc0);
                acc1 = fma(v.y, wr.y, acc1);
            }
            if (b + 2 < NB) {
                // (batches 2 ... 9 come back for the update: normal loads; 10 and 11 stay in the ring: used once from memory)
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    ring[b & 1][i] = (b + 2 >= NB - 2) ? chl_ld<true>(v2, voff, (b + 2) * PB + i) : chl_ld<false>(v2, voff, (b + 2) * PB + i);
                CH_ISSUE_FENCE();
            }
        }
        const unsigned e_ = epoch++;
        double alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                      : grid_sum<XR>(acc0 + acc1, e_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr,
                                                     a.xr.epoch0 + (e_ - a.epoch0));
        if (blockIdx.x == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
        // ---- update phase: w -= alpha * v_j ----
#define CHL_UPD(r, p)                         \
    do {                                      \
        double2 wr_ = W_GET(r);               \
        wr_.x = wr_.x - alpha * (p).x;        \
        wr_.y = wr_.y - alpha * (p).y;        \
        W_PUT(r, wr_);                        \
    } while (0)
        // (a) batch 10 from ring[0], its slot takes the re-read of batch 9; batch 11 from ring[1], its slot that of batch 8
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int b = NB - 2 + s;
#pragma unroll
            for (int i = 0; i < PB; ++i) CHL_UPD(b * PB + i, ring[s][i]);
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[s][i] = chl_ld<true>(v2, voff, (NB - 3 - s) * PB + i);
            CH_ISSUE_FENCE();
        }
        // (b) the parked head of the column from LDS (own entries: no barrier), while the first re-reads are on their way
#pragma unroll
        for (int b = LB - 1; b >= 0; --b) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = park[(b * PB + i) * CH_BS];
                CHL_UPD(b * PB + i, p);
            }
            CH_ISSUE_FENCE();
        }
        // (c) batches 9 ... 2 through the ring: batch 9 - g sits in ring[g & 1]; the slot it leaves takes batch 7 - g, and after
        //     batches 3 and 2 the next column's batches 0 and 1 - which puts the next dot phase's batch b into ring[b & 1]
#pragma unroll
        for (int g = 0; g < ChainShapeLong::NG; ++g) {
            const int b = NB - 3 - g;
#pragma unroll
            for (int i = 0; i < PB; ++i) CHL_UPD(b * PB + i, ring[g & 1][i]);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[g & 1][i] = (b - 2 >= LB) ? chl_ld<true>(v2, voff, (b - 2) * PB + i) : chl_ld<true>(vn, voff, (LB + 1 - b) * PB + i);
            CH_ISSUE_FENCE();
        }
#undef CHL_UPD
    }
    // norm <w, w>, v_{k+1} = w / h from registers: as k_mgs_chain (no preconditioner here: B == V is what lets the column's
    // second use come from the chip, so the launcher sends a Jacobi step to k_mgs_chain)
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const double2 wr = W_GET(r);
        acc = fma(wr.x, wr.x, acc);
        acc = fma(wr.y, wr.y, acc);
    }
    const unsigned en_ = epoch++;
    const double h2 = grid_sum<XR>(acc, en_, a.gran, G, a.err, smd, smu, role, a.xcc_res, nullptr, &a.xr, a.xr.epoch0 + (en_ - a.epoch0));
    const double h = sqrt(fabs(h2));
    if (blockIdx.x == 0 && tid == 0) a.hdev[a.hnext] = h;
    // (stores beyond the vector's end are dropped by the descriptor's range check)
    const __amdgpu_buffer_rsrc_t vn2 = chl_rsrc(a.vnext + 2 * wg0, valid_bytes);
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const double2 wr = W_GET(r);
        double2 o;
        o.x = wr.x / h;
        o.y = wr.y / h;
        chl_st_nt(vn2, voff, r, o);
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
}

}  // namespace kh
