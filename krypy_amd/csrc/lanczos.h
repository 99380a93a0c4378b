// One Lanczos step (MINRES, utils.py:1008-1045 with ortho='lanczos') in one launch with THREE passes over the vector
// instead of six.
//
// The general chain kernel (chain.h) runs a Lanczos step as  w = A v_k | w -= h p_{k-1} | <v_k, w> | sum |
// w -= alpha p_k | <w, D w> | sum | store : six streaming phases, each with its own latency ramp, and v_k, the
// Jacobi diagonal and w's own bytes pass by more often than needed (1.04 GB at N = 10^7, 4.9 TB/s: the launch is
// bound by its phase boundaries, not by HBM).  A step has ONE Gram-Schmidt link, so nothing forces that split:
//
//   pass 1   row by row: (A v_k)_r from the diagonal-major copy (as k_spmv_dia / chain_apply_banded compute it),
//            minus h_{k-1,k} p_{k-1}[r], into the register that holds w_r; the same row of v_k - its cache line was
//            just gathered for the operator - feeds <v_k, w> at once                      (dia + v_k + p_{k-1})
//   sum      alpha = <v_k, w>
//   pass 2   w_r -= alpha p_k[r]  and  <w, D w> (or <w, w>) in the same sweep; the rows of the Jacobi diagonal are
//            parked in LDS as they pass                                                    (p_k + D)
//   sum      h = sqrt <w, D w>
//   pass 3   p_{k+1} = w / h, v_{k+1} = (D w) / h from registers and the parked rows      (rest of D; 2 stores)
//
// 0.92 GB instead of 1.04 GB and three ramps instead of six.  Every floating-point operation and the order of
// every sum are those of k_mgs_chain with presub (rows ascending, one accumulator pair for the dot, one accumulator
// for the norm): H and the vectors come out bit for bit as before (test_lanczos_fused_step_bit_identical).
//
// Optional fourth job, `mr` (MINRES, linsys.py:844-846): the recurrence update of an EARLIER iteration j,
//   z = (v_j - R0 W0 - R1 W1) / R2;  W0 <- z;  yk += y0 z
// whose coefficients the host has produced from H column j in the meantime (the launch for step k is enqueued while
// the host still works on step k-2: look-ahead) - six more streams that touch nothing this step reads or writes,
// issued in pass 3's shadow instead of as a launch of their own.
#pragma once
#include "chain.h"

namespace kh {

struct MinresJob {
    const double* v;      // v_j
    double* w0;           // W0 (overwritten with z)
    const double* w1;     // W1
    double* yk;
    double r0, r1, r2, y0;
    int on;
};

template <int R2>
struct LanczosShape {
    static constexpr int WL = (R2 == 40) ? 8 : 0;                     // rows of w in LDS
    static constexpr int DL_MAX = 19 - WL;                            // 8 KB per row: 19 rows fit beside the static arrays
    static constexpr int DL = R2 < DL_MAX ? R2 : DL_MAX;              // rows of the Jacobi diagonal parked in LDS
    static constexpr size_t LDS_BYTES = (size_t)(WL + DL) * CH_BS * sizeof(double2);
};

template <int R2, int FND, bool JAC, bool MR>
__global__ __launch_bounds__(CH_BS) void k_lanczos_fused(ChainArgs a, MinresJob mr) {
    constexpr int WL = LanczosShape<R2>::WL;
    constexpr int DL = JAC ? LanczosShape<R2>::DL : 0;
    constexpr int RW = R2 - WL;
    extern __shared__ __attribute__((aligned(16))) double2 lsm[];   // [WL rows of w][DL rows of D][CH_BS]
    double2* const wl = lsm;
    double2* const dl = lsm + (size_t)WL * CH_BS;
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
    double2 w[RW];
    unsigned epoch = a.epoch0;
    if (a.debug == 4 && tid == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests: a faked timeout
    // ---- pass 1: w = A v_k - h_{k-1,k} p_{k-1}, <v_k, w> ----
    double acc0 = 0.0, acc1 = 0.0;
    if (a.debug & 8) {               // measurement only (kh_ctx_set "chain_debug"): pass 1 switched off
#pragma unroll
        for (int r = 0; r < R2; ++r) W_PUT(r, make_double2(1.0, 1.0));
    } else {
        const double hk = (a.h_km1_dev != nullptr) ? a.h_km1_dev[0] : a.h_km1;
        const double* __restrict__ xk = a.xk;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(a.bprev);
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + a.col0 * a.ld);
        const int64_t last = a.n_last;
        int64_t fb = first;          // (row r at a constant distance from the opaque base: chain.h, chain_apply_banded)
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
                int64_t c0 = row + off, c1 = row + 1 + off;
                c0 = c0 < 0 ? 0 : (c0 > last ? last : c0);
                c1 = c1 < 0 ? 0 : (c1 > last ? last : c1);
                x0[d] = xk[c0];
                x1[d] = xk[c1];
            }
            // (tried: aligned 16-byte loads of x for the even offsets of a stencil pattern, the +-1 neighbours from the
            // centre pair - 7 load instructions per row pair instead of 12: pass 1 went from 110 to 127 us, not kept;
            // the MINRES job's six streams interleaved with this loop to fill its bubbles, 19 rows of w in LDS to make
            // room: 213 us for the pair against 108 + 73 apart, not kept either)
            const double2 vv = v2[i2];               // (the line the operator's centre entries came from)
            const double2 pp = ld_nt2(p2 + i2);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int d = 0; d < FND; ++d) {
                const double q0 = av[d].x * x0[d], q1 = av[d].y * x1[d];
                s0 = (av[d].x != 0.0) ? s0 + q0 : s0;
                s1 = (av[d].y != 0.0) ? s1 + q1 : s1;
            }
            double2 t;
            t.x = s0 - hk * pp.x;
            t.y = s1 - hk * pp.y;
            W_PUT(r, t);
            acc0 = fma(vv.x, t.x, acc0);
            acc1 = fma(vv.y, t.y, acc1);
            if ((r & KH_RIF) == KH_RIF) asm volatile("" : "+v"(fb) : : "memory");   // four rows of loads in flight
        }
    }
    // ---- pass 2: w -= alpha p_k, <w, D w>; a two-deep ring of PB2 rows, its first batch in flight across the sum ----
    constexpr int PB2 = 4, NB2 = R2 / PB2;
    static_assert(NB2 * PB2 == R2, "rows per lane are a multiple of four");
    const double2* __restrict__ b2 = reinterpret_cast<const double2*>(a.B + a.col0 * a.ld) + first;
    const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
    double2 rp[2][PB2], rd[2][PB2];
    const bool run2 = !(a.debug & 16);
    if (run2) {
#pragma unroll
        for (int i = 0; i < PB2; ++i) {
            rp[0][i] = ld_nt2(b2 + (int64_t)i * CH_BS);
            if (JAC) rd[0][i] = ld_nt2(d2 + (int64_t)i * CH_BS);
        }
        CH_ISSUE_FENCE();
    }
    double alpha = grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
    if (blockIdx.x == 0 && tid == 0) a.hdev[a.col0] = 0.0 + alpha;
    if (a.debug == 4) alpha *= 0.5;      // ... that leaves garbage behind
    double acc = 0.0;
    if (run2) {
#pragma unroll
        for (int b = 0; b < NB2; ++b) {
            if (b + 1 < NB2) {
#pragma unroll
                for (int i = 0; i < PB2; ++i) {
                    rp[(b + 1) & 1][i] = ld_nt2(b2 + (int64_t)((b + 1) * PB2 + i) * CH_BS);
                    if (JAC) rd[(b + 1) & 1][i] = ld_nt2(d2 + (int64_t)((b + 1) * PB2 + i) * CH_BS);
                }
            }
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB2; ++i) {
                const int r = b * PB2 + i;
                const double2 p = rp[b & 1][i];
                double2 wr = W_GET(r);
                wr.x = wr.x - alpha * p.x;
                wr.y = wr.y - alpha * p.y;
                W_PUT(r, wr);
                if (JAC) {
                    const double2 d = rd[b & 1][i];
                    if (r < DL) dl[r * CH_BS + tid] = d;
                    acc = fma(wr.x, d.x * wr.x, acc);
                    acc = fma(wr.y, d.y * wr.y, acc);
                } else {
                    acc = fma(wr.x, wr.x, acc);
                    acc = fma(wr.y, wr.y, acc);
                }
            }
        }
    }
    // ---- pass 3 (+ the MINRES recurrences of an earlier iteration): a two-deep ring of PB3 rows ----
    constexpr int PB3 = 2, NB3 = R2 / PB3;
    double2* __restrict__ vn2 = reinterpret_cast<double2*>(a.vnext) + first;
    double2* __restrict__ pn2 = reinterpret_cast<double2*>(a.pnext) + first;
    const double2* __restrict__ vj = reinterpret_cast<const double2*>(mr.v) + first;
    double2* __restrict__ w0 = reinterpret_cast<double2*>(mr.w0) + first;
    const double2* __restrict__ w1 = reinterpret_cast<const double2*>(mr.w1) + first;
    double2* __restrict__ yk = reinterpret_cast<double2*>(mr.yk) + first;
    double2 qd[2][PB3], qv[2][PB3], qu0[2][PB3], qu1[2][PB3], qy[2][PB3];
    const bool run3 = !(a.debug & 32), runm = MR && !(a.debug & 64);
#define LZ_ST(ptr_, val_) st_nt2(ptr_, val_)          // (kernels.h: nobody reads these before the kernel ends)
#define LZ_ISSUE3(b_, s_)                                                               \
    do {                                                                                \
        _Pragma("unroll") for (int i = 0; i < PB3; ++i) {                               \
            const int r_ = (b_) * PB3 + i;                                              \
            const int64_t o_ = (int64_t)r_ * CH_BS;                                     \
            if (JAC && run3 && r_ >= DL) qd[s_][i] = ld_nt2(d2 + o_);                   \
            if (runm) {                                                                 \
                qv[s_][i] = ld_nt2(vj + o_);                                            \
                qu0[s_][i] = ld_nt2(w0 + o_);                                           \
                qu1[s_][i] = ld_nt2(w1 + o_);                                           \
                qy[s_][i] = ld_nt2(yk + o_);                                            \
            }                                                                           \
        }                                                                               \
    } while (0)
    LZ_ISSUE3(0, 0);
    CH_ISSUE_FENCE();
    const double h2 = grid_sum(acc, epoch++, a.gran, G, a.err, smd, smu, role, a.xcc_res);
    const double h = sqrt(fabs(h2));
    if (blockIdx.x == 0 && tid == 0) a.hdev[a.hnext] = h;
#pragma unroll
    for (int b = 0; b < NB3; ++b) {
        if (b + 1 < NB3) LZ_ISSUE3(b + 1, (b + 1) & 1);
        CH_ISSUE_FENCE();
#pragma unroll
        for (int i = 0; i < PB3; ++i) {
            const int r = b * PB3 + i;
            if (r * CH_BS < rem) {
                if (run3) {
                    const double2 wr = W_GET(r);
                    double2 o;
                    o.x = wr.x / h;
                    o.y = wr.y / h;
                    if (JAC) {
                        const double2 d = (r < DL) ? dl[r * CH_BS + tid] : qd[b & 1][i];
                        double2 m;
                        m.x = (d.x * wr.x) / h;
                        m.y = (d.y * wr.y) / h;
                        LZ_ST(pn2 + (int64_t)r * CH_BS, o);
                        LZ_ST(vn2 + (int64_t)r * CH_BS, m);
                    } else {
                        LZ_ST(vn2 + (int64_t)r * CH_BS, o);
                    }
                }
                if (runm) {      // k_minres_update's formulas (kernels.h), operation for operation
                    const double2 v = qv[b & 1][i], u0 = qu0[b & 1][i], u1 = qu1[b & 1][i], y = qy[b & 1][i];
                    double2 z, yo;
                    z.x = ((v.x - mr.r0 * u0.x) - mr.r1 * u1.x) / mr.r2;
                    z.y = ((v.y - mr.r0 * u0.y) - mr.r1 * u1.y) / mr.r2;
                    LZ_ST(w0 + (int64_t)r * CH_BS, z);
                    yo.x = y.x + mr.y0 * z.x;
                    yo.y = y.y + mr.y0 * z.y;
                    LZ_ST(yk + (int64_t)r * CH_BS, yo);
                }
            }
        }
    }
#undef LZ_ISSUE3
#undef LZ_ST
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

// (Round 6 built the same three passes for complex (c128) vectors - k_zlanczos_fused<R2, FND>: chain_apply_banded_z's products,
// the real pre-subtraction and conj(v_k) w in pass 1, the complex update and |w|^2 in pass 2, the store in pass 3; bit for bit
// the complex chain kernel's H and basis at 16 ... 40 rows per lane, no spilled registers - and measured it on complex Hermitian
// MINRES at N = 5 * 10^6: 3,980-4,120 it/s with two rows of loads in flight, 4,226-4,237 with four, against 4,242-4,275 for the
// general complex chain kernel with the operator in its prologue.  No gain: not kept (EXPERIMENTS.md).)

}  // namespace kh
