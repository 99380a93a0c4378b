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
// error word is set, all workgroups fall through, and the host disables this path.
//
// Residency: one 512-thread workgroup per CU (<= 256 VGPRs per lane); the launch is cooperative
// so that a grid that cannot be co-resident is rejected instead of deadlocking.
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
    unsigned epoch0;
    int* err;
    int debug;         // measurement only: 1 = skip the grid reduction, 2 = skip the streaming phases
    int presub;        // Lanczos: w -= h_km1 * bprev first
    double h_km1;
    const double* h_km1_dev;   // when non-null the coefficient is read from the device (look-ahead)
    const double* bprev;
    // when hpin != nullptr workgroup 0 finally copies the H column (hcount doubles) and the error word
    // straight into pinned host memory: no device-to-host copies behind the launch
    double* hpin;
    int hcount;
    int* errpin;
    // fused operator (k_mgs_chain_lds<..., FND > 0>): w = A x computed by the prologue straight into
    // registers from the diagonal-major copy of a banded operator instead of being loaded
    const double* dia;
    int64_t dia_ld;
    const double* xk;
    int64_t n_last;    // n - 1
    DiaOffs offs;
#ifdef KH_CHAIN_TRACE
    unsigned long long* trace;   // diagnostic build only (make trace): [G][links][2 waves][8] 100 MHz stamps
#endif
};

// Phase stamps of the diagnostic build (tools/chain_trace.py): wave 0 and wave 7 of every workgroup
// note the constant 100 MHz clock at the phase boundaries of every link.  Compiled out of the product.
#ifdef KH_CHAIN_TRACE
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

// Fixed-order sum of one double per workgroup over the whole grid; every thread gets the result.
__device__ __forceinline__ double grid_sum(double part, unsigned epoch, unsigned long long* gran,
                                           int G, int* err, double* smd, unsigned* smu,
                                           unsigned long long* tr = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // 1. workgroup partial (8 waves, fixed order)
    part = wave_sum(part);
    if (lane == 0) smd[wid] = part;
    __syncthreads();
#ifdef KH_CHAIN_TRACE
    if (tr != nullptr && (tid == 0 || tid == CH_BS - 64)) tr[(tid ? 8 : 0) + 2] = wall_clock64();
#endif
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (tid == 0) {
        double s = smd[0];
#pragma unroll
        for (int i = 1; i < CH_BS / 64; ++i) s += smd[i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        st_agent(slot + 2 * blockIdx.x, tag | (bits & 0xffffffffull));
        st_agent(slot + 2 * blockIdx.x + 1, tag | (bits >> 32));
    }
#ifdef KH_CHAIN_TRACE
    if (tr != nullptr && (tid == 0 || tid == CH_BS - 64)) tr[(tid ? 8 : 0) + 3] = wall_clock64();
#endif
    // 2. sweep all granules of this epoch (bounded spin)
    for (int g = tid; g < 2 * G; g += CH_BS) {
        unsigned long long x = ld_agent(slot + g);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(1);
            x = ld_agent(slot + g);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 22)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        smu[g] = (unsigned)x;
    }
#ifdef KH_CHAIN_TRACE
    if (tr != nullptr && (tid == 0 || tid == CH_BS - 64)) tr[(tid ? 8 : 0) + 4] = wall_clock64();
#endif
    __syncthreads();
    // 3. fixed-order sum of the G workgroup partials
    double v = 0.0;
    for (int b = tid; b < G; b += CH_BS) {
        const unsigned long long bits = ((unsigned long long)smu[2 * b + 1] << 32) | smu[2 * b];
        v += __longlong_as_double((long long)bits);
    }
    v = wave_sum(v);
    __syncthreads();          // smd reuse
    if (lane == 0) smd[wid] = v;
    __syncthreads();
    double s = smd[0];
#pragma unroll
    for (int i = 1; i < CH_BS / 64; ++i) s += smd[i];
    __syncthreads();
    return s;
}

// Two sums in one round (the real and imaginary part of a complex coefficient): four granules per
// workgroup, one publish, one sweep.  Needs 4*G <= 2*CH_GMAX words per parity (checked by the
// launcher); same protocol, same fixed summation order as grid_sum.
__device__ __forceinline__ void grid_sum2(double& p0, double& p1, unsigned epoch, unsigned long long* gran,
                                          int G, int* err, double* smd, unsigned* smu) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    const double a0 = wave_sum(p0), a1 = wave_sum(p1);
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
    for (int g = tid; g < 4 * G; g += CH_BS) {
        unsigned long long x = ld_agent(slot + g);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(1);
            x = ld_agent(slot + g);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (spins > (1u << 22)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        smu[g] = (unsigned)x;
    }
    __syncthreads();
    double v0 = 0.0, v1 = 0.0;
    for (int b = tid; b < G; b += CH_BS) {
        const unsigned long long b0 = ((unsigned long long)smu[4 * b + 1] << 32) | smu[4 * b];
        const unsigned long long b1 = ((unsigned long long)smu[4 * b + 3] << 32) | smu[4 * b + 2];
        v0 += __longlong_as_double((long long)b0);
        v1 += __longlong_as_double((long long)b1);
    }
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    __syncthreads();          // smd reuse
    if (lane == 0) {
        smd[wid] = v0;
        smd[NW + wid] = v1;
    }
    __syncthreads();
    double s0 = smd[0], s1 = smd[NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) {
        s0 += smd[i];
        s1 += smd[NW + i];
    }
    __syncthreads();
    p0 = s0;
    p1 = s1;
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
template <int R2, int FND>
__device__ __forceinline__ void chain_apply_banded(const ChainArgs& a, int64_t first, double2 (&w)[R2]) {
    // w = A x_k for this lane's rows, exactly as k_spmv_dia computes them (ascending offsets,
    // separate multiply and add, empty slots skipped): the 80 MB of w are never written nor read.
    // The padding rows behind n hold zeros in every diagonal and come out as w = 0.
    const double* __restrict__ xk = a.xk;
    const int64_t last = a.n_last;
    int64_t i2 = first;      // advanced through an opaque asm every two rows: that is what bounds the
                             // loads in flight (and the live temporaries) of this fully unrolled loop
#pragma unroll
    for (int r = 0; r < R2; ++r) {
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
        w[r].x = s0;
        w[r].y = s1;
        i2 += CH_BS;
        if ((r & 1) == 1) asm volatile("" : "+v"(i2) : : "memory");   // two rows of loads in flight (one: 1023 it/s, two: 1028-1037, four: 1018-1030)
    }
}

template <int R2, bool MASKED, bool CPLX = false, int FND = 0>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain(ChainArgs a) {
    static_assert(FND == 0 || (!MASKED && !CPLX), "the fused operator exists for the padded real kernel");
    constexpr int PB = ChainShape<R2>::PB;
    constexpr int NB = ChainShape<R2>::NB;
    __shared__ double smd[2 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    // chunk2 == R2 * CH_BS: thread `tid` owns elements first + r*CH_BS, r < R2
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[R2];
    double2 ring[2][PB];
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, w);
    } else {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w_in) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = win2[(int64_t)r * CH_BS];
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
            for (int i = 0; i < PB; ++i) ring[(b + 1) & 1][i] = nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 v = ring[b & 1][i];
                if (CPLX) {               // conj(v) * w: acc0 = re, acc1 = im
                    acc0 = fma(v.x, w[b * PB + i].x, acc0);
                    acc0 = fma(v.y, w[b * PB + i].y, acc0);
                    acc1 = fma(v.x, w[b * PB + i].y, acc1);
                    acc1 = fma(-v.y, w[b * PB + i].x, acc1);
                } else {
                    acc0 = fma(v.x, w[b * PB + i].x, acc0);
                    acc1 = fma(v.y, w[b * PB + i].y, acc1);
                }
            }
        }
        double alpha, alpha_i = 0.0;
        if (CPLX) {
            alpha = acc0;
            alpha_i = acc1;
            grid_sum2(alpha, alpha_i, epoch++, a.gran, G, a.err, smd, smu);
            if (blockIdx.x == 0 && tid == 0) {
                // first sweep assigns (the caller does not clear the H column for a chain launch)
                a.hdev[2 * j] = (t < a.ncol ? 0.0 : a.hdev[2 * j]) + alpha;
                a.hdev[2 * j + 1] = (t < a.ncol ? 0.0 : a.hdev[2 * j + 1]) + alpha_i;
            }
        } else {
            alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                   : grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu);
            if (blockIdx.x == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        }
        // ---- update phase: w -= alpha * b_j ----
        const int64_t jn = a.col0 + ((t + 1) % a.ncol);
        const double2* __restrict__ vn = (t + 1 < total)
            ? reinterpret_cast<const double2*>(a.V + jn * a.ld) + first
            : reinterpret_cast<const double2*>(a.w_in) + first;       // harmless: valid memory
        if (a.debug != 2)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2* __restrict__ nx = (b + 1 < NB) ? b2 + (int64_t)(b + 1) * PB * CH_BS : vn;
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[(b + 1) & 1][i] = nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = ring[b & 1][i];
                const int r = b * PB + i;
                if (CPLX) {               // (alpha + i alpha_i) * (p.x + i p.y), NumPy's formula
                    const double tr = alpha * p.x - alpha_i * p.y;
                    const double ti = alpha * p.y + alpha_i * p.x;
                    w[r].x = CH_OK(r) ? w[r].x - tr : 0.0;
                    w[r].y = CH_OK(r) ? w[r].y - ti : 0.0;
                } else {
                    w[r].x = CH_OK(r) ? w[r].x - alpha * p.x : 0.0;
                    w[r].y = CH_OK(r) ? w[r].y - alpha * p.y : 0.0;
                }
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
            acc = fma(w[r].x, d.x * w[r].x, acc);
            acc = fma(w[r].y, d.y * w[r].y, acc);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            acc = fma(w[r].x, w[r].x, acc);
            acc = fma(w[r].y, w[r].y, acc);
        }
    }
    const double h2 = grid_sum(acc, epoch++, a.gran, G, a.err, smd, smu);
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
                double2 o, m;
                o.x = w[r].x / h;
                o.y = w[r].y / h;
                m.x = (d.x * w[r].x) / h;
                m.y = (d.y * w[r].y) / h;
                pn2[(int64_t)r * CH_BS] = o;
                vn2[(int64_t)r * CH_BS] = m;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                double2 o;
                o.x = w[r].x / h;
                o.y = w[r].y / h;
                vn2[(int64_t)r * CH_BS] = o;
            }
        }
    }
    if (blockIdx.x == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
template <int R2, bool CPLX = false>
struct ChainShapeLds {
    // 5 rows per batch for R2 = 40, like the plain kernel (the complex instantiation spills 36 registers
    // with it and is still 14 % faster than the plain kernel; 4-row batches spill 42)
    static constexpr int PB = ChainShape<R2>::PB;
    static constexpr int NB = R2 / PB;
    static constexpr int LB = NB >= 4 ? 3 : 1;                // leading batches kept in LDS
    static constexpr int NG = NB - LB - 1;                    // update batches that still come from memory
    static_assert((NB % 2) == 0 && NG >= 0 && (NG % 2) == 0, "ring parity must reset every phase");
    static constexpr size_t LDS_BYTES = (size_t)LB * PB * CH_BS * sizeof(double2);
};

template <int R2, bool MASKED, bool CPLX = false, int FND = 0>
__global__ __launch_bounds__(CH_BS) void k_mgs_chain_lds(ChainArgs a) {
    static_assert(FND == 0 || (!MASKED && !CPLX), "the fused operator exists for the padded real kernel");
    constexpr int PB = ChainShapeLds<R2, CPLX>::PB;
    constexpr int NB = ChainShapeLds<R2, CPLX>::NB;
    constexpr int LB = ChainShapeLds<R2, CPLX>::LB;
    constexpr int NG = ChainShapeLds<R2, CPLX>::NG;
    extern __shared__ __attribute__((aligned(16))) double2 vlds[];   // [LB*PB][CH_BS]
    __shared__ double smd[2 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    // a batch that will be read again (update phase) is loaded normally so that the Infinity Cache
    // keeps it; everything that is used once (the batches that live on in LDS / the ring, the second
    // read itself, w) is loaded non-temporally and does not evict it
#define CH_LD(ptr, reuse) ((reuse) ? *(ptr) : ld_nt2(ptr))
    double2 w[R2];
    double2 ring[2][PB];
    if constexpr (FND > 0) {
        chain_apply_banded<R2, FND>(a, first, w);
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
    const int total = a.ncol * a.sweeps;
    {
        const double2* __restrict__ v2 = reinterpret_cast<const double2*>(a.V + a.col0 * a.ld) + first;
#pragma unroll
        for (int i = 0; i < PB; ++i) ring[0][i] = CH_LD(v2 + (int64_t)i * CH_BS, false);
        CH_ISSUE_FENCE();
    }
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
                ring[(b + 1) & 1][i] = CH_LD(nx + (int64_t)i * CH_BS, (b + 1 < NB) && (b + 1 >= LB) && (b + 1 <= NB - 2));
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 v = ring[b & 1][i];
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
        }
        double alpha, alpha_i = 0.0;
        CH_STAMP(a, t, total, 1);
        if (CPLX) {
            alpha = acc0;
            alpha_i = acc1;
            grid_sum2(alpha, alpha_i, epoch++, a.gran, G, a.err, smd, smu);
            if (blockIdx.x == 0 && tid == 0) {
                // first sweep assigns (the caller does not clear the H column for a chain launch)
                a.hdev[2 * j] = (t < a.ncol ? 0.0 : a.hdev[2 * j]) + alpha;
                a.hdev[2 * j + 1] = (t < a.ncol ? 0.0 : a.hdev[2 * j + 1]) + alpha_i;
            }
        } else {
#ifdef KH_CHAIN_TRACE
            alpha = grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu,
                             a.trace ? a.trace + (((size_t)blockIdx.x * total) + t) * 16 : nullptr);
#else
            alpha = (a.debug == 1) ? (acc0 + acc1) * 1e-30
                                   : grid_sum(acc0 + acc1, epoch++, a.gran, G, a.err, smd, smu);
#endif
            if (blockIdx.x == 0 && tid == 0) a.hdev[j] = (t < a.ncol ? 0.0 : a.hdev[j]) + alpha;
        }
        CH_STAMP(a, t, total, 5);
        // ---- update phase: w -= alpha * v_j, batches in reverse order ----
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
        // (a) the last batch of the dot phase is still in ring[1]
#pragma unroll
        for (int i = 0; i < PB; ++i) CH_UPD((NB - 1) * PB + i, ring[1][i]);
        // (b) batches NB-2 ... LB from memory through the ring
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int b = NB - 2 - g;
            const double2* __restrict__ nx = (g + 1 < NG) ? v2 + (int64_t)(b - 1) * PB * CH_BS : vn;
#pragma unroll
            for (int i = 0; i < PB; ++i) ring[(g + 1) & 1][i] = CH_LD(nx + (int64_t)i * CH_BS, false);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) CH_UPD(b * PB + i, ring[g & 1][i]);
        }
        CH_STAMP(a, t, total, 6);
        // (c) the head of the column from LDS (own entries: no barrier needed), one batch of reads at
        //     a time (hoisted all together they would spill)
#pragma unroll
        for (int b = LB - 1; b >= 0; --b) {
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 p = vlds[(b * PB + i) * CH_BS + tid];
                CH_UPD(b * PB + i, p);
            }
        }
#undef CH_UPD
        CH_STAMP(a, t, total, 7);
    }
    // norm: <w,w> or <w, D w>
    double acc = 0.0;
    if (a.dg != nullptr) {
        const double2* __restrict__ d2 = reinterpret_cast<const double2*>(a.dg) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 d = d2[(int64_t)r * CH_BS];
            acc = fma(w[r].x, d.x * w[r].x, acc);
            acc = fma(w[r].y, d.y * w[r].y, acc);
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            acc = fma(w[r].x, w[r].x, acc);
            acc = fma(w[r].y, w[r].y, acc);
        }
    }
    const double h2 = grid_sum(acc, epoch++, a.gran, G, a.err, smd, smu);
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
                double2 o, m;
                o.x = w[r].x / h;
                o.y = w[r].y / h;
                m.x = (d.x * w[r].x) / h;
                m.y = (d.y * w[r].y) / h;
                pn2[(int64_t)r * CH_BS] = o;
                vn2[(int64_t)r * CH_BS] = m;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                double2 o;
                o.x = w[r].x / h;
                o.y = w[r].y / h;
                vn2[(int64_t)r * CH_BS] = o;
            }
        }
    }
    if (blockIdx.x == 0 && a.hpin != nullptr) {
        __syncthreads();          // the H entries were written by thread 0 of this workgroup
        for (int i = tid; i < a.hcount; i += CH_BS)
            a.hpin[i] = __hip_atomic_load(a.hdev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) *a.errpin = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef CH_LD
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
};

template <int R2, bool MASKED, bool NTC>
__global__ __launch_bounds__(CH_BS) void k_cgs_dots(CgsArgs a) {
    constexpr int PB = CgsShape<R2>::PB;
    constexpr int NB = CgsShape<R2>::NB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[R2];
    double2 ring[2][PB];
    {
        const double2* __restrict__ win2 = reinterpret_cast<const double2*>(a.w) + first;
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const double2 v = ld_nt2(win2 + (int64_t)r * CH_BS);   // (the columns stay normal loads: the update pass re-reads them)
            w[r].x = CH_OK(r) ? v.x : 0.0;
            w[r].y = CH_OK(r) ? v.y : 0.0;
            if ((r + 1) % 8 == 0) CH_ISSUE_FENCE();
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
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2* __restrict__ nx = (b + 1 < NB) ? v2 + (int64_t)(b + 1) * PB * CH_BS : vn;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                ring[(P0 + b + 1) & 1][i] = NTC ? ld_nt2(nx + (int64_t)i * CH_BS) : nx[(int64_t)i * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const double2 v = ring[(P0 + b) & 1][i];
                acc0 = fma(v.x, w[b * PB + i].x, acc0);
                acc1 = fma(v.y, w[b * PB + i].y, acc1);
            }
        }
        const double s = wave_sum(acc0 + acc1);
        if (lane == 0) a.part[(int64_t)t * a.pstride + slot] = s;
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
#undef CH_OK
}

template <int R2, bool MASKED>
__global__ __launch_bounds__(CH_BS) void k_cgs_update(CgsArgs a) {
    constexpr int PB = CgsShape<R2>::PB;
    constexpr int NB = CgsShape<R2>::NB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t first = (int64_t)blockIdx.x * a.chunk2 + tid;
    const int64_t left = a.n2 - first;
    const int rem = (int)(left < 0 ? 0 : (left > a.chunk2 ? a.chunk2 : left));
#define CH_OK(r) (!MASKED || (r) * CH_BS < rem)
    double2 w[R2];
    double2 ring[2][PB];
    double2* __restrict__ w2 = reinterpret_cast<double2*>(a.w) + first;
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const double2 v = ld_nt2(w2 + (int64_t)r * CH_BS);
        w[r].x = CH_OK(r) ? v.x : 0.0;
        w[r].y = CH_OK(r) ? v.y : 0.0;
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
        const double h = a.coef[t];
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
                w[r].x = CH_OK(r) ? w[r].x - h * p.x : 0.0;
                w[r].y = CH_OK(r) ? w[r].y - h * p.y : 0.0;
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
                double2 m;
                m.x = d.x * w[r].x;
                m.y = d.y * w[r].y;
                acc = fma(w[r].x, m.x, acc);
                acc = fma(w[r].y, m.y, acc);
                m2[(int64_t)r * CH_BS] = m;
                w2[(int64_t)r * CH_BS] = w[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            if (r * CH_BS < rem) {
                acc = fma(w[r].x, w[r].x, acc);
                acc = fma(w[r].y, w[r].y, acc);
                w2[(int64_t)r * CH_BS] = w[r];
            }
        }
    }
    const double s = wave_sum(acc);
    if (lane == 0) a.part[blockIdx.x * (CH_BS / 64) + wid] = s;
#undef CH_OK
}

}  // namespace kh
