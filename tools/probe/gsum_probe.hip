// Probe (not product): latency of the in-kernel grid-wide sum of the chain kernel, and of candidate variants.
// V0 is what the chain kernels use; V1 ... V6 are the alternatives that were measured against each other
// (V3: every workgroup sweeps the fabric, the round-1 scheme without s_sleep).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../krypy_amd/csrc gsum_probe.hip -o gsum_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include "chain.h"

using namespace kh;

struct XcdState {
    unsigned* leader;               // [16] launch id of the last election per XCC
    unsigned long long* res;        // [16][2 parities][2 granules]
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// V1: two polls in flight per thread
__device__ __forceinline__ unsigned long long poll2(const unsigned long long* p, unsigned epoch) {
    unsigned long long x0 = ld_agent(p);
    __builtin_amdgcn_s_sleep(4);
    unsigned long long x1 = ld_agent(p);
    unsigned spins = 0;
    while (true) {
        if ((unsigned)(x0 >> 32) == epoch) return x0;
        x0 = x1;
        x1 = ld_agent(p);
        if (++spins > (1u << 22)) return x0;
    }
}

template <int V>
__device__ __forceinline__ double gsum(double part, unsigned epoch, unsigned long long* gran, int G, int* err,
                                       double* smd, unsigned* smu, XcdState xs, unsigned xcc, bool leader) {
    if (V == 0) {       // the library's grid_sum (XCD leaders)
        GridRole role;
        role.xcc = xcc;
        role.leader = leader;
        return grid_sum(part, epoch, gran, G, err, smd, smu, role, xs.res);
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CH_BS / 64;
    const double ws = wave_sum_dpp(part);
    if (lane == 0) smd[wid] = ws;
    __syncthreads();
    unsigned long long* slot = gran + (size_t)(epoch & 1u) * (2 * CH_GMAX);
    if (tid == 0) {
        double s = smd[0];
        for (int i = 1; i < NW; ++i) s += smd[i];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        st_agent(slot + 2 * blockIdx.x, tag | (bits & 0xffffffffull));
        st_agent(slot + 2 * blockIdx.x + 1, tag | (bits >> 32));
    }
    if (V == 1 || V == 3 || leader) {
        constexpr bool NOSLEEP = (V == 3 || V >= 4);
        unsigned mine = 0;
        if (tid < 2 * G) {
            if (V == 1) mine = (unsigned)poll2(slot + tid, epoch);
            else if (NOSLEEP) {
                unsigned long long x = ld_agent(slot + tid);
                while ((unsigned)(x >> 32) != epoch) x = ld_agent(slot + tid);
                mine = (unsigned)x;
            } else mine = (unsigned)poll_granule(slot + tid, epoch, err);
        }
        const unsigned low = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xf, 0xf, false);
        const unsigned long long bits = ((unsigned long long)mine << 32) | low;
        const double v = ((lane & 1) && tid < 2 * G) ? __longlong_as_double((long long)bits) : 0.0;
        const double wv = wave_sum_dpp(v);
        if (lane == 0) smd[NW + wid] = wv;
        __syncthreads();
        double s = smd[NW];
        for (int i = 1; i < NW; ++i) s += smd[NW + i];
        if (V >= 2 && V != 3 && tid == 0) {       // the XCD's leader hands the total to its neighbours through the shared L2
            const unsigned long long bits2 = (unsigned long long)__double_as_longlong(s);
            const unsigned long long tag = (unsigned long long)epoch << 32;
            unsigned long long* r = xs.res + ((size_t)xcc * 2 + (epoch & 1u)) * 4;
            r[0] = tag | (bits2 & 0xffffffffull);
            r[1] = tag | (bits2 >> 32);
        }
        return s;
    }
    // V == 2, not the leader: poll the XCD's result pair (L2-served)
    const unsigned long long* r = xs.res + ((size_t)xcc * 2 + (epoch & 1u)) * 4;
    if (V >= 5) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2* r2 = reinterpret_cast<const u64x2*>(r);
        u64x2 ab;
        unsigned spins = 0;
        while (true) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ab) : "v"(r2) : "memory");
            if ((unsigned)(ab.x >> 32) == epoch && (unsigned)(ab.y >> 32) == epoch) break;
            if (V == 6) __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        return __longlong_as_double((long long)(((ab.y & 0xffffffffull) << 32) | (ab.x & 0xffffffffull)));
    }
    unsigned long long a = ld_agent(r), b = ld_agent(r + 1);
    unsigned spins = 0;
    while ((unsigned)(a >> 32) != epoch || (unsigned)(b >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        a = ld_agent(r);
        b = ld_agent(r + 1);
        if (++spins > (1u << 22)) {
            __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    return __longlong_as_double((long long)(((b & 0xffffffffull) << 32) | (a & 0xffffffffull)));
}

template <int V>
__global__ __launch_bounds__(CH_BS) void k(unsigned long long* gran, int* err, double* out, int iters, unsigned epoch0,
                                           XcdState xs, unsigned launch_id, const double2* bg, int bg_rows, unsigned* xcc_out) {
    __shared__ double smd[4 * (CH_BS / 64)];
    __shared__ unsigned smu[2 * CH_GMAX];
    __shared__ int lead;
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    unsigned xcc = 0;
    bool leader = false;
    if (V != 1 && V != 3) {
        xcc = xcc_id();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_max(xs.leader + xcc, launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lead = old < launch_id;
            xcc_out[blockIdx.x] = xcc | (lead ? 0x100u : 0u);
        }
        __syncthreads();
        leader = lead != 0;
    }
    unsigned epoch = epoch0;
    double acc = 0.0;
    double2 sink = make_double2(0.0, 0.0);
    for (int it = 0; it < iters; ++it) {
        const double part = (double)((blockIdx.x * CH_BS + tid) % 1000 + it % 7) * 1e-3;
        if (bg_rows > 0) {      // background: HBM loads in flight while the sum runs (what a prefetch would do)
            const double2* p = bg + ((size_t)(blockIdx.x * 64 + (it % 64)) * bg_rows) * CH_BS + tid;
            for (int r = 0; r < bg_rows; ++r) {
                const double2 v = ld_nt2(p + (size_t)r * CH_BS);
                sink.x += v.x;
                sink.y += v.y;
            }
        }
        const double s = gsum<V>(part, epoch++, gran, G, err, smd, smu, xs, xcc, leader);
        acc += s;
    }
    if (tid == 0) out[blockIdx.x] = acc + (sink.x + sink.y) * 1e-300;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
static void run(const char* name, int G, int iters, int bg_rows, const double2* bg, unsigned long long* gran, int* err,
                double* out, XcdState xs, unsigned* xcc_out, unsigned& epoch, unsigned& launch_id) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        ++launch_id;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<V>), dim3(G), dim3(CH_BS), 0, 0, gran, err, out, iters, epoch, xs, launch_id, bg, bg_rows, xcc_out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        epoch += iters;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<double> h(G);
    int herr = 0;
    CK(hipMemcpy(h.data(), out, sizeof(double) * G, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 1; i < G; ++i) if (h[i] != h[0]) same = 0;
    printf("%-44s G=%3d bg_rows=%d: %.3f us per sum   (all workgroups agree: %s, err=%d, value %.6e)\n", name, G, bg_rows,
           best * 1e3 / iters, same ? "yes" : "NO", herr, h[0]);
    if (herr) CK(hipMemset(err, 0, sizeof(int)));
}

int main() {
    const int iters = 2000;
    unsigned long long* gran; int* err; double* out; XcdState xs; unsigned* xcc_out; double2* bg;
    CK(hipMalloc(&gran, sizeof(unsigned long long) * 4 * CH_GMAX));
    CK(hipMemset(gran, 0, sizeof(unsigned long long) * 4 * CH_GMAX));
    CK(hipMalloc(&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int)));
    CK(hipMalloc(&out, sizeof(double) * 512));
    CK(hipMalloc(&xs.leader, sizeof(unsigned) * 16)); CK(hipMemset(xs.leader, 0, sizeof(unsigned) * 16));
    CK(hipMalloc(&xs.res, sizeof(unsigned long long) * 16 * 8)); CK(hipMemset(xs.res, 0, sizeof(unsigned long long) * 16 * 8));
    CK(hipMalloc(&xcc_out, sizeof(unsigned) * 512));
    const size_t bgbytes = (size_t)256 * 64 * 8 * CH_BS * sizeof(double2);      // 1 GB
    CK(hipMalloc(&bg, bgbytes)); CK(hipMemset(bg, 0, bgbytes));
    unsigned epoch = 1, launch_id = 0;
    for (int G : {245, 128, 17}) {
        for (int bgr : {0, 2}) {
            run<0>("V0 the library's grid_sum (chain.h)", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<3>("V3 no s_sleep in the poll loop", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<1>("V1 two polls in flight", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<2>("V2 XCD leaders sweep, neighbours read L2", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<4>("V4 = V2, leaders poll without s_sleep", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<5>("V5 = V4, neighbours: one 16-B load, no sleep", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
            run<6>("V6 = V5 with s_sleep for the neighbours", G, iters, bgr, bg, gran, err, out, xs, xcc_out, epoch, launch_id);
        }
    }
    std::vector<unsigned> hx(256);
    CK(hipMemcpy(hx.data(), xcc_out, sizeof(unsigned) * 256, hipMemcpyDeviceToHost));
    printf("xcc of the first 16 workgroups (0x100 = leader):");
    for (int i = 0; i < 16; ++i) printf(" %x", hx[i]);
    printf("\n");
    return 0;
}
