// Probe (not product): LDS-DMA through inline asm as the chain kernel uses it.
//   each wave copies R rows of 1 KB (64 lanes x 16 B) global -> its own LDS rows with global_load_lds_dwordx4,
//   waits with a counted s_waitcnt, reads the rows back with ordinary ds_reads and stores them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int BS = 512;
constexpr int R = 6;

typedef __attribute__((address_space(3))) double2 lds_double2;

__device__ __forceinline__ void dma16(const double2* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(BS) void k(const double2* __restrict__ src, double2* __restrict__ dst,
                                        const double2* __restrict__ other, double2* __restrict__ dst2) {
    extern __shared__ __attribute__((aligned(16))) double2 vlds[];
    const int tid = threadIdx.x;
    const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned base = (unsigned)(size_t)((lds_double2*)vlds);
    const double2* s = src + (size_t)blockIdx.x * R * BS + tid;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) dma16(s + r * BS, base + (unsigned)((r * BS + wid * 64) * 16));
    // two ordinary loads issued AFTER the DMAs (like the ring loads of the next column)
    const double2 o0 = other[(size_t)blockIdx.x * BS + tid];
    const double2 o1 = other[(size_t)(blockIdx.x + gridDim.x) * BS + tid];
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    double2* d = dst + (size_t)blockIdx.x * R * BS + tid;
#pragma unroll
    for (int r = 0; r < R; ++r) d[r * BS] = vlds[r * BS + tid];
    dst2[(size_t)blockIdx.x * BS + tid] = make_double2(o0.x + o1.x, o0.y + o1.y);
}

int main() {
    const int G = 256;
    const size_t n = (size_t)G * R * BS;
    std::vector<double2> h(n), o(2 * (size_t)G * BS);
    for (size_t i = 0; i < n; ++i) h[i] = make_double2((double)i, -(double)i);
    for (size_t i = 0; i < o.size(); ++i) o[i] = make_double2(1.0 * i, 2.0 * i);
    double2 *src, *dst, *oth, *dst2;
    hipMalloc(&src, n * 16); hipMalloc(&dst, n * 16); hipMalloc(&oth, o.size() * 16); hipMalloc(&dst2, (size_t)G * BS * 16);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipMemcpy(oth, o.data(), o.size() * 16, hipMemcpyHostToDevice);
    hipMemset(dst, 0, n * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, R * BS * 16);
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(k, dim3(G), dim3(BS), R * BS * 16, 0, src, dst, oth, dst2);
    hipDeviceSynchronize();
    std::vector<double2> g(n);
    hipMemcpy(g.data(), dst, n * 16, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) if (g[i].x != h[i].x || g[i].y != h[i].y) ++bad;
    printf("dma_test: %zu mismatches of %zu (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
