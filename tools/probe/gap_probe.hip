// Launch-to-launch gap of dependent kernels on one stream (what an Arnoldi step costs beyond its kernel at small N):
//   (a) kernels back to back, (b) + hipEventRecord after every kernel (what kh_arnoldi_step_begin does),
//   (c) + the kernel writes a word to pinned host memory (the H column of the chain kernels), (d) b + c.
// Each kernel spins for ~20 us on the wall clock so that the host is always ahead; the gap is the time between the end
// stamp of one launch and the start stamp of the next (wall_clock64, 100 MHz), averaged over 2000 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(unsigned long long* stamps, int i, int ticks, double* pin, double* dev) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * i] = t0;
    while (wall_clock64() - t0 < (unsigned long long)ticks) {}
    if (dev != nullptr && threadIdx.x == 0) dev[blockIdx.x] = (double)i;          // something to write back
    if (pin != nullptr && blockIdx.x == 0 && threadIdx.x < 64) pin[threadIdx.x] = (double)i;
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * i + 1] = wall_clock64();
}

int main() {
    const int n = 2000, grid = 256;
    unsigned long long* stamps;
    double *pin, *dev;
    CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * n));
    CK(hipHostMalloc(&pin, 4096, hipHostMallocDefault));
    CK(hipMalloc(&dev, sizeof(double) * grid));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<hipEvent_t> ev(8);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<unsigned long long> h(2 * n);
    const char* names[4] = {"kernels back to back", "+ hipEventRecord after every kernel", "+ 512 B to pinned host memory per kernel",
                            "+ event record and pinned write"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(stamps, 0, sizeof(unsigned long long) * 2 * n, st));
            for (int i = 0; i < n; ++i) {
                hipLaunchKernelGGL(k_spin, dim3(grid), dim3(512), 0, st, stamps, i, 2000, (mode & 2) ? pin : nullptr, dev);
                if (mode & 1) CK(hipEventRecord(ev[i & 7], st));
            }
            CK(hipStreamSynchronize(st));
        }
        CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
        double sum = 0.0, mx = 0.0;
        int cnt = 0;
        for (int i = 100; i + 1 < n; ++i) {
            const double g = (double)(h[2 * (i + 1)] - h[2 * i + 1]) / 100.0;      // us
            sum += g;
            mx = g > mx ? g : mx;
            ++cnt;
        }
        printf("%-46s gap %.2f us (max %.1f)\n", names[mode], sum / cnt, mx);
    }
    return 0;
}
