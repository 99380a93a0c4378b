// Probe (not product): what the second read of a long column costs when it walks the column BACKWARDS.
// Geometry of k_mgs_chain<48>: 245 workgroups x 512 lanes x ROWS rows of 16 B; per link a fresh column is streamed
// (the dot phase), then read again (the update phase) first-to-last or last-to-first.  The rows of the first read
// that are marked "keep" are loaded normally (L2 may keep them), the others non-temporally.
//   reread_probe            -> table: us per link for forward / reverse re-reads and 0 ... ROWS kept rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kh_internal.h"
#include "kernels.h"
#include "chain.h"
using namespace kh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ROWS, int PB>
__global__ __launch_bounds__(CH_BS) void k_reread(const double2* cols, int64_t ld2, int ncols, int links, int keep, int reverse,
                                                  int again, int nt_again, double* out) {
    const int tid = threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * ROWS * CH_BS + tid;
    double acc = 0.0;
    for (int t = 0; t < links; ++t) {
        const double2* c = cols + (int64_t)(t % ncols) * ld2 + first;
        for (int r0 = 0; r0 < ROWS; r0 += PB) {
            double2 v[PB];
            const bool kp = r0 >= ROWS - keep;          // the LAST `keep` rows of the first read stay in L2 (if it wants them)
#pragma unroll
            for (int i = 0; i < PB; ++i) v[i] = kp ? c[(int64_t)(r0 + i) * CH_BS] : ld_nt2(c + (int64_t)(r0 + i) * CH_BS);
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) acc = fma(v[i].x, v[i].y, acc);
        }
        __syncthreads();
        for (int g = 0; g < again; g += PB) {
            const int r0 = reverse ? ROWS - PB - g : g;
            double2 v[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) v[i] = nt_again ? ld_nt2(c + (int64_t)(r0 + i) * CH_BS) : c[(int64_t)(r0 + i) * CH_BS];
            CH_ISSUE_FENCE();
#pragma unroll
            for (int i = 0; i < PB; ++i) acc = fma(v[i].x, v[i].y, acc);
        }
    }
    if (tid == 0) out[blockIdx.x] = acc;
}

int main() {
    constexpr int ROWS = 48, PB = 4;
    const int G = 245, ncols = 48, links = 480;
    const int64_t ld2 = (int64_t)G * ROWS * CH_BS;
    double2* cols; double* out;
    CK(hipMalloc(&cols, sizeof(double2) * ld2 * ncols + (1 << 20)));
    CK(hipMemset(cols, 0, sizeof(double2) * ld2 * ncols + (1 << 20)));
    CK(hipMalloc(&out, sizeof(double) * 512));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("column = %.1f MB, %d workgroups x %d rows\n", ld2 * 16.0 / 1e6, G, ROWS);
    for (int again : {0, ROWS})
    for (int reverse = 0; reverse < 2; ++reverse)
    for (int nt_again = 0; nt_again < 2; ++nt_again)
    for (int keep : {0, 8, 12, 16, 24, 48}) {
        if (again == 0 && (reverse || nt_again || keep)) continue;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_reread<ROWS, PB>), dim3(G), dim3(CH_BS), 0, 0, cols, ld2, ncols, links, keep, reverse, again, nt_again, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("again=%2d reverse=%d nt_again=%d keep=%2d: %.2f us per link, %.2f TB/s requested\n", again, reverse, nt_again, keep,
               best * 1e3 / links, (double)G * CH_BS * 16.0 * (ROWS + again) / (best * 1e-3 / links) * 1e-12);
        fflush(stdout);
    }
    return 0;
}
