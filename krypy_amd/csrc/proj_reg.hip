// Launcher of the register-resident deflation projector (proj_reg.h): a translation unit of its own.
#include <hip/hip_runtime.h>

#include "kh_internal.h"
#include "proj_reg.h"

namespace kh {

static constexpr size_t PR_GRAN_WORDS = (size_t)2 * CH_GMAX * PR_NV * 2;
static constexpr size_t PR_RES_WORDS = (size_t)16 * 2 * PR_NV * 2;
static constexpr size_t PR_LEAD_WORDS = 8;          // 16 unsigned stamps

static hipError_t proj_reg_reset(kh_ctx ctx) {
    const size_t words = PR_GRAN_WORDS + PR_RES_WORDS + PR_LEAD_WORDS;
    if (ctx->proj_gran == nullptr) {
        hipError_t e = hipMalloc(&ctx->proj_gran, sizeof(unsigned long long) * words);
        if (e != hipSuccess) {
            ctx->proj_gran = nullptr;
            return e;
        }
    }
    ctx->proj_epoch = 1;
    return hipMemsetAsync(ctx->proj_gran, 0, sizeof(unsigned long long) * words, ctx->stream);
}

void proj_reg_free(kh_ctx ctx) {
    if (ctx->proj_gran != nullptr) (void)hipFree(ctx->proj_gran);
    ctx->proj_gran = nullptr;
    if (ctx->proj_err != nullptr) (void)hipFree(ctx->proj_err);
    if (ctx->proj_err_pin != nullptr) (void)hipHostFree(ctx->proj_err_pin);
    ctx->proj_err = nullptr;
    ctx->proj_err_pin = nullptr;
}

template <int R2>
static hipError_t launch_proj(kh_ctx ctx, int G, ProjRegArgs& a) {
    static int blocks_per_cu = -1;
    constexpr size_t lds = ProjRegShape<R2>::LDS_BYTES;
    if (blocks_per_cu < 0) {
        if (lds > 0) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_proj_reg<R2>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return e0;
        }
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_proj_reg<R2>, CH_BS, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb;
    }
    if ((int64_t)blocks_per_cu * ctx->ncu < G) return hipErrorCooperativeLaunchTooLarge;      // the sum needs every workgroup resident
    hipLaunchKernelGGL((k_proj_reg<R2>), dim3(G), dim3(CH_BS), lds, ctx->stream, a);
    return hipGetLastError();
}

// z <- complement projection of z with z register-resident (one launch); ya_dev: d doubles on the device or nullptr.
// r2 / G: the chain-kernel geometry of vectors of this length (krylov_hip.hip: chain_geometry).  Returns 1 when the
// launch was made, 0 when this shape / state is not served (the caller runs the four-launch form), negative on error.
int proj_reg_apply(kh_ctx ctx, kh_proj p, double* z, int64_t zld, int r2, int G, double* ya_dev) {
    // (ctx->proj_reg_why: why the last call declined - kh_ctx_get "proj_reg_why"; 0 = it ran)
    ctx->proj_reg_why = 1;
    if (!ctx->proj_reg || !ctx->chain_enabled || kh_multi(ctx) || p->cplx || p->d < 1 || p->d > PR_NV || p->iterations < 1) return 0;
    ctx->proj_reg_why = 2;
    if (r2 < 16 || G > CH_GMAX / 2 || G > 256) return 0;       // (short vectors: the four launches are latency-bound either way)
    const int64_t n = p->W->n;
    const int64_t chunk2 = (int64_t)r2 * CH_BS;
    const int64_t need_ld = (int64_t)G * chunk2 * 2;
    ctx->proj_reg_why = 3;
    if (p->W->ld != p->V->ld || p->W->ld < need_ld || zld < need_ld) return 0;      // padded blocks only
    ctx->proj_reg_why = 4;
    if (ctx->proj_gran == nullptr || ctx->proj_epoch > 0xfff00000u) {
        if (ctx->proj_gran != nullptr) KH_HIP(hipStreamSynchronize(ctx->stream));
        KH_HIP(proj_reg_reset(ctx));
    }
    if (ctx->proj_err == nullptr) {
        KH_HIP(hipMalloc(&ctx->proj_err, sizeof(int)));
        KH_HIP(hipMemsetAsync(ctx->proj_err, 0, sizeof(int), ctx->stream));
        KH_HIP(hipHostMalloc(&ctx->proj_err_pin, sizeof(int), hipHostMallocDefault));
        *ctx->proj_err_pin = 0;
    }
    ProjRegArgs a;
    a.n2 = (n + 1) >> 1;
    a.chunk2 = chunk2;
    a.Wb = p->W->d;
    a.Vb = p->V->d;
    a.ld = p->W->ld;
    a.d = (int)p->d;
    a.iterations = p->iterations;
    a.z = z;
    a.T = p->T;
    a.WRH = p->WRH;
    a.ya = ya_dev;
    a.gran = ctx->proj_gran;
    a.res = ctx->proj_gran + PR_GRAN_WORDS;
    a.xcc_leader = reinterpret_cast<unsigned*>(ctx->proj_gran + PR_GRAN_WORDS + PR_RES_WORDS);
    a.epoch0 = ctx->proj_epoch;
    a.err = ctx->proj_err;      // (its own word: a timeout here must not make the chain kernels that follow break out of THEIR waits)
    hipError_t e;
    switch (r2) {
        case 16: e = launch_proj<16>(ctx, G, a); break;
        case 24: e = launch_proj<24>(ctx, G, a); break;
        case 32: e = launch_proj<32>(ctx, G, a); break;
        case 40: e = launch_proj<40>(ctx, G, a); break;
        case 48: e = launch_proj<48>(ctx, G, a); break;
        case 56: e = launch_proj<56>(ctx, G, a); break;
        default: return 0;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    ctx->proj_reg_why = 0;
    ctx->proj_epoch += (unsigned)p->iterations;
    ctx->n_proj_reg += 1;
    if (ctx->proj_fault) {          // tests: what a timed-out sum leaves behind - the error word set, garbage in z
        ctx->proj_fault = 0;
        KH_HIP(hipMemsetAsync(ctx->proj_err, 1, 1, ctx->stream));
        KH_HIP(hipMemsetAsync(z, 0, sizeof(double) * (size_t)(n < 4096 ? n : 4096), ctx->stream));
    }
    // the word travels to pinned memory behind the launch: whoever synchronises next (kh_arnoldi_step_end for a deflated
    // step whatever its Gram-Schmidt variant, kh_proj_apply_complement) sees it and runs the four-launch projector instead
    KH_HIP(hipMemcpyAsync(ctx->proj_err_pin, ctx->proj_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return 1;
}

}  // namespace kh
