// xr: sums across the ranks of one node inside small kernels of our own - IPC-mapped mailboxes, tagged granules over xGMI.
//
// Why.  On N ranks every Arnoldi step needs the sum over the ranks of a panel of inner products (the inner
// products themselves: /root/reference/krypy/utils.py:182-183) and of the norm.  As `ncclAllReduce` that is one RCCL
// kernel per panel - 14-29 us each on this part when measured through a 1-rank communicator (profiles/r03_halo_overlap.md) -
// behind a reduction launch of our own, twice per step: a quarter of a step at N/8 of the benchmark problem.  The panels
// are tiny (1 ... 2 k + 1 doubles); what they need is latency, not bandwidth, and no ring.
//
// How.  Each rank allocates a MAILBOX in fine-grained device memory and exports it (hipIpcGetMemHandle); the handles
// travel over the launcher's rendezvous (krypy_amd/dist.py: enable_xr) and every rank maps every peer's mailbox
// (hipIpcOpenMemHandle).  An exchange of `count` values is ONE kernel on the compute stream:
//     lane v:  for every rank r:  store {epoch | low word}, {epoch | high word} of my value v into r's mailbox,
//                                 slot [epoch parity][my rank][v]           (relaxed SYSTEM-scope 8-byte stores: xGMI writes)
//              for r = 0 .. nranks-1:  poll MY mailbox slot [parity][r][v] until both tags are this epoch
//              total = (...((x_0 + x_1) + x_2) ...)                           rank order: the same bits on every rank
// A granule carries its own tag, so no flag, no fence and no ordering between the stores is needed (the guide's R2
// granule).  Two parities: a rank can be one exchange ahead of a peer, never two - to publish epoch e + 2 it must have
// gathered epoch e + 1, which the slowest peer publishes only after it has finished gathering epoch e (stream order).
// Fused form: the workgroup that adds up the partial sums of value v (k_reduce_partials) exchanges v itself - reduction
// and all-reduce in one launch (`reduce_partials_allreduce`).
//
// Rank invariance.  Whether a sum goes this way or through RCCL changes nothing in the PATTERN of collectives a peer
// sees only if every rank decides alike: the transport is switched on by the host layer after every rank has reported a
// successful attach (kh_ctx_set "xr"), and per call only `count` - the same on every rank - decides between the fused
// kernel, the chunked exchange and (transport off) RCCL.  A timeout (a peer that never arrives: 60 s) is an ERROR
// reported at the next host synchronisation (xr_check), never a fallback.
//
// What is measured and what is not: one rank in loopback and two PROCESSES on one GPU exchanging through IPC handles
// (tests/test_gpu_xr.py) run the protocol, the tags, the rank-ordered sum and the timeout path.  Nothing has run between
// two GPUs: the latency of a system-scope store over an xGMI link is unmeasured.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "kh_internal.h"
#include "kernels.h"
#include "xr_dev.h"

namespace kh {

struct XrArgs {
    unsigned long long* peer[XR_MAXRANKS];
    int rank, nranks;
    unsigned epoch;
    int* err;                 // mapped pinned host word
    long long timeout_ticks;  // of the 100 MHz wall clock
};

static constexpr size_t XR_BOX_WORDS = (size_t)2 * XR_MAXRANKS * XR_MAXV * 2;      // [parity][sender][value][lo, hi]

// my value v to every rank's mailbox (mine included: the gather below reads every contribution from the mailbox, so one
// code path - and a 1-rank loopback runs all of it)
__device__ __forceinline__ void xr_publish(const XrArgs& a, int v, double x) {
    xr_put_all(a.peer, a.rank, a.nranks, a.epoch, v, x);
}

// the sum over the ranks of value v, contributions added in rank order
__device__ __forceinline__ double xr_gather(const XrArgs& a, int v) {
    int bad = 0;
    const double t = xr_take_all(a.peer[a.rank], a.nranks, a.epoch, v, a.timeout_ticks, &bad);
    if (bad) __hip_atomic_store(a.err, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return t;
}

// vals[i] <- sum over the ranks of vals[i], i < count <= XR_MAXV; one lane per value
static __global__ __launch_bounds__(64) void k_xr_allreduce(double* __restrict__ vals, int count, XrArgs a) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= count) return;
    xr_publish(a, v, vals[v]);
    vals[v] = xr_gather(a, v);
}

// k_reduce_partials (mode 0) and the exchange of the result in one launch: workgroup c adds up the partial sums of value c
// in the order k_reduce_partials adds them, lane 0 exchanges the sum
static __global__ __launch_bounds__(BS) void k_reduce_partials_xr(const double* __restrict__ part, int nb, int pstride,
                                                                  double* __restrict__ out, XrArgs a) {
    __shared__ double sm[8];
    const double* p = part + (int64_t)blockIdx.x * pstride;
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += BS) v += p[i];
    const double r = block_sum(v, sm);
    if (threadIdx.x == 0) {
        xr_publish(a, (int)blockIdx.x, r);
        out[blockIdx.x] = xr_gather(a, (int)blockIdx.x);
    }
}

static int xr_args(kh_ctx ctx, XrArgs& a) {
    if (ctx->xr_epoch > 0xfff00000u)
        return fail(KH_ERR_COMM, "xr: the epoch counter of the cross-rank exchange is exhausted (4e9 sums in one context); "
                                 "create a new context");
    for (int r = 0; r < XR_MAXRANKS; ++r) a.peer[r] = r < ctx->xr_nranks ? ctx->xr_peer[r] : nullptr;
    a.rank = ctx->xr_rank;
    a.nranks = ctx->xr_nranks;
    a.epoch = ctx->xr_epoch++;
    {
        void* dp = nullptr;
        KH_HIP(hipHostGetDevicePointer(&dp, ctx->xr_err_pin, 0));
        a.err = static_cast<int*>(dp);
    }
    if (ctx->xr_timeout_ms <= 0) {
        const char* e = getenv("KRYPY_AMD_XR_TIMEOUT_S");
        ctx->xr_timeout_ms = (int64_t)((e ? atof(e) : 60.0) * 1e3);
    }
    a.timeout_ticks = (long long)ctx->xr_timeout_ms * 100000ll;      // wall_clock64: 100 MHz
    return 0;
}

int xr_allreduce_dev(kh_ctx ctx, double* dev, int64_t count) {
    for (int64_t done = 0; done < count; done += XR_MAXV) {
        const int nc = (int)std::min<int64_t>(XR_MAXV, count - done);
        XrArgs a;
        KH_TRY(xr_args(ctx, a));
        hipLaunchKernelGGL(k_xr_allreduce, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, ctx->stream, dev + done, nc, a);
        KH_HIP(hipGetLastError());
        ctx->n_xr += 1;
    }
    return 0;
}

int reduce_partials_allreduce(kh_ctx ctx, const double* part, int nb, int pstride, double* out, int count, bool multi) {
    if (count <= 0) return 0;
    if (multi && ctx->xr_on && count <= XR_MAXV) {
        XrArgs a;
        KH_TRY(xr_args(ctx, a));
        hipLaunchKernelGGL(k_reduce_partials_xr, dim3((unsigned)count), dim3(BS), 0, ctx->stream, part, nb, pstride, out, a);
        KH_HIP(hipGetLastError());
        ctx->n_xr += 1;
        ctx->n_xr_fused += 1;
        ctx->n_allreduce += 1;
        return 0;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)count), dim3(BS), 0, ctx->stream, part, nb, pstride, out, 0);
    KH_HIP(hipGetLastError());
    if (multi) KH_TRY(comm_allreduce_dev(ctx, out, count));
    return 0;
}

int xr_check(kh_ctx ctx) {
    if (ctx->xr_err_pin == nullptr || *ctx->xr_err_pin == 0) return 0;
    const int who = *ctx->xr_err_pin - 1;
    *ctx->xr_err_pin = 0;
    if (who == 99)
        return fail(KH_ERR_COMM, "xh: a neighbour's boundary rows did not arrive within the timeout in a sharded SpMV on rank %d of %d "
                                 "(KRYPY_AMD_XR_TIMEOUT_S): a peer has died or applies a different sequence of operators", ctx->rank, ctx->nranks);
    return fail(KH_ERR_COMM, "xr: rank %d's contribution to a cross-rank sum did not arrive within the timeout (rank %d of %d "
                             "waited; KRYPY_AMD_XR_TIMEOUT_S): a peer has died or runs a different sequence of collectives",
                who, ctx->xr_rank, ctx->xr_nranks);
}

void xr_free(kh_ctx ctx) {
    ctx->xr_on = 0;
    for (int r = 0; r < XR_MAXRANKS; ++r) {
        if (ctx->xr_peer[r] != nullptr && r != ctx->xr_rank) (void)hipIpcCloseMemHandle(ctx->xr_peer[r]);
        ctx->xr_peer[r] = nullptr;
    }
    if (ctx->xr_box != nullptr) (void)hipFree(ctx->xr_box);
    ctx->xr_box = nullptr;
    // (the error word stays: an operator whose halo travels inside its launch - xh - reports through it too; kh_ctx_destroy frees it)
    if (ctx->xr_own_comm) {
        ctx->rank = 0;
        ctx->nranks = 1;
        ctx->xr_own_comm = 0;
    }
    ctx->xr_nranks = 0;
}

void xh_free(kh_mat A) {
    if (A == nullptr) return;
    A->xh_on = 0;
    if (!A->xh_self) {
        if (A->xh_prev != nullptr) (void)hipIpcCloseMemHandle(A->xh_prev);
        if (A->xh_next != nullptr) (void)hipIpcCloseMemHandle(A->xh_next);
    }
    A->xh_prev = A->xh_next = nullptr;
    if (A->xh_box != nullptr) (void)hipFree(A->xh_box);
    A->xh_box = nullptr;
}

}  // namespace kh

using namespace kh;

extern "C" {

// ---- xh: the halo of a block-row shard through IPC-mapped granules (kernels.h: k_spmv_dia<..., XH>) ----
int kh_mat_xh_export(kh_ctx ctx, kh_mat A, unsigned char handle[64]) {
    KH_ARG(ctx && A && handle, "kh_mat_xh_export: NULL");
    KH_ARG(A->kind == KH_MAT_CSR, "kh_mat_xh_export: a real CSR shard expected");
    KH_HIP(hipSetDevice(ctx->device));
    if (A->xh_box == nullptr) {
        const int64_t ng = A->nrecv_prev + A->nrecv_next;
        const size_t bytes = sizeof(unsigned long long) * (size_t)std::max<int64_t>(2 * ng * 2, 2);
        void* p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
        }
        KH_HIP(e);
        A->xh_box = static_cast<unsigned long long*>(p);
        KH_HIP(hipMemset(A->xh_box, 0, bytes));
        KH_HIP(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    KH_HIP(hipIpcGetMemHandle(&h, A->xh_box));
    memcpy(handle, &h, 64);
    return 0;
}

// prev / next: the neighbours' 64-byte handles (NULL: no such neighbour, or - self_loop - this rank's own box: the slab of an
// operator that is periodic across the slab boundary, kh_ctx_set "halo_loopback"); prev_ng / next_ng: the ghost entries
// (nrecv_prev + nrecv_next) of THEIR boxes, prev_off: the previous rank's nrecv_prev (my first rows are its ghosts from its
// next rank, which sit behind its ghosts from its previous one)
int kh_mat_xh_attach(kh_ctx ctx, kh_mat A, const unsigned char* prev, int64_t prev_ng, int64_t prev_off, const unsigned char* next,
                     int64_t next_ng, int self_loop) {
    KH_ARG(ctx && A, "kh_mat_xh_attach: NULL");
    KH_ARG(A->xh_box != nullptr, "kh_mat_xh_attach: kh_mat_xh_export first");
    KH_ARG(A->xh_prev == nullptr && A->xh_next == nullptr, "kh_mat_xh_attach: already attached");
    KH_HIP(hipSetDevice(ctx->device));
    if (self_loop) {
        A->xh_self = 1;
        A->xh_prev = A->xh_next = A->xh_box;
        A->xh_prev_ng = A->xh_next_ng = A->nrecv_prev + A->nrecv_next;
        A->xh_prev_off = A->nrecv_prev;
        return 0;
    }
    auto open = [&](const unsigned char* hb, unsigned long long** out) -> int {
        hipIpcMemHandle_t h;
        memcpy(&h, hb, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(KH_ERR_COMM, "kh_mat_xh_attach: cannot map a neighbour's ghost granules (%s)", hipGetErrorString(e));
        }
        *out = static_cast<unsigned long long*>(p);
        return 0;
    };
    if (prev != nullptr && A->nsend_prev > 0) KH_TRY(open(prev, &A->xh_prev));
    if (next != nullptr && A->nsend_next > 0) {
        const int rc = open(next, &A->xh_next);
        if (rc != 0) {
            if (A->xh_prev != nullptr) (void)hipIpcCloseMemHandle(A->xh_prev);
            A->xh_prev = nullptr;
            return rc;
        }
    }
    A->xh_prev_ng = prev_ng;
    A->xh_next_ng = next_ng;
    A->xh_prev_off = prev_off;
    return 0;
}

// on: the sharded SpMV of this operator exchanges its halo inside its own launch from now on (EVERY rank must make the same
// setting: krypy_amd/dist.py switches it on after all ranks have attached)
int kh_mat_xh_enable(kh_ctx ctx, kh_mat A, int on) {
    KH_ARG(ctx && A, "kh_mat_xh_enable: NULL");
    if (on) {
        KH_ARG(A->xh_box != nullptr, "kh_mat_xh_enable: kh_mat_xh_export / _attach first");
        KH_ARG(A->dia != nullptr, "kh_mat_xh_enable: the banded kernel carries the exchange; this shard has no diagonal-major copy");
        KH_ARG((A->nsend_prev == 0 || A->xh_prev != nullptr) && (A->nsend_next == 0 || A->xh_next != nullptr),
               "kh_mat_xh_enable: a neighbour this slab sends to is not attached");
        if (ctx->xr_err_pin == nullptr) {
            KH_HIP(hipHostMalloc(&ctx->xr_err_pin, sizeof(int), hipHostMallocMapped));
            *ctx->xr_err_pin = 0;
        }
    }
    A->xh_on = on ? 1 : 0;
    return 0;
}

// the neighbours' granules unmapped again (the own box stays until kh_mat_free: a neighbour may still have it mapped): what the
// host layer calls on EVERY rank when the collective decision about the in-launch halo comes out "off" after some ranks had
// attached already - so that a later attempt does not find the operator "already attached" 
int kh_mat_xh_detach(kh_ctx ctx, kh_mat A) {
    KH_ARG(ctx && A, "kh_mat_xh_detach: NULL");
    A->xh_on = 0;
    if (!A->xh_self) {
        if (A->xh_prev != nullptr) (void)hipIpcCloseMemHandle(A->xh_prev);
        if (A->xh_next != nullptr) (void)hipIpcCloseMemHandle(A->xh_next);
        (void)hipGetLastError();
    }
    A->xh_prev = A->xh_next = nullptr;
    A->xh_self = 0;
    return 0;
}

int kh_xr_export(kh_ctx ctx, unsigned char handle[64]) {
    KH_ARG(ctx && handle, "kh_xr_export: NULL");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    KH_HIP(hipSetDevice(ctx->device));
    if (ctx->xr_box == nullptr) {
        // fine-grained (uncached) device memory: a peer's system-scope stores over xGMI are visible to a kernel that is
        // already running here and polls with system-scope loads (what RCCL allocates for its own flags); plain hipMalloc
        // memory - coarse-grained - only as the last resort (good enough between two processes on ONE device)
        void* p = nullptr;
        const size_t bytes = sizeof(unsigned long long) * XR_BOX_WORDS;
        hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
        }
        KH_HIP(e);
        ctx->xr_box = static_cast<unsigned long long*>(p);
        KH_HIP(hipMemset(ctx->xr_box, 0, bytes));
        KH_HIP(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    KH_HIP(hipIpcGetMemHandle(&h, ctx->xr_box));
    memcpy(handle, &h, 64);
    return 0;
}

int kh_xr_attach(kh_ctx ctx, int rank, int nranks, const unsigned char* handles) {
    KH_ARG(ctx && handles, "kh_xr_attach: NULL");
    KH_ARG(nranks >= 1 && nranks <= XR_MAXRANKS && rank >= 0 && rank < nranks, "kh_xr_attach: rank %d of %d (at most %d ranks)",
           rank, nranks, XR_MAXRANKS);
    KH_ARG(ctx->xr_box != nullptr, "kh_xr_attach: kh_xr_export first");
    KH_ARG(ctx->xr_nranks == 0, "kh_xr_attach: already attached");
    KH_ARG(ctx->comm == nullptr || (ctx->rank == rank && ctx->nranks == nranks),
           "kh_xr_attach: rank %d of %d, but the RCCL communicator says %d of %d", rank, nranks, ctx->rank, ctx->nranks);
    KH_HIP(hipSetDevice(ctx->device));
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) {
            ctx->xr_peer[r] = ctx->xr_box;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)64 * r, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; ++q) {
                if (q != rank && ctx->xr_peer[q] != nullptr) (void)hipIpcCloseMemHandle(ctx->xr_peer[q]);
                ctx->xr_peer[q] = nullptr;
            }
            return fail(KH_ERR_COMM, "kh_xr_attach: cannot map rank %d's mailbox (%s)", r, hipGetErrorString(e));
        }
        ctx->xr_peer[r] = static_cast<unsigned long long*>(p);
    }
    if (ctx->xr_err_pin == nullptr) {
        KH_HIP(hipHostMalloc(&ctx->xr_err_pin, sizeof(int), hipHostMallocMapped));
        *ctx->xr_err_pin = 0;
    }
    ctx->xr_rank = rank;
    ctx->xr_nranks = nranks;
    return 0;
}

int kh_xr_detach(kh_ctx ctx) {
    if (!ctx) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    xr_free(ctx);
    return 0;
}

}  // extern "C"
