// Multi-GPU plumbing of libkrylov_hip: one process per GPU, RCCL over xGMI.
//
// The Krylov path shards block-row-wise: vectors and matrix rows are split
// into contiguous slabs, H/R/Givens stay replicated on every rank's host.  Two exchanges exist:
//   * all-reduce(sum, fp64) of tiny dot-product panels (1 .. k+1 values): latency bound, so
//     the panel Gram-Schmidt mode (one all-reduce per sweep) is the default when nranks > 1;
//   * nearest-neighbour halo exchange for the stencil SpMV (ncclSend/ncclRecv in one group:
//     each neighbour pair has its own point-to-point xGMI link).
// librccl is resolved with dlopen at kh_comm_init so that single-GPU processes never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include "kh_internal.h"

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void* lib = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return kh::fail(KH_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define KH_SYM(field, name)                                                            \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));         \
    if (!g_rccl.field) return kh::fail(KH_ERR_COMM, "librccl lacks symbol %s", name)
    KH_SYM(GetUniqueId, "ncclGetUniqueId");
    KH_SYM(CommInitRank, "ncclCommInitRank");
    KH_SYM(CommDestroy, "ncclCommDestroy");
    KH_SYM(AllReduce, "ncclAllReduce");
    KH_SYM(Send, "ncclSend");
    KH_SYM(Recv, "ncclRecv");
    KH_SYM(GroupStart, "ncclGroupStart");
    KH_SYM(GroupEnd, "ncclGroupEnd");
    KH_SYM(GetErrorString, "ncclGetErrorString");
#undef KH_SYM
    g_rccl.lib = lib;
    return 0;
}

#define KH_NCCL(call)                                                                     \
    do {                                                                                  \
        ncclResult_t r_ = (call);                                                         \
        if (r_ != ncclSuccess)                                                            \
            return kh::fail(KH_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

namespace kh {

int comm_allreduce_dev(kh_ctx ctx, double* dev, int64_t count) {
    if (count == 0) return 0;
    if (ctx->xr_on) {                                   // mailboxes over IPC / xGMI (xr.hip): no library call
        ctx->n_allreduce += 1;
        return xr_allreduce_dev(ctx, dev, count);
    }
    if (ctx->comm == nullptr) return 0;                 // a 1-rank communicator still goes through RCCL
    ctx->n_allreduce += 1;
    KH_NCCL(g_rccl.AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm,
                             ctx->stream));
    return 0;
}

int comm_halo_exchange(kh_ctx ctx, kh_mat A, const double* x, hipStream_t stream, int width) {
    // width: doubles per vector entry (1 real, 2 complex); counts and offsets below are in entries
    if (ctx->comm == nullptr) {
        // ranks joined through the xr transport alone: sums cross them, and a halo only inside the banded kernel's launch
        // (kh_mat_xh_*) - which this call is not
        if (ctx->nranks > 1)
            return fail(KH_ERR_COMM, "sharded SpMV: %d ranks without an RCCL communicator and this operator's halo is not exchanged "
                                     "inside its launch (kh_mat_xh_enable)", ctx->nranks);
        return 0;
    }
    const int64_t nloc = A->n_rows;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (ctx->halo_loopback && ctx->nranks == 1) {
        // test mode (kh_ctx_set "halo_loopback"): the one rank is its own previous and next neighbour, i.e. the slab
        // of an operator that is periodic across the slab boundary.  The same grouped ncclSend / ncclRecv calls as
        // between real neighbours, on the same stream with the same event ordering - on one GPU.  Point-to-point
        // operations between the same pair match in issue order: what goes "to the previous rank" (my first rows)
        // is what my next neighbour's ghost region receives, and the other way round, so the receives are posted
        // next-region first.
        KH_ARG(A->nsend_prev == A->nrecv_next && A->nsend_next == A->nrecv_prev,
               "halo_loopback: a periodic slab sends what it receives (send %lld/%lld, receive %lld/%lld)",
               (long long)A->nsend_prev, (long long)A->nsend_next, (long long)A->nrecv_prev, (long long)A->nrecv_next);
        KH_NCCL(g_rccl.GroupStart());
        if (A->nsend_prev) KH_NCCL(g_rccl.Send(x, width * A->nsend_prev, ncclDouble, 0, comm, stream));
        if (A->nsend_next)
            KH_NCCL(g_rccl.Send(x + width * (nloc - A->nsend_next), width * A->nsend_next, ncclDouble, 0, comm, stream));
        if (A->nrecv_next)
            KH_NCCL(g_rccl.Recv(A->ghost + width * A->nrecv_prev, width * A->nrecv_next, ncclDouble, 0, comm, stream));
        if (A->nrecv_prev) KH_NCCL(g_rccl.Recv(A->ghost, width * A->nrecv_prev, ncclDouble, 0, comm, stream));
        KH_NCCL(g_rccl.GroupEnd());
        ctx->n_halo_exchange += 1;
        return 0;
    }
    KH_NCCL(g_rccl.GroupStart());
    if (ctx->rank > 0) {
        if (A->nsend_prev) KH_NCCL(g_rccl.Send(x, width * A->nsend_prev, ncclDouble, ctx->rank - 1, comm, stream));
        if (A->nrecv_prev) KH_NCCL(g_rccl.Recv(A->ghost, width * A->nrecv_prev, ncclDouble, ctx->rank - 1, comm, stream));
    }
    if (ctx->rank + 1 < ctx->nranks) {
        if (A->nsend_next)
            KH_NCCL(g_rccl.Send(x + width * (nloc - A->nsend_next), width * A->nsend_next, ncclDouble, ctx->rank + 1, comm, stream));
        if (A->nrecv_next)
            KH_NCCL(g_rccl.Recv(A->ghost + width * A->nrecv_prev, width * A->nrecv_next, ncclDouble, ctx->rank + 1, comm, stream));
    }
    KH_NCCL(g_rccl.GroupEnd());
    if (ctx->nranks > 1) ctx->n_halo_exchange += 1;
    return 0;
}

// the halo of a BLOCK of vectors in one grouped exchange: column c's ghost entries land in
// A->ghost_panel + c * (nrecv_prev + nrecv_next) (the caller sized it); 2 sends + 2 receives per column, all in one
// group - RCCL fuses them into one kernel
int comm_halo_exchange_panel(kh_ctx ctx, kh_mat A, const double* X, int64_t ldx, int64_t ncols, hipStream_t stream) {
    if (ctx->comm == nullptr) return 0;
    const int64_t nloc = A->n_rows, ng = A->nrecv_prev + A->nrecv_next;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const bool loop = ctx->halo_loopback && ctx->nranks == 1;
    if (loop)
        KH_ARG(A->nsend_prev == A->nrecv_next && A->nsend_next == A->nrecv_prev, "halo_loopback: a periodic slab sends what it receives");
    if (!loop && ctx->nranks == 1) return 0;
    KH_NCCL(g_rccl.GroupStart());
    for (int64_t c = 0; c < ncols; ++c) {
        const double* x = X + c * ldx;
        double* g = A->ghost_panel + c * ng;
        if (loop) {         // (see comm_halo_exchange: receives posted next-region first)
            if (A->nsend_prev) KH_NCCL(g_rccl.Send(x, A->nsend_prev, ncclDouble, 0, comm, stream));
            if (A->nsend_next) KH_NCCL(g_rccl.Send(x + (nloc - A->nsend_next), A->nsend_next, ncclDouble, 0, comm, stream));
            if (A->nrecv_next) KH_NCCL(g_rccl.Recv(g + A->nrecv_prev, A->nrecv_next, ncclDouble, 0, comm, stream));
            if (A->nrecv_prev) KH_NCCL(g_rccl.Recv(g, A->nrecv_prev, ncclDouble, 0, comm, stream));
            continue;
        }
        if (ctx->rank > 0) {
            if (A->nsend_prev) KH_NCCL(g_rccl.Send(x, A->nsend_prev, ncclDouble, ctx->rank - 1, comm, stream));
            if (A->nrecv_prev) KH_NCCL(g_rccl.Recv(g, A->nrecv_prev, ncclDouble, ctx->rank - 1, comm, stream));
        }
        if (ctx->rank + 1 < ctx->nranks) {
            if (A->nsend_next) KH_NCCL(g_rccl.Send(x + (nloc - A->nsend_next), A->nsend_next, ncclDouble, ctx->rank + 1, comm, stream));
            if (A->nrecv_next) KH_NCCL(g_rccl.Recv(g + A->nrecv_prev, A->nrecv_next, ncclDouble, ctx->rank + 1, comm, stream));
        }
    }
    KH_NCCL(g_rccl.GroupEnd());
    ctx->n_halo_exchange += 1;
    return 0;
}

}  // namespace kh

extern "C" {

int kh_comm_unique_id(unsigned char id[128]) {
    KH_ARG(id != nullptr, "kh_comm_unique_id: NULL");
    KH_TRY(load_rccl());
    ncclUniqueId uid;
    KH_NCCL(g_rccl.GetUniqueId(&uid));
    static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &uid, 128);
    return 0;
}

int kh_comm_init(kh_ctx ctx, int rank, int nranks, const unsigned char id[128]) {
    KH_ARG(ctx && id, "kh_comm_init: NULL");
    KH_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "kh_comm_init: rank %d of %d", rank, nranks);
    KH_ARG(ctx->comm == nullptr, "kh_comm_init: communicator already initialised");
    KH_TRY(load_rccl());
    KH_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclComm_t comm = nullptr;
    KH_NCCL(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    {
        const char* e = getenv("KRYPY_AMD_FORCE_MULTI");
        ctx->force_multi = (e != nullptr && atoi(e) != 0) ? 1 : 0;
    }
    KH_HIP(hipMalloc(&ctx->commbuf, sizeof(double) * kh::SCAL_CAP));
    KH_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    KH_HIP(hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
    KH_HIP(hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
    {
        const char* e = getenv("KRYPY_AMD_SPMV_SPLIT");
        ctx->spmv_split = (e == nullptr) ? 1 : atoi(e);
    }
    return 0;
}

int kh_comm_destroy(kh_ctx ctx) {
    if (!ctx) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    kh::xr_free(ctx);
    if (!ctx->comm) {
        if (ctx->commbuf) (void)hipFree(ctx->commbuf);
        ctx->commbuf = nullptr;
        return 0;
    }
    g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->force_multi = 0;
    ctx->nranks = 1;
    ctx->rank = 0;
    (void)hipFree(ctx->commbuf);
    ctx->commbuf = nullptr;
    if (ctx->comm_stream) {
        (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamDestroy(ctx->comm_stream);
        (void)hipEventDestroy(ctx->ev_x);
        (void)hipEventDestroy(ctx->ev_halo);
        ctx->comm_stream = nullptr;
        ctx->ev_x = ctx->ev_halo = nullptr;
    }
    return 0;
}

int kh_comm_allreduce_host(kh_ctx ctx, double* vals, int64_t count) {
    KH_ARG(ctx && (vals || count == 0), "kh_comm_allreduce_host: NULL");
    if ((ctx->comm == nullptr && !ctx->xr_on) || count == 0) return 0;
    KH_ARG(count <= kh::SCAL_CAP, "kh_comm_allreduce_host: at most %d values", kh::SCAL_CAP);
    if (ctx->commbuf == nullptr) KH_HIP(hipMalloc(&ctx->commbuf, sizeof(double) * kh::SCAL_CAP));      // (xr without a communicator)
    KH_HIP(hipMemcpyAsync(ctx->commbuf, vals, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    KH_TRY(kh::comm_allreduce_dev(ctx, ctx->commbuf, count));
    KH_HIP(hipMemcpyAsync(vals, ctx->commbuf, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    KH_HIP(hipStreamSynchronize(ctx->stream));
    return kh::xr_check(ctx);
}

int kh_mat_set_halo(kh_ctx ctx, kh_mat A, int64_t nsend_prev, int64_t nsend_next, int64_t nrecv_prev,
                    int64_t nrecv_next) {
    KH_ARG(ctx && A, "kh_mat_set_halo: NULL");
    KH_ARG(A->kind == KH_MAT_CSR || A->kind == KH_MAT_ZCSR, "kh_mat_set_halo: CSR operators only");
    const int width = A->kind == KH_MAT_ZCSR ? 2 : 1;
    KH_ARG(nsend_prev >= 0 && nsend_next >= 0 && nrecv_prev >= 0 && nrecv_next >= 0, "negative halo");
    KH_ARG(nsend_prev <= A->n_rows && nsend_next <= A->n_rows, "halo wider than the local slab");
    KH_ARG(A->n_cols == A->n_rows + nrecv_prev + nrecv_next,
           "kh_mat_set_halo: n_cols %lld != n_rows %lld + ghosts %lld", (long long)A->n_cols,
           (long long)A->n_rows, (long long)(nrecv_prev + nrecv_next));
    A->nsend_prev = nsend_prev;
    A->nsend_next = nsend_next;
    A->nrecv_prev = nrecv_prev;
    A->nrecv_next = nrecv_next;
    (void)hipFree(A->ghost);
    A->ghost = nullptr;
    kh::xh_free(A);                 // (granules sized for another halo)
    const int64_t ng = nrecv_prev + nrecv_next;
    if (ng > 0) {
        KH_HIP(hipMalloc(&A->ghost, sizeof(double) * width * ng));
        // (on the context's stream: a memset on the null stream is not ordered with the non-blocking streams the
        // halo exchange and kh_mat_set_ghost write this buffer on)
        KH_HIP(hipMemsetAsync(A->ghost, 0, sizeof(double) * width * ng, ctx->stream));
    }
    if (A->kind == KH_MAT_ZCSR) return 0;       // (no banded copy / split launches for complex operators)
    return kh::dia_rebuild_for_halo(ctx, A);
}

}  // extern "C"
