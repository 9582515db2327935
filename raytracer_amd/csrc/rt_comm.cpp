// Multi-GPU collectives of the C ABI: one process per GPU, RCCL over xGMI, called directly (no
// framework in between).  The reference renders on one device and has no counterpart.
//
// librccl is opened at the first rt_comm_* call (dlopen), so the render library itself loads on
// machines without it and single-GPU users never pay for it.  All collectives are queued on the
// caller's context stream after joining its trace streams, i.e. they see every frame rendered so
// far, and they are followed by a host wait: on return the result is in place.

#include "../../include/rt_mi355x.h"
#include "rt_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;

int fail(const std::string &m) {
    rt::set_last_error(m);
    return 1;
}

// 0 on success; the library stays open for the life of the process
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_rccl.handle) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(std::string("rt_comm: cannot open librccl: ") + dlerror());
    Rccl r;
    r.handle = h;
#define SYM(field, name)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));              \
    if (!r.field) return fail(std::string("rt_comm: librccl lacks ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Reduce, "ncclReduce");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
#undef SYM
    g_rccl = r;
    return 0;
}

#define NCCL_TRY(expr)                                                                           \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) return fail(std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

}  // namespace

namespace rt {

void comm_destroy(void *comm) {
    if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
}

}  // namespace rt

extern "C" {

int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]) {
    if (!id) return fail("rt_comm_unique_id: null argument");
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return 0;
}

int rt_comm_init(rt_ctx *ctx, int rank, int world, const char id[RT_COMM_ID_BYTES]) {
    if (!ctx || !id) return fail("rt_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail("rt_comm_init: rank must be in [0, world)");
    if (rt::ctx_comm(ctx)) return fail("rt_comm_init: the context already has a communicator");
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(rt::ctx_device(ctx)));
    ncclUniqueId u;
    memcpy(u.internal, id, RT_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, world, u, rank));
    rt::ctx_set_comm(ctx, comm, rank, world);
    return 0;
}

int rt_comm_reduce_sum(rt_ctx *ctx, float *sum_rgba, int root) {
    if (!ctx) return fail("rt_comm_reduce_sum: null context");
    ncclComm_t comm = static_cast<ncclComm_t>(rt::ctx_comm(ctx));
    if (!comm) return fail("rt_comm_reduce_sum: rt_comm_init has not been called on this context");
    HIP_TRY(hipSetDevice(rt::ctx_device(ctx)));
    hipStream_t s = nullptr;
    float *own = nullptr;
    size_t nfloats = 0;
    if (int rc = rt::ctx_join_for_collective(ctx, &s, &own, &nfloats)) return rc;
    float *buf = sum_rgba ? sum_rgba : own;
    NCCL_TRY(g_rccl.Reduce(buf, buf, nfloats, ncclFloat32, ncclSum, root, comm, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int rt_comm_all_reduce_f64(rt_ctx *ctx, double *values, int count, int op) {
    if (!ctx || !values) return fail("rt_comm_all_reduce_f64: null argument");
    if (count < 1 || count > RT_COMM_MAX_SCALARS) return fail("rt_comm_all_reduce_f64: count out of range");
    if (op != RT_COMM_SUM && op != RT_COMM_MAX) return fail("rt_comm_all_reduce_f64: unknown op");
    ncclComm_t comm = static_cast<ncclComm_t>(rt::ctx_comm(ctx));
    if (!comm) return fail("rt_comm_all_reduce_f64: rt_comm_init has not been called on this context");
    HIP_TRY(hipSetDevice(rt::ctx_device(ctx)));
    hipStream_t s = nullptr;
    float *own = nullptr;
    size_t nfloats = 0;
    if (int rc = rt::ctx_join_for_collective(ctx, &s, &own, &nfloats)) return rc;
    double *dev = rt::ctx_comm_scratch(ctx);
    HIP_TRY(hipMemcpyAsync(dev, values, sizeof(double) * count, hipMemcpyHostToDevice, s));
    NCCL_TRY(g_rccl.AllReduce(dev, dev, (size_t)count, ncclFloat64, op == RT_COMM_SUM ? ncclSum : ncclMax, comm, s));
    HIP_TRY(hipMemcpyAsync(values, dev, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int rt_comm_info(rt_ctx *ctx, int *rank, int *nranks) {
    if (!ctx || !rank || !nranks) return fail("rt_comm_info: null argument");
    ncclComm_t comm = static_cast<ncclComm_t>(rt::ctx_comm(ctx));
    if (!comm) return fail("rt_comm_info: rt_comm_init has not been called on this context");
    NCCL_TRY(g_rccl.CommUserRank(comm, rank));
    NCCL_TRY(g_rccl.CommCount(comm, nranks));
    return 0;
}

int rt_comm_barrier(rt_ctx *ctx) {
    double one = 1.0;
    return rt_comm_all_reduce_f64(ctx, &one, 1, RT_COMM_SUM);
}

}  // extern "C"
