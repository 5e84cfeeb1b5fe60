// In-library RCCL binding of libepx.so: the ONE reduction of an EP iteration across the GPUs of
// a node -- Q = sum_k Qi2 + Q0, r = sum_k ri2 + r0 over ALL sites (/root/reference/epstan/
// method.py:1073-1074) and the logical AND of the cavity flags (:1145) -- runs as
// ncclAllReduce on the context's own stream, in stream order with the kernels that produce and
// consume the packed sums: no host round trip, no second stream, no PyTorch.
//
// librccl is bound at run time (dlopen): a process that never calls epx_comm_init does not
// need it, and when another component of the process (e.g. torch) has already loaded an RCCL,
// that one is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "epx_ctx.h"

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL); if (h) break; }   // already in the process?
    for (size_t i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail("cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                               \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));           \
    if (!g_rccl.field) return fail("librccl has no symbol %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.h = h;
    return 0;
}

#define NCCLCHK(x)                                                                                      \
    do {                                                                                                \
        ncclResult_t r_ = (x);                                                                          \
        if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #x, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

int stage(epx_ctx *c, size_t n) {
    if (c->comm_stage_n >= n) return 0;
    if (c->comm_stage) (void)hipFree(c->comm_stage);
    c->comm_stage = nullptr; c->comm_stage_n = 0;
    HIPCHK(dalloc(&c->comm_stage, n));
    c->comm_stage_n = n;
    return 0;
}

}  // namespace

int epx_comm_unique_id(void *id_out) {
    if (!id_out) return fail("null id buffer");
    if (load_rccl()) return -1;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof id == EPX_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int epx_comm_init(epx_ctx *c, const void *id, int rank, int nranks) {
    CTX(c);
    if (c->comm || c->comm_ext) return fail("the context already has a communicator");
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail("bad communicator arguments (rank %d of %d)", rank, nranks);
    if (load_rccl()) return -1;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm;
    NCCLCHK(g_rccl.CommInitRank(&comm, nranks, uid, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_size = nranks;
    return 0;
}

int epx_comm_init_host(epx_ctx *c, int rank, int nranks, epx_host_allreduce_fn fn, void *user) {
    CTX(c);
    if (c->comm || c->comm_ext) return fail("the context already has a communicator");
    if (!fn || nranks < 1 || rank < 0 || rank >= nranks) return fail("bad communicator arguments (rank %d of %d)", rank, nranks);
    c->comm_ext = fn; c->comm_ext_user = user; c->comm_rank = rank; c->comm_size = nranks;
    return 0;
}

// all-reduce through the caller's host transport: device -> pinned-less host copy, callback, back; in stream order
static int host_allreduce_dev(epx_ctx *c, double *buf, size_t n, int op) {
    std::vector<double> h(n);
    HIPCHK(hipMemcpyAsync(h.data(), buf, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->comm_ext(h.data(), (long long)n, op, c->comm_ext_user)) return fail("the host transport of the communicator failed");
    HIPCHK(hipMemcpyAsync(buf, h.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));        // `h` goes away
    return 0;
}

int epx_comm_destroy(epx_ctx *c) {
    if (!c) return fail("null context");
    if (c->comm_ext) { c->comm_ext = nullptr; c->comm_ext_user = nullptr; c->comm_size = 0; c->comm_rank = 0; return 0; }
    if (!c->comm) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    ncclResult_t r = g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr; c->comm_size = 0; c->comm_rank = 0;
    if (r != ncclSuccess) return fail("ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return 0;
}

int epx_comm_size(epx_ctx *c, int *rank, int *nranks) {
    if (!c) return fail("null context");
    int n = 1;
    if (c->comm) NCCLCHK(g_rccl.CommCount(static_cast<ncclComm_t>(c->comm), &n));
    if (c->comm_ext) n = c->comm_size;
    if (rank) *rank = (c->comm || c->comm_ext) ? c->comm_rank : 0;
    if (nranks) *nranks = n;
    return 0;
}

// used by epx_api.hip: all-reduce of a device buffer in stream order (no synchronisation)
int epx_comm_allreduce_dev(epx_ctx *c, double *buf, size_t n, int op) {
    if (c->comm_ext) return host_allreduce_dev(c, buf, n, op);
    if (!c->comm) return 0;
    const ncclRedOp_t ops[] = {ncclSum, ncclMin, ncclMax};
    NCCLCHK(g_rccl.AllReduce(buf, buf, n, ncclDouble, ops[op], static_cast<ncclComm_t>(c->comm), c->stream));
    return 0;
}

int epx_comm_allreduce(epx_ctx *c, double *buf, int n, int op) {
    CTX(c);
    if (!buf || n < 1) return fail("bad buffer");
    if (op < EPX_OP_SUM || op > EPX_OP_MAX) return fail("unknown reduction %d", op);
    if (c->comm_ext) {
        if (c->comm_ext(buf, n, op, c->comm_ext_user)) return fail("the host transport of the communicator failed");
        return 0;
    }
    if (!c->comm) return 0;                         // one rank: identity
    if (stage(c, (size_t)n)) return -1;
    const ncclRedOp_t ops[] = {ncclSum, ncclMin, ncclMax};
    HIPCHK(hipMemcpyAsync(c->comm_stage, buf, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(g_rccl.AllReduce(c->comm_stage, c->comm_stage, (size_t)n, ncclDouble, ops[op],
                             static_cast<ncclComm_t>(c->comm), c->stream));
    HIPCHK(hipMemcpyAsync(buf, c->comm_stage, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int epx_comm_allgather(epx_ctx *c, const double *in, int n, double *out) {
    CTX(c);
    if (!in || !out || n < 1) return fail("bad buffer");
    if (c->comm_ext) {
        // every rank's block in its own slot, zeros elsewhere: the sum over the ranks is the gather
        memset(out, 0, (size_t)n * c->comm_size * 8);
        memcpy(out + (size_t)n * c->comm_rank, in, (size_t)n * 8);
        if (c->comm_ext(out, (long long)n * c->comm_size, EPX_OP_SUM, c->comm_ext_user)) return fail("the host transport of the communicator failed");
        return 0;
    }
    if (!c->comm) { memcpy(out, in, (size_t)n * 8); return 0; }
    const size_t tot = (size_t)n * (1 + c->comm_size);
    if (stage(c, tot)) return -1;
    HIPCHK(hipMemcpyAsync(c->comm_stage, in, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(g_rccl.AllGather(c->comm_stage, c->comm_stage + n, (size_t)n, ncclDouble,
                             static_cast<ncclComm_t>(c->comm), c->stream));
    HIPCHK(hipMemcpyAsync(out, c->comm_stage + n, (size_t)n * c->comm_size * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
