// csr_host_comm.inl -- part of csr_lib.hip (one translation unit; included in this order): the ONE collective of the path
// (SURVEY.md 8(e)): the final gather of the per-bin output tracks over RCCL / xGMI, one communicator per rank (process).
//
// The reference has nothing to mirror here (chromosomes are a sequential loop in one process, consenrich.py:8809).  RCCL is
// bound at run time (dlopen of librccl.so.1): the library loads, and every single-GPU entry point works, on a machine
// without RCCL; only the csr_comm_* entry points need it.  The caller hands over the 128-byte unique id
// (created by rank 0 with csr_comm_unique_id and distributed by the caller's launcher -- bench.py uses a file on the node).

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only; every function is resolved with dlsym

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rcclMutex;

static int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rcclMutex);
    if (g_rccl.handle) return 0;
    const char *cands[] = {getenv("CONSENRICH_AMD_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *name : cands) {
        if (!name || !*name) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail("RCCL not found (dlopen librccl.so.1 failed: %s): multi-GPU gather unavailable", dlerror());
    RcclApi a;
    a.handle = h;
#define CSR_RCCL_SYM(field, sym)                                                   \
    do {                                                                           \
        a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym));              \
        if (!a.field) { dlclose(h); return fail("RCCL symbol %s not found", sym); } \
    } while (0)
    CSR_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    CSR_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    CSR_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    CSR_RCCL_SYM(AllGather, "ncclAllGather");
    CSR_RCCL_SYM(AllReduce, "ncclAllReduce");
    CSR_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef CSR_RCCL_SYM
    g_rccl = a;
    return 0;
}
#define NCCLOK(expr)                                                                                          \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct csr_comm {
    csr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    double *dScalar = nullptr;      // 2 doubles of device scratch for the scalar collectives
    DevBuf send, recv;              // gather buffers (grown on demand, freed with the communicator)
    int64_t *dPackPos = nullptr;    // packed start of every chain of the batch the buffers were sized for
    int packChains = 0;
};

extern "C" int csr_comm_unique_id(char *id128) {
    if (!id128) return fail("null argument");
    CHECK(rccl_load());
    ncclUniqueId id;
    NCCLOK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, id.internal, 128);
    return 0;
}

extern "C" csr_comm *csr_comm_create(csr_ctx *c, const char *id128, int32_t world, int32_t rank) {
    if (!c || !id128) { fail("null argument"); return nullptr; }
    if (world <= 0 || rank < 0 || rank >= world) { fail("bad world size / rank (%d / %d)", world, rank); return nullptr; }
    if (rccl_load() != 0 || ctx_select(c) != 0) return nullptr;
    csr_comm *k = new csr_comm();
    k->ctx = c;
    k->world = world;
    k->rank = rank;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&k->comm, world, id, rank);
    if (r != ncclSuccess) {
        fail("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, g_rccl.GetErrorString(r));
        delete k;
        return nullptr;
    }
    if (hipMalloc((void **)&k->dScalar, 2 * sizeof(double)) != hipSuccess) {
        fail("hipMalloc failed for the communicator scratch");
        (void)g_rccl.CommDestroy(k->comm);
        delete k;
        return nullptr;
    }
    return k;
}

extern "C" void csr_comm_destroy(csr_comm *k) {
    if (!k) return;
    if (k->ctx) {
        (void)hipSetDevice(k->ctx->device);
        (void)hipStreamSynchronize(k->ctx->stream);
    }
    if (k->comm) (void)g_rccl.CommDestroy(k->comm);
    if (k->dScalar) (void)hipFree(k->dScalar);
    if (k->dPackPos) (void)hipFree(k->dPackPos);
    for (DevBuf *b : {&k->send, &k->recv})
        if (b->ptr) (void)hipFree(b->ptr);
    delete k;
}

extern "C" int csr_comm_world(csr_comm *k) { return k ? k->world : 0; }
extern "C" int csr_comm_rank(csr_comm *k) { return k ? k->rank : -1; }

// max over ranks of *value (in place); also the barrier of the timed region: an all-reduce of one double on the library's
// stream followed by a stream synchronisation completes only when every rank has entered it with its queue drained.
static int comm_allreduce(csr_comm *k, double *value, ncclRedOp_t op) {
    if (!k || !value) return fail("null argument");
    csr_ctx *c = k->ctx;
    CHECK(ctx_select(c));
    if (c->configured) CHECK(settle(c));
    HIPOK(hipMemcpyAsync(k->dScalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLOK(g_rccl.AllReduce(k->dScalar, k->dScalar + 1, 1, ncclDouble, op, k->comm, c->stream));
    HIPOK(hipMemcpyAsync(value, k->dScalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int csr_comm_allreduce_max(csr_comm *k, double *value) { return comm_allreduce(k, value, ncclMax); }
// sum over ranks of *value (in place): an all-reduce of 1.0 tells a caller how many ranks the communicator really spans
extern "C" int csr_comm_allreduce_sum(csr_comm *k, double *value) { return comm_allreduce(k, value, ncclSum); }
extern "C" int csr_comm_barrier(csr_comm *k) {
    double v = 0.0;
    return csr_comm_allreduce_max(k, &v);
}

// (smoothed level, its variance P00) of every bin of the rank's chains, packed chain after chain without padding:
// out[2 * (pos[chain] + k)] = xs[off + k][0], out[.. + 1] = Ps[off + k][0][0]
__global__ __launch_bounds__(256) void k_pack_tracks(const float *xs, int xsStride, const float *ps, int psStride,
                                                    const int64_t *chainOff, const int64_t *chainLen, const int64_t *packPos,
                                                    int nchains, int64_t Npad, float2 *out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= Npad) return;
    int lo = 0, hi = nchains - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chainOff[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const int64_t k = g - chainOff[lo];
    if (k >= chainLen[lo]) return;
    out[packPos[lo] + k] = make_float2(xs[g * xsStride], ps[g * psStride]);
}

// Final track gather.  The smoothed state / covariance of the batch's last pass must have been exported
// (CSR_EXPORT_SMOOTH).  cap_bins = max over ranks of the rank's total bins (every rank passes the same value): each rank
// contributes a buffer of cap_bins (level, variance) pairs, packed on the device straight from the exported arrays, and
// ncclAllGather leaves world x cap_bins pairs on every rank.  host_out (may be NULL): world * cap_bins * 2 floats.
extern "C" int csr_batch_gather_tracks(csr_ctx *c, csr_comm *k, int64_t cap_bins, float *host_out) {
    CHECK(need(c));
    if (!k || k->ctx != c) return fail("communicator does not belong to this context");
    CHECK(settle(c));
    if (!c->nat[CSR_ARR_XS] || !c->nat[CSR_ARR_PS]) return fail("smoothed tracks were not exported (CSR_EXPORT_SMOOTH)");
    // the natural arrays must hold the RESIDENT fit: a later ecm / forward_backward / background_apply without an export
    // would otherwise be gathered as stale tracks
    if (!c->haveBwd || c->natSmoothGen != c->fitGen)
        return fail("the exported smoothed tracks are not those of the resident fit: export (CSR_EXPORT_SMOOTH) after the last pass");
    const int nc = (int)c->chains.size();
    int64_t mine = 0;
    std::vector<int64_t> pos(nc);
    for (int i = 0; i < nc; ++i) { pos[i] = mine; mine += c->chains[i].n; }
    if (cap_bins < mine) return fail("cap_bins (%lld) is smaller than this rank's %lld bins", (long long)cap_bins, (long long)mine);
    const size_t sendBytes = sizeof(float2) * (size_t)cap_bins;
    CHECK(k->send.reserve(sendBytes));
    CHECK(k->recv.reserve(sendBytes * (size_t)k->world));
    if (k->packChains < nc) {
        if (k->dPackPos) (void)hipFree(k->dPackPos);
        k->dPackPos = nullptr;
        HIPOK(hipMalloc((void **)&k->dPackPos, sizeof(int64_t) * (size_t)nc));
        k->packChains = nc;
    }
    HIPOK(hipMemcpyAsync(k->dPackPos, pos.data(), sizeof(int64_t) * (size_t)nc, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemsetAsync(k->send.ptr, 0, sendBytes, c->stream));
    const int d = c->mdl.state_dim;
    {
        Scope sc(c, "gather_pack");
        hipLaunchKernelGGL(k_pack_tracks, dim3((unsigned)((c->Npad + 255) / 256)), dim3(256), 0, c->stream, c->nat[CSR_ARR_XS], d,
                           c->nat[CSR_ARR_PS], d * d, c->dChainOff, c->dChainLen, k->dPackPos, nc, c->Npad,
                           reinterpret_cast<float2 *>(k->send.ptr));
    }
    LAUNCH_CHECK("k_pack_tracks");
    {
        Scope sc(c, "gather_allgather");
        NCCLOK(g_rccl.AllGather(k->send.ptr, k->recv.ptr, (size_t)cap_bins * 2, ncclFloat, k->comm, c->stream));
    }
    if (host_out)
        HIPOK(hipMemcpyAsync(host_out, k->recv.ptr, sendBytes * (size_t)k->world, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}
