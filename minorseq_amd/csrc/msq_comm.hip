// msq_comm.hip -- the one exchange step of the path when reads are sharded over the GPUs of a node: RCCL
// collectives over xGMI on the context's stream (nothing in the reference corresponds -- it is a
// single process).  One process per GPU, one communicator per context.
//
//   step 0   window bounds: element-wise MAX of the device-resident {-begin0, end0} pair  (2 x int32)
//   step 2   count matrices: element-wise SUM of the u32 [W][70] buffer                    (0.8 .. 5.6 MB)
//
// Both are integer and order-free, so every rank ends up with bit-identical matrices and the testing pass can run
// replicated.  When the caller knows the window beforehand (a target region; the bench after set-up) the two are
// issued as ONE RCCL group -- one launch -- behind the pileup (msq_comm_allreduce_step).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU user of the library needs no RCCL at all, and a
// process that already carries an RCCL (PyTorch's) shares it instead of loading a second copy.
#include "msq_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return MSQ_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return msq_fail(MSQ_ERR_NODEVICE, "msq_comm: cannot load librccl.so.1 (%s)", dlerror());
    RcclApi api;
    api.handle = h;
#define MSQ_SYM(field, name)                                                                        \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));                              \
    if (!api.field) return msq_fail(MSQ_ERR_NODEVICE, "msq_comm: librccl lacks %s", name)
    MSQ_SYM(GetUniqueId, "ncclGetUniqueId");
    MSQ_SYM(CommInitRank, "ncclCommInitRank");
    MSQ_SYM(CommDestroy, "ncclCommDestroy");
    MSQ_SYM(AllReduce, "ncclAllReduce");
    MSQ_SYM(AllGather, "ncclAllGather");
    MSQ_SYM(GroupStart, "ncclGroupStart");
    MSQ_SYM(GroupEnd, "ncclGroupEnd");
    MSQ_SYM(GetErrorString, "ncclGetErrorString");
#undef MSQ_SYM
    g_rccl = api;
    return MSQ_OK;
}

#define MSQ_RCCL(call)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess) return msq_fail(MSQ_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));     \
    } while (0)

}  // namespace

struct msq_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static_assert(MSQ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "msq_comm_unique_id hands out an ncclUniqueId");

// Can this process reach RCCL at all?  Local and non-collective: ranks agree on the answer over their control plane
// BEFORE any of them enters msq_comm_create (ncclCommInitRank blocks until every rank has arrived).
extern "C" int msq_comm_available(void) { return load_rccl(); }

extern "C" int msq_comm_unique_id(uint8_t* id_out)
{
    if (!id_out) return msq_fail(MSQ_ERR_ARG, "msq_comm_unique_id: id_out is NULL");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    MSQ_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, MSQ_COMM_ID_BYTES);
    return MSQ_OK;
}

extern "C" int msq_comm_create(msq_ctx* ctx, int rank, int world, const uint8_t* id)
{
    if (!ctx || !id) return msq_fail(MSQ_ERR_ARG, "msq_comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return msq_fail(MSQ_ERR_ARG, "msq_comm_create: rank %d of %d", rank, world);
    if (ctx->comm) return msq_fail(MSQ_ERR_ARG, "msq_comm_create: the context already has a communicator");
    if (int rc = load_rccl()) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, MSQ_COMM_ID_BYTES);
    msq_comm* c = new msq_comm;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return msq_fail(MSQ_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    ctx->comm = c;
    return MSQ_OK;
}

extern "C" int msq_comm_destroy(msq_ctx* ctx)
{
    if (!ctx || !ctx->comm) return MSQ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm->comm) (void)g_rccl.CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
    return MSQ_OK;
}

extern "C" int msq_comm_active(const msq_ctx* ctx) { return ctx && ctx->comm ? 1 : 0; }
extern "C" int msq_comm_rank(const msq_ctx* ctx) { return ctx && ctx->comm ? ctx->comm->rank : 0; }
extern "C" int msq_comm_world(const msq_ctx* ctx) { return ctx && ctx->comm ? ctx->comm->world : 1; }

// Contiguous read range of a rank: [rank * N / world, (rank + 1) * N / world)  
extern "C" int msq_comm_shard(uint64_t n_reads, int rank, int world, uint64_t* first, uint64_t* count)
{
    if (!first || !count || world < 1 || rank < 0 || rank >= world) return msq_fail(MSQ_ERR_ARG, "msq_comm_shard: bad argument");
    const unsigned __int128 n = n_reads;
    const uint64_t lo = (uint64_t)(n * (unsigned)rank / (unsigned)world), hi = (uint64_t)(n * (unsigned)(rank + 1) / (unsigned)world);
    *first = lo;
    *count = hi - lo;
    return MSQ_OK;
}

// Contiguous read range of a rank, balanced by BASE count ("balanced by base count via base_offset"):
// with prefix[i] = units (bases or 32-base groups) of the reads before read i, rank r starts at the first read i with
// prefix[i] >= r * prefix[n] / world.  Boundaries are monotone in r, so the ranges tile [0, n); reads of equal length
// give msq_comm_shard's ranges up to one read.  Host; every rank computes every boundary from the same table.
extern "C" int msq_comm_shard_bases(const uint64_t* prefix, uint64_t n_reads, int rank, int world, uint64_t* first, uint64_t* count)
{
    if (!first || !count || world < 1 || rank < 0 || rank >= world || (!prefix && n_reads))
        return msq_fail(MSQ_ERR_ARG, "msq_comm_shard_bases: bad argument");
    if (n_reads == 0) {
        *first = *count = 0;
        return MSQ_OK;
    }
    const unsigned __int128 total = prefix[n_reads];
    auto boundary = [&](int r) -> uint64_t {
        if (r <= 0) return 0;
        if (r >= world) return n_reads;
        const uint64_t want = (uint64_t)((total * (unsigned)r + (unsigned)world - 1) / (unsigned)world);  // ceil(r total / world)
        uint64_t lo = 0, hi = n_reads;  // first i in [0, n] with prefix[i] >= want
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (prefix[mid] >= want) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const uint64_t lo = boundary(rank), hi = boundary(rank + 1);
    *first = lo;
    *count = hi > lo ? hi - lo : 0;
    return MSQ_OK;
}

// One rank's status -> every rank's status.  Collective.  An error that one rank alone detects ("Unexpected cigar" in
// its share, a deferred "Unexpected base") must not make that rank leave while the others wait in the next
// collective: every rank calls this with its local return code (and msq_last_error() text), and every rank gets back
// the code and text of the LOWEST rank that failed -- the error a single-process run over the same reads would have
// reported first -- or MSQ_OK.  local_text: the failing rank's message, NULL = msq_last_error() (an error raised above
// the ABI, e.g. an exception in a fill callback, brings its own).  256 bytes per rank, all-gathered on the context's
// stream; synchronises.
extern "C" int msq_comm_agree_status(msq_ctx* ctx, int local_rc, const char* local_text, int* agreed_rc)
{
    if (!ctx || !agreed_rc) return msq_fail(MSQ_ERR_ARG, "msq_comm_agree_status: NULL argument");
    *agreed_rc = local_rc;
    if (!ctx->comm) {  // one process: its status is the status
        if (local_rc != MSQ_OK && local_text) msq_set_error("%s", local_text);
        return MSQ_OK;
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    struct Slot { int32_t rc; char text[252]; };
    static_assert(sizeof(Slot) == 256, "one status slot");
    const int world = ctx->comm->world;
    std::vector<Slot> all((size_t)world);
    Slot mine;
    memset(&mine, 0, sizeof(mine));
    mine.rc = local_rc;
    if (local_rc != MSQ_OK) snprintf(mine.text, sizeof(mine.text), "%s", local_text ? local_text : msq_last_error());
    void* d = nullptr;
    if (int rc = msq_scratch(ctx, sizeof(Slot) * (size_t)(world + 1), &d)) return rc;
    Slot* d_mine = static_cast<Slot*>(d);
    Slot* d_all = d_mine + 1;
    MSQ_HIP(hipMemcpyAsync(d_mine, &mine, sizeof(Slot), hipMemcpyHostToDevice, ctx->stream));
    MSQ_RCCL(g_rccl.AllGather(d_mine, d_all, sizeof(Slot), ncclUint8, ctx->comm->comm, ctx->stream));
    MSQ_HIP(hipMemcpyAsync(all.data(), d_all, sizeof(Slot) * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < world; ++r) {
        if (all[(size_t)r].rc == MSQ_OK) continue;
        all[(size_t)r].text[sizeof(all[(size_t)r].text) - 1] = 0;
        *agreed_rc = all[(size_t)r].rc;
        msq_set_error("%s", all[(size_t)r].text);
        return MSQ_OK;
    }
    *agreed_rc = MSQ_OK;
    return MSQ_OK;
}

extern "C" int msq_comm_allreduce_bounds(msq_ctx* ctx, int32_t* d_bounds)
{
    if (!ctx || !d_bounds) return msq_fail(MSQ_ERR_ARG, "msq_comm_allreduce_bounds: NULL argument");
    if (!ctx->comm) return MSQ_OK;  // one GPU: the shard's window is the window
    MSQ_HIP(hipSetDevice(ctx->device));
    MSQ_RCCL(g_rccl.AllReduce(d_bounds, d_bounds, 2, ncclInt32, ncclMax, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_comm_allreduce_counts(msq_ctx* ctx, uint32_t* d_counts, size_t n)
{
    if (!ctx || (!d_counts && n)) return msq_fail(MSQ_ERR_ARG, "msq_comm_allreduce_counts: NULL argument");
    if (!ctx->comm || n == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    MSQ_RCCL(g_rccl.AllReduce(d_counts, d_counts, n, ncclUint32, ncclSum, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_comm_allreduce_step(msq_ctx* ctx, int32_t* d_bounds, uint32_t* d_counts, size_t n)
{
    if (!ctx || !d_bounds || (!d_counts && n)) return msq_fail(MSQ_ERR_ARG, "msq_comm_allreduce_step: NULL argument");
    if (!ctx->comm) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    MSQ_RCCL(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.AllReduce(d_bounds, d_bounds, 2, ncclInt32, ncclMax, ctx->comm->comm, ctx->stream);
    ncclResult_t r2 = n ? g_rccl.AllReduce(d_counts, d_counts, n, ncclUint32, ncclSum, ctx->comm->comm, ctx->stream) : ncclSuccess;
    ncclResult_t r3 = g_rccl.GroupEnd();
    if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess)
        return msq_fail(MSQ_ERR_HIP, "msq_comm_allreduce_step: %s",
                        g_rccl.GetErrorString(r1 != ncclSuccess ? r1 : r2 != ncclSuccess ? r2 : r3));
    return MSQ_OK;
}

extern "C" int msq_comm_allgather(msq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank)
{
    if (!ctx || ((!d_send || !d_recv) && bytes_per_rank)) return msq_fail(MSQ_ERR_ARG, "msq_comm_allgather: NULL argument");
    if (bytes_per_rank == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm) {
        if (d_send != d_recv) MSQ_HIP(hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return MSQ_OK;
    }
    MSQ_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}
