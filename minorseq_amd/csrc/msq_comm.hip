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
//
// A second transport exists for REHEARSALS only (MSQ_COMM_TRANSPORT=shm in the environment when the id is made and the
// communicator created): the ranks are processes of one host that may share one GPU -- which RCCL refuses -- and
// exchange through a POSIX shared-memory segment (device -> segment, a barrier, every rank reduces or gathers all
// slots, -> device).  Same entry points, same results, host-synchronous; it lets a one-GPU box run the sharded front
// and the bench's exchange call sites with world > 1.  It is no data path: nothing selects it but that variable.
#include "msq_internal.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>

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

// the rehearsal transport's segment: a header, then one slot per rank
struct ShmHeader {
    std::atomic<uint32_t> ready;       // kShmMagic once rank 0 has set the segment up
    std::atomic<uint32_t> arrived;     // barrier: ranks in this round
    std::atomic<uint32_t> generation;  // barrier: rounds completed
    uint32_t world;
    uint64_t slot_bytes;
};
constexpr uint32_t kShmMagic = 0x4d535143u;
constexpr size_t kShmSlotBytes = size_t(8) << 20, kShmHeaderBytes = 4096;

struct msq_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // rehearsal transport
    bool shm = false;
    void* base = nullptr;
    size_t map_bytes = 0;
    std::vector<uint32_t> acc;
    ShmHeader* hdr() const { return static_cast<ShmHeader*>(base); }
    uint8_t* slot(int r) const { return static_cast<uint8_t*>(base) + kShmHeaderBytes + (size_t)r * kShmSlotBytes; }
};

namespace {

bool shm_transport()
{
    const char* t = getenv("MSQ_COMM_TRANSPORT");
    return t && strcmp(t, "shm") == 0;
}

void shm_name(const uint8_t* id, char* out, size_t n)
{
    size_t at = (size_t)snprintf(out, n, "/msq_comm_");
    for (int k = 0; k < 12 && at + 3 < n; ++k) at += (size_t)snprintf(out + at, n - at, "%02x", id[k]);
}

// every rank of the communicator; gives up after two minutes (a rank that died must not hang the others for ever)
int shm_barrier(msq_comm* c)
{
    ShmHeader* h = c->hdr();
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
        return MSQ_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; h->generation.load(std::memory_order_acquire) == gen; ++spins) {
        if ((spins & 1023u) == 1023u) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return msq_fail(MSQ_ERR_HIP, "msq_comm (shm rehearsal transport): rank %d waited two minutes at a barrier", c->rank);
        }
    }
    return MSQ_OK;
}

// in place over n 32-bit elements on the device: op 0 = int32 MAX, 1 = uint32 SUM
int shm_allreduce(msq_ctx* ctx, void* d_buf, size_t n, int op)
{
    msq_comm* c = ctx->comm;
    const size_t per = kShmSlotBytes / 4;
    for (size_t at = 0; at < n; at += per) {
        const size_t m = std::min(per, n - at);
        MSQ_HIP(hipMemcpyAsync(c->slot(c->rank), (char*)d_buf + at * 4, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (int rc = shm_barrier(c)) return rc;
        c->acc.resize(m);
        memcpy(c->acc.data(), c->slot(0), m * 4);
        for (int r = 1; r < c->world; ++r) {
            const uint32_t* o = reinterpret_cast<const uint32_t*>(c->slot(r));
            if (op == 0)
                for (size_t i = 0; i < m; ++i) c->acc[i] = (uint32_t)std::max((int32_t)c->acc[i], (int32_t)o[i]);
            else
                for (size_t i = 0; i < m; ++i) c->acc[i] += o[i];
        }
        if (int rc = shm_barrier(c)) return rc;  // every rank has read every slot
        MSQ_HIP(hipMemcpyAsync((char*)d_buf + at * 4, c->acc.data(), m * 4, hipMemcpyHostToDevice, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MSQ_OK;
}

int shm_allgather(msq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank)
{
    msq_comm* c = ctx->comm;
    for (size_t at = 0; at < bytes_per_rank; at += kShmSlotBytes) {
        const size_t m = std::min(kShmSlotBytes, bytes_per_rank - at);
        MSQ_HIP(hipMemcpyAsync(c->slot(c->rank), (const char*)d_send + at, m, hipMemcpyDeviceToHost, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (int rc = shm_barrier(c)) return rc;
        for (int r = 0; r < c->world; ++r)
            MSQ_HIP(hipMemcpyAsync((char*)d_recv + (size_t)r * bytes_per_rank + at, c->slot(r), m, hipMemcpyHostToDevice, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (int rc = shm_barrier(c)) return rc;
    }
    return MSQ_OK;
}

int shm_create(msq_ctx* ctx, int rank, int world, const uint8_t* id)
{
    char name[64];
    shm_name(id, name, sizeof name);
    msq_comm* c = new msq_comm;
    c->rank = rank;
    c->world = world;
    c->shm = true;
    c->map_bytes = kShmHeaderBytes + (size_t)world * kShmSlotBytes;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) {
            close(fd);
            shm_unlink(name);
            fd = -1;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(120)) usleep(2000);
        struct stat st;  // rank 0 sizes the segment right after creating it
        while (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size < c->map_bytes && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(120)) usleep(2000);
    }
    if (fd < 0) {
        delete c;
        return msq_fail(MSQ_ERR_HIP, "msq_comm (shm rehearsal transport): rank %d cannot open %s", rank, name);
    }
    c->base = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) {
        if (rank == 0) shm_unlink(name);
        delete c;
        return msq_fail(MSQ_ERR_NOMEM, "msq_comm (shm rehearsal transport): cannot map %s", name);
    }
    ShmHeader* h = c->hdr();
    if (rank == 0) {
        h->arrived.store(0);
        h->generation.store(0);
        h->world = (uint32_t)world;
        h->slot_bytes = kShmSlotBytes;
        h->ready.store(kShmMagic, std::memory_order_release);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (h->ready.load(std::memory_order_acquire) != kShmMagic) {
            usleep(1000);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                munmap(c->base, c->map_bytes);
                delete c;
                return msq_fail(MSQ_ERR_HIP, "msq_comm (shm rehearsal transport): rank %d never saw the segment ready", rank);
            }
        }
        if (h->world != (uint32_t)world) {
            munmap(c->base, c->map_bytes);
            delete c;
            return msq_fail(MSQ_ERR_ARG, "msq_comm (shm rehearsal transport): the segment is for %u ranks, not %d", h->world, world);
        }
    }
    ctx->comm = c;
    int rc = shm_barrier(c);  // everyone is attached: the name can go
    if (rank == 0) shm_unlink(name);
    if (rc != MSQ_OK) {
        munmap(c->base, c->map_bytes);
        delete c;
        ctx->comm = nullptr;
    }
    return rc;
}

}  // namespace

static_assert(MSQ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "msq_comm_unique_id hands out an ncclUniqueId");

// Can this process reach RCCL at all?  Local and non-collective: ranks agree on the answer over their control plane
// BEFORE any of them enters msq_comm_create (ncclCommInitRank blocks until every rank has arrived).
extern "C" int msq_comm_available(void) { return shm_transport() ? MSQ_OK : load_rccl(); }

extern "C" int msq_comm_unique_id(uint8_t* id_out)
{
    if (!id_out) return msq_fail(MSQ_ERR_ARG, "msq_comm_unique_id: id_out is NULL");
    if (shm_transport()) {  // rehearsal: random bytes name the segment
        memset(id_out, 0, MSQ_COMM_ID_BYTES);
        const int fd = open("/dev/urandom", O_RDONLY);
        const bool ok = fd >= 0 && read(fd, id_out, 16) == 16;
        if (fd >= 0) close(fd);
        if (!ok) return msq_fail(MSQ_ERR_HIP, "msq_comm_unique_id: cannot read /dev/urandom");
        return MSQ_OK;
    }
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
    MSQ_HIP(hipSetDevice(ctx->device));
    if (shm_transport()) return shm_create(ctx, rank, world, id);
    if (int rc = load_rccl()) return rc;
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
    if (ctx->comm->shm && ctx->comm->base) munmap(ctx->comm->base, ctx->comm->map_bytes);
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
    if (ctx->comm->shm) {
        if (int rc = shm_allgather(ctx, d_mine, d_all, sizeof(Slot))) return rc;
    } else {
        MSQ_RCCL(g_rccl.AllGather(d_mine, d_all, sizeof(Slot), ncclUint8, ctx->comm->comm, ctx->stream));
    }
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
    if (ctx->comm->shm) return shm_allreduce(ctx, d_bounds, 2, 0);
    MSQ_RCCL(g_rccl.AllReduce(d_bounds, d_bounds, 2, ncclInt32, ncclMax, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_comm_allreduce_counts(msq_ctx* ctx, uint32_t* d_counts, size_t n)
{
    if (!ctx || (!d_counts && n)) return msq_fail(MSQ_ERR_ARG, "msq_comm_allreduce_counts: NULL argument");
    if (!ctx->comm || n == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    if (ctx->comm->shm) return shm_allreduce(ctx, d_counts, n, 1);
    MSQ_RCCL(g_rccl.AllReduce(d_counts, d_counts, n, ncclUint32, ncclSum, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_comm_allreduce_step(msq_ctx* ctx, int32_t* d_bounds, uint32_t* d_counts, size_t n)
{
    if (!ctx || !d_bounds || (!d_counts && n)) return msq_fail(MSQ_ERR_ARG, "msq_comm_allreduce_step: NULL argument");
    if (!ctx->comm) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    if (ctx->comm->shm) {
        if (int rc = shm_allreduce(ctx, d_bounds, 2, 0)) return rc;
        return n ? shm_allreduce(ctx, d_counts, n, 1) : MSQ_OK;
    }
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
    if (ctx->comm->shm) return shm_allgather(ctx, d_send, d_recv, bytes_per_rank);
    MSQ_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, ctx->comm->comm, ctx->stream));
    return MSQ_OK;
}
