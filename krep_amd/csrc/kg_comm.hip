// kg_comm.hip — the one collective of the path: per-shard counters meet in ONE RCCL all-reduce over xGMI.
//
// The reference merges its chunks on the host thread that joined the workers (krep.c:2930-3016: sums of count_result,
// concatenation of the local lists).  Here a shard is a GPU, and what has to meet is a handful of 64-bit counters per
// shard {matches, lines, line-carry bits}: one ncclAllReduce(uint64, sum) over slotted vectors (slot g is written by shard
// g only, so the sum is at the same time the all-gather the left-to-right line fold needs — SURVEY §8e "one exchange step").
//
// Two ways in, one implementation:
//   * ONE PROCESS, several devices (search_buffer(num_gpus > 1), the search_func_t operators with num_gpus > 1 — i.e. the
//     reference CLI with KREP_GPU_NUM=8): ncclCommInitAll over the devices used, cached per device list;
//   * ONE PROCESS PER GPU (bench.py under torch.distributed.run, one rank per GPU): krep_gpu_comm_unique_id() on rank 0, the
//     128-byte id travels through whatever bootstrap the host program has, krep_gpu_comm_init_rank() everywhere, then
//     krep_gpu_comm_allreduce_u64() per scan.
// librccl is opened on FIRST USE (dlopen), not linked: it is a 570 MB library whose load costs 0.2 s, which every
// single-GPU invocation of the CLI would pay for nothing; in a process that already holds an RCCL (PyTorch bundles one,
// same soname) that copy is the one found, so there are never two.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_internal.h"

namespace {
struct Rccl
{
    void *h = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why; // load failure
};
std::mutex g_mu;                 // RCCL communicators are used by one thread at a time
std::atomic<uint64_t> g_calls{0}; // collectives issued (diagnostic, krep_gpu_rccl_calls)

// RCCL prints a banner ("RCCL version : ... Hostname ... Librccl path ...") to STDOUT when a process creates its first
// communicator.  A drop-in must not add a byte to its host's output (krep's stdout IS its result).  The banner belongs to RCCL's
// logging: a process STARTED with NCCL_DEBUG=NONE does not print it (measured, round 5) — and then nothing is done here.  Putting
// the variable into the environment from inside the process, before librccl is opened, does NOT work (measured with RCCL 2.27.7:
// the patched CLI still printed the banner), so without it file descriptor 1 points to /dev/null while the communicator is
// created: stdio is flushed on both sides of the switch, the host's pending output reaches the real stdout and whatever RCCL left
// in a stdio buffer goes down the drain.  The switch is process-wide — a host that prints from other threads while a search runs
// should export NCCL_DEBUG=NONE — and it happens once, at selector time (comm_warmup), before the host has
// printed anything.  ($KREP_GPU_RCCL_BANNER=1 keeps the banner.)
struct StdoutMute
{
    int saved = -1;
    StdoutMute()
    {
        const char *dbg = getenv("NCCL_DEBUG");
        if (getenv("KREP_GPU_RCCL_BANNER") || (dbg && (!strcmp(dbg, "NONE") || !strcmp(dbg, "none"))))
            return;
        fflush(stdout);
        saved = dup(1);
        const int nul = open("/dev/null", O_WRONLY);
        if (saved >= 0 && nul >= 0)
            (void)dup2(nul, 1);
        if (nul >= 0)
            close(nul);
    }
    ~StdoutMute()
    {
        if (saved < 0)
            return;
        fflush(stdout);
        (void)dup2(saved, 1);
        close(saved);
    }
};

Rccl *g_rccl = nullptr;
Rccl *rccl() // g_mu held
{
    Rccl *&r = g_rccl;
    if (r)
        return r->h ? r : nullptr;
    r = new Rccl();
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((r->h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!r->h)
    {
        r->why = std::string("cannot load librccl: ") + dlerror();
        return nullptr;
    }
    auto sym = [&](const char *n) {
        void *p = dlsym(r->h, n);
        if (!p && r->why.empty())
            r->why = std::string("librccl lacks ") + n;
        return p;
    };
    r->GetVersion = (decltype(r->GetVersion))sym("ncclGetVersion");
    r->GetUniqueId = (decltype(r->GetUniqueId))sym("ncclGetUniqueId");
    r->CommInitRank = (decltype(r->CommInitRank))sym("ncclCommInitRank");
    r->CommInitAll = (decltype(r->CommInitAll))sym("ncclCommInitAll");
    r->CommDestroy = (decltype(r->CommDestroy))sym("ncclCommDestroy");
    r->GroupStart = (decltype(r->GroupStart))sym("ncclGroupStart");
    r->GroupEnd = (decltype(r->GroupEnd))sym("ncclGroupEnd");
    r->AllReduce = (decltype(r->AllReduce))sym("ncclAllReduce");
    r->GetErrorString = (decltype(r->GetErrorString))sym("ncclGetErrorString");
    if (!r->why.empty())
    {
        dlclose(r->h);
        r->h = nullptr;
        return nullptr;
    }
    return r;
}
const char *rccl_why() { return g_rccl && !g_rccl->why.empty() ? g_rccl->why.c_str() : "librccl is not loadable"; }

#define NCHK(R, x)                                                                                 \
    do                                                                                             \
    {                                                                                              \
        ncclResult_t e_ = (x);                                                                     \
        if (e_ != ncclSuccess)                                                                     \
            return kg::fail("%s failed: %s (%s:%d)", #x, (R)->GetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define HCHK(x)                                                                                   \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess)                                                                     \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- one process, several devices ---------------------------------------------------------------------------------
struct Clique
{
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<unsigned long long *> d_vec; // per device: the slotted counter vector
    size_t cap = 0;                          // u64 entries of d_vec
};
std::map<std::vector<int>, Clique *> *g_cliques = nullptr; // leaked at exit: RCCL / HIP may be gone by then

int clique_for(Rccl *R, const std::vector<int> &devs, size_t n, Clique **out) // g_mu held
{
    if (!g_cliques)
        g_cliques = new std::map<std::vector<int>, Clique *>();
    Clique *&c = (*g_cliques)[devs];
    if (!c)
    {
        Clique *nc = new Clique();
        nc->devs = devs;
        nc->comms.resize(devs.size());
        ncclResult_t e;
        {
            StdoutMute mute;
            e = R->CommInitAll(nc->comms.data(), (int)devs.size(), devs.data());
        }
        if (e != ncclSuccess)
        {
            delete nc;
            g_cliques->erase(devs);
            return kg::fail("ncclCommInitAll over %zu devices failed: %s", devs.size(), R->GetErrorString(e));
        }
        nc->streams.resize(devs.size(), nullptr);
        nc->d_vec.resize(devs.size(), nullptr);
        bool ok = true;
        for (size_t i = 0; i < devs.size() && ok; ++i)
            ok = hipSetDevice(devs[i]) == hipSuccess && hipStreamCreateWithFlags(&nc->streams[i], hipStreamNonBlocking) == hipSuccess;
        if (!ok)
        { // a half-built clique is torn down completely: no leaked communicators, no nullptr entry left in the map
            const hipError_t he = hipGetLastError();
            for (size_t i = 0; i < devs.size(); ++i)
            {
                if (nc->streams[i]) (void)hipStreamDestroy(nc->streams[i]);
                if (nc->comms[i]) (void)R->CommDestroy(nc->comms[i]);
            }
            delete nc;
            g_cliques->erase(devs);
            return kg::fail("cannot create the per-device streams of the %zu-device communicator: %s", devs.size(), hipGetErrorString(he));
        }
        c = nc;
    }
    if (c->cap < n)
    {
        for (size_t i = 0; i < devs.size(); ++i)
        {
            HCHK(hipSetDevice(devs[i]));
            if (c->d_vec[i]) (void)hipFree(c->d_vec[i]);
            c->d_vec[i] = nullptr;
            HCHK(hipMalloc(&c->d_vec[i], n * sizeof(unsigned long long)));
        }
        c->cap = n;
    }
    *out = c;
    return 0;
}

// ---- one process per GPU ------------------------------------------------------------------------------------------
struct RankComm
{
    ncclComm_t comm = nullptr;
    int device = 0, nranks = 0, rank = 0;
    hipStream_t stream = nullptr;
    unsigned long long *d_vec = nullptr;
    size_t cap = 0;
} g_rank;
} // namespace

namespace kg {
// vecs[i] (n entries, host) belongs to devs[i]; on return every vecs[i] holds the element-wise sum over all devices.
// The counters travel host -> HBM -> one grouped ncclAllReduce on the devices' streams -> host.
int allreduce_across_devices(const std::vector<int> &devs, std::vector<std::vector<unsigned long long>> &vecs)
{
    const size_t n = vecs.empty() ? 0 : vecs[0].size();
    if (devs.empty() || n == 0)
        return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (!R)
        return kg::fail("%s", rccl_why());
    int prev = -1;
    (void)hipGetDevice(&prev);
    Clique *c = nullptr;
    int rc = clique_for(R, devs, n, &c);
    auto body = [&]() -> int {
        for (size_t i = 0; i < devs.size(); ++i)
        {
            HCHK(hipSetDevice(devs[i]));
            HCHK(hipMemcpyAsync(c->d_vec[i], vecs[i].data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice, c->streams[i]));
        }
        NCHK(R, R->GroupStart());
        for (size_t i = 0; i < devs.size(); ++i)
            NCHK(R, R->AllReduce(c->d_vec[i], c->d_vec[i], n, ncclUint64, ncclSum, c->comms[i], c->streams[i]));
        NCHK(R, R->GroupEnd());
        g_calls.fetch_add(1);
        for (size_t i = 0; i < devs.size(); ++i)
        {
            HCHK(hipSetDevice(devs[i]));
            HCHK(hipMemcpyAsync(vecs[i].data(), c->d_vec[i], n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->streams[i]));
        }
        for (size_t i = 0; i < devs.size(); ++i)
        {
            HCHK(hipSetDevice(devs[i]));
            HCHK(hipStreamSynchronize(c->streams[i]));
        }
        return 0;
    };
    if (!rc)
        rc = body();
    if (prev >= 0)
        (void)hipSetDevice(prev);
    return rc;
}

// Creates (and caches) the communicator of a device list ahead of the first search (ncclCommInitAll costs ~1 s): a host calls
// this — through krep_gpu_select_search_algorithm() — so that no search ever creates one.
int comm_warmup(const std::vector<int> &devs)
{
    if (devs.size() < 2)
        return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (!R)
        return kg::fail("%s", rccl_why());
    int prev = -1;
    (void)hipGetDevice(&prev);
    Clique *c = nullptr;
    const int rc = clique_for(R, devs, 8, &c);
    if (prev >= 0)
        (void)hipSetDevice(prev);
    return rc;
}
// ranks of the cached communicator over `devs` (0: none yet)
int comm_clique_ranks(const std::vector<int> &devs)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_cliques)
        return 0;
    auto it = g_cliques->find(devs);
    return it == g_cliques->end() || !it->second ? 0 : (int)it->second->comms.size();
}
} // namespace kg

extern "C" uint64_t krep_gpu_rccl_calls(void) { return g_calls.load(); }

extern "C" int krep_gpu_rccl_version(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    int v = 0;
    if (!R || R->GetVersion(&v) != ncclSuccess)
        return 0;
    return v;
}

extern "C" int krep_gpu_comm_unique_id(void *id128)
{
    if (!id128)
        return kg::fail("comm_unique_id: NULL buffer");
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (!R)
        return kg::fail("%s", rccl_why());
    ncclUniqueId id;
    NCHK(R, R->GetUniqueId(&id));
    static_assert(sizeof id == KREP_GPU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int krep_gpu_comm_init_rank(const void *id128, int nranks, int rank, int device)
{
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return kg::fail("comm_init_rank: bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (!R)
        return kg::fail("%s", rccl_why());
    if (g_rank.comm)
        return kg::fail("comm_init_rank: a rank communicator already exists (krep_gpu_comm_destroy first)");
    struct Restore // the caller's device comes back on EVERY path out of here
    {
        int prev = -1;
        Restore() { (void)hipGetDevice(&prev); }
        ~Restore()
        {
            if (prev >= 0)
                (void)hipSetDevice(prev);
        }
    } restore;
    HCHK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    {
        StdoutMute mute;
        const ncclResult_t e = R->CommInitRank(&g_rank.comm, nranks, id, rank);
        if (e != ncclSuccess)
        {
            g_rank = RankComm{};
            return kg::fail("ncclCommInitRank failed: %s", R->GetErrorString(e));
        }
    }
    if (hipStreamCreateWithFlags(&g_rank.stream, hipStreamNonBlocking) != hipSuccess)
    { // no half-initialised communicator stays behind (a set comm with a NULL stream)
        const hipError_t he = hipGetLastError();
        (void)R->CommDestroy(g_rank.comm);
        g_rank = RankComm{};
        return kg::fail("hipStreamCreate for the rank communicator failed: %s", hipGetErrorString(he));
    }
    g_rank.device = device;
    g_rank.nranks = nranks;
    g_rank.rank = rank;
    return 0;
}

// in place, device-resident: d_values[n] on the communicator's device, on `stream` (NULL: the communicator's own stream)
extern "C" int krep_gpu_comm_allreduce_device_u64(void *d_values, int n, void *stream)
{
    if (!d_values || n <= 0)
        return kg::fail("comm_allreduce: bad arguments");
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (!R || !g_rank.comm)
        return kg::fail("comm_allreduce: no rank communicator (krep_gpu_comm_init_rank)");
    NCHK(R, R->AllReduce(d_values, d_values, (size_t)n, ncclUint64, ncclSum, g_rank.comm, stream ? (hipStream_t)stream : g_rank.stream));
    g_calls.fetch_add(1);
    return 0;
}

// host values in and out: H2D, the all-reduce, D2H on the communicator's stream; returns when the sums are in `values`
extern "C" int krep_gpu_comm_allreduce_u64(uint64_t *values, int n)
{
    if (!values || n <= 0)
        return kg::fail("comm_allreduce: bad arguments");
    struct Restore // the communicator's device for the whole call, the caller's device afterwards
    {
        int prev = -1;
        Restore() { (void)hipGetDevice(&prev); }
        ~Restore()
        {
            if (prev >= 0)
                (void)hipSetDevice(prev);
        }
    } restore;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_rank.comm)
            return kg::fail("comm_allreduce: no rank communicator (krep_gpu_comm_init_rank)");
        HCHK(hipSetDevice(g_rank.device));
        if (g_rank.cap < (size_t)n)
        {
            if (g_rank.d_vec) (void)hipFree(g_rank.d_vec);
            g_rank.d_vec = nullptr;
            g_rank.cap = 0;
            HCHK(hipMalloc(&g_rank.d_vec, (size_t)n * sizeof(unsigned long long)));
            g_rank.cap = (size_t)n;
        }
        HCHK(hipMemcpyAsync(g_rank.d_vec, values, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, g_rank.stream));
    }
    if (krep_gpu_comm_allreduce_device_u64(g_rank.d_vec, n, nullptr))
        return 2;
    HCHK(hipMemcpyAsync(values, g_rank.d_vec, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, g_rank.stream));
    HCHK(hipStreamSynchronize(g_rank.stream));
    return 0;
}

extern "C" void krep_gpu_comm_destroy(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Rccl *R = rccl();
    if (g_rank.comm && R)
        (void)R->CommDestroy(g_rank.comm);
    if (g_rank.stream) (void)hipStreamDestroy(g_rank.stream);
    if (g_rank.d_vec) (void)hipFree(g_rank.d_vec);
    g_rank = RankComm{};
}
