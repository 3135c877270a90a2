// fake_rccl.cpp — TEST DOUBLE for the five RCCL entry points libcem_mpc_gfx950.so binds (cem_capi.hip: rccl()).
//
// Why: a gpurun lease has ONE GPU and RCCL refuses two ranks on one device, so the library's multi-rank path (candidate shards,
// ncclAllGather of the scores inside the plan, the captured graph with the collective in it, bench.py's multi-rank leg) could
// never execute with world_size > 1 on hardware.  With CEM_RCCL_LIBRARY pointing here, N processes that share the one GPU run
// exactly that path; only the collective itself is replaced: the all-gather goes through a POSIX shared-memory segment (device ->
// pinned host slot, a host-function barrier across the processes, host -> device), every piece a stream operation, so it is
// captured into the plan's hipGraph like the real call.  Nothing here is product code; nothing in the product loads it unless
// that variable is set.  What stays unverified is RCCL itself (its call is the same five arguments).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>

namespace {
constexpr size_t kSlotBytes = 1u << 20;            // per rank and parity: 262 144 floats (B5 has 65 536 scores in all)
constexpr int kMaxRanks = 16;
struct Shared {
    std::atomic<uint32_t> attached;                // ranks that mapped the segment
    std::atomic<uint32_t> arrived[2];              // barrier counters by generation parity (they only grow)
    std::atomic<uint32_t> failed;
    char pad[4096 - 16];
    char slots[kMaxRanks][kSlotBytes];             // rank r's shard of the gather in flight
};
struct Comm {
    int n, rank;
    Shared *sh;
    char name[64];
    uint64_t generation;                           // barriers passed; advanced by the host function, i.e. in stream order
};
struct NcclId { char internal[128]; };

void barrier_fn(void *p)
{
    Comm *c = (Comm *)p;
    const uint64_t g = c->generation++;
    const int par = (int)(g & 1);
    std::atomic<uint32_t> &ctr = c->sh->arrived[par];
    // counters only grow: generation g of parity par completes at (g / 2 + 1) * n arrivals
    const uint32_t target = (uint32_t)((g / 2 + 1) * (uint64_t)c->n);
    ctr.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (ctr.load(std::memory_order_acquire) < target) {
        if (c->sh->failed.load()) return;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { c->sh->failed.store(1); fprintf(stderr, "fake_rccl: rank %d timed out in generation %llu\n", c->rank, (unsigned long long)g); return; }
        std::this_thread::yield();
    }
}
}  // namespace

extern "C" {

int ncclGetUniqueId(NcclId *id)
{
    std::memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/cem_fake_rccl_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}

int ncclCommInitRank(void **comm, int n, NcclId id, int rank)
{
    if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return 4;                 // ncclInvalidArgument
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;                                                          // ncclSystemError
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }               // new pages read as zero: the counters start at 0
    void *m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 2;
    if (hipHostRegister(m, sizeof(Shared), hipHostRegisterDefault) != hipSuccess) { munmap(m, sizeof(Shared)); return 1; }
    Comm *c = new Comm{};
    c->n = n; c->rank = rank; c->sh = (Shared *)m; c->generation = 0;
    std::strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->sh->attached.fetch_add(1);
    // like ncclCommInitRank, return when every rank has joined
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < (uint32_t)n) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 2;
        std::this_thread::yield();
    }
    *comm = c;
    return 0;
}

// Failure knobs (environment, read per call; every rank of a test sets the same ones):
//   CEM_FAKE_RCCL_CAPTURE=error       ncclAllGather on a CAPTURING stream returns ncclInvalidUsage and enqueues nothing — an RCCL that
//                                     refuses to be captured;
//   CEM_FAKE_RCCL_CAPTURE=invalidate  ... performs an operation that is illegal under capture (hipStreamSynchronize of the capturing
//                                     stream) and returns success — an RCCL whose call silently invalidates the capture: the caller's
//                                     later launches and its hipStreamEndCapture fail;
//   CEM_FAKE_RCCL_COUNT_OFFSET=d      ncclCommCount reports n + d — a communicator that is not the one the launcher asked for.
// (libfake_rccl_nocount.so is this file built with -DCEM_FAKE_NO_COMMCOUNT: an RCCL without the optional ncclCommCount export.)
static int env_int(const char *name) { const char *v = std::getenv(name); return v ? std::atoi(v) : 0; }
static bool env_is(const char *name, const char *val) { const char *v = std::getenv(name); return v && std::strcmp(v, val) == 0; }

#ifndef CEM_FAKE_NO_COMMCOUNT
int ncclCommCount(void *comm, int *n) { *n = ((Comm *)comm)->n + env_int("CEM_FAKE_RCCL_COUNT_OFFSET"); return 0; }
#endif

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister(c->sh);
    munmap(c->sh, sizeof(Shared));
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

// ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream): rank r's `count` elements land at recvbuff + r * count
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (dtype != 7) return 4;                                                      // ncclFloat32 only
    const size_t bytes = count * 4;
    if (bytes > kSlotBytes) return 4;
    if (c->sh->failed.load()) return 3;                                            // ncclInternalError
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
            if (env_is("CEM_FAKE_RCCL_CAPTURE", "error")) return 5;                // ncclInvalidUsage
            if (env_is("CEM_FAKE_RCCL_CAPTURE", "invalidate")) { (void)hipStreamSynchronize(stream); (void)hipGetLastError(); return 0; }
        } else (void)hipGetLastError();
    }
    // device -> this rank's host slot; barrier (every slot written); all slots -> device; barrier (every rank has copied the slots
    // out before anyone's next gather overwrites one).  Stream operations only, and nothing but pointers in their arguments, so
    // the sequence can be captured into a hipGraph and replayed.
    if (hipMemcpyAsync(c->sh->slots[c->rank], send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    if (hipLaunchHostFunc(stream, barrier_fn, c) != hipSuccess) return 1;
    for (int r = 0; r < c->n; ++r)
        if (hipMemcpyAsync((char *)recv + (size_t)r * bytes, c->sh->slots[r], bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    if (hipLaunchHostFunc(stream, barrier_fn, c) != hipSuccess) return 1;
    return 0;
}

const char *ncclGetErrorString(int) { return "fake_rccl"; }

}  // extern "C"
