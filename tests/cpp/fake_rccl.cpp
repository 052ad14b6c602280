// fake_rccl.cpp -- TEST DOUBLE of the four RCCL entry points libk16.so's one-process-per-GPU leg uses (csrc/msm_sharded.hip:
// ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, + ncclCommAbort / ncclGetErrorString), so that
// k16_rank_comm_* can run at world size > 1 on a ONE-GPU box: real RCCL refuses two ranks on one device.  Built by
// tests/test_gpu_rank_comm.py into a temporary directory as librccl.so.1 and handed to the library through K16_RCCL_LIB; it
// is never part of libk16.so / libk16.a (tests/test_boundary.py checks that).
//
// The ranks are separate PROCESSES (all on whatever device they selected); the "fabric" is a POSIX shared-memory segment named
// by the unique id: a header of per-rank generation counters and two banks of per-rank 4 KB slots.  ncclAllGather is
// stream-ordered like the real one from the caller's point of view -- it drains the stream, copies the send buffer to the
// rank's slot, waits (bounded: FAKE_RCCL_TIMEOUT_MS, default 20 s) until every rank has published the same generation, and
// copies all slots into the receive buffer -- but it blocks the calling thread while it does so.  A rank that never arrives
// makes the others return ncclSystemError, which the library maps to K16_ERR_HIP.
// FAKE_RCCL_ASYNC=1: the collective is ENQUEUED like the real one (the segment is page-locked with hipHostRegister: an async
// copy of the send buffer into the rank's slot, a host function on the stream that publishes the generation and waits for
// the peers, async copies of all slots into the receive buffer) and ncclAllGather returns at once -- a missing rank then
// shows up as a stream that never drains, which is what the library's own bounded wait (K16_RANK_COMM_TIMEOUT_MS) and its
// ncclCommAbort are for.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

namespace {
constexpr int    MAX_RANKS  = 64;
constexpr size_t SLOT_BYTES = 4096;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

struct Fabric {
    std::atomic<uint32_t> arrived;              // ranks that have mapped the segment
    std::atomic<uint32_t> calls;                // diagnostic: collectives completed by rank 0
    std::atomic<uint64_t> written[MAX_RANKS];   // generation whose payload rank r has published
    std::atomic<uint64_t> read_done[MAX_RANKS]; // generation rank r has finished copying out
    unsigned char         slots[2][MAX_RANKS][SLOT_BYTES];
};
struct Comm {
    Fabric*  f     = nullptr;
    int      rank  = 0, world = 1;
    uint64_t gen   = 0;
    char     name[128];
    void*    h_tmp = nullptr; // page-locked bounce buffer, world x SLOT_BYTES
    bool     async = false;   // FAKE_RCCL_ASYNC=1: the segment is registered, collectives are enqueued
    std::atomic<bool>     aborted{false};
    std::atomic<unsigned> pending{0}; // host functions enqueued and not yet returned
    hipStream_t           last_stream = nullptr;
};
struct AsyncStep {
    Comm*    c;
    uint64_t g;
};
void async_publish_and_wait(void* arg)
{
    AsyncStep* st = (AsyncStep*)arg;
    Comm*      c  = st->c;
    Fabric*    f  = c->f;
    f->written[c->rank].store(st->g, std::memory_order_release);
    for (unsigned spin = 0;; spin++) { // (unbounded on purpose: the CALLER bounds the wait and aborts)
        if (c->aborted.load()) break;
        bool all = true;
        for (int r = 0; r < c->world; r++)
            if (f->written[r].load(std::memory_order_acquire) < st->g) all = false;
        if (all) break;
        if (spin > 1000) usleep(50);
    }
    delete st;
    c->pending.fetch_sub(1);
}
void async_mark_read(void* arg)
{
    AsyncStep* st = (AsyncStep*)arg;
    if (!st->c->aborted.load()) st->c->f->read_done[st->c->rank].store(st->g, std::memory_order_release);
    st->c->pending.fetch_sub(1);
    delete st;
}
struct Id {
    char internal[128];
};

long timeout_ms()
{
    const char* e = getenv("FAKE_RCCL_TIMEOUT_MS");
    return (e && atol(e) > 0) ? atol(e) : 20000;
}
double now_ms()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
template <class P>
bool wait_until(P pred)
{
    const double end = now_ms() + (double)timeout_ms();
    for (unsigned spin = 0; !pred(); spin++) {
        if (now_ms() > end) return false;
        if (spin > 1000) usleep(50);
    }
    return true;
}
} // namespace

extern "C" {

int ncclGetUniqueId(Id* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    static std::atomic<unsigned> counter{0};
    timespec                     t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/k16_fake_rccl_%d_%u_%lx", (int)getpid(), counter++, (unsigned long)t.tv_nsec);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)sizeof(Fabric)) != 0) { // (a fresh segment is zero-filled: every counter starts at 0)
        close(fd);
        shm_unlink(id->internal);
        return ncclSystemError;
    }
    close(fd);
    return ncclSuccess;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/k16_fake_rccl_", 15) != 0) return ncclInvalidArgument;
    int fd = -1;
    if (!wait_until([&] { return (fd = shm_open(id.internal, O_RDWR, 0600)) >= 0; })) return ncclSystemError;
    void* m = mmap(nullptr, sizeof(Fabric), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Comm* c  = new Comm;
    c->f     = (Fabric*)m;
    c->rank  = rank;
    c->world = nranks;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    const char* as = getenv("FAKE_RCCL_ASYNC");
    c->async       = as && *as == '1';
    if (hipHostMalloc(&c->h_tmp, (size_t)nranks * SLOT_BYTES, hipHostMallocDefault) != hipSuccess ||
        (c->async && hipHostRegister(m, sizeof(Fabric), hipHostRegisterPortable) != hipSuccess)) {
        if (c->h_tmp) (void)hipHostFree(c->h_tmp);
        munmap(m, sizeof(Fabric));
        delete c;
        return ncclUnhandledCudaError;
    }
    c->f->arrived.fetch_add(1);
    const bool all = wait_until([&] { return c->f->arrived.load() >= (uint32_t)nranks; }); // collective, like the real one
    if (rank == 0) shm_unlink(c->name); // everyone who will ever map it has (or has given up): nothing is left in /dev/shm
    if (!all) {
        (void)hipHostFree(c->h_tmp);
        if (c->async) (void)hipHostUnregister(m);
        munmap(m, sizeof(Fabric));
        delete c;
        return ncclSystemError;
    }
    *comm = c;
    return ncclSuccess;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, int datatype, void* comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || !sendbuff || !recvbuff || (datatype != 0 && datatype != 1) /* ncclInt8 / ncclUint8 */ || count == 0 || count > SLOT_BYTES)
        return ncclInvalidArgument;
    Fabric*        f = c->f;
    const uint64_t g = ++c->gen;
    if (c->async) {
        c->last_stream = stream;
        // bank g % 2 is free once every rank has read generation g - 2; with at most one collective in flight per rank (the
        // library waits for each) that holds whenever generation g - 1 completed, so no wait is needed here
        if (hipMemcpyAsync(f->slots[g & 1][c->rank], sendbuff, count, hipMemcpyDeviceToHost, stream) != hipSuccess)
            return ncclUnhandledCudaError;
        c->pending.fetch_add(2);
        if (hipLaunchHostFunc(stream, async_publish_and_wait, new AsyncStep{c, g}) != hipSuccess) return ncclUnhandledCudaError;
        for (int r = 0; r < c->world; r++)
            if (hipMemcpyAsync((char*)recvbuff + (size_t)r * count, f->slots[g & 1][r], count, hipMemcpyHostToDevice, stream) != hipSuccess)
                return ncclUnhandledCudaError;
        if (hipLaunchHostFunc(stream, async_mark_read, new AsyncStep{c, g}) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError; // what was queued before the collective
    // bank g % 2 still holds generation g - 2: wait until every rank has copied that one out
    if (g > 2 && !wait_until([&] {
            for (int r = 0; r < c->world; r++)
                if (f->read_done[r].load() < g - 2) return false;
            return true;
        }))
        return ncclSystemError;
    unsigned char* mine = f->slots[g & 1][c->rank];
    if (hipMemcpy(mine, sendbuff, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    f->written[c->rank].store(g, std::memory_order_release);
    if (!wait_until([&] {
            for (int r = 0; r < c->world; r++)
                if (f->written[r].load(std::memory_order_acquire) < g) return false;
            return true;
        }))
        return ncclSystemError; // a rank never arrived
    for (int r = 0; r < c->world; r++) memcpy((char*)c->h_tmp + (size_t)r * count, f->slots[g & 1][r], count);
    if (hipMemcpy(recvbuff, c->h_tmp, (size_t)c->world * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    f->read_done[c->rank].store(g, std::memory_order_release);
    if (c->rank == 0) f->calls.fetch_add(1);
    return ncclSuccess;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    c->aborted.store(true); // (a host function still waiting for a peer leaves its loop; nothing touches the segment after that)
    for (unsigned spin = 0; c->pending.load() != 0 && spin < 200000; spin++) usleep(50);
    if (c->async && c->last_stream) (void)hipStreamSynchronize(c->last_stream); // the copies queued behind the host functions
    if (c->h_tmp) (void)hipHostFree(c->h_tmp);
    if (c->async) (void)hipHostUnregister(c->f);
    munmap(c->f, sizeof(Fabric));
    delete c;
    return ncclSuccess;
}
int ncclCommAbort(void* comm) { return ncclCommDestroy(comm); }

const char* ncclGetErrorString(int e)
{
    switch (e) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: a rank did not arrive in time";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    default: return "fake rccl: internal error";
    }
}

// only the double has this: lets a test prove that the collective really went through it
const char* k16_fake_rccl_marker(void) { return "k16 fake rccl test double"; }
}
