// fake_rccl.cpp -- TEST INFRASTRUCTURE, never linked into the product: a stand-in for librccl whose "ranks" are THREADS of one
// process, each driving a context of its own on the same GPU.  RCCL refuses two ranks on one device, so on a one-GPU box this
// is the only way the > 1-rank branch of prt_hip_gather_rccl (prt_amd/csrc/prt_gather.hip: pack, grouped ncclSend / ncclRecv,
// the staging offsets, the de-interleave per peer) can execute at all.  The product binds it through PRT_RCCL_LIB.
//
// Semantics kept from RCCL: calls between ncclGroupStart and ncclGroupEnd are queued and issued at the outermost ncclGroupEnd;
// a send matches the receive posted by the peer for the same (communicator, source, destination) in posting order; the bytes
// move on the RECEIVER's stream after everything the sender had enqueued on its stream before the send; a receive whose count
// differs from the matching send's fails (ncclInvalidArgument) -- which catches a wrong staging offset table.
// Simplification: ncclGroupEnd blocks the host until the thread's sends have been consumed and its receives have landed.
//
// Test hooks (not part of rccl.h): fake_rccl_open_groups() = the calling thread's group depth; fake_rccl_fail_next(what)
// makes the calling thread's next ncclSend (1) or ncclRecv (2) return ncclInternalError; fake_rccl_counts(sends, recvs, bytes).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct Comm {
    uint64_t id;
    int rank, n;
};

struct Post { // one send waiting for its receive
    const void* src;
    size_t bytes;
    hipEvent_t ready; // recorded on the sender's stream at the send
    bool done = false, failed = false;
};

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};

std::mutex mu;
std::condition_variable cv;
std::map<std::tuple<uint64_t, int, int>, std::deque<Post*>> mailbox; // (communicator, from, to)
std::atomic<uint64_t> nextId{1}, nSends{0}, nRecvs{0}, nBytes{0};
thread_local int depth = 0;
thread_local int failNext = 0;
thread_local std::vector<Op> queued;

const int kSuccess = 0, kInternalError = 3, kInvalidArgument = 4;

size_t type_size(int t) { return (t == 0 || t == 1) ? 1 : (t == 6 || t == 9) ? 2 : (t == 4 || t == 5 || t == 8) ? 8 : 4; }

int run(std::vector<Op>& ops)
{
    int rc = kSuccess;
    std::vector<Post*> mine;
    // sends first: nothing below may block before every send of this thread is visible to its peer
    for (Op& o : ops) {
        if (!o.send) continue;
        Post* p = new Post{o.buf, o.bytes, nullptr};
        if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(p->ready, o.stream) != hipSuccess) rc = kInternalError;
        {
            std::lock_guard<std::mutex> g(mu);
            mailbox[{o.comm->id, o.comm->rank, o.peer}].push_back(p);
        }
        cv.notify_all();
        mine.push_back(p);
        nSends++;
    }
    for (Op& o : ops) {
        if (o.send) continue;
        Post* p = nullptr;
        {
            std::unique_lock<std::mutex> g(mu);
            auto& q = mailbox[{o.comm->id, o.peer, o.comm->rank}];
            if (!cv.wait_for(g, std::chrono::seconds(60), [&] { return !q.empty(); })) return kInternalError; // a missing peer must not hang the test
            p = q.front();
            q.pop_front();
        }
        bool ok = p->bytes == o.bytes;
        if (ok) {
            ok = hipStreamWaitEvent(o.stream, p->ready, 0) == hipSuccess && hipMemcpyAsync(o.buf, p->src, o.bytes, hipMemcpyDeviceToDevice, o.stream) == hipSuccess &&
                 hipStreamSynchronize(o.stream) == hipSuccess;
            nBytes += o.bytes;
        }
        {
            std::lock_guard<std::mutex> g(mu);
            p->done = true;
            p->failed = !ok;
        }
        cv.notify_all();
        if (!ok) rc = kInvalidArgument;
        nRecvs++;
    }
    for (Post* p : mine) {
        std::unique_lock<std::mutex> g(mu);
        if (!cv.wait_for(g, std::chrono::seconds(60), [&] { return p->done; })) return kInternalError;
        if (p->failed) rc = kInvalidArgument;
        g.unlock();
        (void)hipEventDestroy(p->ready);
        delete p;
    }
    return rc;
}

int submit(const Op& o)
{
    if (depth > 0) {
        queued.push_back(o);
        return kSuccess;
    }
    std::vector<Op> one{o};
    return run(one);
}

} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    const uint64_t v = nextId++;
    memcpy(id->internal, &v, sizeof(v));
    return kSuccess;
}

int ncclCommInitRank(void** comm, int n, ncclUniqueId id, int rank)
{
    if (!comm || n <= 0 || rank < 0 || rank >= n) return kInvalidArgument;
    uint64_t v;
    memcpy(&v, id.internal, sizeof(v));
    *comm = new Comm{v, rank, n};
    return kSuccess;
}

int ncclCommDestroy(void* comm)
{
    delete (Comm*)comm;
    return kSuccess;
}

static std::atomic<int> abortedComms{0};
int ncclCommAbort(void* comm)
{
    abortedComms++;
    delete (Comm*)comm;
    return kSuccess;
}
int fake_rccl_aborted_comms() { return abortedComms.load(); }

int ncclCommCount(void* comm, int* n)
{
    *n = ((Comm*)comm)->n;
    return kSuccess;
}

int ncclCommUserRank(void* comm, int* r)
{
    *r = ((Comm*)comm)->rank;
    return kSuccess;
}

int ncclGroupStart()
{
    depth++;
    return kSuccess;
}

int ncclGroupEnd()
{
    if (depth <= 0) return kInvalidArgument;
    if (--depth > 0) return kSuccess;
    std::vector<Op> ops;
    ops.swap(queued);
    return run(ops);
}

int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    if (failNext == 1) {
        failNext = 0;
        return kInternalError;
    }
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->n || peer == c->rank) return kInvalidArgument;
    return submit(Op{true, (void*)buf, count * type_size(type), peer, c, stream});
}

int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    if (failNext == 2) {
        failNext = 0;
        return kInternalError;
    }
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->n || peer == c->rank) return kInvalidArgument;
    return submit(Op{false, buf, count * type_size(type), peer, c, stream});
}

const char* ncclGetErrorString(int r)
{
    return r == kSuccess ? "no error" : r == kInvalidArgument ? "invalid argument (stand-in: unmatched send/recv size or bad peer)" : "internal error (stand-in)";
}

int fake_rccl_open_groups(void) { return depth; }
void fake_rccl_fail_next(int what) { failNext = what; }
void fake_rccl_counts(uint64_t* sends, uint64_t* recvs, uint64_t* bytes)
{
    *sends = nSends;
    *recvs = nRecvs;
    *bytes = nBytes;
}

} // extern "C"
