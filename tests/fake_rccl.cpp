// tests/fake_rccl.cpp -- TEST INFRASTRUCTURE: a loopback communicator with RCCL's entry points for ranks that are THREADS of one
// process sharing ONE device.  Loaded through the engine's collectives hook (sextans_dist_bind_library / SEXTANS_RCCL_PATH,
// csrc/engine_dist.hip), it lets `pytest -m gpu` run sextans_dist_spmm / _rm / _bell with world = 2, 3, 8 on a single-GPU box:
// every byte the library moves between ranks, every cut list, row table and padding rule is exercised exactly as with RCCL; only
// the transport differs (device-to-device copies between the ranks' buffers instead of xGMI).  Not a product path, never timed.
//
// Semantics kept from RCCL (what the library relies on):
//   * collectives are STREAM-ORDERED and asynchronous to the host: a call enqueues work on `stream` and returns;
//   * every rank calls the same collectives in the same order with the same counts (checked: ncclInvalidArgument otherwise);
//   * ncclCommInitRank blocks until all ranks of the id have joined; ncclGroupStart/End defer the calls in between;
//   * in-place forms (sendbuff == recvbuff + rank * count for all-gather, sendbuff == recvbuff for broadcast).
// How a collective is built from stream primitives: every rank records a "send buffer ready" event on its stream, all ranks meet
// at a host barrier (which publishes the buffer pointers), every rank makes its stream wait for all peers' ready events, enqueues
// its device-to-device copies, records a "done" event; after a second host barrier every stream waits for all peers' done events,
// so a rank's later writes to its send buffer are ordered behind the peers' reads.  Host threads only ever block in the barriers.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
enum { kSuccess = 0, kUnhandledHipError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct Slot {
    const void *send = nullptr;
    void *recv = nullptr;
    size_t bytes = 0;
    int root = -1, kind = 0;
    hipEvent_t ready = nullptr, done = nullptr;
};
struct Group {
    int world = 0, joined = 0, refs = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::vector<Slot> slots;
    bool mismatch = false;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = generation;
        if (++waiting == world) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};
struct Comm {
    Group *g = nullptr;
    int rank = 0;
};
struct Op {
    int kind;   // 0 all-gather, 1 broadcast
    const void *send;
    void *recv;
    size_t bytes;
    int root;
    Comm *comm;
    hipStream_t stream;
};
std::mutex g_table_mutex;
std::map<std::string, Group *> g_groups;   // by unique id
uint64_t g_next_id = 1;
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_pending;

size_t dtype_bytes(int t) {   // ncclDataType_t
    switch (t) {
        case 0: case 1: return 1;            // int8, uint8
        case 2: case 3: return 4;            // int32, uint32
        case 4: case 5: return 8;            // int64, uint64
        case 6: case 9: return 2;            // half, bfloat16
        case 7: return 4;                    // float
        case 8: return 8;                    // double
        default: return 0;
    }
}

int run(const Op &op) {
    Group *g = op.comm->g;
    const int me = op.comm->rank, world = g->world;
    Slot &mine = g->slots[(size_t)me];
    mine.send = op.send; mine.recv = op.recv; mine.bytes = op.bytes; mine.root = op.root; mine.kind = op.kind;
    if (hipEventRecord(mine.ready, op.stream) != hipSuccess) return kUnhandledHipError;
    g->barrier();                                           // every rank's buffers and ready events are published
    bool bad = false;
    for (int p = 0; p < world; ++p) {
        const Slot &s = g->slots[(size_t)p];
        bad = bad || s.bytes != op.bytes || s.kind != op.kind || s.root != op.root;
    }
    int rc = kSuccess;
    if (!bad) {
        for (int p = 0; p < world && rc == kSuccess; ++p)
            if (p != me && hipStreamWaitEvent(op.stream, g->slots[(size_t)p].ready, 0) != hipSuccess) rc = kUnhandledHipError;
        if (op.kind == 0) {
            for (int p = 0; p < world && rc == kSuccess; ++p) {
                char *dst = static_cast<char *>(op.recv) + (size_t)p * op.bytes;
                const void *src = g->slots[(size_t)p].send;
                if (op.bytes && dst != src && hipMemcpyAsync(dst, src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) rc = kUnhandledHipError;
            }
        } else {
            const void *src = g->slots[(size_t)op.root].send;
            if (op.bytes && op.recv != src && hipMemcpyAsync(op.recv, src, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) rc = kUnhandledHipError;
        }
    }
    if (hipEventRecord(mine.done, op.stream) != hipSuccess) rc = kUnhandledHipError;
    g->barrier();                                           // every rank's copies are enqueued and its done event recorded
    for (int p = 0; p < world; ++p)
        if (p != me && hipStreamWaitEvent(op.stream, g->slots[(size_t)p].done, 0) != hipSuccess) rc = kUnhandledHipError;
    return bad ? kInvalidArgument : rc;
}

int submit(const Op &op) {
    if (!op.comm || !op.comm->g) return kInvalidArgument;
    if (t_group_depth > 0) { t_pending.push_back(op); return kSuccess; }
    return run(op);
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void *id) {
    if (!id) return kInvalidArgument;
    std::lock_guard<std::mutex> lk(g_table_mutex);
    memset(id, 0, 128);
    const uint64_t v = g_next_id++;
    memcpy(id, "loopback", 8);
    memcpy(static_cast<char *>(id) + 8, &v, sizeof v);
    return kSuccess;
}

struct FakeId { char b[128]; };
int ncclCommInitRank(void **comm, int world, FakeId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world || memcmp(id.b, "loopback", 8) != 0) return kInvalidArgument;
    Group *g = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_table_mutex);
        const std::string key(id.b, 128);
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group();
            g->world = world;
            g->slots.resize((size_t)world);
            g_groups[key] = g;
        } else {
            g = it->second;
        }
        if (g->world != world || g->joined >= world) return kInvalidArgument;
        ++g->joined;
        ++g->refs;
        if (g->joined == world) g_groups.erase(key);        // complete: the id cannot be joined again
    }
    Slot &s = g->slots[(size_t)rank];
    if (s.ready) return kInvalidArgument;                   // rank joined twice
    if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
        return kUnhandledHipError;
    auto *c = new Comm();
    c->g = g;
    c->rank = rank;
    g->barrier();                                           // like RCCL: returns once every rank has joined
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(void *comm) {
    auto *c = static_cast<Comm *>(comm);
    if (!c) return kInvalidArgument;
    Group *g = c->g;
    Slot &s = g->slots[(size_t)c->rank];
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(g_table_mutex);
        last = --g->refs == 0;
    }
    // (events of this rank stay alive until the whole group is gone: a peer's stream may still hold a wait on them)
    if (last) {
        (void)hipDeviceSynchronize();
        for (Slot &t : g->slots) {
            if (t.ready) (void)hipEventDestroy(t.ready);
            if (t.done) (void)hipEventDestroy(t.done);
        }
        delete g;
    }
    (void)s;
    delete c;
    return kSuccess;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
    const size_t b = dtype_bytes(dtype);
    if (!b || (count && (!send || !recv))) return kInvalidArgument;
    return submit(Op{0, send, recv, count * b, -1, static_cast<Comm *>(comm), stream});
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t stream) {
    const size_t b = dtype_bytes(dtype);
    auto *c = static_cast<Comm *>(comm);
    if (!b || !c || root < 0 || root >= c->g->world || (count && (!send || !recv))) return kInvalidArgument;
    return submit(Op{1, send, recv, count * b, root, c, stream});
}

int ncclGroupStart() { ++t_group_depth; return kSuccess; }

int ncclGroupEnd() {
    if (t_group_depth <= 0) return kInvalidUsage;
    if (--t_group_depth > 0) return kSuccess;
    int rc = kSuccess;
    std::vector<Op> ops;
    ops.swap(t_pending);
    for (const Op &op : ops) {   // (every rank issues the same list: run them all even after an error so that nobody is left in a barrier)
        const int r = run(op);
        if (rc == kSuccess) rc = r;
    }
    return rc;
}

const char *ncclGetErrorString(int rc) {
    switch (rc) {
        case kSuccess: return "no error";
        case kUnhandledHipError: return "loopback communicator: HIP call failed";
        case kInvalidArgument: return "loopback communicator: invalid argument (or ranks disagree on a collective's count / root)";
        case kInvalidUsage: return "loopback communicator: invalid usage";
        default: return "loopback communicator: error";
    }
}

}  // extern "C"
