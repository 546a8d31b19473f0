// mock_rccl.hip -- TEST INFRASTRUCTURE: the six RCCL entry points libsfhip.so's step groups bind (sfhip_rccl_bind), for
// several "ranks" that are THREADS of one process on one GPU.  RCCL refuses two ranks on one device, so the one-GPU
// test box can run the multi-rank step of sfhip_step_group_rk2 only with every slab sent to the own rank; this
// library lets two or more step groups with different rank numbers exchange their slabs and reduce their wavespeed
// the way ranks do -- sends and receives matched per (source, destination) pair in posting order, nothing moving
// before ncclGroupEnd, the all-reduce a rendezvous of all ranks -- so that the peers, the message names and their
// posting order, the streams and the events of the C step are exercised with real neighbours.
// Not RCCL: no claim about RCCL's own behaviour is tested here.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Message { void *staging; size_t bytes; hipEvent_t ready; };

struct World {
    int nranks = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<Message>> box; // (source, destination) -> messages in posting order
    // all-reduce rendezvous
    int arrived = 0, generation = 0;
    uint64_t value = 0, result = 0;
    long sends = 0, recvs = 0, reductions = 0;
};

struct Comm { World *world; int rank; };

struct Op { int send; void *ptr; size_t bytes; int peer; Comm *comm; hipStream_t stream; };

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
World *g_world = nullptr;
std::mutex g_world_mutex;

int do_send(const Op &op)
{
    Message msg{nullptr, op.bytes, nullptr};
    if (hipMalloc(&msg.staging, op.bytes) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&msg.ready, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipMemcpyAsync(msg.staging, op.ptr, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return 1;
    if (hipEventRecord(msg.ready, op.stream) != hipSuccess) return 1;
    World &w = *op.comm->world;
    {
        std::lock_guard<std::mutex> lock(w.m);
        w.box[{op.comm->rank, op.peer}].push_back(msg);
        w.sends += 1;
    }
    w.cv.notify_all();
    return 0;
}

int do_recv(const Op &op)
{
    World &w = *op.comm->world;
    Message msg;
    {
        std::unique_lock<std::mutex> lock(w.m);
        auto &q = w.box[{op.peer, op.comm->rank}];
        if (!w.cv.wait_for(lock, std::chrono::seconds(20), [&] { return !q.empty(); })) return 2; // no matching send
        msg = q.front();
        q.pop_front();
        w.recvs += 1;
    }
    if (msg.bytes != op.bytes) return 3; // a send and a receive of different sizes were matched: wrong posting order
    if (hipStreamWaitEvent(op.stream, msg.ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(op.ptr, msg.staging, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) return 1;
    // (the staging buffer and its event are left to the process: a few hundred KB per test)
    return 0;
}

} // namespace

extern "C" {

// test hook: a communicator handle for `rank` of `nranks` (all ranks of one process share one world)
void *mock_rccl_comm(int rank, int nranks)
{
    std::lock_guard<std::mutex> lock(g_world_mutex);
    if (!g_world || g_world->nranks != nranks) { g_world = new World(); g_world->nranks = nranks; }
    return new Comm{g_world, rank};
}

// test hook: sends, receives, all-reduces seen so far and messages still unmatched
void mock_rccl_counts(void *comm, long *out4)
{
    World &w = *((Comm *)comm)->world;
    std::lock_guard<std::mutex> lock(w.m);
    long pending = 0;
    for (auto &kv : w.box) pending += (long)kv.second.size();
    out4[0] = w.sends; out4[1] = w.recvs; out4[2] = w.reductions; out4[3] = pending;
}

int ncclGroupStart() { t_depth += 1; return 0; }

int ncclGroupEnd()
{
    if (--t_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op &op : ops) if (op.send) { int r = do_send(op); if (r) return r; }   // sends never block ...
    for (const Op &op : ops) if (!op.send) { int r = do_recv(op); if (r) return r; }  // ... so receives cannot deadlock
    return 0;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (dtype != 8) return 4; // ncclFloat64
    Op op{1, (void *)buf, count * 8, peer, (Comm *)comm, stream};
    if (t_depth > 0) { t_ops.push_back(op); return 0; }
    return do_send(op);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (dtype != 8) return 4;
    Op op{0, buf, count * 8, peer, (Comm *)comm, stream};
    if (t_depth > 0) { t_ops.push_back(op); return 0; }
    return do_recv(op);
}

// one uint64, MAX: what the step group asks for.  A rendezvous of all ranks (blocks the host, which a test can afford).
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream)
{
    if (count != 1 || dtype != 5 /* ncclUint64 */ || op != 2 /* ncclMax */) return 4;
    World &w = *((Comm *)comm)->world;
    uint64_t mine = 0;
    if (hipMemcpyAsync(&mine, send, 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    uint64_t result;
    {
        std::unique_lock<std::mutex> lock(w.m);
        const int gen = w.generation;
        w.value = w.arrived == 0 ? mine : (mine > w.value ? mine : w.value);
        if (++w.arrived == w.nranks) {
            w.result = w.value; w.arrived = 0; w.generation += 1; w.reductions += 1;
            w.cv.notify_all();
        } else if (!w.cv.wait_for(lock, std::chrono::seconds(20), [&] { return w.generation != gen; })) {
            return 2;
        }
        result = w.result;
    }
    if (hipMemcpyAsync(recv, &result, 8, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1;
}

const char *ncclGetErrorString(int status)
{
    switch (status) {
    case 0: return "ok";
    case 1: return "mock rccl: a HIP call failed";
    case 2: return "mock rccl: no matching operation from the peer within 20 s";
    case 3: return "mock rccl: matched a send and a receive of different sizes (posting order)";
    default: return "mock rccl: unsupported datatype, count or operation";
    }
}

} // extern "C"
