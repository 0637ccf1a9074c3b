// smallk_amd/csrc/comm.cpp -- collectives of the column-sharded solver, issued from C on the
// solver's own streams: RCCL (ncclAllReduce / ncclAllGather over xGMI) or, for several shards on ONE device
// (tests, a box with fewer GPUs than shards), an in-process stand-in with the same semantics.
//
// The reference has no distributed code; north_star fixes the exchange steps: sum-all-reduce of HH' (k x k), the sum of
// (AH')' (k x m) per iteration in row chunks (all-reduce for HALS; reduce-scatter for BPP / MU, whose ranks each solve their
// own row blocks of W), an all-reduce of W'W, an all-gather per chunk of the packed streaming operand of those blocks
// (the fp64 rows only when results are read), and one 3-element all-reduce when the stopping rule is evaluated.
#include "common.h"
#include "comm.h"
#include "../../include/smallk_amd.h"

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace smk {

// ---- in-process stand-in: `world` ranks = `world` host threads, all buffers reachable from rank 0's device
// (same device, or peers with access enabled).  Rank 0 sums in fixed rank order and writes the result back to
// every rank, so all ranks hold identical bits, as after a real all-reduce.
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    int err = 0;                          // sticky: a rank that failed (or a wait that timed out) fails every later collective of the group
    std::vector<void*> ptr, ptr2;          // send / receive buffer of every rank for the collective in flight
    std::vector<hipStream_t> stream;
    int refs = 0;
    // all ranks meet here; `my_err` is published before the meeting so that every rank leaves with the same verdict.
    // A rank that never arrives (it returned early on an error of its own) does not strand the others for ever.
    int barrier(int my_err)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (my_err && !err) err = my_err;
        if (err) {                                // a failed group fails fast: nobody waits for a rank that has left
            arrived = 0; ++generation; cv.notify_all();
            return err;
        }
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::seconds(300), [&] { return generation != gen; })) {
            if (!err) err = SMK_DEVICE_ERROR;
            arrived = 0; ++generation; cv.notify_all();       // release whoever else is waiting
        }
        return err;
    }
};

int launch_local_allreduce(void* const* ptrs, int world, i64 count, int f64, hipStream_t st);               // kernels.hip
int launch_local_allgather(void* const* sends, void* const* recvs, int world, i64 count_per_rank, int f64, hipStream_t st);
int launch_local_reduce_scatter(void* const* sends, void* const* recvs, int world, i64 count_per_rank, int f64, hipStream_t st);

enum { LOCAL_ALLREDUCE = 0, LOCAL_ALLGATHER = 1, LOCAL_REDUCE_SCATTER = 2 };
static int local_collective(smk_comm* c, const void* send, void* recv, i64 count, int f64, hipStream_t st, int op)
{
    LocalGroup* g = c->local;
    int mine = 0;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("local communicator: hipStreamSynchronize failed"); mine = SMK_DEVICE_ERROR; }   // this rank's contribution is complete
    g->ptr[c->rank] = const_cast<void*>(send);
    g->ptr2[c->rank] = recv;
    g->stream[c->rank] = st;
    int rc = g->barrier(mine);
    if (c->rank == 0 && !rc) {
        rc = op == LOCAL_ALLGATHER        ? launch_local_allgather(g->ptr.data(), g->ptr2.data(), g->world, count, f64, st)
             : op == LOCAL_REDUCE_SCATTER ? launch_local_reduce_scatter(g->ptr.data(), g->ptr2.data(), g->world, count, f64, st)
                                          : launch_local_allreduce(g->ptr2.data(), g->world, count, f64, st);
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    }
    rc = g->barrier(rc);                        // results are in every rank's buffer -- or every rank learns that they are not
    if (rc) set_error("local communicator: a rank failed inside a collective");
    return rc;
}

bool comm_forced()
{
    const char* e = getenv("SMK_COMM_FORCE");       // read per call: tests switch it inside one process
    return e && atoi(e) != 0;
}
static inline bool skip_collective(const smk_comm* c) { return !c || (c->world == 1 && !comm_forced()); }

int comm_allreduce(smk_comm* c, void* ptr, i64 count, int f64, hipStream_t st)
{
    if (skip_collective(c) || count <= 0) return 0;
    if (c->local) return local_collective(c, ptr, ptr, count, f64, st, LOCAL_ALLREDUCE);
    const ncclResult_t r = ncclAllReduce(ptr, ptr, (size_t)count, f64 ? ncclDouble : ncclFloat, ncclSum, (ncclComm_t)c->nccl, st);
    if (r != ncclSuccess) { set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    return 0;
}

// every rank contributes `count_per_rank` elements at `send`; afterwards recv holds all contributions in rank order
// (send == recv + rank * count_per_rank: in place)
int comm_allgather_to(smk_comm* c, const void* send, void* recv, i64 count_per_rank, int f64, hipStream_t st)
{
    const size_t es = f64 ? 8 : 4;
    if (count_per_rank <= 0 || !c) return 0;
    if (skip_collective(c)) {        // one rank: the gather is a copy (or nothing, in place)
        if (send != recv) SMK_HIP(hipMemcpyAsync(recv, send, (size_t)count_per_rank * es, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (c->local) return local_collective(c, send, recv, count_per_rank, f64, st, LOCAL_ALLGATHER);
    const ncclResult_t r = ncclAllGather(send, recv, (size_t)count_per_rank, f64 ? ncclDouble : ncclFloat, (ncclComm_t)c->nccl, st);
    if (r != ncclSuccess) { set_error(std::string("ncclAllGather: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    return 0;
}
int comm_allgather(smk_comm* c, void* buf, i64 count_per_rank, int f64, hipStream_t st)
{
    if (!c) return 0;
    return comm_allgather_to(c, (const char*)buf + (size_t)c->rank * count_per_rank * (f64 ? 8 : 4), buf, count_per_rank, f64, st);
}

// every rank holds world * count_per_rank elements at `send`; afterwards recv (count_per_rank elements) is the sum over
// ranks of their slice `rank` (recv == send + rank * count_per_rank: in place)
int comm_reduce_scatter_to(smk_comm* c, const void* send, void* recv, i64 count_per_rank, int f64, hipStream_t st)
{
    const size_t es = f64 ? 8 : 4;
    if (count_per_rank <= 0 || !c) return 0;
    if (skip_collective(c)) {
        if (send != recv) SMK_HIP(hipMemcpyAsync(recv, send, (size_t)count_per_rank * es, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (c->local) return local_collective(c, send, recv, count_per_rank, f64, st, LOCAL_REDUCE_SCATTER);
    const ncclResult_t r = ncclReduceScatter(send, recv, (size_t)count_per_rank, f64 ? ncclDouble : ncclFloat, ncclSum, (ncclComm_t)c->nccl, st);
    if (r != ncclSuccess) { set_error(std::string("ncclReduceScatter: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    return 0;
}
int comm_reduce_scatter(smk_comm* c, void* buf, i64 count_per_rank, int f64, hipStream_t st)
{
    if (!c) return 0;
    return comm_reduce_scatter_to(c, buf, (char*)buf + (size_t)c->rank * count_per_rank * (f64 ? 8 : 4), count_per_rank, f64, st);
}

}  // namespace smk

using namespace smk;

extern "C" {

int smk_comm_unique_id(void* id128)
{
    if (!id128) return SMK_BAD_PARAM;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { set_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    std::memcpy(id128, &id, sizeof(id));
    return SMK_OK;
}

int smk_comm_init_rank(smk_comm** out, const void* id128, int rank, int world)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return SMK_BAD_PARAM;
    *out = nullptr;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t nc = nullptr;
    const ncclResult_t r = ncclCommInitRank(&nc, world, id, rank);
    if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    smk_comm* c = new smk_comm;
    c->rank = rank; c->world = world; c->nccl = nc;
    *out = c;
    return SMK_OK;
}

int smk_comm_init_all(smk_comm** out, int ndev, const int* devices)
{
    if (!out || ndev < 1) return SMK_BAD_PARAM;
    std::vector<ncclComm_t> nc((size_t)ndev, nullptr);
    std::vector<int> devs((size_t)ndev);
    for (int i = 0; i < ndev; ++i) devs[(size_t)i] = devices ? devices[i] : i;
    const ncclResult_t r = ncclCommInitAll(nc.data(), ndev, devs.data());
    if (r != ncclSuccess) { set_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    for (int i = 0; i < ndev; ++i) {
        smk_comm* c = new smk_comm;
        c->rank = i; c->world = ndev; c->nccl = nc[(size_t)i]; c->device = devs[(size_t)i];
        out[i] = c;
    }
    return SMK_OK;
}

int smk_comm_init_local(smk_comm** out, int nranks)
{
    if (!out || nranks < 1) return SMK_BAD_PARAM;
    LocalGroup* g = new LocalGroup;
    g->world = nranks;
    g->ptr.assign((size_t)nranks, nullptr);
    g->ptr2.assign((size_t)nranks, nullptr);
    g->stream.assign((size_t)nranks, nullptr);
    g->refs = nranks;
    for (int i = 0; i < nranks; ++i) {
        smk_comm* c = new smk_comm;
        c->rank = i; c->world = nranks; c->local = g;
        out[i] = c;
    }
    return SMK_OK;
}

// One sum-all-reduce (fp64 and fp32) and one all-gather of known values on a private stream of the current device,
// checked on the host.  Every rank of the communicator must call it.  SMK_OK, or SMK_DEVICE_ERROR with the reason in
// smk_last_error(): callers (bench.py) use it to decide whether the native path can be trusted on this node.
int smk_comm_selftest(smk_comm* c)
{
    if (!c) return SMK_BAD_PARAM;
    const int W = c->world;
    hipStream_t st = nullptr;
    double* d = nullptr;
    std::vector<double> h((size_t)(2 * W + 2), 0.0);      // [0] sum f64, [1] 2 x f32, [2, 2+W) gather, [2+W, 2+2W) reduce-scatter
    int rc = SMK_OK;
    auto fail = [&](const std::string& m) { set_error("communicator self-test: " + m); rc = SMK_DEVICE_ERROR; };
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { fail("hipStreamCreate"); return rc; }
    if (smk::dev_malloc((void**)&d, h.size() * sizeof(double)) != hipSuccess) { (void)hipStreamDestroy(st); fail("hipMalloc"); return rc; }
    // layout: [0] fp64 sum slot, [1] two fp32 sum slots, [2 .. 2 + W) gather slots (fp64)
    h[0] = (double)(c->rank + 1);
    float f2[2] = {(float)(c->rank + 1), 0.5f};
    std::memcpy(&h[1], f2, sizeof(f2));
    h[(size_t)(2 + c->rank)] = 100.0 + c->rank;
    for (int r = 0; r < W; ++r) h[(size_t)(2 + W + r)] = (double)((c->rank + 1) * (r + 1));     // slot r sums to (r+1) W (W+1) / 2
    if (hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) fail("upload");
    if (!rc && c->nccl) {          // also with one rank: the calls must at least go through the library
        ncclComm_t nc = (ncclComm_t)c->nccl;
        ncclResult_t r = ncclAllReduce(d, d, 1, ncclDouble, ncclSum, nc, st);
        if (r == ncclSuccess) r = ncclAllReduce(d + 1, d + 1, 2, ncclFloat, ncclSum, nc, st);
        if (r == ncclSuccess) r = ncclAllGather(d + 2 + c->rank, d + 2, 1, ncclDouble, nc, st);
        if (r == ncclSuccess) r = ncclReduceScatter(d + 2 + W, d + 2 + W + c->rank, 1, ncclDouble, ncclSum, nc, st);
        if (r != ncclSuccess) fail(std::string("RCCL: ") + ncclGetErrorString(r));
    } else if (!rc) {
        if (comm_allreduce(c, d, 1, 1, st) || comm_allreduce(c, d + 1, 2, 0, st) || comm_allgather(c, d + 2, 1, 1, st) ||
            comm_reduce_scatter(c, d + 2 + W, 1, 1, st))
            rc = SMK_DEVICE_ERROR;
    }
    if (!rc && hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) fail("download");
    if (!rc && hipStreamSynchronize(st) != hipSuccess) fail("synchronize");
    if (!rc) {
        std::memcpy(f2, &h[1], sizeof(f2));
        const double want = 0.5 * W * (W + 1);
        if (h[0] != want) fail("fp64 all-reduce returned " + std::to_string(h[0]) + ", expected " + std::to_string(want));
        else if (f2[0] != (float)want || f2[1] != 0.5f * W) fail("fp32 all-reduce returned wrong sums");
        else
            for (int r = 0; r < W; ++r)
                if (h[(size_t)(2 + r)] != 100.0 + r) { fail("all-gather slot " + std::to_string(r) + " is wrong"); break; }
        if (!rc && h[(size_t)(2 + W + c->rank)] != (c->rank + 1) * want) fail("reduce-scatter returned a wrong sum");
    }
    (void)smk::dev_free(d);
    (void)hipStreamDestroy(st);
    return rc;
}

// A rank that leaves its iteration loop with an error calls this so that peers blocked in a collective are released
// (ncclCommAbort for RCCL; the stand-in marks the group failed and wakes every waiter).
void smk_comm_abort(smk_comm* c)
{
    if (!c) return;
    if (c->nccl) { (void)ncclCommAbort((ncclComm_t)c->nccl); c->nccl = nullptr; }
    if (c->local) {
        std::lock_guard<std::mutex> lk(c->local->mu);
        if (!c->local->err) c->local->err = SMK_DEVICE_ERROR;
        c->local->arrived = 0; ++c->local->generation;
        c->local->cv.notify_all();
    }
}

int smk_comm_rank(const smk_comm* c) { return c ? c->rank : 0; }
int smk_comm_world(const smk_comm* c) { return c ? c->world : 1; }

void smk_comm_destroy(smk_comm* c)
{
    if (!c) return;
    if (c->nccl) (void)ncclCommDestroy((ncclComm_t)c->nccl);
    if (c->local) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->local->mu); last = (--c->local->refs == 0); }
        if (last) delete c->local;
    }
    delete c;
}

}  // extern "C"
