// smallk_amd/csrc/comm.cpp -- collectives of the column-sharded solver, issued from C on the
// solver's own streams: RCCL (ncclAllReduce / ncclAllGather over xGMI) or, for several shards on ONE device
// (tests, a box with fewer GPUs than shards), an in-process stand-in with the same semantics.
//
// The reference has no distributed code; north_star fixes the exchange steps: sum-all-reduce of HH' (k x k) and
// of (AH')' (k x m) per iteration, one scalar when the stopping rule is evaluated, and -- BPP only -- an
// all-gather of the row slices of W each rank solved.
#include "common.h"
#include "comm.h"
#include "../../include/smallk_amd.h"

#include <rccl/rccl.h>

#include <condition_variable>
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
    std::vector<void*> ptr;
    std::vector<hipStream_t> stream;
    int refs = 0;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const long gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

int launch_local_allreduce(void* const* ptrs, int world, i64 count, int f64, hipStream_t st);               // kernels.hip
int launch_local_allgather(void* const* ptrs, int world, i64 count_per_rank, int f64, hipStream_t st);

static int local_collective(smk_comm* c, void* ptr, i64 count, int f64, hipStream_t st, bool gather)
{
    LocalGroup* g = c->local;
    SMK_HIP(hipStreamSynchronize(st));          // this rank's contribution is complete
    g->ptr[c->rank] = ptr;
    g->stream[c->rank] = st;
    g->barrier();
    int rc = 0;
    if (c->rank == 0) {
        rc = gather ? launch_local_allgather(g->ptr.data(), g->world, count, f64, st)
                    : launch_local_allreduce(g->ptr.data(), g->world, count, f64, st);
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = SMK_DEVICE_ERROR;
    }
    g->barrier();                               // results are in every rank's buffer
    return rc;
}

int comm_allreduce(smk_comm* c, void* ptr, i64 count, int f64, hipStream_t st)
{
    if (!c || c->world == 1) return 0;
    if (c->local) return local_collective(c, ptr, count, f64, st, false);
    const ncclResult_t r = ncclAllReduce(ptr, ptr, (size_t)count, f64 ? ncclDouble : ncclFloat, ncclSum, (ncclComm_t)c->nccl, st);
    if (r != ncclSuccess) { set_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    return 0;
}

// every rank contributes `count_per_rank` elements at buf + rank * count_per_rank (in place)
int comm_allgather(smk_comm* c, void* buf, i64 count_per_rank, int f64, hipStream_t st)
{
    if (!c || c->world == 1) return 0;
    if (c->local) return local_collective(c, buf, count_per_rank, f64, st, true);
    const size_t es = f64 ? 8 : 4;
    const ncclResult_t r = ncclAllGather((const char*)buf + (size_t)c->rank * count_per_rank * es, buf, (size_t)count_per_rank,
                                         f64 ? ncclDouble : ncclFloat, (ncclComm_t)c->nccl, st);
    if (r != ncclSuccess) { set_error(std::string("ncclAllGather: ") + ncclGetErrorString(r)); return SMK_DEVICE_ERROR; }
    return 0;
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
    g->stream.assign((size_t)nranks, nullptr);
    g->refs = nranks;
    for (int i = 0; i < nranks; ++i) {
        smk_comm* c = new smk_comm;
        c->rank = i; c->world = nranks; c->local = g;
        out[i] = c;
    }
    return SMK_OK;
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
