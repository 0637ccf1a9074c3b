// smallk_amd/csrc/comm.h -- communicator object shared by comm.cpp and solver.cpp
#pragma once
#include "common.h"

namespace smk { struct LocalGroup; }

struct smk_comm {
    int rank = 0, world = 1;
    int device = -1;                 // single-process groups: the HIP device of this rank
    void* nccl = nullptr;            // ncclComm_t
    smk::LocalGroup* local = nullptr;
};

namespace smk {
// SMK_COMM_FORCE=1: a world of one rank issues its collectives too (tests: the real nccl* calls on the solver's streams)
bool comm_forced();
// in-place sum over ranks of `count` elements (f64 != 0: doubles, else floats), ordered on stream `st`
int comm_allreduce(smk_comm* c, void* ptr, i64 count, int f64, hipStream_t st);
// in place: rank r contributes buf[r * count_per_rank ...]; afterwards every rank holds all slices
int comm_allgather(smk_comm* c, void* buf, i64 count_per_rank, int f64, hipStream_t st);
// in place: afterwards slice `rank` (count_per_rank elements) of this rank's buffer holds the sum over ranks of that slice
int comm_reduce_scatter(smk_comm* c, void* buf, i64 count_per_rank, int f64, hipStream_t st);
// the same two with separate send and receive buffers (send: count_per_rank / world * count_per_rank elements)
int comm_allgather_to(smk_comm* c, const void* send, void* recv, i64 count_per_rank, int f64, hipStream_t st);
int comm_reduce_scatter_to(smk_comm* c, const void* send, void* recv, i64 count_per_rank, int f64, hipStream_t st);
}  // namespace smk
