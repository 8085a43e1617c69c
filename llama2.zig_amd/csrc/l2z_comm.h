// l2z_comm.h -- shard group object shared by comm.cpp and api.cpp
#pragma once
#include "l2z_internal.h"

struct l2z_comm {
    int rank;
    int world;
    int device;
    void *nccl;  // ncclComm_t, null when world == 1
};

namespace l2z {
// In-place all-gather of `count_per_rank` floats per rank over buf[0 .. world*count).
// No-op for a null comm or world == 1.
int comm_allgather_inplace(const l2z_comm *c, float *buf, size_t count_per_rank, hipStream_t st);
}  // namespace l2z
