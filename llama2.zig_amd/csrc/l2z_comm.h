// l2z_comm.h -- shard group object shared by comm.cpp and the host sources (l2z_state.h)
#pragma once
#include "l2z_internal.h"

constexpr int kMaxWorld = 16;
constexpr size_t kP2pFlagBytes = 4096;  // reserved head of every arena

struct l2z_comm {
    int rank;
    int world;
    int device;
    void *nccl;  // ncclComm_t, null when world == 1 / peer-write only / emulated
    // ---- peer-write all-gather (comm.cpp, p2p.hip) ----
    bool p2p = false;                 // connected: gathers go through peer stores + flags
    char *arena = nullptr;            // this rank's arena: reserved | landing slot 0 | landing slot 1
    size_t slot_floats = 0;           // 8-byte words per landing slot (>= the longest gathered vector)
    char *peer_arena[kMaxWorld] = {}; // every rank's arena mapped here ([rank] == arena)
    int *d_epoch = nullptr;           // [kMaxWorld] gathers completed with peer p (device)
    int *h_err = nullptr;             // pinned host int: a wait timed out (peer died / desync)
};

namespace l2z {
// In-place all-gather of `count_per_rank` floats per rank over buf[0 .. world*count).
// No-op for a null comm or world == 1.
int comm_allgather_inplace(const l2z_comm *c, float *buf, size_t count_per_rank, hipStream_t st);
// 0, or L2Z_ERR_COMM once a peer-write gather has timed out (checked after synchronising)
int comm_check(const l2z_comm *c);

struct P2pArgs {
    float *buf;
    size_t count;        // floats per rank
    int rank, world;
    size_t slot_floats;
    char *peer_arena[kMaxWorld];
    int *epoch;
    int *err;
    long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
};
// pushed: the producing kernel has already written this rank's words (MatvecArgs::push)
hipError_t launch_p2p_allgather(const P2pArgs &a, hipStream_t st, bool pushed = false);
bool comm_p2p_args(const l2z_comm *c, float *buf, size_t count_per_rank, P2pArgs *out);
int comm_allgather_inplace_pushed(const l2z_comm *c, float *buf, size_t count_per_rank, hipStream_t st);

#ifdef __HIPCC__
// LL word of gather `e` for element `idx` of the gathered vector, to every peer (one lane)
__device__ __forceinline__ void p2p_ll_push(const P2pArgs *a, int e, size_t idx, float v)
{
    const unsigned long long w = ((unsigned long long)(unsigned)e << 32) | (unsigned long long)__float_as_uint(v);
    const int world = a->world, rank = a->rank;
    const size_t off = (size_t)(e & 1) * a->slot_floats + idx;
    for (int p = 0; p < world; p++) {
        if (p == rank) continue;
        unsigned long long *dst = (unsigned long long *)(a->peer_arena[p] + kP2pFlagBytes) + off;
        __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// number of the gather this launch feeds (all peers are at the same count)
__device__ __forceinline__ int p2p_ll_epoch(const P2pArgs *a) { return a->epoch[a->rank == 0 ? 1 : 0] + 1; }
#endif
}  // namespace l2z
