// l2z_comm.h -- shard group object shared by comm.cpp and api.cpp
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
hipError_t launch_p2p_allgather(const P2pArgs &a, hipStream_t st);
}  // namespace l2z
