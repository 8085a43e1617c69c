// l2z_comm.h -- shard group object shared by comm.cpp and the host sources (l2z_state.h), and the
// device side of the peer-write ("LL") all-gather: protocol in p2p.hip.
#pragma once
#include "l2z_internal.h"

constexpr int kMaxWorld = 16;
constexpr size_t kP2pFlagBytes = 4096;  // reserved head of every arena: bulk flags [kMaxWorld] u64 at 0, probe words [kMaxWorld] u64 at kP2pProbeOff
constexpr size_t kP2pProbeOff = 2048;   // l2z_comm_p2p_pingpong's words (one per sender)

// ints of l2z_comm::d_ctl (device memory, shared by every runstate of the group)
constexpr int kCtlEpoch = 0;  // epochs used up by completed passes: gather gi of the running pass is ctl[0] + gi
constexpr int kCtlDone = 1;   // blocks of the pass-closing gather that have finished
constexpr int kCtlErr = 2;    // != 0 once a wait has timed out: later waits give up at once
constexpr int kCtlBulkDone = 3;  // blocks of the running bulk push that have finished
constexpr int kCtlInts = 8;

struct l2z_comm {
    int rank;
    int world;
    int device;
    void *nccl;  // ncclComm_t, null when world == 1 / peer-write only / emulated
    // ---- peer-write all-gather (comm.cpp, p2p.hip) ----
    bool p2p = false;                 // connected: gathers go through peer stores of LL words
    char *arena = nullptr;            // this rank's arena: reserved | landing slot 0 | landing slot 1
    size_t slot_floats = 0;           // 8-byte words per landing slot (>= the longest gathered vector)
    char *peer_arena[kMaxWorld] = {}; // every rank's arena mapped here ([rank] == arena)
    int *d_ctl = nullptr;             // [kCtlInts] epoch counter, arrival counter, error latch (device)
    int *h_err = nullptr;             // pinned host int: a wait timed out (peer died / desync)
    // bulk landing regions behind the two LL slots (sharded prefill: [P, n] activation blocks travel
    // as plain 16-byte stores + one flag per sender in the arena's reserved head); 0: none
    size_t bulk_floats = 0;
    mutable unsigned long long bulk_epoch = 0;  // bulk gathers issued (every rank issues the same sequence)
    // SOLO (measurement: l2z_comm_p2p_connect_solo): one rank of an N-rank group alone on a GPU, every "peer" arena
    // mapped to a local sink (peer stores hit distinct local addresses instead of N devices).  Every hand-over of every pass gets index 0, so its epoch is the counter's initial value and
    // the zeroed landing slots already "carry" it: no wait ever blocks, the peers' slices read as 0.0.  The rank's
    // launches, pushes, polls and gather / reduce launches all run -- the per-rank time of a sharded pass with free
    // hand-overs (bench.py extra.scaling_model); the results mean nothing.
    bool solo = false;
    char *solo_sink = nullptr;        // solo: where the pushes "to the peers" land (one slot pair per peer, distinct addresses)
    unsigned long long probe_seq[kMaxWorld] = {};  // round trips exchanged with each rank so far (l2z_comm_p2p_pingpong)
};
// the index a hand-over is given on this group (solo: always 0)
inline int comm_gi(const l2z_comm *c, int gi) { return c != nullptr && c->solo ? 0 : gi; }

namespace l2z {
// In-place all-gather of `count_per_rank` floats per rank over buf[0 .. world*count) as its own
// launch (RCCL, or the peer-write gather kernel).  gi = 1-based index of this gather in the
// forward pass, n_gathers = gathers per pass; the peer-write form derives its epoch from them
// and the pass-closing gather (gi == n_gathers) advances the group's epoch counter.
// No-op for a null comm or world == 1.
int comm_allgather_inplace(const l2z_comm *c, float *buf, size_t count_per_rank, int gi, int n_gathers,
                           bool pushed, hipStream_t st);
// Scheme B: out[i] = the sum over ranks, in rank order, of every rank's part[i] (count floats each), as its own launch:
// the peer-write reduce kernel (p2p.hip; pushed: the producing mat-vec already stored this rank's words), or
// ncclAllReduce (sum order RCCL's; every rank receives the same result).  gi as for the gathers.
int comm_allreduce(const l2z_comm *c, const float *part, float *out, size_t count, int gi, bool pushed, hipStream_t st);
// 0, or L2Z_ERR_COMM once a peer-write wait has timed out (checked after synchronising)
int comm_check(const l2z_comm *c);

// device-side description of one gathered vector (lives in device memory: l2z_runstate::d_push)
struct P2pArgs {
    float *buf;
    size_t count;        // floats per rank
    int rank, world;
    size_t slot_floats;
    char *peer_arena[kMaxWorld];
    int *ctl;            // l2z_comm::d_ctl
    int *err;            // l2z_comm::h_err
    long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
    int self;            // 1: producers also write this rank's own landing slot (consumers read LL words)
};

// pushed: the producing kernel has already written this rank's words (MatvecArgs::push)
hipError_t launch_p2p_allgather(const P2pArgs &a, int gi, int n_gathers, bool pushed, hipStream_t st);
hipError_t launch_p2p_pingpong(char *mine, char *theirs, int me, int other, bool initiator, int iters,
                               unsigned long long base, long long timeout_ticks, long long *ticks_out, int *err,
                               hipStream_t st);
hipError_t launch_p2p_allreduce(const P2pArgs &a, float *out, int gi, bool pushed, hipStream_t st);
bool comm_p2p_args(const l2z_comm *c, float *buf, size_t count_per_rank, bool self, P2pArgs *out);
LLIn comm_ll_in(const l2z_comm *c, int gi, size_t count_per_rank);
// the peer-write transport is connected and not overridden by L2Z_COMM=rccl
bool comm_uses_p2p(const l2z_comm *c);

// ---- bulk all-gather of activation blocks (sharded prefill) ----
// Every rank holds its [P, n_loc] block (contiguous) at stage + rank * P * n_loc; afterwards
// dst[t * ldd + p * n_loc + j] = rank p's block[t][j] for every p: the row-major [P, world * n_loc]
// matrix the next GEMM reads.  Peer-write transport: a push launch stores the block into every peer's
// bulk region and then one flag per peer; the unpack launch waits for each sender's flag.  RCCL: an
// in-place ncclAllGather over the staging buffer, then the same unpack without waits.
int comm_bulk_allgather(const l2z_comm *c, float *stage, int P, int n_loc, float *dst, int ldd, hipStream_t st);
// Scheme B's batched prefill: dst[P, n] (ldd floats per row) = the sum over ranks, in rank order, of every rank's partial
// [P, n] (contiguous rows of n).  Peer-write transport: reduce-scatter through the bulk regions (launch_bulk_scatter_push,
// launch_bulk_reduce into stage + rank * P * n / world), then comm_bulk_allgather of the summed slices; RCCL: ncclAllReduce
// (ldd == n).  Every rank ends with the same bits.
int comm_bulk_allreduce(const l2z_comm *c, float *part, int P, int n, float *stage, float *dst, int ldd, hipStream_t st);
// whether comm_bulk_allgather can carry P x n_total floats (a transport is there and, peer-write, the
// bulk regions are large enough)
bool comm_bulk_ok(const l2z_comm *c, size_t floats);

struct BulkArgs {
    float *stage;        // [world][P * n_loc]
    int P, n_loc;
    int rank, world;
    char *peer_arena[kMaxWorld];
    size_t bulk_off;     // bytes from the arena base to bulk region 0
    size_t bulk_floats;  // floats per region
    int *ctl, *err;
    long long timeout_ticks;
};
hipError_t launch_bulk_push(const BulkArgs &a, unsigned long long e, hipStream_t st);
// scheme B's bulk all-reduce, first half (p2p.hip): a.stage = this rank's partial [P, world * n_loc]; every peer gets the
// columns of ITS slice; then the owner's sum over ranks, in rank order, into out [P, n_loc]
hipError_t launch_bulk_scatter_push(const BulkArgs &a, unsigned long long e, hipStream_t st);
hipError_t launch_bulk_reduce(const BulkArgs &a, unsigned long long e, float *out, hipStream_t st);
// wait != 0: blocks of peers' data come from this rank's bulk region (e & 1) once the sender's flag
// says e; else every block is read from the staging buffer
hipError_t launch_bulk_unpack(const BulkArgs &a, unsigned long long e, int wait, float *dst, int ldd,
                              hipStream_t st);

#ifdef __HIPCC__
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// LL word of gather epoch `e` for element `idx` of the gathered vector, to every peer (one lane);
// with a->self also into this rank's own slot
__device__ __forceinline__ void p2p_ll_push(const P2pArgs *a, int e, size_t idx, float v)
{
    const unsigned long long w = ((unsigned long long)(unsigned)e << 32) | (unsigned long long)__float_as_uint(v);
    const int world = a->world, rank = a->rank, self = a->self;
    const size_t off = (size_t)(e & 1) * a->slot_floats + idx;
    for (int p = 0; p < world; p++) {
        if (p == rank && !self) continue;
        unsigned long long *dst = (unsigned long long *)(a->peer_arena[p] + kP2pFlagBytes) + off;
        __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// epoch of gather `gi` of the pass that is running (every rank counts the same gathers)
__device__ __forceinline__ int p2p_ll_epoch(const P2pArgs *a, int gi) { return a->ctl[kCtlEpoch] + gi; }

// Two adjacent LL words (16 bytes) in one system-scope load.  Each 8-byte half validates itself
// through its own epoch, so nothing is assumed about the two halves having been written together.
__device__ __forceinline__ v4u ll_load2(const unsigned long long *slot_base, size_t word_idx)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned long long *>(slot_base), 0, 0x7ffffff0, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(word_idx * 8), 0, 17 /* sc0 sc1 */);
}
__device__ __forceinline__ bool ll_ready2(v4u w, unsigned e) { return w.y == e && w.w == e; }
#endif
}  // namespace l2z
