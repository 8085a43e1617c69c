// comm.cpp -- multi-GPU shard group: one process per GPU, RCCL over xGMI.
//
// The reference (cgbur/llama2.zig) is single-threaded and single-device; this
// is what the build adds (DESIGN.md "Sharding").  RCCL is bound
// with dlopen at l2z_comm_init time, so the single-GPU library has no RCCL
// dependency at all and loads on machines without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <chrono>
#include <cstring>

#include "l2z_comm.h"
#include "tunables.h"

namespace l2z {

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

RcclApi g_api;

int load_rccl()
{
    if (g_api.handle) return L2Z_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    L2Z_CHECK(h != nullptr, L2Z_ERR_COMM, "cannot dlopen librccl.so.1: %s", dlerror());
#define SYM(field, name)                                                             \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, name));           \
    L2Z_CHECK(g_api.field != nullptr, L2Z_ERR_COMM, "librccl: missing symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_api.handle = h;
    return L2Z_OK;
}
}  // namespace

int comm_check(const l2z_comm *c)
{
    if (c && c->h_err && *(volatile int *)c->h_err != 0) {
        set_error("peer-write all-gather timed out waiting for rank %d (peer gone or call sequences differ)",
                  *(volatile int *)c->h_err - 1);
        return L2Z_ERR_COMM;
    }
    return L2Z_OK;
}

bool comm_p2p_args(const l2z_comm *c, float *buf, size_t count_per_rank, bool self, P2pArgs *out)
{
    if (c == nullptr || !c->p2p || c->world <= 1) return false;
    P2pArgs a = {};
    a.buf = buf; a.count = count_per_rank; a.rank = c->rank; a.world = c->world;
    a.slot_floats = c->slot_floats;
    for (int r = 0; r < c->world; r++) a.peer_arena[r] = c->peer_arena[r];
    a.ctl = c->d_ctl; a.err = c->h_err;
    a.timeout_ticks = tunables().p2p_timeout_s * 100000000LL;
    a.self = self ? 1 : 0;
    *out = a;
    return true;
}

// where a consumer finds gathered vector `gi` as LL words: this rank's own landing slots
LLIn comm_ll_in(const l2z_comm *c, int gi, size_t count_per_rank)
{
    LLIn in = {};
    in.slots = reinterpret_cast<const unsigned long long *>(c->arena + kP2pFlagBytes);
    in.slot_floats = (unsigned)c->slot_floats;
    in.count = (unsigned)count_per_rank;
    in.gi = comm_gi(c, gi);
    in.ctl = c->d_ctl;
    in.h_err = c->h_err;
    in.timeout_ticks = tunables().p2p_timeout_s * 100000000LL;
    return in;
}

bool comm_uses_p2p(const l2z_comm *c)
{
    return c != nullptr && c->p2p && c->world > 1 && !(tunables().prefer_rccl && c->nccl);
}

int comm_allgather_inplace(const l2z_comm *c, float *buf, size_t count_per_rank, int gi, int n_gathers,
                           bool pushed, hipStream_t st)
{
    if (comm_uses_p2p(c)) {
        L2Z_CHECK(count_per_rank * (size_t)c->world <= c->slot_floats, L2Z_ERR_COMM,
                  "peer-write gather of %zu floats exceeds the landing slot (%zu)",
                  count_per_rank * (size_t)c->world, c->slot_floats);
        P2pArgs a = {};
        comm_p2p_args(c, buf, count_per_rank, false, &a);
        // (solo: index 0 and no pass-closing advance of the epoch counter)
        hipError_t e = c->solo ? launch_p2p_allgather(a, 0, -1, pushed, st) : launch_p2p_allgather(a, gi, n_gathers, pushed, st);
        L2Z_CHECK(e == hipSuccess, L2Z_ERR_HIP, "peer-write gather launch failed: %s", hipGetErrorString(e));
        return L2Z_OK;
    }
    L2Z_CHECK(!pushed, L2Z_ERR_STATE, "pushed gather without the peer-write transport");
    if (c == nullptr || c->nccl == nullptr) return L2Z_OK;  // single GPU without a communicator
    // in-place form: sendbuff == recvbuff + rank * count
    ncclResult_t r = g_api.AllGather(buf + (size_t)c->rank * count_per_rank, buf, count_per_rank,
                                     ncclFloat, static_cast<ncclComm_t>(c->nccl), st);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclAllGather failed: %s", g_api.GetErrorString(r));
    return L2Z_OK;
}

int comm_allreduce(const l2z_comm *c, const float *part, float *out, size_t count, int gi, bool pushed, hipStream_t st)
{
    if (comm_uses_p2p(c)) {
        L2Z_CHECK(count * (size_t)c->world <= c->slot_floats, L2Z_ERR_COMM,
                  "peer-write all-reduce of %d x %zu floats exceeds the landing slot (%zu)", c->world, count, c->slot_floats);
        P2pArgs a = {};
        comm_p2p_args(c, const_cast<float *>(part), count, false, &a);
        hipError_t e = launch_p2p_allreduce(a, out, comm_gi(c, gi), pushed, st);
        L2Z_CHECK(e == hipSuccess, L2Z_ERR_HIP, "peer-write all-reduce launch failed: %s", hipGetErrorString(e));
        return L2Z_OK;
    }
    L2Z_CHECK(!pushed, L2Z_ERR_STATE, "pushed all-reduce without the peer-write transport");
    L2Z_CHECK(c != nullptr && c->nccl != nullptr, L2Z_ERR_COMM, "all-reduce: the shard group has no transport");
    ncclResult_t r = g_api.AllReduce(part, out, count, ncclFloat, ncclSum, static_cast<ncclComm_t>(c->nccl), st);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclAllReduce failed: %s", g_api.GetErrorString(r));
    return L2Z_OK;
}

bool comm_bulk_ok(const l2z_comm *c, size_t floats)
{
    if (c == nullptr || c->world <= 1) return false;
    if (comm_uses_p2p(c)) return floats <= c->bulk_floats;
    return c->nccl != nullptr;
}

int comm_bulk_allgather(const l2z_comm *c, float *stage, int P, int n_loc, float *dst, int ldd, hipStream_t st)
{
    L2Z_CHECK(c != nullptr && c->world > 1, L2Z_ERR_STATE, "bulk all-gather without a shard group");
    const size_t count = (size_t)P * (size_t)n_loc;
    BulkArgs a = {};
    a.stage = stage; a.P = P; a.n_loc = n_loc; a.rank = c->rank; a.world = c->world;
    if (comm_uses_p2p(c)) {
        L2Z_CHECK(count * (size_t)c->world <= c->bulk_floats, L2Z_ERR_COMM,
                  "bulk gather of %zu floats exceeds the bulk landing region (%zu)", count * (size_t)c->world,
                  c->bulk_floats);
        for (int r = 0; r < c->world; r++) a.peer_arena[r] = c->peer_arena[r];
        a.bulk_off = kP2pFlagBytes + 2 * c->slot_floats * 8;
        a.bulk_floats = c->bulk_floats;
        a.ctl = c->d_ctl; a.err = c->h_err;
        a.timeout_ticks = tunables().p2p_timeout_s * 100000000LL;
        const unsigned long long e = ++c->bulk_epoch;
        hipError_t he = launch_bulk_push(a, e, st);
        if (he == hipSuccess) he = launch_bulk_unpack(a, e, 1, dst, ldd, st);
        L2Z_CHECK(he == hipSuccess, L2Z_ERR_HIP, "bulk gather launch failed: %s", hipGetErrorString(he));
        return L2Z_OK;
    }
    L2Z_CHECK(c->nccl != nullptr, L2Z_ERR_COMM, "bulk all-gather: the shard group has no transport");
    ncclResult_t r = g_api.AllGather(stage + (size_t)c->rank * count, stage, count, ncclFloat,
                                     static_cast<ncclComm_t>(c->nccl), st);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclAllGather failed: %s", g_api.GetErrorString(r));
    hipError_t he = launch_bulk_unpack(a, 0, 0, dst, ldd, st);
    L2Z_CHECK(he == hipSuccess, L2Z_ERR_HIP, "bulk unpack launch failed: %s", hipGetErrorString(he));
    return L2Z_OK;
}

int comm_bulk_allreduce(const l2z_comm *c, float *part, int P, int n, float *stage, float *dst, int ldd, hipStream_t st)
{
    L2Z_CHECK(c != nullptr, L2Z_ERR_STATE, "bulk all-reduce without a shard group");
    L2Z_CHECK(n % c->world == 0, L2Z_ERR_INVALID, "bulk all-reduce: %d columns over %d ranks", n, c->world);
    const int n_loc = n / c->world;
    if (comm_uses_p2p(c)) {
        L2Z_CHECK((size_t)P * (size_t)n <= c->bulk_floats, L2Z_ERR_COMM,
                  "bulk all-reduce of %zu floats exceeds the bulk landing region (%zu)", (size_t)P * (size_t)n, c->bulk_floats);
        BulkArgs a = {};
        a.stage = part; a.P = P; a.n_loc = n_loc; a.rank = c->rank; a.world = c->world;
        for (int r = 0; r < c->world; r++) a.peer_arena[r] = c->peer_arena[r];
        a.bulk_off = kP2pFlagBytes + 2 * c->slot_floats * 8;
        a.bulk_floats = c->bulk_floats;
        a.ctl = c->d_ctl; a.err = c->h_err;
        a.timeout_ticks = tunables().p2p_timeout_s * 100000000LL;
        const unsigned long long e = ++c->bulk_epoch;
        hipError_t he = launch_bulk_scatter_push(a, e, st);
        if (he == hipSuccess) he = launch_bulk_reduce(a, e, stage + (size_t)c->rank * P * n_loc, st);
        L2Z_CHECK(he == hipSuccess, L2Z_ERR_HIP, "bulk all-reduce launch failed: %s", hipGetErrorString(he));
        return comm_bulk_allgather(c, stage, P, n_loc, dst, ldd, st);
    }
    L2Z_CHECK(c->nccl != nullptr, L2Z_ERR_COMM, "bulk all-reduce: the shard group has no transport");
    L2Z_CHECK(ldd == n, L2Z_ERR_INVALID, "bulk all-reduce over RCCL: the destination rows must be contiguous");
    ncclResult_t r = g_api.AllReduce(part, dst, (size_t)P * (size_t)n, ncclFloat, ncclSum, static_cast<ncclComm_t>(c->nccl), st);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclAllReduce failed: %s", g_api.GetErrorString(r));
    return L2Z_OK;
}

}  // namespace l2z

using namespace l2z;

extern "C" int l2z_comm_unique_id(void *out_id)
{
    L2Z_CHECK(out_id != nullptr, L2Z_ERR_INVALID, "l2z_comm_unique_id: null out");
    static_assert(sizeof(ncclUniqueId) == L2Z_COMM_ID_BYTES, "ncclUniqueId size");
    L2Z_TRY(load_rccl());
    ncclUniqueId id;
    ncclResult_t r = g_api.GetUniqueId(&id);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclGetUniqueId failed: %s", g_api.GetErrorString(r));
    std::memcpy(out_id, &id, sizeof id);
    return L2Z_OK;
}

extern "C" int l2z_comm_init(int rank, int world, const void *id, int device, l2z_comm **out)
{
    L2Z_CHECK(out != nullptr && world >= 1 && rank >= 0 && rank < world, L2Z_ERR_INVALID,
              "l2z_comm_init: bad rank/world %d/%d", rank, world);
    L2Z_HIP(hipSetDevice(device));
    l2z_comm *c = new l2z_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->nccl = nullptr;
    // world == 1 with an id still builds a 1-rank RCCL communicator: the same call path as
    // N > 1 (dlopen, ncclCommInitRank, ncclAllGather on the stream), testable on one GPU
    // id == NULL with world > 1: no RCCL communicator; the group must then be connected for
    // peer-write gathers (l2z_comm_p2p_export / _connect) before it is used
    if (id != nullptr) {
        int s = load_rccl();
        if (s != L2Z_OK) {
            delete c;
            return s;
        }
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        ncclComm_t comm = nullptr;
        ncclResult_t r = g_api.CommInitRank(&comm, world, uid, rank);
        if (r != ncclSuccess) {
            delete c;
            set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(r));
            return L2Z_ERR_COMM;
        }
        c->nccl = comm;
    }
    *out = c;
    return L2Z_OK;
}

// A rank descriptor without a communicator: for l2z_emu_transformer (one process, one GPU).
extern "C" int l2z_comm_init_emulated(int rank, int world, int device, l2z_comm **out)
{
    L2Z_CHECK(out != nullptr && world >= 1 && rank >= 0 && rank < world, L2Z_ERR_INVALID,
              "l2z_comm_init_emulated: bad rank/world %d/%d", rank, world);
    l2z_comm *c = new l2z_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->nccl = nullptr;
    *out = c;
    return L2Z_OK;
}

// ---- peer-write all-gather: arena export / connect (protocol in p2p.hip) ----
extern "C" int l2z_comm_p2p_export(l2z_comm *c, size_t max_vector_floats, void *handle_out)
{
    // without a separate width the bulk regions are sized for matrices as wide as the longest vector
    return l2z_comm_p2p_export_sized(c, max_vector_floats, max_vector_floats, handle_out);
}

extern "C" int l2z_comm_p2p_export_sized(l2z_comm *c, size_t max_vector_floats, size_t max_matrix_width,
                                         void *handle_out)
{
    L2Z_CHECK(c != nullptr && handle_out != nullptr && max_vector_floats > 0, L2Z_ERR_INVALID,
              "l2z_comm_p2p_export: bad arguments");
    L2Z_CHECK(c->world <= kMaxWorld, L2Z_ERR_INVALID, "peer-write gathers support up to %d ranks", kMaxWorld);
    L2Z_CHECK(c->arena == nullptr, L2Z_ERR_STATE, "l2z_comm_p2p_export: called twice");
    static_assert(sizeof(hipIpcMemHandle_t) == L2Z_COMM_IPC_BYTES, "hipIpcMemHandle_t size");
    L2Z_HIP(hipSetDevice(c->device));
    c->slot_floats = (max_vector_floats + 1023) & ~(size_t)1023;
    // bulk regions for the sharded prefill's [chunk tokens, dim | hidden_dim] activation matrices
    // (plain floats): max_matrix_width = max(dim, hidden_dim) -- the vocabulary is only ever gathered as
    // a vector (7B shape: 2 x 11008 x 1024 floats = 90 MB instead of 256 MB at the vocabulary's width).
    // L2Z_P2P_BULK_MB overrides, 0 = none (sharded runstates then step their prompts token by token).
    const size_t width = ((max_matrix_width ? max_matrix_width : max_vector_floats) + 63) & ~(size_t)63;
    c->bulk_floats = tunables().p2p_bulk_mb >= 0 ? ((size_t)tunables().p2p_bulk_mb << 20) / 4
                                                 : width * (size_t)prefill_chunk_tokens();
    const size_t bytes = kP2pFlagBytes + 2 * c->slot_floats * 8 + 2 * c->bulk_floats * 4;  // LL: 8-byte {value, epoch} words
    // fine-grained: peers' stores and this rank's flag polls / landing reads are coherent inside
    // a running kernel (ordinary hipMalloc memory is only coherent at kernel boundaries)
    L2Z_HIP(hipExtMallocWithFlags((void **)&c->arena, bytes, hipDeviceMallocFinegrained));
    L2Z_HIP(hipMemset(c->arena, 0, bytes));
    L2Z_HIP(hipMalloc((void **)&c->d_ctl, kCtlInts * sizeof(int)));
    L2Z_HIP(hipMemset(c->d_ctl, 0, kCtlInts * sizeof(int)));
    L2Z_HIP(hipHostMalloc((void **)&c->h_err, sizeof(int), hipHostMallocDefault));
    *c->h_err = 0;
    L2Z_HIP(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    L2Z_HIP(hipIpcGetMemHandle(&h, c->arena));
    std::memcpy(handle_out, &h, sizeof h);
    return L2Z_OK;
}

extern "C" int l2z_comm_p2p_connect(l2z_comm *c, const void *handles)
{
    L2Z_CHECK(c != nullptr && handles != nullptr, L2Z_ERR_INVALID, "l2z_comm_p2p_connect: bad arguments");
    L2Z_CHECK(c->arena != nullptr && !c->p2p, L2Z_ERR_STATE,
              "l2z_comm_p2p_connect: call l2z_comm_p2p_export first (once)");
    L2Z_HIP(hipSetDevice(c->device));
    {   // best effort: kernels here store straight into the peers' memory, so make sure peer access
        // is on for every other visible GPU (hipIpcOpenMemHandle's lazy flag covers copies; errors
        // such as "already enabled" or "not supported" are not fatal -- the open below decides)
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) == hipSuccess)
            for (int d = 0; d < n_dev; d++) {
                int can = 0;
                if (d != c->device && hipDeviceCanAccessPeer(&can, c->device, d) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(d, 0);
            }
        (void)hipGetLastError();  // clear a sticky "peer access already enabled"
    }
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) {
            c->peer_arena[r] = c->arena;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char *>(handles) + (size_t)r * L2Z_COMM_IPC_BYTES, sizeof h);
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        L2Z_CHECK(e == hipSuccess, L2Z_ERR_COMM, "hipIpcOpenMemHandle(rank %d) failed: %s", r,
                  hipGetErrorString(e));
        c->peer_arena[r] = static_cast<char *>(p);
    }
    c->p2p = true;
    return L2Z_OK;
}

// Measurement support (llama2_hip_test.h): this rank ALONE, every peer's arena mapped to its own -- see l2z_comm::solo.
extern "C" int l2z_comm_p2p_connect_solo(l2z_comm *c)
{
    L2Z_CHECK(c != nullptr, L2Z_ERR_INVALID, "l2z_comm_p2p_connect_solo: null comm");
    L2Z_CHECK(c->arena != nullptr && !c->p2p, L2Z_ERR_STATE, "l2z_comm_p2p_connect_solo: call l2z_comm_p2p_export first (once)");
    // the peers' "arenas": a local sink with one reserved head + slot pair per peer, so that the N - 1 stores of a pushed
    // word go to N - 1 different addresses as they would to N - 1 devices (they are never read); no bulk regions there:
    // the batched prefill is refused on a solo group
    const size_t per_peer = kP2pFlagBytes + 2 * c->slot_floats * 8;
    L2Z_HIP(hipSetDevice(c->device));
    if (c->world > 1) {
        L2Z_HIP(hipExtMallocWithFlags((void **)&c->solo_sink, per_peer * (size_t)(c->world - 1), hipDeviceMallocFinegrained));
        L2Z_HIP(hipMemset(c->solo_sink, 0, per_peer * (size_t)(c->world - 1)));
    }
    for (int r = 0, k = 0; r < c->world; r++) c->peer_arena[r] = r == c->rank ? c->arena : c->solo_sink + per_peer * (size_t)(k++);
    c->bulk_floats = 0;
    c->p2p = true;
    c->solo = true;
    return L2Z_OK;
}

// Diagnostics of the peer-write transport (include/llama2_hip_test.h): what a hand-over costs between two ranks.
extern "C" int l2z_comm_p2p_pingpong(l2z_comm *c, int other, int initiator, int iters, double *rtt_us)
{
    L2Z_CHECK(c != nullptr && rtt_us != nullptr && iters >= 1, L2Z_ERR_INVALID, "l2z_comm_p2p_pingpong: bad arguments");
    L2Z_CHECK(c->p2p && !c->solo, L2Z_ERR_STATE, "l2z_comm_p2p_pingpong: the peer-write arenas are not connected");
    L2Z_CHECK(other >= 0 && other < c->world && other != c->rank, L2Z_ERR_INVALID, "l2z_comm_p2p_pingpong: peer %d", other);
    L2Z_HIP(hipSetDevice(c->device));
    long long *d_ticks = nullptr;
    L2Z_HIP(hipMalloc((void **)&d_ticks, sizeof(long long)));
    const unsigned long long base = c->probe_seq[other];
    c->probe_seq[other] += (unsigned long long)iters;
    hipError_t e = launch_p2p_pingpong(c->arena, c->peer_arena[other], c->rank, other, initiator != 0, iters, base,
                                       tunables().p2p_timeout_s * 100000000LL, d_ticks, c->h_err, nullptr);
    long long ticks = -1;
    if (e == hipSuccess) e = hipMemcpy(&ticks, d_ticks, sizeof ticks, hipMemcpyDeviceToHost);
    (void)hipFree(d_ticks);
    L2Z_HIP(e);
    L2Z_CHECK(ticks >= 0, L2Z_ERR_COMM, "l2z_comm_p2p_pingpong: rank %d did not answer within %lld s", other,
              tunables().p2p_timeout_s);
    *rtt_us = (double)ticks / 100.0 / (double)iters;   // wall_clock64: 100 MHz
    return L2Z_OK;
}

// `bytes` from a local buffer into rank `other`'s arena (bulk region 1, which only the sharded prefill uses and
// rewrites before it flags) by the runtime's device-to-device copy: `iters` copies, each synchronised -- the latency
// of ONE small peer copy (the transport RCCL-free hosts would reach for) beside the LL word's.
extern "C" int l2z_comm_peer_copy_probe(l2z_comm *c, int other, size_t bytes, int iters, double *us_per_copy)
{
    L2Z_CHECK(c != nullptr && us_per_copy != nullptr && iters >= 1 && bytes >= 4, L2Z_ERR_INVALID,
              "l2z_comm_peer_copy_probe: bad arguments");
    L2Z_CHECK(c->p2p && !c->solo, L2Z_ERR_STATE, "l2z_comm_peer_copy_probe: the peer-write arenas are not connected");
    L2Z_CHECK(other >= 0 && other < c->world && other != c->rank, L2Z_ERR_INVALID, "l2z_comm_peer_copy_probe: peer %d", other);
    L2Z_CHECK(bytes <= c->bulk_floats * 4, L2Z_ERR_INVALID, "l2z_comm_peer_copy_probe: %zu bytes do not fit the bulk region", bytes);
    L2Z_HIP(hipSetDevice(c->device));
    char *dst = c->peer_arena[other] + kP2pFlagBytes + 2 * c->slot_floats * 8 + c->bulk_floats * 4;
    void *src = nullptr;
    L2Z_HIP(hipMalloc(&src, bytes));
    hipError_t e = hipMemset(src, 0, bytes);
    hipStream_t st = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 3 && e == hipSuccess; i++) {  // warm-up
        e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters && e == hipSuccess; i++) {
        e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(src);
    L2Z_HIP(e);
    *us_per_copy = us / (double)iters;
    return L2Z_OK;
}

extern "C" int l2z_comm_rank(const l2z_comm *c, int *rank, int *world)
{
    L2Z_CHECK(c != nullptr, L2Z_ERR_INVALID, "l2z_comm_rank: null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return L2Z_OK;
}

// Loads RCCL now (l2z_comm_init would on first use) and says WHICH library the process got: a process that has imported
// PyTorch holds torch's own bundled copies of librccl / libamdhip64, and a dlopen by SONAME resolves to whichever copy
// was loaded first.  bench.py's legs import torch first (gloo control plane; the other order leaves HIP without a visible
// device on this image), so their RCCL is torch's bundled one: they put this answer in their record.
extern "C" int l2z_comm_rccl_info(char *path_out, size_t cap, int *version)
{
    L2Z_TRY(load_rccl());
    if (version) {
        *version = 0;
        (void)g_api.GetVersion(version);
    }
    if (path_out && cap) {
        Dl_info info;
        snprintf(path_out, cap, "%s", dladdr(reinterpret_cast<void *>(g_api.AllGather), &info) && info.dli_fname ? info.dli_fname : "?");
    }
    return L2Z_OK;
}

// What the group's transports really are, for the bench line: the rank count RCCL itself reports for
// the communicator (ncclCommCount; 0 without one) and whether the peer-write arenas are connected.
extern "C" int l2z_comm_transports(const l2z_comm *c, int *rccl_ranks, int *p2p_connected)
{
    L2Z_CHECK(c != nullptr, L2Z_ERR_INVALID, "l2z_comm_transports: null comm");
    int n = 0;
    if (c->nccl != nullptr) {
        ncclResult_t r = g_api.CommCount(static_cast<ncclComm_t>(c->nccl), &n);
        L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclCommCount failed: %s", g_api.GetErrorString(r));
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (p2p_connected) *p2p_connected = c->p2p ? 1 : 0;
    return L2Z_OK;
}

extern "C" void l2z_comm_free(l2z_comm *c)
{
    if (!c) return;
    for (int r = 0; r < c->world && r < kMaxWorld; r++)
        if (r != c->rank && c->peer_arena[r] && !c->solo) (void)hipIpcCloseMemHandle(c->peer_arena[r]);
    if (c->solo_sink) (void)hipFree(c->solo_sink);
    if (c->arena) (void)hipFree(c->arena);
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->nccl && g_api.CommDestroy) g_api.CommDestroy(static_cast<ncclComm_t>(c->nccl));
    delete c;
}

// Pure host logic (no GPU): contiguous equal split in units of `granule` rows.
extern "C" int l2z_shard_range(int64_t rows, int64_t granule, int rank, int world, int64_t *r0,
                               int64_t *r1)
{
    L2Z_CHECK(r0 && r1 && world >= 1 && rank >= 0 && rank < world && granule >= 1 && rows >= 0,
              L2Z_ERR_INVALID, "l2z_shard_range: bad arguments");
    L2Z_CHECK(rows % granule == 0 && (rows / granule) % world == 0, L2Z_ERR_INVALID,
              "l2z_shard_range: %lld rows in granules of %lld do not split over %d ranks",
              (long long)rows, (long long)granule, world);
    const int64_t per = rows / world;
    *r0 = per * rank;
    *r1 = per * (rank + 1);
    return L2Z_OK;
}
