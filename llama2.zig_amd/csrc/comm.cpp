// comm.cpp -- multi-GPU shard group: one process per GPU, RCCL over xGMI.
//
// The reference (cgbur/llama2.zig) is single-threaded and single-device; this
// is what the build adds (DESIGN.md "Sharding").  RCCL is bound
// with dlopen at l2z_comm_init time, so the single-GPU library has no RCCL
// dependency at all and loads on machines without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "l2z_comm.h"

namespace l2z {

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
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
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.handle = h;
    return L2Z_OK;
}
}  // namespace

int comm_allgather_inplace(const l2z_comm *c, float *buf, size_t count_per_rank, hipStream_t st)
{
    if (c == nullptr || c->nccl == nullptr) return L2Z_OK;  // single GPU without a communicator
    // in-place form: sendbuff == recvbuff + rank * count
    ncclResult_t r = g_api.AllGather(buf + (size_t)c->rank * count_per_rank, buf, count_per_rank,
                                     ncclFloat, static_cast<ncclComm_t>(c->nccl), st);
    L2Z_CHECK(r == ncclSuccess, L2Z_ERR_COMM, "ncclAllGather failed: %s", g_api.GetErrorString(r));
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
    if (world > 1 || id != nullptr) {
        if (id == nullptr) {
            delete c;
            set_error("l2z_comm_init: id required when world > 1");
            return L2Z_ERR_INVALID;
        }
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

extern "C" int l2z_comm_rank(const l2z_comm *c, int *rank, int *world)
{
    L2Z_CHECK(c != nullptr, L2Z_ERR_INVALID, "l2z_comm_rank: null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return L2Z_OK;
}

extern "C" void l2z_comm_free(l2z_comm *c)
{
    if (!c) return;
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
