// p2p.hip -- all-gather by direct peer stores (xGMI between GPUs), no collective library.
//
// Every rank owns an "arena" of fine-grained device memory that all peers map through HIP IPC:
//   (4 KB reserved) | landing slot 0 | landing slot 1     slot = 8 bytes per float of the longest
//                                                          gathered vector
// "LL" form: a float travels as ONE 8-byte word {value bits, epoch}, written with a single relaxed
// system-scope atomic store and read with a system-scope load.  An 8-byte access is atomic, so the
// receiver either sees the old word or the complete new one: no fence, no separate flag, no
// assumption about the order in which different stores arrive -- and none of the L2 write-backs /
// invalidates a release / acquire pair costs on this part (per-XCD L2s).
//
// Epochs.  Every rank runs the same sequence of forward passes, each with the same n gathers, so
// gather gi (1-based) of the running pass has epoch  ctl[kCtlEpoch] + gi  on every rank; the
// pass-closing gather (gi == n: the logits) adds n to the counter once all its blocks are done.
// Epochs are consecutive, and a gather lands in slot (epoch & 1).
//
// Who writes: the PRODUCING kernel -- the writer lane of a mat-vec epilogue or of the attention
// output stage stores {value, epoch} for each of its outputs straight into every peer's slot
// (p2p_ll_push), so the values travel while the launch is still running.  (A producer that cannot
// push leaves it to the gather launch below.)
// Who reads, two forms:
//   * consumer-side (default, P2pArgs::self = 1): the producers also store into their OWN slot, and
//     the next mat-vec reads its whole input vector as LL words out of this rank's slot while it
//     stages x in LDS (kernel_common.h xload_issue_ll / xstage_finish_ll), re-reading until every
//     word carries the epoch.  No gather launch at all: a layer stays 5 graph nodes at any N.  Only
//     the logits are collected by a launch (the host and argmax read them as a plain buffer).
//   * gather launch (L2Z_P2P_CONSUME=0, or shapes only the scalar kernels take): one launch per
//     gathered vector; block p polls peer p's slice in this rank's slot and copies the values into
//     the ordinary (cached) activation buffer the consumer then reads.
// Slot reuse is safe with two slots in both forms: a rank writes gather g+2 (same slot as g) only
// from a kernel that has read ALL of gather g+1, and a peer's g+1 words are all there only after
// every block of the peer's kernel that read g has finished reading it (each consumer block is
// also a producer of the next gather, and it produces after it has staged its input).
// No block waits for another block of the same launch, so nothing here can deadlock on scheduling;
// a peer that never arrives trips the timeout, latches ctl[kCtlErr] (every later wait gives up at
// once), sets *err for the host and lets the kernel end.
// The gathered values are bit copies -- results stay identical to the unsharded pass.
#include <hip/hip_runtime.h>

#include "l2z_comm.h"
#include "tunables.h"

namespace l2z {
namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 *slot_words(char *arena, int e, size_t slot_floats)
{
    return (u64 *)(arena + kP2pFlagBytes) + (size_t)(e & 1) * slot_floats;
}

__global__ __launch_bounds__(256) void p2p_allgather_kernel(const P2pArgs a, int gi, int n_gathers,
                                                            int pushed)
{
    const int p = blockIdx.x;
    __shared__ int s_timeout;
    if (threadIdx.x == 0) s_timeout = 0;
    const int e = a.ctl[kCtlEpoch] + gi;
    const bool dead = __hip_atomic_load(a.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    const u64 tag = (u64)(unsigned)e << 32;
    if (p != a.rank && !dead) {  // own slice is already in place
        // 1. my slice -> peer p's slot (unless the producing kernel wrote the words itself)
        if (!pushed) {
            u64 *dst = slot_words(a.peer_arena[p], e, a.slot_floats) + (size_t)a.rank * a.count;
            const float *src = a.buf + (size_t)a.rank * a.count;
            for (size_t i = threadIdx.x; i < a.count; i += blockDim.x)
                __hip_atomic_store(dst + i, tag | (u64)__float_as_uint(src[i]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // 2. peer p's slice <- my slot: up to 4 words per lane in flight, re-polled until all carry e
        const u64 *src = slot_words(a.peer_arena[a.rank], e, a.slot_floats) + (size_t)p * a.count;
        float *dst = a.buf + (size_t)p * a.count;
        const long long t0 = wall_clock64();
        bool give_up = false;
        for (size_t base = threadIdx.x; base < a.count; base += 4 * (size_t)blockDim.x) {
            u64 w[4];
            bool ready;
            do {
                ready = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const size_t i = base + (size_t)k * blockDim.x;
                    w[k] = i < a.count ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                       : tag;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) ready = ready && (unsigned)(w[k] >> 32) == (unsigned)e;
                if (!ready && (give_up || wall_clock64() - t0 > a.timeout_ticks)) {
                    give_up = true;
                    s_timeout = 1;
                    break;
                }
            } while (!ready);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t i = base + (size_t)k * blockDim.x;
                if (i < a.count) dst[i] = __uint_as_float((unsigned)w[k]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_timeout) {
            __hip_atomic_store(a.ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.err = 1 + p;
        }
        // the pass-closing gather advances the epoch counter once every block has read it
        if (gi == n_gathers &&
            __hip_atomic_fetch_add(a.ctl + kCtlDone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                (int)gridDim.x - 1) {
            __hip_atomic_store(a.ctl + kCtlDone, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctl + kCtlEpoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- all-reduce (scheme B: column-sharded Wo / W2) ----
// Every rank holds a partial [count] vector (rank 0's includes the residual); afterwards out[i] = part_0[i] + part_1[i] + ...
// + part_{N-1}[i], summed in RANK order by every rank -- so all ranks end with the same bits, whatever the order the
// words arrive in.  One hop: a rank's partial travels as LL words to word rank * count + i of every peer's slot -- sent by
// this launch (the default: a peer store in a mat-vec's epilogue holds up the wave's loads behind it; L2Z_P2P_PUSH=2 makes
// the producing mat-vec send instead) -- and the thread that owns element i polls the N - 1 peers' words for it.  Slot reuse and deadlock freedom: as for the gather launch above.
__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(const P2pArgs a, float *out, int gi, int pushed)
{
    __shared__ int s_timeout;
    if (threadIdx.x == 0) s_timeout = 0;
    const int e = a.ctl[kCtlEpoch] + gi;
    const bool dead = __hip_atomic_load(a.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    const u64 tag = (u64)(unsigned)e << 32;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int late = -1;
    if (i < a.count && !dead) {
        const float own = a.buf[i];
        if (!pushed)
            for (int p = 0; p < a.world; p++)
                if (p != a.rank)
                    __hip_atomic_store(slot_words(a.peer_arena[p], e, a.slot_floats) + (size_t)a.rank * a.count + i,
                                       tag | (u64)__float_as_uint(own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // all peers' words for element i in flight at once, re-polled until each carries the epoch
        const u64 *mine = slot_words(a.peer_arena[a.rank], e, a.slot_floats) + i;
        const long long t0 = wall_clock64();
        u64 w[kMaxWorld];
#pragma unroll
        for (int p = 0; p < kMaxWorld; p++)
            w[p] = (p < a.world && p != a.rank) ? __hip_atomic_load(mine + (size_t)p * a.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                 : tag;
        for (;;) {
            bool ready = true;
#pragma unroll
            for (int p = 0; p < kMaxWorld; p++)
                if ((unsigned)(w[p] >> 32) != (unsigned)e) {
                    w[p] = __hip_atomic_load(mine + (size_t)p * a.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(w[p] >> 32) != (unsigned)e) {
                        ready = false;
                        late = p;
                    }
                }
            if (ready) {
                late = -1;
                break;
            }
            if (wall_clock64() - t0 > a.timeout_ticks) break;
            __builtin_amdgcn_s_sleep(1);
        }
        float acc = 0.0f;
#pragma unroll
        for (int p = 0; p < kMaxWorld; p++)
            if (p < a.world) {
                const float v = p == a.rank ? own : __uint_as_float((unsigned)w[p]);
                acc = p == 0 ? v : __fadd_rn(acc, v);
            }
        out[i] = acc;
        if (late >= 0) s_timeout = 1 + late;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_timeout) {
        __hip_atomic_store(a.ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *a.err = s_timeout;
    }
}

// ---- bulk form (sharded prefill): [P, n_loc] blocks of floats, plain stores, one flag per sender ----
// A gather of a whole activation matrix is bandwidth-, not latency-bound: the LL form's 8 bytes per
// float would double the xGMI traffic for nothing.  The sender stores its block into region (e & 1)
// of every peer's arena (16-byte stores), makes them visible (system-scope release fence, which also
// waits for the stores to be acknowledged), and the LAST block of the launch to get there writes
// flag[rank] = e into every peer's reserved head.  The receiver's unpack launch waits per sender for
// flag[p] >= e (epochs only grow), acquires, and copies that block into the row-major matrix.  Two
// regions suffice for the same reason two LL slots do: a rank pushes gather g+2 only after its own
// unpack of g+1 has run, and that waited for every peer's push of g+1, which each peer issued
// behind its unpack of g.
constexpr int kBulkBlocksPerPeer = 16;

__device__ __forceinline__ float *bulk_region(char *arena, const BulkArgs &a, u64 e)
{
    return (float *)(arena + a.bulk_off) + (size_t)(e & 1) * a.bulk_floats;
}

__global__ __launch_bounds__(256) void bulk_push_kernel(const BulkArgs a, u64 e)
{
    // peers in ring order from this rank, so that the ranks do not all start on the same receiver
    const int pi = blockIdx.x / kBulkBlocksPerPeer, part = blockIdx.x % kBulkBlocksPerPeer;
    const int p = (a.rank + 1 + pi) % a.world;
    const size_t count = (size_t)a.P * a.n_loc;
    const float *src = a.stage + (size_t)a.rank * count;
    float *dst = bulk_region(a.peer_arena[p], a, e) + (size_t)a.rank * count;
    if ((count & 3) == 0) {  // blocks start 16-byte aligned
        const size_t n4 = count >> 2;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < n4; i += (size_t)kBulkBlocksPerPeer * 256)
            ((float4 *)dst)[i] = ((const float4 *)src)[i];
    } else {
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < count; i += (size_t)kBulkBlocksPerPeer * 256)
            dst[i] = src[i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(a.ctl + kCtlBulkDone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(a.ctl + kCtlBulkDone, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        for (int q = 0; q < a.world; q++)
            if (q != a.rank)
                __hip_atomic_store((u64 *)a.peer_arena[q] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void bulk_unpack_kernel(const BulkArgs a, u64 e, int wait, float *dst, int ldd)
{
    const int p = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const size_t count = (size_t)a.P * a.n_loc;
    const float *src = a.stage + (size_t)p * count;
    if (wait && p != a.rank) {
        __shared__ int s_fail;
        if (threadIdx.x == 0) {
            s_fail = 0;
            const u64 *flag = (const u64 *)a.peer_arena[a.rank] + p;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
                if (__hip_atomic_load(a.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > a.timeout_ticks) {
                    s_fail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (s_fail) {
                __hip_atomic_store(a.ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *a.err = 1 + p;
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the sender's stores, not a stale line
        src = bulk_region(a.peer_arena[a.rank], a, e) + (size_t)p * count;
    }
    // rows of the block -> columns [p * n_loc, (p + 1) * n_loc) of the row-major matrix
    const int n_loc = a.n_loc;
    if ((n_loc & 3) == 0 && (ldd & 3) == 0) {
        const int n4 = n_loc >> 2;
        const size_t total = (size_t)a.P * n4;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < total; i += (size_t)parts * 256) {
            const size_t t = i / n4, j = i - t * n4;
            ((float4 *)(dst + t * ldd + (size_t)p * n_loc))[j] = ((const float4 *)(src + t * n_loc))[j];
        }
    } else {
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < count; i += (size_t)parts * 256) {
            const size_t t = i / n_loc, j = i - t * n_loc;
            dst[t * ldd + (size_t)p * n_loc + j] = src[i];
        }
    }
}

// ---- bulk all-reduce (scheme B's batched prefill): reduce-scatter here, then the bulk all-gather above ----
// Every rank holds a partial [P, n] matrix (n = world * n_loc; rank 0's carries the residual).  Rank r sends the columns of
// peer p's slice to peer p -- a [P, n_loc] block into block `rank` of p's region (e & 1) -- and flags as bulk_push_kernel
// does; the owner of a slice then adds the world blocks IN RANK ORDER (its own straight from its partial), so every rank
// that later receives the slice holds the same bits.  Region reuse: as for the gathers (every bulk operation is a push
// followed by a wait for every peer's push).
__global__ __launch_bounds__(256) void bulk_scatter_push_kernel(const BulkArgs a, u64 e)
{
    const int pi = blockIdx.x / kBulkBlocksPerPeer, part = blockIdx.x % kBulkBlocksPerPeer;
    const int p = (a.rank + 1 + pi) % a.world;
    const int n_loc = a.n_loc, n = n_loc * a.world;
    const size_t count = (size_t)a.P * n_loc;
    float *dst = bulk_region(a.peer_arena[p], a, e) + (size_t)a.rank * count;
    const float *src = a.stage + (size_t)p * n_loc;   // columns of p's slice, rows n floats apart
    if ((n_loc & 3) == 0) {
        const int n4 = n_loc >> 2;
        const size_t total = (size_t)a.P * n4;
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < total; i += (size_t)kBulkBlocksPerPeer * 256) {
            const size_t t = i / n4, j = i - t * n4;
            ((float4 *)(dst + t * n_loc))[j] = ((const float4 *)(src + t * n))[j];
        }
    } else {
        for (size_t i = (size_t)part * 256 + threadIdx.x; i < count; i += (size_t)kBulkBlocksPerPeer * 256) {
            const size_t t = i / n_loc, j = i - t * n_loc;
            dst[t * n_loc + j] = src[t * n + j];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(a.ctl + kCtlBulkDone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(a.ctl + kCtlBulkDone, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        for (int q = 0; q < a.world; q++)
            if (q != a.rank)
                __hip_atomic_store((u64 *)a.peer_arena[q] + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// out[t][j] = sum over ranks, in rank order, of their blocks of THIS rank's slice; out: [P, n_loc] contiguous
__global__ __launch_bounds__(256) void bulk_reduce_kernel(const BulkArgs a, u64 e, int wait, float *out)
{
    __shared__ int s_fail;
    if (threadIdx.x == 0) {
        s_fail = 0;
        for (int p = 0; p < a.world && wait && !s_fail; p++) {
            if (p == a.rank) continue;
            const u64 *flag = (const u64 *)a.peer_arena[a.rank] + p;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
                if (__hip_atomic_load(a.ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > a.timeout_ticks) {
                    s_fail = 1 + p;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (s_fail) {
            __hip_atomic_store(a.ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *a.err = s_fail;
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the senders' stores, not a stale line
    const int n_loc = a.n_loc, n = n_loc * a.world;
    const size_t count = (size_t)a.P * n_loc;
    // wait == 0 (emulated ranks, tests): the peers' blocks already lie in a.stage-style buffers the caller copied
    const float *region = bulk_region(a.peer_arena[a.rank], a, e);
    const float *own = a.stage + (size_t)a.rank * n_loc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const size_t t = i / n_loc, j = i - t * n_loc;
        float acc = 0.0f;
        for (int p = 0; p < a.world; p++) {
            const float v = p == a.rank ? own[t * n + j] : region[(size_t)p * count + i];
            acc = p == 0 ? v : __fadd_rn(acc, v);
        }
        out[i] = acc;
    }
}

}  // namespace

hipError_t launch_bulk_scatter_push(const BulkArgs &a, unsigned long long e, hipStream_t st)
{
    if (a.world < 2) return hipSuccess;
    hipLaunchKernelGGL(bulk_scatter_push_kernel, dim3((a.world - 1) * kBulkBlocksPerPeer), dim3(256), 0, st, a, (u64)e);
    return hipGetLastError();
}

hipError_t launch_bulk_reduce(const BulkArgs &a, unsigned long long e, float *out, hipStream_t st)
{
    const int cap = tunables().grid_cap;
    const int blocks = cap > 0 ? (cap < 64 ? cap : 64) : 128;
    hipLaunchKernelGGL(bulk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a, (u64)e, 1, out);
    return hipGetLastError();
}

hipError_t launch_bulk_push(const BulkArgs &a, unsigned long long e, hipStream_t st)
{
    if (a.world < 2) return hipSuccess;
    hipLaunchKernelGGL(bulk_push_kernel, dim3((a.world - 1) * kBulkBlocksPerPeer), dim3(256), 0, st, a, (u64)e);
    return hipGetLastError();
}

hipError_t launch_bulk_unpack(const BulkArgs &a, unsigned long long e, int wait, float *dst, int ldd,
                              hipStream_t st)
{
    // enough blocks to move the matrix at memory speed; when ranks share a GPU (L2Z_GRID_CAP set: the
    // one-GPU tests) a waiting launch stays small, so that the senders' launches have room to run
    const int blocks = wait && tunables().grid_cap > 0 ? (tunables().grid_cap < 128 ? tunables().grid_cap : 128) : 512;
    const int parts = (blocks + a.world - 1) / a.world;
    hipLaunchKernelGGL(bulk_unpack_kernel, dim3(a.world, parts), dim3(256), 0, st, a, (u64)e, wait, dst, ldd);
    return hipGetLastError();
}

hipError_t launch_p2p_allreduce(const P2pArgs &a, float *out, int gi, bool pushed, hipStream_t st)
{
    int nt = tunables().reduce_block;
    nt = nt < 64 ? 64 : nt > 1024 ? 1024 : (nt / 64) * 64;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3((unsigned)((a.count + nt - 1) / nt)), dim3(nt), 0, st, a, out, gi, pushed ? 1 : 0);
    return hipGetLastError();
}

// ---- diagnostic: LL-word round trips between two ranks (include/llama2_hip_test.h l2z_comm_p2p_pingpong) ----
// One thread per side.  The initiator stores word base + i into the other side's probe word (reserved head of its arena,
// kP2pProbeOff + 8 * sender), the responder answers with the same value into the initiator's; `iters` round trips
// between two reads of the 100-MHz wall clock.  The words are the gathers' own kind of traffic: one 8-byte system-scope
// store to the peer's fine-grained memory, polled by a system-scope load -- over xGMI when the ranks sit on two GPUs.
__global__ void p2p_pingpong_kernel(char *mine, char *theirs, int me, int other, int initiator, int iters, u64 base,
                                    long long timeout_ticks, long long *ticks_out, int *err)
{
    u64 *out = (u64 *)(theirs + kP2pProbeOff) + me;
    const u64 *in = (const u64 *)(mine + kP2pProbeOff) + other;
    const long long t0 = wall_clock64();
    bool dead = false;
    for (int i = 1; i <= iters && !dead; i++) {
        const u64 v = base + (u64)i;
        if (initiator) __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (__hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v)
            if (wall_clock64() - t0 > timeout_ticks) { dead = true; break; }
        if (!initiator && !dead) __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *ticks_out = dead ? -1 : wall_clock64() - t0;
    if (dead) *err = 1 + other;
}

hipError_t launch_p2p_pingpong(char *mine, char *theirs, int me, int other, bool initiator, int iters,
                               unsigned long long base, long long timeout_ticks, long long *ticks_out, int *err,
                               hipStream_t st)
{
    hipLaunchKernelGGL(p2p_pingpong_kernel, dim3(1), dim3(1), 0, st, mine, theirs, me, other, initiator ? 1 : 0, iters,
                       (u64)base, timeout_ticks, ticks_out, err);
    return hipGetLastError();
}

hipError_t launch_p2p_allgather(const P2pArgs &a, int gi, int n_gathers, bool pushed, hipStream_t st)
{
    hipLaunchKernelGGL(p2p_allgather_kernel, dim3(a.world), dim3(256), 0, st, a, gi, n_gathers,
                       pushed ? 1 : 0);
    return hipGetLastError();
}

}  // namespace l2z
