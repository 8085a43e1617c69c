// p2p.hip -- all-gather by direct peer stores (xGMI between GPUs), no collective library.
//
// Every rank owns an "arena" of fine-grained device memory that all peers map through HIP IPC:
//   (4 KB reserved) | landing slot 0 | landing slot 1     slot = 8 bytes per float of the longest
//                                                          gathered vector
// "LL" form: a float travels as ONE 8-byte word {value bits, epoch}, written with a single relaxed
// system-scope atomic store and read with a relaxed system-scope atomic load.  An 8-byte access is
// atomic, so the receiver either sees the old word or the complete new one: no fence, no separate
// flag, no assumption about the order in which different stores arrive -- and none of the L2
// write-backs / invalidates a release / acquire pair costs on this part (per-XCD L2s).
// One launch gathers one vector.  Block p of rank r talks to peer p only:
//   1. writes {slice[i], e} for r's slice into p's slot (at r's offset), e = number of this gather
//      between r and p;
//   2. polls p's slice in its OWN slot until every word carries e, copying the values into the
//      ordinary (cached) activation buffer.
// No block waits for another block of the same launch, so nothing here can deadlock on
// scheduling; a peer that never arrives trips the timeout, sets *err and lets the kernel end.
// Slots alternate with e: a rank can only be writing gather g+2 (same slot as g) after it has
// received every peer's words of g+1, which a peer writes only after it finished reading gather g.
// The gathered values are bit copies -- results stay identical to the unsharded pass.
#include <hip/hip_runtime.h>

#include "l2z_comm.h"

namespace l2z {
namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 *slot_words(char *arena, int e, size_t slot_floats)
{
    return (u64 *)(arena + kP2pFlagBytes) + (size_t)(e & 1) * slot_floats;
}

__global__ __launch_bounds__(256) void p2p_allgather_kernel(const P2pArgs a, int pushed)
{
    const int p = blockIdx.x;
    if (p == a.rank) return;  // own slice is already in place
    __shared__ int s_epoch, s_timeout;
    if (threadIdx.x == 0) {
        s_epoch = a.epoch[p] + 1;
        s_timeout = 0;
    }
    __syncthreads();
    const int e = s_epoch;
    const u64 tag = (u64)(unsigned)e << 32;
    // 1. my slice -> peer p's slot (unless the producing kernel wrote the words itself)
    if (!pushed) {
        u64 *dst = slot_words(a.peer_arena[p], e, a.slot_floats) + (size_t)a.rank * a.count;
        const float *src = a.buf + (size_t)a.rank * a.count;
        for (size_t i = threadIdx.x; i < a.count; i += blockDim.x)
            __hip_atomic_store(dst + i, tag | (u64)__float_as_uint(src[i]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. peer p's slice <- my slot: up to 4 words per lane in flight, re-polled until all carry e
    {
        const u64 *src = slot_words(a.peer_arena[a.rank], e, a.slot_floats) + (size_t)p * a.count;
        float *dst = a.buf + (size_t)p * a.count;
        const long long t0 = wall_clock64();
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
                if (!ready && wall_clock64() - t0 > a.timeout_ticks) {
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
        a.epoch[p] = e;
        if (s_timeout) *a.err = 1 + p;
    }
}

}  // namespace

hipError_t launch_p2p_allgather(const P2pArgs &a, hipStream_t st, bool pushed)
{
    hipLaunchKernelGGL(p2p_allgather_kernel, dim3(a.world), dim3(256), 0, st, a, pushed ? 1 : 0);
    return hipGetLastError();
}

}  // namespace l2z
