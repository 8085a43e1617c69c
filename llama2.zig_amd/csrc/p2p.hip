// p2p.hip -- all-gather by direct peer stores (xGMI between GPUs), no collective library.
//
// Every rank owns an "arena" of fine-grained device memory that all peers map through HIP IPC:
//   int flag[kMaxWorld] | landing slot 0 | landing slot 1      (slot = longest gathered vector)
// One launch gathers one vector.  Block p of rank r talks to peer p only:
//   1. stores r's slice into p's landing slot (at r's offset), fences at system scope and writes
//      flag[r] = epoch in p's arena;
//   2. waits until flag[p] in its OWN arena reaches the epoch (p's slice has landed here);
//   3. copies that slice from the landing slot into the ordinary (cached) activation buffer.
// No block waits for another block of the same launch, so nothing here can deadlock on
// scheduling; a peer that never arrives trips the timeout, sets *err and lets the kernel end.
// Slots alternate with the epoch: a rank can only be pushing gather g+2 after it has seen every
// peer's flag for g+1, which the peer raises after it finished copying gather g out of that slot.
// The gathered values are copies -- results stay bit-identical to the unsharded pass.
#include <hip/hip_runtime.h>

#include "l2z_comm.h"

namespace l2z {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void copy_floats(float *dst, const float *src, size_t n)
{
    const bool vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    if (vec) {
        const size_t n4 = n >> 2;
        for (size_t i = threadIdx.x; i < n4; i += blockDim.x) ((v4f *)dst)[i] = ((const v4f *)src)[i];
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    } else {
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void p2p_allgather_kernel(const P2pArgs a)
{
    const int p = blockIdx.x;
    if (p == a.rank) return;  // own slice is already in place
    __shared__ int s_epoch;
    if (threadIdx.x == 0) s_epoch = a.epoch[p] + 1;
    __syncthreads();
    const int e = s_epoch;
    const size_t slot_off = (size_t)(e & 1) * a.slot_floats;
    // 1. my slice -> peer p's landing slot
    float *dst = (float *)(a.peer_arena[p] + kP2pFlagBytes) + slot_off + (size_t)a.rank * a.count;
    copy_floats(dst, a.buf + (size_t)a.rank * a.count, a.count);
    __threadfence_system();  // every lane's stores are out before the flag
    __syncthreads();
    if (threadIdx.x == 0) {
        int *peer_flag = (int *)a.peer_arena[p] + a.rank;
        __hip_atomic_store(peer_flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // 2. peer p's slice has landed here?
        int *my_flag = (int *)a.peer_arena[a.rank] + p;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(my_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                *a.err = 1 + p;
                break;
            }
        }
        a.epoch[p] = e;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: no lane reads a stale landing line
    // 3. landing slot -> activation buffer
    const float *src = (const float *)(a.peer_arena[a.rank] + kP2pFlagBytes) + slot_off + (size_t)p * a.count;
    copy_floats(a.buf + (size_t)p * a.count, src, a.count);
}

}  // namespace

hipError_t launch_p2p_allgather(const P2pArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(p2p_allgather_kernel, dim3(a.world), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace l2z
