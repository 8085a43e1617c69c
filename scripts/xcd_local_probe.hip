// xcd_local_probe.hip -- what does synchronisation cost when every participant sits on ONE XCD?
// (round 4: every hand-over measured so far crossed XCDs and therefore went through the memory side: 1.6-11 us per
// device-wide barrier, scripts/barrier_probe.hip.  The 32 CUs of one XCD share one L2, which is their point of
// coherence: an atomic without sc1 executes there, a store is visible there once it is acknowledged.)
//
// Launch: 256 blocks x 256 threads with enough LDS that a CU holds one; a block reads its XCC id
// (s_getreg HW_REG_XCC_ID) and takes part only when that is `xcd`; the others leave at once.
//  1. placement: blocks per XCC id, and whether block b sits on XCC b % 8.
//  2. barrier among the XCD's blocks, R rounds: arrive = global atomic add (no scope bits: executed in the L2), wait =
//     polling form `poll`: 0 an atomic add of 0 that returns the value, 1 a load with sc0, 2 a load with sc1 (agent
//     scope: the cross-XCD recipe, for comparison).  Every round each block also publishes a word (plain store,
//     s_waitcnt vmcnt(0)), and after the barrier reads its neighbour's word (buffer_inv sc1, plain load): a stale
//     value counts as an error.  Printed: us per barrier, errors.
//  3. one XCD streaming: the XCD's blocks read `mb` MB with non-temporal 16-byte loads (8 in flight per lane), twice
//     over the same buffer (the second pass finds <= 256 MB in the Infinity Cache) -> GB/s into one XCD.
//
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_local_probe scripts/xcd_local_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id()
{
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 0xf;
}

__device__ __forceinline__ int l2_atomic_add(int *p, int v)   // executed in this XCD's L2, returns the old value
{
    int old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ int load_sc0(const int *p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ int load_sc1(const int *p)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ void placement_kernel(int *xcc) { if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id(); }

// ctl[0] ticket, ctl[1] barrier counter, ctl[2] errors, ctl[3] timeouts
__global__ __launch_bounds__(256) void barrier_kernel(int xcd, int W, int R, int poll, int *ctl, int *words, long long *ticks)
{
    extern __shared__ float lds[];
    if (xcc_id() != xcd) return;
    __shared__ int s_me;
    if (threadIdx.x == 0) s_me = l2_atomic_add(ctl + 0, 1);
    __syncthreads();
    const int me = s_me;
    if (me >= W) return;   // more blocks on this XCD than expected: the extras stay out
    int errors = 0;
    long long t0 = 0;
    for (int r = 0; r < R; r++) {
        if (r == 8 && threadIdx.x == 0) t0 = wall_clock64();
        if (threadIdx.x == 0) {
            words[me * 32] = r + 1;                              // plain store ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... acknowledged by the L2
            l2_atomic_add(ctl + 1, 1);
            const int want = W * (r + 1);
            const long long ts = wall_clock64();
            for (;;) {
                const int v = poll == 0 ? l2_atomic_add(ctl + 1, 0) : poll == 1 ? load_sc0(ctl + 1) : load_sc1(ctl + 1);
                if (v >= want) break;
                if (wall_clock64() - ts > 2000000LL) { atomicAdd(ctl + 3, 1); s_me = -1; break; }   // 20 ms: give up
            }
        }
        __syncthreads();
        if (s_me < 0) break;
        asm volatile("buffer_inv sc1" ::: "memory");             // drop this CU's L1 lines
        if (threadIdx.x == 0) {
            const int got = words[((me + 1) % W) * 32];
            if (got < r + 1) errors++;
        }
    }
    if (threadIdx.x == 0) {
        ticks[me] = wall_clock64() - t0;
        if (errors) atomicAdd(ctl + 2, errors);
    }
    if (lds[threadIdx.x] == 123.0f) ticks[0] = 0;
}

__global__ __launch_bounds__(256) void stream_kernel(int xcd, int W, const v4f *__restrict__ p, size_t n4, int *ctl, float *out, long long *ticks)
{
    extern __shared__ float lds[];
    if (xcc_id() != xcd) return;
    __shared__ int s_me;
    if (threadIdx.x == 0) s_me = l2_atomic_add(ctl + 0, 1);
    __syncthreads();
    const int me = s_me;
    if (me >= W) return;
    const long long t0 = wall_clock64();
    constexpr int U = 8;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)me * 256 * U + threadIdx.x; i + 256 * (U - 1) < n4; i += (size_t)W * 256 * U) {
        v4f r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
        for (int k = 0; k < U; k++) acc += r[k];
    }
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 123.456f) out[me] = s;
    __syncthreads();
    if (threadIdx.x == 0) ticks[me] = wall_clock64() - t0;
    if (lds[threadIdx.x] == 123.0f) ticks[0] = 0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int grid = 256, lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void *)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int *xcc, *ctl, *words;
    long long *ticks;
    float *out;
    CK(hipMalloc(&xcc, 4096 * 4)); CK(hipMalloc(&ctl, 64)); CK(hipMalloc(&words, 64 * 32 * 4)); CK(hipMalloc(&ticks, 64 * 8));
    CK(hipMalloc(&out, 4096));
    // 1. placement
    for (int g : {256, 512, 1024}) {
        hipLaunchKernelGGL(placement_kernel, dim3(g), dim3(64), 0, 0, xcc);
        std::vector<int> h(g);
        CK(hipMemcpy(h.data(), xcc, g * 4, hipMemcpyDeviceToHost));
        int cnt[16] = {}, rr = 0;
        for (int b = 0; b < g; b++) { cnt[h[b]]++; rr += h[b] == b % 8; }
        printf("placement, %4d blocks: per XCC", g);
        for (int x = 0; x < 8; x++) printf(" %d", cnt[x]);
        printf("; block b on XCC b %% 8: %d of %d\n", rr, g);
    }
    // 2. barrier
    std::vector<long long> ht(64);
    for (int W : {32, 16, 8}) {
        for (int poll = 0; poll < 3; poll++) {
            const int R = 2008;
            CK(hipMemset(ctl, 0, 64)); CK(hipMemset(words, 0, 64 * 32 * 4));
            hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), lds, 0, 0, W, R, poll, ctl, words, ticks);
            CK(hipDeviceSynchronize());
            int hc[4];
            CK(hipMemcpy(hc, ctl, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ht.data(), ticks, W * 8, hipMemcpyDeviceToHost));
            long long mx = 0;
            for (int i = 0; i < W; i++) mx = ht[i] > mx ? ht[i] : mx;
            printf("barrier among %2d blocks of XCD 0, poll by %s: %.3f us per barrier (publish + arrive + wait + invalidate + read), "
                   "blocks that took a ticket %d, stale reads %d, timeouts %d\n", W,
                   poll == 0 ? "atomic add 0 " : poll == 1 ? "load sc0     " : "load sc1     ", mx / 100.0 / (R - 8), hc[0], hc[2], hc[3]);
        }
    }
    // 3. one XCD streaming
    for (size_t mb : {60, 200, 1024}) {
        const size_t bytes = mb << 20;
        char *buf;
        CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
        for (int W : {32, 16}) {
            for (int pass = 0; pass < 3; pass++) {
                CK(hipMemset(ctl, 0, 64));
                hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), lds, 0, 0, W, (const v4f *)buf, bytes / 16, ctl, out, ticks);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(ht.data(), ticks, W * 8, hipMemcpyDeviceToHost));
                long long mx = 0;
                for (int i = 0; i < W; i++) mx = ht[i] > mx ? ht[i] : mx;
                printf("one XCD, %2d blocks of 256 threads stream %4zu MB, pass %d: %.1f us = %.0f GB/s\n", W, mb, pass, mx / 100.0,
                       bytes / (mx / 100.0 * 1e-6) / 1e9);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
