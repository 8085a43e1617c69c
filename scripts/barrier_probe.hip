// barrier_probe.hip -- what does a device-wide barrier cost inside a persistent kernel on MI355X
// (8 XCDs, L2 per XCD)?  One block per CU, N barriers; flat counter vs per-XCD counters + a
// second level.  Bounded spins: a bug ends the kernel instead of hanging the GPU.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool spin_until(const int *p, int target, long long limit)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
        if (wall_clock64() - t0 > limit) return false;
    }
    return true;
}

// flat: every block adds to one counter, spins on it
__global__ void flat_kernel(int *ctr, int n_bar, int *err, long long *ticks)
{
    const int nb = gridDim.x;
    const long long t0 = wall_clock64();
    for (int b = 1; b <= n_bar; b++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!spin_until(ctr, b * nb, 20000000LL)) { *err = b; return; }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = wall_clock64() - t0;
}

// two-level: blocks of one XCD meet on that XCD's counter; the last arrival adds to the global
// counter and, when the global count completes, bumps the generation flag everyone spins on
__global__ void tree_kernel(int *xcd_ctr /*[8]*/, int *glob, int *gen, int n_bar, int *err, long long *ticks, int *xcd_count /*[8] blocks per xcd*/)
{
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    __shared__ int s_n;
    if (threadIdx.x == 0) {  // count the blocks of this XCD (once)
        __hip_atomic_fetch_add(&xcd_count[xcc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // first barrier (flat) so that xcd_count is final
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(glob + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_until(glob + 1, (int)gridDim.x, 200000000LL)) *err = -1;
        s_n = __hip_atomic_load(&xcd_count[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int n_local = s_n;
    const long long t0 = wall_clock64();
    for (int b = 1; b <= n_bar; b++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(&xcd_ctr[xcc * 32], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == b * n_local) {  // last of this XCD
                const int g = __hip_atomic_fetch_add(glob, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (g + 1 == b * 8) __hip_atomic_store(gen, b, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!spin_until(gen, b, 20000000LL)) { *err = b; return; }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = wall_clock64() - t0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int *d; CK(hipMalloc(&d, 4096 * 4));
    int *err; long long *ticks;  // managed by the device; read back after the kernel
    CK(hipMallocManaged(&err, 4)); CK(hipMallocManaged(&ticks, 8));
    const int n_bar = 2000;
    for (int grid : {32, 64, 128, cus, 2 * cus}) {
        for (int threads : {64, 256}) {
            CK(hipMemset(d, 0, 4096 * 4)); *err = 0; *ticks = 0;
            void *args[] = {&d, (void *)&n_bar, &err, &ticks};
            int *ctr = d;
            void *a2[] = {&ctr, (void *)&n_bar, &err, &ticks};
            CK(hipLaunchCooperativeKernel((const void *)flat_kernel, dim3(grid), dim3(threads), a2, 0, 0));
            CK(hipDeviceSynchronize());
            printf("flat  grid %3d x %3d thr: err %d, %.2f us per barrier\n", grid, threads, *err, *ticks / 100.0 / n_bar);
            (void)args;
            CK(hipMemset(d, 0, 4096 * 4)); *err = 0; *ticks = 0;
            int *xc = d, *glob = d + 1024, *gen = d + 1100, *xcount = d + 1200;
            void *a3[] = {&xc, &glob, &gen, (void *)&n_bar, &err, &ticks, &xcount};
            CK(hipLaunchCooperativeKernel((const void *)tree_kernel, dim3(grid), dim3(threads), a3, 0, 0));
            CK(hipDeviceSynchronize());
            printf("tree  grid %3d x %3d thr: err %d, %.2f us per barrier\n", grid, threads, *err, *ticks / 100.0 / n_bar);
        }
    }
    return 0;
}
