// ipc_probe.cpp -- can two PROCESSES on one GPU ping-pong through IPC-mapped fine-grained device
// memory from inside running kernels?  (feasibility of the peer-write all-gather; not product code)
//   hipcc --offload-arch=gfx950 -O2 ipc_probe.cpp -o ipc_probe && ./ipc_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", getpid(), #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void pingpong(volatile int *mine, volatile int *peer, int rounds, int first, long long *cycles, int *err)
{
    long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        if (first) {
            __hip_atomic_store((int *)peer, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            long long w0 = wall_clock64();
            while (__hip_atomic_load((int *)mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < r)
                if (wall_clock64() - w0 > 500000000LL) { *err = r; return; }  // 5 s at 100 MHz
        } else {
            long long w0 = wall_clock64();
            while (__hip_atomic_load((int *)mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < r)
                if (wall_clock64() - w0 > 500000000LL) { *err = r; return; }
            __hip_atomic_store((int *)peer, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    *cycles = wall_clock64() - t0;
}

int run(int me, int rd, int wr, int finegrained)
{
    CK(hipSetDevice(0));
    int *mine = nullptr;
    if (finegrained) CK(hipExtMallocWithFlags((void **)&mine, 4096, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void **)&mine, 4096));
    CK(hipMemset(mine, 0, 4096));
    hipIpcMemHandle_t h, ph;
    CK(hipIpcGetMemHandle(&h, mine));
    if (write(wr, &h, sizeof h) != (ssize_t)sizeof h) return 3;
    if (read(rd, &ph, sizeof ph) != (ssize_t)sizeof ph) return 3;
    int *peer = nullptr;
    CK(hipIpcOpenMemHandle((void **)&peer, ph, hipIpcMemLazyEnablePeerAccess));
    long long *cyc; int *err;
    CK(hipHostMalloc((void **)&cyc, 8)); CK(hipHostMalloc((void **)&err, 4));
    *cyc = 0; *err = 0;
    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, mine, peer, rounds, me == 0, cyc, err);
    CK(hipDeviceSynchronize());
    printf("[rank %d, %s] err=%d  %d round trips in %lld ticks (100 MHz) = %.2f us per round trip\n", me,
           finegrained ? "fine-grained" : "coarse", *err, rounds, *cyc, *cyc / 100.0 / rounds);
    CK(hipIpcCloseMemHandle(peer));
    return *err ? 4 : 0;
}

int main(int argc, char **argv)
{
    const int fg = argc > 1 ? atoi(argv[1]) : 1;
    int a[2], b[2];
    if (pipe(a) || pipe(b)) return 1;
    pid_t pid = fork();  // before any HIP call
    if (pid == 0) return run(1, a[0], b[1], fg);
    int rc = run(0, b[0], a[1], fg);
    int st = 0;
    waitpid(pid, &st, 0);
    return rc ? rc : WEXITSTATUS(st);
}
