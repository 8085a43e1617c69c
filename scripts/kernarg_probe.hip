// kernarg_probe.hip -- does kernarg preloading (gfx950: -mllvm -amdgpu-kernarg-preload-count=N)
// shorten tiny dependent kernels in a hipGraph?  Chain of 200 one-block-per-CU kernels, each
// reading a pointer argument and doing one dependent load + store.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
struct A { const float *p; float *o; int n; int pad[24]; };
__global__ void k_struct(const A a) { if ((int)threadIdx.x < a.n) a.o[threadIdx.x + blockIdx.x * 256] = a.p[threadIdx.x] + 1.0f; }
__global__ void k_scalar(const float *p, float *o, int n) { if ((int)threadIdx.x < n) o[threadIdx.x + blockIdx.x * 256] = p[threadIdx.x] + 1.0f; }
__global__ void k_empty() {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    float *p, *o;
    CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&o, 1 << 24)); CK(hipMemset(p, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 200;
    for (int variant = 0; variant < 3; variant++) {
        for (int grid : {1, 64, 512}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < N; i++) {
                if (variant == 0) { A a = {}; a.p = p; a.o = o; a.n = 256; hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, a); }
                else if (variant == 1) hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, st, (const float *)p, o, 256);
                else hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e0, st));
            const int reps = 20;
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-8s grid %3d: %.2f us per kernel in a %d-node graph\n", variant == 0 ? "struct" : variant == 1 ? "scalars" : "empty", grid, ms * 1e3 / (reps * N), N);
        }
    }
    return 0;
}
