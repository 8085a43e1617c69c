// Does data written by one kernel come back faster to the NEXT kernel when the reader runs on the
// SAME XCD as the writer?  (Per-XCD L2s; what happens to their contents at a kernel boundary decides
// whether pinning the small models' tiny launches to one XCD would shorten their x-load latency.)
// Block b runs on XCD b % 8 (observed).  Writer blocks on XCD `xw` store 16 KB; reader blocks on XCD
// `xr` time one dependent 16-byte load per lane of that data with s_memtime.  Also: end-to-end time
// of a 2-kernel chain per placement (graph of 40 pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void writer(float *buf, int xw, float v)
{
    if ((int)(blockIdx.x & 7) != xw) return;
    const int slot = blockIdx.x >> 3;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[slot * 4096 + i] = v + i;
}
__global__ void reader(const float *buf, int xr, long long *cyc, float *sink, int n_slots)
{
    if ((int)(blockIdx.x & 7) != xr) return;
    const int slot = (blockIdx.x >> 3) % n_slots;
    const long long t0 = clock64();
    const v4f a = ((const v4f *)(buf + slot * 4096))[threadIdx.x];
    const float s = a.x + a.y + a.z + a.w;
    const long long t1 = clock64();
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x >> 3] = t1 - t0 + (long long)(s == 7.0f);
}
int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    float *buf, *sink; long long *cyc;
    const int n_slots = 8;
    CK(hipMalloc(&buf, n_slots * 4096 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, n_slots * 8));
    for (int xw : {0, 3}) for (int xr : {0, 3, 5}) {
        // latency seen by the reader's first load
        double tot = 0; int cnt = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(writer, dim3(8 * n_slots), dim3(256), 0, st, buf, xw, (float)rep);
            hipLaunchKernelGGL(reader, dim3(8 * n_slots), dim3(256), 0, st, buf, xr, cyc, sink, n_slots);
            CK(hipStreamSynchronize(st));
            long long h[8]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            if (rep >= 4) for (int i = 0; i < n_slots; i++) { tot += (double)h[i]; cnt++; }
        }
        // chain time
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 40; i++) {
            hipLaunchKernelGGL(writer, dim3(8 * n_slots), dim3(256), 0, st, buf, xw, (float)i);
            hipLaunchKernelGGL(reader, dim3(8 * n_slots), dim3(256), 0, st, buf, xr, cyc, sink, n_slots);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("writer XCD %d -> reader XCD %d: first load %.0f cycles (s_memtime), chain %.2f us per kernel\n", xw, xr, tot / cnt, ms / 80 * 1e3);
    }
    return 0;
}
