// mfma_probe.hip -- what fp32 MFMA rate does an MI355X sustain?  (hipcc --offload-arch=gfx950 -O3)
// Pure v_mfma_f32_32x32x2_f32 loops, NACC independent accumulators per wave, W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void probe(float *out, int iters, float a, float b)
{
    v16f acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_cu, int cus)
{
    float *out;
    hipMalloc(&out, (size_t)cus * 64 * waves_per_cu * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(cus), dim3(64 * waves_per_cu), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 2 * 8.0 * NACC * iters * waves_per_cu * cus;
    printf("NACC %d, %2d waves/CU: %.1f TFLOP/s (%.2f ms)\n", NACC, waves_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    run<1>(4, p.multiProcessorCount);
    run<2>(4, p.multiProcessorCount);
    run<4>(4, p.multiProcessorCount);
    run<1>(8, p.multiProcessorCount);
    run<2>(8, p.multiProcessorCount);
    run<4>(8, p.multiProcessorCount);
    run<1>(16, p.multiProcessorCount);
    return 0;
}
