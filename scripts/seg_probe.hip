// seg_probe.hip -- read rate of a matrix streamed in 64-byte row segments (the 16x16x4 "skinny"
// prefill kernel's pattern: lane (j, q) loads W[row j][k0 + 4q .. +3]) against 256-byte and
// 1-KB segments.  Pure reads, nt loads.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
// block = 16*RPW... rows; wave w of 8 handles k chunks w, w+8, ...; SEG = lanes per row (4, 16, 64)
template <int SEG>
__global__ __launch_bounds__(512) void rd(const float *w, int K, float *out)
{
    constexpr int ROWS = 64 / SEG;            // rows covered by one wave-wide load
    constexpr int NU = 16 / ROWS;             // loads to cover 16 rows x (SEG*4) k
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16;
    const int r = lane / SEG, c = lane % SEG;
    const int chunk_k = SEG * 4 * (SEG == 4 ? 4 : 1);   // k per chunk: 64 for SEG 4 (4 loads along k), else SEG*4
    v4f acc = {0, 0, 0, 0};
    for (int k0 = wave * chunk_k; k0 + chunk_k <= K; k0 += 8 * chunk_k) {
        v4f v[4 > NU ? 4 : NU];
        if (SEG == 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load((const v4f *)(w + (size_t)(n0 + r) * K + k0 + 16 * u + 4 * c));
#pragma unroll
            for (int u = 0; u < 4; u++) acc += v[u];
        } else {
#pragma unroll
            for (int u = 0; u < NU; u++) v[u] = __builtin_nontemporal_load((const v4f *)(w + (size_t)(n0 + ROWS * u + r) * K + k0 + 4 * c));
#pragma unroll
            for (int u = 0; u < NU; u++) acc += v[u];
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;
}
int main()
{
    const int N = 11008, K = 4096, NM = 24;  // 24 different matrices: nothing is re-read from cache
    float *w; hipMalloc(&w, (size_t)N * K * 4 * NM); hipMemset(w, 0, (size_t)N * K * 4 * NM);
    float *out; hipMalloc(&out, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char *name) {
        float tot = 0; int n = 0;
        for (int it = 0; it < NM; it++) {
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(N / 16), dim3(512), 0, 0, w + (size_t)it * N * K, K, out); hipEventRecord(b);
            hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 4) { tot += ms; n++; }
        }
        printf("%-28s %.1f us = %.2f TB/s\n", name, tot / n * 1e3, (double)N * K * 4 / (tot / n * 1e-3) / 1e12);
    };
    run(rd<4>, "64-byte row segments");
    run(rd<16>, "256-byte row segments");
    run(rd<64>, "1-KB row segments");
    return 0;
}
