// coresident_probe.hip -- what does a kernel that is merely RESIDENT in another HW queue cost a chain of
// streaming kernels?  (round 4: the overlapped decode chain ran every launch ~4-5 us longer than the single
// chain; this isolates the platform's share from the hand-over's.)
//
// Chain: K launches on stream A, each a 512-thread-per-CU non-temporal streaming read of `mb` MB (a different
// slice of a 4 GB buffer every launch, so nothing is re-read from cache), optionally publishing 16 LL words per
// block at its end (system-scope 8-byte stores into fine-grained memory, like the decode chain's hand-overs).
// Resident kernel on stream B, alive for the whole chain: G blocks x T threads that sleep (s_sleep) until a
// wall-clock deadline; mode 1: lane 0 of every block also polls one fine-grained word per wake-up.
// Printed: us per launch of the chain alone and beside each resident form.
//
// build: hipcc --offload-arch=gfx950 -O3 -o coresident_probe scripts/coresident_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__global__ __launch_bounds__(512, 4) void stream_kernel(const v4f *__restrict__ p, size_t n4, float *out, u64 *ll, unsigned epoch)
{
    constexpr int U = 8;
    size_t i = (size_t)blockIdx.x * 512 * U + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 512 * U;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 512 * (U - 1) < n4; i += stride) {
        v4f r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = __builtin_nontemporal_load(p + i + 512 * k);
#pragma unroll
        for (int k = 0; k < U; k++) acc += r[k];
    }
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 123.456f) out[blockIdx.x] = s;
    if (ll != nullptr && threadIdx.x < 16)
        __hip_atomic_store(ll + blockIdx.x * 16 + threadIdx.x, ((u64)epoch << 32) | threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void resident_kernel(long long ticks, int mode, const u64 *word, int *sink)
{
    const long long t0 = wall_clock64();
    unsigned seen = 0;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 1 && threadIdx.x == 0)
            seen += (unsigned)(__hip_atomic_load(word + blockIdx.x % 4096, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> 32);
        __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_s_sleep(8);
    }
    if (seen == 0xffffffffu) *sink = 1;
}

int main(int argc, char **argv)
{
    const int K = 640, reps = 3;
    const double mbs[4] = {201.3, 67.1, 360.7, 180.4};  // the 7B layer's four mat-vecs
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t buf_bytes = (size_t)4 << 30;
    float *buf, *out; u64 *ll; int *sink;
    CK(hipMalloc(&buf, buf_bytes)); CK(hipMemset(buf, 0, buf_bytes));
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipExtMallocWithFlags((void **)&ll, 4096 * 16 * 8, hipDeviceMallocFinegrained)); CK(hipMemset(ll, 0, 4096 * 16 * 8));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // the chain as a graph (what the product replays)
    auto enqueue_chain = [&](bool publish) {
        size_t off = 0;
        for (int k = 0; k < K; k++) {
            const size_t bytes = ((size_t)(mbs[k % 4] * 1e6) / 65536) * 65536;
            if (off + bytes > buf_bytes) off = 0;
            hipLaunchKernelGGL(stream_kernel, dim3(cus), dim3(512), 0, sa, (const v4f *)(buf + off / 4), bytes / 16, out,
                               publish ? ll : nullptr, (unsigned)(k + 1));
            off += bytes;
        }
    };
    hipGraphExec_t gx[2];
    for (int pub = 0; pub < 2; pub++) {
        hipGraph_t g; CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal)); enqueue_chain(pub != 0);
        CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&gx[pub], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    }
    double total_mb = 0; for (int k = 0; k < K; k++) total_mb += mbs[k % 4];
    struct Res { int g, t, mode; };
    const Res forms[] = {{0, 0, 0}, {1, 64, 0}, {cus, 64, 0}, {cus, 512, 0}, {cus, 512, 1}, {0, 0, 0}};
    for (int pub = 0; pub < 2; pub++) {
        for (const Res &f : forms) {
            double best = 1e30, sum = 0;
            for (int r = 0; r < reps + 1; r++) {
                CK(hipDeviceSynchronize());
                if (f.g) hipLaunchKernelGGL(resident_kernel, dim3(f.g), dim3(f.t), 0, sb, (long long)(0.25 * 1e8), f.mode, ll, sink);  // 250 ms
                CK(hipEventRecord(e0, sa));
                CK(hipGraphLaunch(gx[pub], sa));
                CK(hipEventRecord(e1, sa));
                CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipDeviceSynchronize());
                if (r > 0) { sum += ms; if (ms < best) best = ms; }
            }
            const double us = sum / reps * 1e3 / K;
            printf("chain of %d launches (%s): resident %3d blocks x %3d threads mode %d: %7.2f us per launch (best %7.2f), %5.2f TB/s\n", K,
                   pub ? "publishing LL words" : "no publish", f.g, f.t, f.mode, us, best * 1e3 / K, total_mb / K * 1e6 / (us * 1e-6) / 1e12);
        }
    }
    return 0;
}
