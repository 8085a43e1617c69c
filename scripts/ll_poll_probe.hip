// What does it cost every block of a launch to sweep a gathered vector of LL words {value, epoch}
// out of memory?  The consumer-side gather of the sharded decode (kernel_common.h xstage_finish_ll)
// does exactly this in every mat-vec block: 4096 .. 11008 words of 8 bytes, two per 16-byte load.
// Variants: memory kind (fine-grained = what peers can write into, coarse-grained) x load flavour
// (system scope sc0 sc1, agent scope sc1, plain) x blocks.  Prints us per launch (graph of 50).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int AUX>
__global__ __launch_bounds__(256) void sweep(const unsigned long long *slot, int n_words, unsigned e, float *out)
{
    __shared__ float xs[11008 + 64];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long *>(slot), 0, 0x7ffffff0, 0x00020000);
    float acc = 0.f;
    for (int j = threadIdx.x; j < n_words / 4; j += 256 * 4) {
        v4u a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int jj = j + 256 * k < n_words / 4 ? j + 256 * k : 0;
            a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, jj * 32, 0, AUX);
            b[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, jj * 32 + 16, 0, AUX);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int jj = j + 256 * k;
            if (jj < n_words / 4) {
                const bool ok = a[k].y == e && a[k].w == e && b[k].y == e && b[k].w == e;
                xs[4 * jj] = ok ? __uint_as_float(a[k].x) : 0.f;
                xs[4 * jj + 1] = __uint_as_float(a[k].z);
                xs[4 * jj + 2] = __uint_as_float(b[k].x);
                xs[4 * jj + 3] = __uint_as_float(b[k].z);
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_words; j += 256) acc += xs[j];
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

__global__ void fill(unsigned long long *slot, int n, unsigned e)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        slot[i] = ((unsigned long long)e << 32) | (unsigned)__float_as_uint(1.0f + i);
}

template <int AUX>
float run(const unsigned long long *slot, int n_words, int blocks, float *out, hipStream_t st)
{
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(sweep<AUX>, dim3(blocks), dim3(256), 0, st, slot, n_words, 7u, out);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 50 * 1e3f;
}

int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long *fine, *coarse; float *out;
    const int cap = 65536;
    CK(hipExtMallocWithFlags((void **)&fine, cap * 8, hipDeviceMallocFinegrained));
    CK(hipMalloc((void **)&coarse, cap * 8)); CK(hipMalloc((void **)&out, 4096 * 4));
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, fine, cap, 7u);
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, coarse, cap, 7u);
    CK(hipStreamSynchronize(st));
    for (int n_words : {4096, 11008})
        for (int blocks : {64, 256, 512}) {
            printf("words %5d blocks %3d | fine: sys %6.2f  agent %6.2f  plain %6.2f | coarse: sys %6.2f  agent %6.2f  plain %6.2f  us (an empty graph node costs ~1.6)\n",
                   n_words, blocks, run<17>(fine, n_words, blocks, out, st), run<16>(fine, n_words, blocks, out, st),
                   run<0>(fine, n_words, blocks, out, st), run<17>(coarse, n_words, blocks, out, st),
                   run<16>(coarse, n_words, blocks, out, st), run<0>(coarse, n_words, blocks, out, st));
        }
    return 0;
}
