// bw_probe2.hip -- ceiling of a launch-per-matrix pipeline: the 7B layer's five weight
// streams (qkv 201 MB, wo 67 MB, ffn13 361 MB, ffn2 180 MB) as back-to-back pure
// streaming-read kernels with NO prologue, over 32 "layers" of distinct memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  v4f acc = {0,0,0,0};
  for (; i + 256 * (U - 1) < n4; i += stride) {
    v4f r[U];
#pragma unroll
    for (int k = 0; k < U; k++) r[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
    for (int k = 0; k < U; k++) acc += r[k];
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[blockIdx.x] = s;
}
int main() {
  const size_t sz[4] = {201326592, 67108864, 360710144, 180355072};
  size_t layer = 0; for (size_t s : sz) layer += s;
  const int L = 32;
  char* buf; hipMalloc(&buf, layer * L); hipMemset(buf, 0, layer * L);
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int grid : {512, 1024, 2048}) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      for (int l = 0; l < L; l++) {
        size_t off = layer * l;
        for (int k = 0; k < 4; k++) { hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + off), sz[k] / 16, out); off += sz[k]; }
      }
      hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      printf("grid %d: %d layers x 4 streams = %.2f GB in %.3f ms = %.2f TB/s (%.1f us per launch overhead vs 6.9 TB/s)\n", grid, L, layer * L / 1e9, ms, layer * L / (ms * 1e-3) / 1e12,
             (ms * 1e3 - layer * L / 6.9e6) / (L * 4));
    }
  }
  return 0;
}
