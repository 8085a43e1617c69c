// bw_probe.hip -- what read bandwidth can a pure streaming kernel reach on this GPU?
// (ceiling for the mat-vec kernels).  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  v4f acc = {0,0,0,0};
  for (; i + 256 * (U - 1) < n4; i += stride) {
    v4f r[U];
#pragma unroll
    for (int k = 0; k < U; k++) r[k] = NT ? __builtin_nontemporal_load(p + i + 256 * k) : p[i + 256 * k];
#pragma unroll
    for (int k = 0; k < U; k++) acc += r[k];
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[blockIdx.x] = s;
}
int main() {
  const size_t slice = 360710144, nsl = 12;  // ffn13-sized slices, 4.3 GB total
  char* buf; hipMalloc(&buf, slice * nsl); hipMemset(buf, 0, slice * nsl);
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name, int grid) {
    float best = 1e9, tot = 0; int n = 0;
    for (int it = 0; it < 24; it++) {
      const v4f* p = (const v4f*)(buf + slice * (it % nsl));
      hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p, slice / 16, out); hipEventRecord(b);
      hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
      if (it >= 4) { tot += ms; n++; if (ms < best) best = ms; }
    }
    printf("%-28s grid %5d  avg %.1f us = %.2f TB/s   best %.1f us = %.2f TB/s\n", name, grid, tot / n * 1e3, slice / (tot / n * 1e-3) / 1e12, best * 1e3, slice / (best * 1e-3) / 1e12);
  };
  for (int grid : {256, 512, 1024, 2048, 4096, 8192}) {
    run(rd<4, true>, "U4 nt", grid); run(rd<8, true>, "U8 nt", grid); run(rd<4, false>, "U4 plain", grid); run(rd<2, true>, "U2 nt", grid);
  }
  return 0;
}
