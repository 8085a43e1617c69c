// mv_probe2.cpp -- per-shape launch time of the library's mat-vec vs a pure streaming read
// of the same bytes, 32 back-to-back launches over distinct memory.  Not product code.
#include <cstdio>
#include <cstdlib>
#include "../llama2.zig_amd/csrc/l2z_internal.h"
using namespace l2z;
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; const size_t stride = (size_t)gridDim.x * 1024;
  v4f acc = {0,0,0,0};
  for (; i + 768 < n4; i += stride) { v4f r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) acc += r[k]; }
  float s = acc.x + acc.y + acc.z + acc.w; if (s == 123.456f) out[blockIdx.x] = s;
}
int main() {
  const int dim = 4096, hid = 11008, L = 32;
  struct Shape { const char* name; int segs, rows, n, pro, epi; } shapes[] = {
    {"qkv  3x(4096,4096) rms+store", 3, dim, dim, PRO_RMS, EPI_STORE}, {"wo   (4096,4096) resid", 1, dim, dim, PRO_NONE, EPI_RESID},
    {"ffn13 2x(11008,4096) rms+swiglu", 2, hid, dim, PRO_RMS, EPI_SWIGLU}, {"ffn2 (4096,11008) resid", 1, dim, hid, PRO_NONE, EPI_RESID},
    {"wo   (4096,4096) store", 1, dim, dim, PRO_NONE, EPI_STORE}};
  float *x, *o, *rms, *out; hipMalloc(&x, hid * 4); hipMalloc(&o, 3 * hid * 4); hipMalloc(&rms, hid * 4); hipMalloc(&out, 1 << 20);
  hipMemset(x, 0, hid * 4); hipMemset(rms, 0, hid * 4); hipMemset(o, 0, 3 * hid * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (auto& s : shapes) {
    const size_t bytes = (size_t)4 * s.segs * s.rows * s.n;
    float* w; hipMalloc(&w, bytes * L); hipMemset(w, 0, bytes * L);
    for (int mode = 0; mode < 2; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        for (int l = 0; l < L; l++) {
          float* p = w + (bytes / 4) * l;
          if (mode == 0) hipLaunchKernelGGL(rd, dim3(512), dim3(256), 0, 0, (const v4f*)p, bytes / 16, out);
          else { MatvecArgs m = {}; m.w0 = p; m.out0 = o; m.rows0 = s.rows; m.n = s.n; m.x = x; m.rms_w = rms; m.resid = o;
            if (s.segs > 1) { m.w1 = p + (size_t)s.rows * s.n; m.out1 = o + hid; m.rows1 = s.rows; }
            if (s.segs > 2) { m.w2 = p + 2 * (size_t)s.rows * s.n; m.out2 = o + 2 * hid; m.rows2 = s.rows; }
            launch_matvec(m, s.pro, s.epi, 8, 256, nullptr); }
        }
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      printf("%-34s %s: %.1f us/launch = %.2f TB/s\n", s.name, mode ? "matvec" : "pure  ", best * 1e3 / L, bytes * L / (best * 1e-3) / 1e12);
    }
    hipFree(w);
  }
  return 0;
}
