// mv_probe.cpp -- the library's own mat-vec launcher on the 7B layer shapes, back to back
// over 32 layers of distinct memory, without attention: how far is the kernel from the
// pure-stream pipeline of bw_probe2 (6.8 TB/s)?   Not product code.
#include <cstdio>
#include <cstdlib>
#include "../llama2.zig_amd/csrc/l2z_internal.h"
using namespace l2z;
int main(int argc, char** argv) {
  const int pro = argc > 1 ? atoi(argv[1]) : 0;
  const int maxb = argc > 2 ? atoi(argv[2]) : 8;
  const int dim = 4096, hid = 11008, L = 32;
  const size_t per_layer = (size_t)4 * (4 * (size_t)dim * dim + 3 * (size_t)hid * dim);
  float* w; if (hipMalloc(&w, per_layer * L) != hipSuccess) { printf("oom\n"); return 1; }
  hipMemset(w, 0, per_layer * L);
  float *x, *xh, *q, *hb, *rms; hipMalloc(&x, hid * 4); hipMalloc(&xh, hid * 4); hipMalloc(&q, 3 * dim * 4); hipMalloc(&hb, 2 * hid * 4); hipMalloc(&rms, hid * 4);
  hipMemset(x, 0, hid * 4); hipMemset(xh, 0, hid * 4); hipMemset(rms, 0, hid * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(a);
    for (int l = 0; l < L; l++) {
      float* p = w + (per_layer / 4) * l;
      MatvecArgs m = {};
      // qkv: 3 x (dim, dim)
      m = {}; m.w0 = p; m.w1 = p + (size_t)dim * dim; m.w2 = p + 2 * (size_t)dim * dim; m.out0 = q; m.out1 = q + dim; m.out2 = q + 2 * dim;
      m.rows0 = m.rows1 = m.rows2 = dim; m.n = dim; m.x = x; m.rms_w = rms;
      if (launch_matvec(m, pro, EPI_STORE, maxb, 256, nullptr) != hipSuccess) { printf("launch failed\n"); return 1; }
      p += 3 * (size_t)dim * dim;
      m = {}; m.w0 = p; m.out0 = q; m.rows0 = dim; m.n = dim; m.x = x; m.rms_w = rms;          // wo
      launch_matvec(m, PRO_NONE, EPI_STORE, maxb, 256, nullptr);
      p += (size_t)dim * dim;
      m = {}; m.w0 = p; m.w1 = p + (size_t)hid * dim; m.out0 = hb; m.out1 = hb + hid; m.rows0 = m.rows1 = hid; m.n = dim; m.x = x; m.rms_w = rms;  // w1,w3
      launch_matvec(m, pro, EPI_STORE, maxb, 256, nullptr);
      p += 2 * (size_t)hid * dim;
      m = {}; m.w0 = p; m.out0 = q; m.rows0 = dim; m.n = hid; m.x = xh; m.rms_w = rms;          // w2
      launch_matvec(m, PRO_NONE, EPI_STORE, maxb, 256, nullptr);
    }
    hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    printf("pro=%d maxb=%d: %.2f GB in %.3f ms = %.2f TB/s\n", pro, maxb, per_layer * L / 1e9, ms, per_layer * L / (ms * 1e-3) / 1e12);
  }
  return 0;
}
