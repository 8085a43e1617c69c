// xcd_probe.hip -- is the per-XCD streaming rate imbalance stable?  512 blocks each stream a
// fixed, equal share of a 524 MB buffer (static striding like the row kernel); per-block
// wall-clock durations are averaged by XCC id.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int ROT>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n4, float* out, long long* ts, int* xcc) {
  const long long t0 = wall_clock64();
  const size_t nchunks = n4 / 2048;
  v4f acc = {0,0,0,0};
  for (size_t it = 0; it * gridDim.x < nchunks; it++) {
    const size_t c = it * gridDim.x + (blockIdx.x + it * ROT) % gridDim.x;   // ROT: rotate the block->chunk map per round
    if (c >= nchunks) break;
    const size_t i = c * 2048 + threadIdx.x;
    v4f r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 8; k++) acc += r[k]; }
  float s = acc.x + acc.y + acc.z + acc.w; if (s == 123.456f) out[blockIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = wall_clock64();
    int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id & 0xf; }
}
template <int WE, int WO>
__global__ __launch_bounds__(256) void rdw(const v4f* __restrict__ p, size_t n4, float* out, long long* ts, int* xcc) {
  const long long t0 = wall_clock64();
  const size_t nchunks = n4 / 2048;
  const int pair = blockIdx.x >> 1, odd = blockIdx.x & 1, mine = odd ? WO : WE, first = odd ? WE : 0;
  v4f acc = {0,0,0,0};
  for (size_t j = 0;; j++) {
    const size_t li = (j / mine) * (WE + WO) + first + (j % mine);   // index in the pair's merged list
    const size_t c = (li >> 1) * gridDim.x + 2 * pair + (li & 1);
    if (c >= nchunks) break;
    const size_t i = c * 2048 + threadIdx.x;
    v4f r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = __builtin_nontemporal_load(p + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 8; k++) acc += r[k]; }
  float s = acc.x + acc.y + acc.z + acc.w; if (s == 123.456f) out[blockIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = wall_clock64();
    int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id & 0xf; }
}
int main() {
  const size_t bytes = 524288000, nsl = 6; const int grid = 512;
  char* buf; hipMalloc(&buf, bytes * nsl); hipMemset(buf, 0, bytes * nsl);
  float* out; hipMalloc(&out, 1 << 20); long long* ts; hipMalloc(&ts, grid * 16); int* xcc; hipMalloc(&xcc, grid * 4);
  std::vector<long long> h(grid * 2); std::vector<int> hx(grid);
  for (int it = 0; it < 12; it++) {
    if (it % 3 == 0) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + bytes * ((it / 3) % nsl)), bytes / 16, out, ts, xcc);
    else if (it % 3 == 1) hipLaunchKernelGGL((rdw<31, 29>), dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + bytes * ((it / 3) % nsl)), bytes / 16, out, ts, xcc);
    else if (it >= 6) hipLaunchKernelGGL((rdw<32, 30>), dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + bytes * ((it / 3) % nsl)), bytes / 16, out, ts, xcc);
    else hipLaunchKernelGGL(rd<3>, dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + bytes * ((it / 3) % nsl)), bytes / 16, out, ts, xcc);
    if (0) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, (const v4f*)(buf + bytes * (it % nsl)), bytes / 16, out, ts, xcc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ts, grid * 16, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
    double sum[16] = {0}; int cnt[16] = {0}; long long t0 = h[0], t1 = h[1]; int mism = 0;
    for (int b = 0; b < grid; b++) { sum[hx[b]] += (h[2*b+1] - h[2*b]) / 100.0; cnt[hx[b]]++; if (h[2*b] < t0) t0 = h[2*b]; if (h[2*b+1] > t1) t1 = h[2*b+1]; if (hx[b] != b % 8) mism++; }
    printf("launch %d rot %d: kernel %.1f us; mean block time by XCC:", it, it % 3 == 0 ? 0 : (it % 3 == 1 ? 1 : 3), (t1 - t0) / 100.0);
    for (int x = 0; x < 8; x++) printf(" %.1f", cnt[x] ? sum[x] / cnt[x] : 0.0);
    printf("  (blocks not on XCC b%%8: %d)\n", mism);
  }
  return 0;
}
