// dma_pattern_probe.hip -- how fast does a [N x 4096] fp32 matrix stream into the CUs for the access patterns
// the short-prompt GEMM forms can choose from?  No arithmetic: loads only (direct-to-LDS or to registers), rings
// and waits as the kernels have them.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 scripts/dma_pattern_probe.hip -o scripts/dma_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int K = 4096, PITCH = 264;

__device__ __forceinline__ void dma16(const float *g, float *l, bool nt)
{
    if (nt) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Pattern A: a block owns groups of 16 rows; a stage is 16 rows x 256 k (1 KB per row); W waves, wave w brings
// rows (16 / W) w ...; ring of SW stages, a barrier per stage (skinny / slab).  Groups g = b, b + nb, ...;
// k window [k0, k0 + kc) per block (kc = K: whole rows).
template <int W, int SW, bool BARRIER>
__global__ __launch_bounds__(64 * W) void pat_stage(const float *w, int n_groups, int kc, int nslice)
{
    extern __shared__ float smem[];
    constexpr int RPW = 16 / W;  // rows per wave and stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.y, nb = gridDim.x, b = blockIdx.x;
    const int nst = kc / 256;
    int n_units = 0;
    for (int g = b; g < n_groups; g += nb) n_units++;
    const int total = n_units * nst;
    int issued = 0, done = 0, nbuf = 0, iu = 0, ist = 0;
    auto issue = [&]() {
        const int g = b + iu * nb;
#pragma unroll
        for (int r = 0; r < RPW; r++)
            dma16(w + (size_t)(16 * g + RPW * wave + r) * K + (size_t)s * kc + (size_t)ist * 256 + 4 * lane,
                  smem + nbuf * 16 * PITCH + (RPW * wave + r) * PITCH, true);
        nbuf = nbuf + 1 == SW ? 0 : nbuf + 1;
        issued++;
        if (++ist == nst) { ist = 0; iu++; }
    };
    for (int p = 0; p < SW - 1 && issued < total; p++) issue();
    for (; done < total; done++) {
        const int younger = issued - done - 1;
        if (younger >= 6) wait_vm<6 * RPW>();
        else if (younger == 5) wait_vm<5 * RPW>();
        else if (younger == 4) wait_vm<4 * RPW>();
        else if (younger == 3) wait_vm<3 * RPW>();
        else if (younger == 2) wait_vm<2 * RPW>();
        else if (younger == 1) wait_vm<RPW>();
        else wait_vm<0>();
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (issued < total) issue();
    }
    (void)nslice;
}

// Pattern B: every wave streams whole rows (or its k window of them) as consecutive 1-KB loads, a private ring of
// SLOTS 1-KB slots, no barrier: wave (b, w) takes rows r = (b W + w), + nb W, ...
template <int W, int SLOTS>
__global__ __launch_bounds__(64 * W) void pat_rows(const float *w, int n_rows, int kc, int nslice)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mine = smem + wave * SLOTS * PITCH;
    const int s = blockIdx.y, stride = gridDim.x * W;
    const int per_row = kc / 256;
    int n_mine = 0;
    for (int r = blockIdx.x * W + wave; r < n_rows; r += stride) n_mine++;
    const int total = n_mine * per_row;
    int issued = 0, ir = 0, ik = 0, slot = 0;
    auto issue = [&]() {
        const int r = blockIdx.x * W + wave + ir * stride;
        dma16(w + (size_t)r * K + (size_t)s * kc + (size_t)ik * 256 + 4 * lane, mine + slot * PITCH, true);
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
        issued++;
        if (++ik == per_row) { ik = 0; ir++; }
    };
    for (int p = 0; p < SLOTS - 1 && issued < total; p++) issue();
    for (int done = 0; done < total; done++) {
        if (issued - done - 1 >= SLOTS - 2) wait_vm<SLOTS - 2>();
        else wait_vm<0>();
        if (issued < total) issue();
    }
    (void)nslice;
}

// Pattern C: the same row streams through registers (global_load_dwordx4 nt, U loads in flight per lane)
template <int W, int U>
__global__ __launch_bounds__(64 * W) void pat_rows_reg(const float *w, int n_rows, int kc, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.y, stride = gridDim.x * W;
    v4f acc = {0, 0, 0, 0};
    for (int r = blockIdx.x * W + wave; r < n_rows; r += stride) {
        const v4f *p = (const v4f *)(w + (size_t)r * K + (size_t)s * kc) + lane;
        for (int k = 0; k < kc / 256; k += U) {
            v4f v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * 64);
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}


// Pattern D (round 5): the panel kernel's walk -- items (k range, group of R x W rows) range-major, dealt in equal
// contiguous spans to one persistent block per CU; wave w streams R rows of the item, a stage is R rows x SKB bytes,
// a private ring of DEPTH stages, no barrier.  (R, SKB) = (16, 512): prefill_panel at one / two token tiles;
// (4, 1024): what a 4x4x1-MFMA form (4 rows per wave) would read.
template <int W, int R, int SKB, int DEPTH, int ORDER = 0>
__global__ __launch_bounds__(64 * W) void pat_panel(const float *w, int n_rows, int kr)
{
    extern __shared__ float smem[];
    constexpr int LPR = SKB / 1024 > 0 ? SKB / 1024 : 1;       // wave loads per row and stage
    constexpr int RPL = SKB >= 1024 ? 1 : 1024 / SKB;           // rows per wave load
    constexpr int LOADS = R * SKB / 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *ring = smem + wave * DEPTH * (R * SKB / 4);
    const int n_groups = n_rows / (R * W), n_ranges = K / kr, n_items = n_ranges * n_groups, nst = kr * 4 / SKB;
    const int i0 = (int)((long long)blockIdx.x * n_items / gridDim.x), i1 = (int)((long long)(blockIdx.x + 1) * n_items / gridDim.x);
    const int total = (i1 - i0) * nst;
    int issued = 0, item = i0, st = 0, buf = 0;
    auto issue = [&]() {
        // ORDER 0: range-major (the shipped walk); 1: row-group-major, ranges 0, 1, ...; 2: row-group-major, group g
        // starting at range g % n_ranges and wrapping (what a no-partials form with a double-buffered panel would read)
        int r = item / n_groups, g = item - r * n_groups;
        if (ORDER >= 1) { g = item / n_ranges; r = item - g * n_ranges; }
        if (ORDER == 2) r = (r + g) % n_ranges;
        const float *base = w + (size_t)(g * R * W + wave * R) * K + (size_t)r * kr + (size_t)st * (SKB / 4);
        float *dst = ring + buf * (R * SKB / 4);
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            const int row = SKB >= 1024 ? i / LPR : RPL * i + lane / (SKB / 16);
            const int off = SKB >= 1024 ? (i % LPR) * 256 + 4 * lane : 4 * (lane % (SKB / 16));
            dma16(base + (size_t)row * K + off, dst + i * 256, true);
        }
        buf = buf + 1 == DEPTH ? 0 : buf + 1;
        issued++;
        if (++st == nst) { st = 0; item++; }
    };
    for (int p = 0; p < DEPTH - 1 && issued < total; p++) issue();
    for (int done = 0; done < total; done++) {
        const int younger = issued - done - 1;
        if (younger >= 2 && DEPTH >= 4) wait_vm<2 * LOADS>();
        else if (younger >= 1) wait_vm<LOADS>();
        else wait_vm<0>();
        if (issued < total) issue();
    }
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int idx = 0;
    const int N = 22016;  // W1 | W3 of the 7B shape: 361 MB
    const size_t bytes = (size_t)N * K * 4, nsl = 8;
    char *buf; hipMalloc(&buf, bytes * nsl); hipMemset(buf, 0, bytes * nsl);
    float *out; hipMalloc(&out, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto launch) {
        if (only >= 0 && idx++ != only) return;
        printf("%s ...\n", name); fflush(stdout);
        float tot = 0, best = 1e9; int n = 0;
        for (int it = 0; it < 20; it++) {
            const float *p = (const float *)(buf + bytes * (it % nsl));
            hipEventRecord(a); launch(p); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 4) { tot += ms; n++; if (ms < best) best = ms; }
        }
        printf("%-64s avg %6.1f us = %.2f TB/s   best %6.1f us = %.2f TB/s\n", name, tot / n * 1e3, bytes / (tot / n * 1e-3) / 1e12,
               best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
#define STAGE(W, SW, BAR, nbx, S)                                                                                         \
    {                                                                                                                     \
        auto k = pat_stage<W, SW, BAR>;                                                                                   \
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, SW * 16 * PITCH * 4);            \
        char nm[96]; snprintf(nm, 96, "stage-major 16 rows x 1 KB: %d waves ring %d %s, %d x %d blocks", W, SW, BAR ? "barrier" : "no barrier", nbx, S); \
        run(nm, [&](const float *p) { hipLaunchKernelGGL(k, dim3(nbx, S), dim3(64 * W), SW * 16 * PITCH * 4, 0, p, N / 16, K / S, S); }); \
    }
    STAGE(4, 8, true, 256, 1) STAGE(4, 8, false, 256, 1) STAGE(8, 8, true, 256, 1) STAGE(4, 4, true, 256, 1)
    STAGE(4, 4, true, 512, 1) STAGE(4, 8, true, 64, 4) STAGE(8, 8, true, 64, 4) STAGE(4, 8, true, 32, 8) STAGE(16, 8, true, 256, 1)
#define ROWS(W, SLOTS, nbx, S)                                                                                            \
    {                                                                                                                     \
        auto k = pat_rows<W, SLOTS>;                                                                                      \
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, W * SLOTS * PITCH * 4);          \
        char nm[96]; snprintf(nm, 96, "row streams by direct-to-LDS: %d waves x %d slots, %d x %d blocks", W, SLOTS, nbx, S); \
        run(nm, [&](const float *p) { hipLaunchKernelGGL(k, dim3(nbx, S), dim3(64 * W), W * SLOTS * PITCH * 4, 0, p, N, K / S, S); }); \
    }
    ROWS(4, 16, 256, 1) ROWS(8, 16, 256, 1) ROWS(4, 32, 256, 1) ROWS(8, 8, 256, 1) ROWS(4, 16, 512, 1) ROWS(16, 8, 256, 1) ROWS(8, 16, 64, 4)
#define REG(W, U, nbx, S)                                                                                                 \
    {                                                                                                                     \
        char nm[96]; snprintf(nm, 96, "row streams through registers: %d waves x %d loads, %d x %d blocks", W, U, nbx, S);   \
        run(nm, [&](const float *p) { hipLaunchKernelGGL((pat_rows_reg<W, U>), dim3(nbx, S), dim3(64 * W), 0, 0, p, N, K / S, out); }); \
    }
    REG(4, 8, 256, 1) REG(4, 8, 512, 1) REG(8, 8, 256, 1) REG(4, 16, 256, 1) REG(4, 8, 1024, 1) REG(4, 4, 1024, 1)
#define PANEL(W, R, SKB, DEPTH, KR)                                                                                      \
    {                                                                                                                     \
        auto k = pat_panel<W, R, SKB, DEPTH>;                                                                             \
        const int lds = W * DEPTH * R * SKB;                                                                              \
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                            \
        char nm[96]; snprintf(nm, 96, "panel walk: %d waves x %d rows x %d B stages, ring %d, ranges of %d k", W, R, SKB, DEPTH, KR); \
        run(nm, [&](const float *p) { hipLaunchKernelGGL(k, dim3(256), dim3(64 * W), lds, 0, p, N, KR); });            \
    }
#define PANELO(W, R, SKB, DEPTH, KR, ORD)                                                                                \
    {                                                                                                                     \
        auto k = pat_panel<W, R, SKB, DEPTH, ORD>;                                                                        \
        const int lds = W * DEPTH * R * SKB;                                                                              \
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                            \
        char nm[120]; snprintf(nm, 120, "panel walk, row-group-major%s: %d waves x %d rows x %d B stages, ring %d, ranges of %d k", ORD == 2 ? " skewed" : "", W, R, SKB, DEPTH, KR); \
        run(nm, [&](const float *p) { hipLaunchKernelGGL(k, dim3(256), dim3(64 * W), lds, 0, p, N, KR); });            \
    }
    PANELO(4, 16, 512, 3, 512, 1) PANELO(4, 16, 512, 3, 512, 2) PANELO(4, 16, 512, 3, 1024, 2) PANELO(4, 16, 512, 3, 256, 2) PANELO(8, 16, 256, 3, 256, 2)
    PANEL(4, 16, 512, 3, 512) PANEL(4, 16, 512, 3, 2048) PANEL(4, 16, 512, 3, 4096) PANEL(8, 16, 256, 3, 256)
    PANEL(4, 4, 1024, 3, 1024) PANEL(4, 4, 1024, 3, 2048) PANEL(4, 4, 1024, 3, 4096) PANEL(8, 4, 1024, 3, 2048) PANEL(8, 4, 1024, 3, 4096)
    PANEL(4, 4, 2048, 3, 2048) PANEL(4, 4, 2048, 3, 4096) PANEL(16, 4, 512, 3, 4096) PANEL(4, 4, 1024, 6, 4096) PANEL(8, 2, 1024, 4, 4096)
    PANEL(4, 8, 1024, 3, 4096) PANEL(4, 1, 4096, 3, 4096) PANEL(8, 1, 4096, 3, 4096) PANEL(16, 1, 2048, 3, 4096)
    return 0;
}
