// lds_fill_probe.hip -- how many bytes per clock does ONE CU take in from the L2 (a buffer every block re-reads: the
// activation planes of the stream-form prefill GEMM) and from HBM (a buffer read once: the weights), by direct-to-LDS loads
// and by loads into registers?  One block of 8 waves per CU, every wave requests 1-KB pieces with `ahead` requests in
// flight; no arithmetic.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_fill_probe.hip -o /tmp/lds_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: direct-to-LDS (global_load_lds_dwordx4), 1: the same, non-temporal, 2: to registers (global_load_dwordx4)
// shared != 0: every block walks the SAME `span` bytes (L2 / MALL resident after the first touch); 0: block b walks its own
// span (HBM stream).  pieces: 1-KB requests per wave.
template <int MODE, int AHEAD>
__global__ __launch_bounds__(512) void fill(const float *src, size_t span_floats, int pieces, int shared, float *out)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = src + (shared ? 0 : (size_t)blockIdx.x * span_floats);
    const size_t span_pieces = span_floats / 256;   // 1-KB pieces in the span
    v4f acc = {0, 0, 0, 0};
    v4f r[AHEAD];
    // wave w takes pieces w, w + 8, ... (consecutive waves: consecutive KB, as a stage's rows are dealt)
    size_t p = wave;
    if (shared) p += (size_t)blockIdx.x * 37 % span_pieces;   // blocks start at different places of the shared span
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < AHEAD; k++) r[k] = v4f{0, 0, 0, 0};
        for (int i = 0; i < pieces; i += AHEAD) {
#pragma unroll
            for (int k = 0; k < AHEAD; k++) {
                acc += r[k];   // (the compiler waits for exactly that load: AHEAD - 1 stay in flight)
                r[k] = *(const v4f *)(base + (p % span_pieces) * 256 + 4 * lane);
                p += 8;
            }
        }
    } else {
        for (int i = 0; i < pieces; i++) {
            const float *g = base + (p % span_pieces) * 256 + 4 * lane;
            float *l = smem + ((wave * AHEAD + i % AHEAD) * 256);
            if (i >= AHEAD) wait_vm<AHEAD - 1>();
            if (MODE == 1) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 2);
            else __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
            p += 8;
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < AHEAD; k++) acc += r[k];
    } else {
        wait_vm<0>();
        __syncthreads();
        acc.x = smem[threadIdx.x];
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;
}

// The stream form's mix: per step one 1-KB piece of the block's OWN span (weights: HBM) by direct-to-LDS load, and XPW pieces
// of the SHARED span (activation planes: L2) -- XREG 0: direct-to-LDS too (what the kernel does), 1: into registers.
template <int XREG, int XPW, int AHEAD>
__global__ __launch_bounds__(512) void mix(const float *src, size_t own_floats, size_t shared_floats, int pieces, float *out)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *own = src + shared_floats + (size_t)blockIdx.x * own_floats;
    const size_t own_pieces = own_floats / 256, sh_pieces = shared_floats / 256;
    v4f acc = {0, 0, 0, 0};
    v4f r[AHEAD][XPW];
#pragma unroll
    for (int k = 0; k < AHEAD; k++)
#pragma unroll
        for (int x = 0; x < XPW; x++) r[k][x] = v4f{0, 0, 0, 0};
    size_t p = wave, q = wave + (size_t)blockIdx.x * 37;
    constexpr int PER = XREG ? 1 : 1 + XPW;   // direct-to-LDS loads per step
    for (int i = 0; i < pieces; i += AHEAD) {
#pragma unroll
        for (int k = 0; k < AHEAD; k++) {
            if (XREG) {
#pragma unroll
                for (int x = 0; x < XPW; x++) acc += r[k][x];
            } else {
                if (i > 0) wait_vm<(AHEAD - 1) * PER>();
            }
            __builtin_amdgcn_global_load_lds(own + (p % own_pieces) * 256 + 4 * lane, smem + (wave * AHEAD + k) * (1 + XPW) * 256, 16, 0, 0);
#pragma unroll
            for (int x = 0; x < XPW; x++) {
                const float *g = src + (q % sh_pieces) * 256 + 4 * lane;
                if (XREG) r[k][x] = *(const v4f *)g;
                else __builtin_amdgcn_global_load_lds(g, smem + ((wave * AHEAD + k) * (1 + XPW) + 1 + x) * 256, 16, 0, 0);
                q += 8;
            }
            p += 8;
        }
    }
    wait_vm<0>();
    __syncthreads();
    acc.x += smem[threadIdx.x];
#pragma unroll
    for (int k = 0; k < AHEAD; k++)
#pragma unroll
        for (int x = 0; x < XPW; x++) acc += r[k][x];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;
}

// The stream form's W walk: a block owns 128 rows of a [rows][4096] f32 matrix (16 KB apart) and walks k in stages of PIECE
// bytes per row; a wave-instruction brings 1024 / PIECE rows x PIECE bytes.  AHEAD stages in flight per wave.
template <int PIECE, int AHEAD>
__global__ __launch_bounds__(512) void rowwalk(const float *src, int tiles_per_block, float *out)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = PIECE / 16, RPI = 64 / LPR, IPS = 16 / RPI;   // lanes per row, rows per instruction, instructions per wave and stage
    constexpr int NST = 16384 / PIECE;                               // stages per tile
    int issued = 0;
    for (int t = 0; t < tiles_per_block; t++) {
        const float *tile = src + ((size_t)(blockIdx.x * tiles_per_block + t) * 128) * 4096;
        for (int st = 0; st < NST; st++) {
#pragma unroll
            for (int i = 0; i < IPS; i++) {
                const int row = wave * 16 + i * RPI + lane / LPR;
                if (issued >= AHEAD * IPS) wait_vm<AHEAD * IPS - 1>();
                __builtin_amdgcn_global_load_lds(tile + (size_t)row * 4096 + st * (PIECE / 4) + 4 * (lane % LPR),
                                                 smem + ((wave * AHEAD + (st % AHEAD)) * IPS + i) * 256, 16, 0, 2);
                issued++;
            }
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (smem[threadIdx.x] == 123.456f) out[blockIdx.x] = 1.0f;
}

int main()
{
    const size_t total = (size_t)3 << 30;   // 3 GB
    float *buf; hipMalloc(&buf, total); hipMemset(buf, 0, total);
    float *out; hipMalloc(&out, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int dev = 0, clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev);
    auto run = [&](auto kern, const char *name, int grid, size_t span_bytes, int pieces, int shared, int ahead) {
        float best = 1e9;
        const size_t lds = (size_t)8 * ahead * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int it = 0; it < 6; it++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, buf, span_bytes / 4, pieces, shared, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 1 && ms < best) best = ms;
        }
        const double bytes = (double)grid * 8 * pieces * 1024;
        printf("%-44s grid %4d  %8.1f us  %6.2f TB/s  %5.1f GB/s per block  (%4.1f B/clk at 2.4 GHz)\n", name, grid, best * 1e3,
               bytes / (best * 1e-3) / 1e12, bytes / grid / (best * 1e-3) / 1e9, bytes / grid / (best * 1e-3) / 2.4e9);
    };
    // every block re-reads the same 3 MB (the planes of 128 tokens x 4096 k); 8 MB per block requested
    const int pc = 1024;
    for (int grid : {256, 64, 8}) {
        run(fill<0, 4>, "shared 3 MB  direct-to-LDS, 4 ahead", grid, (size_t)3 << 20, pc, 1, 4);
        run(fill<0, 8>, "shared 3 MB  direct-to-LDS, 8 ahead", grid, (size_t)3 << 20, pc, 1, 8);
        run(fill<0, 16>, "shared 3 MB  direct-to-LDS, 16 ahead", grid, (size_t)3 << 20, pc, 1, 16);
        run(fill<2, 4>, "shared 3 MB  to registers, 4 ahead", grid, (size_t)3 << 20, pc, 1, 4);
        run(fill<2, 8>, "shared 3 MB  to registers, 8 ahead", grid, (size_t)3 << 20, pc, 1, 8);
    }
    // every block streams its own 8 MB (2 GB in all at 256 blocks): HBM
    for (int grid : {256, 192, 128, 64, 16}) {
        run(fill<0, 2>, "own 8 MB  direct-to-LDS, 2 ahead", grid, (size_t)8 << 20, pc, 0, 2);
        run(fill<0, 4>, "own 8 MB  direct-to-LDS, 4 ahead", grid, (size_t)8 << 20, pc, 0, 4);
        run(fill<0, 8>, "own 8 MB  direct-to-LDS, 8 ahead", grid, (size_t)8 << 20, pc, 0, 8);
        run(fill<1, 8>, "own 8 MB  direct-to-LDS nt, 8 ahead", grid, (size_t)8 << 20, pc, 0, 8);
        run(fill<0, 16>, "own 8 MB  direct-to-LDS, 16 ahead", grid, (size_t)8 << 20, pc, 0, 16);
        run(fill<2, 8>, "own 8 MB  to registers, 8 ahead", grid, (size_t)8 << 20, pc, 0, 8);
    }
    // the mix: per block 4 MB of its own span (1 GB in all) + XPW x as much of the shared 3 MB
    auto runmix = [&](auto kern, const char *name, int xpw, int ahead) {
        float best = 1e9;
        const size_t lds = (size_t)8 * ahead * (1 + xpw) * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int pieces = 512;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, buf, ((size_t)4 << 20) / 4, ((size_t)3 << 20) / 4, pieces, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 1 && ms < best) best = ms;
        }
        const double wb = 256.0 * 8 * pieces * 1024;
        printf("%-52s %8.1f us  weights %5.2f TB/s, planes %5.2f TB/s, per block %5.1f GB/s (%4.1f B/clk)\n", name, best * 1e3,
               wb / (best * 1e-3) / 1e12, wb * xpw / (best * 1e-3) / 1e12, wb * (1 + xpw) / 256 / (best * 1e-3) / 1e9,
               wb * (1 + xpw) / 256 / (best * 1e-3) / 2.4e9);
    };
    runmix(mix<0, 1, 4>, "mix 1 : 1, planes direct-to-LDS, 4 steps ahead", 1, 4);
    runmix(mix<1, 1, 4>, "mix 1 : 1, planes to registers, 4 steps ahead", 1, 4);
    runmix(mix<0, 2, 4>, "mix 1 : 2, planes direct-to-LDS, 4 steps ahead", 2, 4);
    runmix(mix<1, 2, 4>, "mix 1 : 2, planes to registers, 4 steps ahead", 2, 4);
    runmix(mix<0, 1, 8>, "mix 1 : 1, planes direct-to-LDS, 8 steps ahead", 1, 8);
    runmix(mix<1, 1, 8>, "mix 1 : 1, planes to registers, 8 steps ahead", 1, 8);
    // the stream form's W walk (non-temporal): 256 blocks x 2 tiles of 128 rows x 16 KB = 1 GB
    auto runrow = [&](auto kern, const char *name, int ahead, int ips) {
        float best = 1e9;
        const size_t lds = (size_t)8 * ahead * ips * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int it = 0; it < 6; it++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, buf, 2, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 1 && ms < best) best = ms;
        }
        const double bytes = 256.0 * 2 * 128 * 16384;
        printf("%-60s %8.1f us  %5.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    runrow(rowwalk<128, 4>, "row walk, 128-byte pieces (32 k), 4 stages ahead", 4, 2);
    runrow(rowwalk<128, 8>, "row walk, 128-byte pieces (32 k), 8 stages ahead", 8, 2);
    runrow(rowwalk<256, 2>, "row walk, 256-byte pieces (64 k), 2 stages ahead", 2, 4);
    runrow(rowwalk<256, 4>, "row walk, 256-byte pieces (64 k), 4 stages ahead", 4, 4);
    runrow(rowwalk<512, 2>, "row walk, 512-byte pieces (128 k), 2 stages ahead", 2, 8);
    runrow(rowwalk<1024, 1>, "row walk, 1-KB pieces (256 k), 1 stage ahead", 1, 16);
    return 0;
}
