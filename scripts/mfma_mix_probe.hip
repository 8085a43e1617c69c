// mfma_mix_probe.hip -- what the prefill tile GEMM's stage loop can sustain, piece by piece
// (DESIGN.md 4.5).  One block of 8 waves per CU; every wave runs the GEMM's instruction mix per
// super-step -- 3 ds_read_b128 (2 X tiles + 1 W tile) feeding 8 v_mfma_f32_32x32x2_f32 -- four
// super-steps per 64-k stage, a __syncthreads() per stage.  On top of that, per wave and stage:
//   NL direct-to-LDS loads of 1 KB (global_load_lds_dwordx4) from an L2-resident source, waited for
//   with vmcnt(0) before the barrier (the GEMM's 128 x 64 tile needs NL = 6);
// and variants: the loads never waited for, 4-byte loads, the operands not read from LDS at all,
// loads to VGPRs (+ ds_write_b128), and loader waves beside 8 MFMA-only waves.
// Prints TFLOP/s, the shader clock during the run (s_memtime / s_memrealtime) and the cycles each
// stage takes beyond the load-free loop, per KB brought into the CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_mix_probe.hip -o scripts/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void lds_dma16(const float *g, float *lds) { __builtin_amdgcn_global_load_lds(g, lds, 16, 0, 0); }
__device__ __forceinline__ void lds_dma4(const float *g, float *lds) { __builtin_amdgcn_global_load_lds(g, lds, 4, 0, 0); }

enum { PLAIN = 0, NOWAIT = 1, DWORD = 2, NOREAD = 3, VGPR = 4, VGPR_ONLY = 5, LOADERS_DMA = 6, LOADERS_VGPR = 7 };
constexpr int kStageFloats = 192 * 64;  // 128 X rows + 64 W rows of 64 k
constexpr int kSrcKB = 96;              // per CU: stays in the XCD's L2 (32 CUs x 96 KB)

template <int NL, int KIND>
__global__ __launch_bounds__(KIND >= LOADERS_DMA ? 768 : 512) void probe(float *out, const float *gsrc, int stages, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * kStageFloats; i += blockDim.x) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    const float *src = gsrc + (size_t)blockIdx.x * kSrcKB * 256;
    v4f *l4 = (v4f *)smem;
    int buf = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    if (KIND >= LOADERS_DMA && wave >= 8) {  // 4 loader waves: the block's 8 * NL KB per stage, no MFMA
        const int lw = wave - 8;
        for (int st = 0; st < stages; st++) {
            if (KIND == LOADERS_DMA) {
#pragma unroll
                for (int j = 0; j < 2 * NL; j++)
                    lds_dma16(src + ((st * 8 * NL + lw * 2 * NL + j) % kSrcKB) * 256 + lane * 4, smem + (buf ^ 1) * kStageFloats + ((lw * 2 * NL + j) % 48) * 256);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                v4f r[NL > 0 ? 2 * NL : 1];
#pragma unroll
                for (int j = 0; j < 2 * NL; j++) r[j] = ((const v4f *)(src + ((st * 8 * NL + lw * 2 * NL + j) % kSrcKB) * 256))[lane];
#pragma unroll
                for (int j = 0; j < 2 * NL; j++) l4[(buf ^ 1) * (kStageFloats / 4) + ((lw * 2 * NL + j) % 48) * 64 + lane] = r[j];
            }
            __syncthreads();
            buf ^= 1;
        }
        return;
    }
    v16f acc[2];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int ra = ((wave & 1) * 64 + (lane & 31)) * 16, rb = (128 + (wave >> 1 & 1) * 32 + (lane & 31)) * 16;
    const int sw = lane & 15, hl = lane >> 5, kg = wave >> 2;
    float sink = 0.f;
    for (int st = 0; st < stages; st++) {
        v4f rs[NL > 0 ? NL : 1];
        if (KIND == PLAIN || KIND == NOWAIT || KIND == NOREAD) {
#pragma unroll
            for (int j = 0; j < NL; j++)
                lds_dma16(src + (((st * 8 + wave) * NL + j) % kSrcKB) * 256 + lane * 4, smem + (buf ^ 1) * kStageFloats + ((wave * NL + j) % 48) * 256);
        } else if (KIND == DWORD) {
#pragma unroll
            for (int j = 0; j < 4 * NL; j++)
                lds_dma4(src + (((st * 8 + wave) * 4 * NL + j) % (4 * kSrcKB)) * 64 + lane, smem + (buf ^ 1) * kStageFloats + ((wave * 4 * NL + j) % 192) * 64);
        } else if (KIND == VGPR || KIND == VGPR_ONLY) {
#pragma unroll
            for (int j = 0; j < NL; j++) rs[j] = ((const v4f *)(src + (((st * 8 + wave) * NL + j) % kSrcKB) * 256))[lane];
        }
        const v4f *xr = l4 + buf * (kStageFloats / 4);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int slot = kg * 8 + 2 * s + hl;
            v4f a0, a1, b0;
            if (KIND == NOREAD) {
                a0 = v4f{acc[0][0], acc[0][1], acc[0][2], acc[0][3]}; a1 = a0; b0 = a0;
            } else {
                a0 = xr[ra + (slot ^ sw)]; a1 = xr[ra + 32 * 16 + (slot ^ sw)]; b0 = xr[rb + (slot ^ sw)];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1], 0, 0, 0);
        }
        if (KIND == VGPR) {
#pragma unroll
            for (int j = 0; j < NL; j++) l4[(buf ^ 1) * (kStageFloats / 4) + ((wave * NL + j) % 48) * 64 + lane] = rs[j];
        } else if (KIND == VGPR_ONLY) {
#pragma unroll
            for (int j = 0; j < NL; j++) sink += rs[j][0] + rs[j][3];
        }
        if (NL > 0 && (KIND == PLAIN || KIND == DWORD || KIND == NOREAD)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NL > 0) buf ^= 1;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = sink;
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static long long *g_clk;
static float *g_out, *g_src;
static int g_cus;
static double g_base_cycles = 0;

template <int NL, int KIND>
void run(const char *what)
{
    const int stages = 4000, threads = KIND >= LOADERS_DMA ? 768 : 512;
    const void *fn = (const void *)probe<NL, KIND>;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageFloats * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NL, KIND>), dim3(g_cus), dim3(threads), 2 * kStageFloats * 4, 0, g_out, g_src, stages, g_clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    long long h[1024];
    (void)hipMemcpy(h, g_clk, sizeof(long long) * 2 * g_cus, hipMemcpyDeviceToHost);
    double mhz = 0, cyc = 0;
    for (int i = 0; i < g_cus; i++) { mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; cyc += (double)h[2 * i]; }
    mhz /= g_cus; cyc /= (double)g_cus * stages;
    const double flops = 2.0 * 32 * 32 * 2 * 32.0 * stages * 8 * g_cus;  // 32 MFMAs per wave and stage
    const double tf = flops / ms / 1e9;
    if (NL == 0) g_base_cycles = cyc;
    printf("%-58s %6.1f TFLOP/s = %.3f of 157.3 | %4.0f MHz | %5.0f cycles/stage", what, tf, tf / 157.3, mhz, cyc);
    if (NL > 0) printf(" = +%4.0f = %4.1f per KB (%d KB)", cyc - g_base_cycles, (cyc - g_base_cycles) / (8.0 * NL), 8 * NL);
    printf("\n");
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    (void)hipMalloc(&g_clk, 1024 * 8);
    (void)hipMalloc(&g_out, (size_t)g_cus * 512 * 4);
    (void)hipMalloc(&g_src, (size_t)g_cus * kSrcKB * 1024 + 4096);
    (void)hipMemset(g_src, 0, (size_t)g_cus * kSrcKB * 1024 + 4096);
    printf("%d CUs, 8 waves per CU, stage = 4 x (3 ds_read_b128 + 8 MFMA 32x32x2 f32) per wave = 4096 MFMA cycles per SIMD\n", g_cus);
    run<0, PLAIN>("no loads");
    run<2, PLAIN>("2 direct-to-LDS KB per wave and stage");
    run<4, PLAIN>("4");
    run<6, PLAIN>("6 (the 128 x 64 tile)");
    run<8, PLAIN>("8 (the 128 x (64 + 64) pair tile)");
    run<12, PLAIN>("12");
    run<6, NOWAIT>("6, never waited for");
    run<6, NOREAD>("6, MFMA operands not read from LDS");
    run<6, DWORD>("6 KB as 24 loads of 4 B per lane");
    run<6, VGPR_ONLY>("6 KB to VGPRs (global_load_dwordx4), not stored");
    run<6, VGPR>("6 KB to VGPRs, ds_write_b128 at the end of the stage");
    run<6, LOADERS_DMA>("8 MFMA waves + 4 loader waves x 12 direct-to-LDS KB");
    run<6, LOADERS_VGPR>("8 MFMA waves + 4 loader waves x 12 KB via VGPRs");
    return 0;
}
